#!/bin/bash
# the loop's launch forms side by side on one box: plain launches from two threads (default), graph replay from
# two threads, graph replay from one thread, event-ordered graphs (round-2's first form)
O=gpurun_out/hs; mkdir -p $O
for w in C1 C2 C3; do
  steps=1000; [ $w = C3 ] && steps=100
  for mode in default graph serial event; do
    unset AMMSB_LOOP_LAUNCH AMMSB_LOOP_HANDSHAKE
    [ $mode = graph ] && export AMMSB_LOOP_LAUNCH=graph
    [ $mode = serial ] && export AMMSB_LOOP_LAUNCH=serial
    [ $mode = event ] && export AMMSB_LOOP_HANDSHAKE=event
    timeout -k 10 400 python bench.py --workload $w --steps $steps --warmup 100 --no-cpu-baseline --cpp-dropin 0 > $O/${mode}_$w.log 2>&1 || exit 1
    echo "== $w $mode"; tail -1 $O/${mode}_$w.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k in ('ms_per_step','host_enqueue_ms_per_step','value')}, {k:round(v['ms_per_step'],4) for k,v in d['step_classes'].items() if isinstance(v,dict)})"
  done
done
