#!/bin/bash
# C3 (the default bench workload) under the loop's forms and the eager loop, alternating, one box
for i in 1 2 3; do
  for form in default event graph eager; do
    unset AMMSB_LOOP_LAUNCH AMMSB_LOOP_HANDSHAKE; extra=""
    [ $form = event ] && export AMMSB_LOOP_HANDSHAKE=event
    [ $form = graph ] && export AMMSB_LOOP_LAUNCH=graph
    [ $form = eager ] && extra="--loop eager"
    timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --cpp-dropin 0 $extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$form', 'ms/step %.4f' % d['ms_per_step'], 'phi %.4f' % r['avg_launch_ms'], {k:round(v['ms_per_step'],4) for k,v in (d.get('step_classes') or {}).items() if isinstance(v,dict)})"
  done
done
