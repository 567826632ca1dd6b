#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
BETA_TRACE_N=1000000 timeout -k 10 300 tools/beta_trace.sh run 1024 65536 > gpurun_out/r04/beta_trace_N1M.txt 2>&1 || { tail -20 gpurun_out/r04/beta_trace_N1M.txt; exit 1; }
tail -7 gpurun_out/r04/beta_trace_N1M.txt
for i in 1 2; do
for f in 1 2; do
  [ $f = 2 ] && export AMMSB_LOOP_FUSE_PI=2 || unset AMMSB_LOOP_FUSE_PI
  timeout -k 10 300 python bench.py --steps 100 --warmup 10 --large none --extras 0 --no-cpu-baseline --cpp-dropin 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('fuse=$f', round(d['value']/1e6,3), d['step_classes']['nonlink']['ms_per_step'], {k:v.get('avg_ms') for k,v in r['kernels'].items() if isinstance(v,dict) and 'avg_ms' in v})" || exit 1
done
done
