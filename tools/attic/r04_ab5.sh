#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "graph_loop or loop_robust or configs or learner or cli or state or cpp" > gpurun_out/r04/gputests5.log 2>&1 || { tail -40 gpurun_out/r04/gputests5.log; exit 1; }
tail -3 gpurun_out/r04/gputests5.log
for i in 1 2; do
  timeout -k 10 300 python bench.py --steps 100 --warmup 10 --large none --no-cpu-baseline --cpp-dropin 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,3), round(d['value_per_class']['value']/1e6,3), d['step_classes']['nonlink']['ms_per_step'], d['step_classes']['link']['ms_per_step'], {k:v.get('avg_ms') for k,v in r['kernels'].items() if isinstance(v,dict) and 'avg_ms' in v}); print('wg32', round(d['reference_default_wg']['value']/1e6,3), {k:v.get('avg_ms') for k,v in d['reference_default_wg']['roofline']['kernels'].items() if isinstance(v,dict) and 'avg_ms' in v}); print('cpp', {k:(v.get('edges_per_s') if isinstance(v,dict) else None) for k,v in d['cpp_dropin'].items()})" || exit 1
done
