#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
for p in 0 64 128 0 32 96; do
  AMMSB_LOOP_SIDE_CUS=$p timeout -k 10 300 python bench.py --extras 0 --no-cpu-baseline --cpp-dropin 0 --steps 200 --warmup 20 --sustained-s 0 > gpurun_out/r04/bench_cus_$p.json 2> gpurun_out/r04/bench_cus_$p.err || { tail -20 gpurun_out/r04/bench_cus_$p.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04/bench_cus_$p.json').read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print('side CUs $p: value', round(d['value']), 'vpc', round(d['value_per_class']['value']), 'classes', {a:round(b['ms_per_step'],4) for a,b in d['step_classes'].items() if isinstance(b,dict)}, {a:b.get('avg_ms') for a,b in k.items() if isinstance(b,dict) and 'avg_ms' in b}, 'link wait', k['link_steps_ms']['wait_for_sampler'], 'kept', d['pi_placement']['kept_ms'])
PY
done
