#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_final7.json 2> gpurun_out/r04/bench_final7.err || { tail -20 gpurun_out/r04/bench_final7.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_final7.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value', d['value'], 'vpc', d['value_per_class']['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'avg_launch_ms', r['avg_launch_ms'])
su=d['sustained']; print('sustained', {k:su.get(k) for k in ('value','value_per_class','update_phi_ms','frac')})
print('settle', d['settle']['auto']); print('placement', d['pi_placement'])
print('C5', {k:(v.get('value'), v['roofline']['frac']) for k,v in d['large_configs'].items()}, 'wg32', d['reference_default_wg']['value'])
print('cpp', {k:({a:v.get(a) for a in ('edges_per_s','total_s','iterations')} if isinstance(v,dict) else None) for k,v in d['cpp_dropin'].items()})
PY
grep "bench " gpurun_out/r04/bench_final7.err | tail -1
