#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_bench_launcher.py -x -q -m gpu > gpurun_out/r04/launcher5.log 2>&1 || { tail -40 gpurun_out/r04/launcher5.log; exit 1; }
tail -2 gpurun_out/r04/launcher5.log
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_final5.json 2> gpurun_out/r04/bench_final5.err || { tail -20 gpurun_out/r04/bench_final5.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_final5.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value', d['value'], 'vpc', d['value_per_class']['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'avg_launch_ms', r['avg_launch_ms'])
print({k:(v.get('avg_ms'),v.get('frac')) for k,v in r['kernels'].items() if isinstance(v,dict)}, 'ppx', d['ppx_eval_ms'])
su=d['sustained']; print('sustained', {k:su.get(k) for k in ('after_s','untimed_steps','value','value_per_class','update_phi_ms','frac')}, su['device_state'].get('power_w'), su['device_state'].get('shader_clock_under_load_mhz'))
print('settle', d['settle']); print('placement', d['pi_placement'])
print('state', {k:r['device_state'].get(k) for k in ('sclk_mhz','power_w','shader_clock_under_load_mhz')})
print('C5', {k:(v.get('value'), v['roofline']['frac']) for k,v in d['large_configs'].items()}, 'wg32', d['reference_default_wg']['value'], d['reference_default_wg']['roofline']['kernels']['update_phi']['avg_ms'])
print('cpp', {k:(v.get('edges_per_s') if isinstance(v,dict) else None) for k,v in d['cpp_dropin'].items()})
PY
grep "bench " gpurun_out/r04/bench_final5.err | tail -2
