#!/bin/bash
# the driver's two commands, back to back, on the final tree of the round
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gputests_final.log 2>&1 || { tail -40 gpurun_out/r04/gputests_final.log; exit 1; }
tail -3 gpurun_out/r04/gputests_final.log
timeout -k 10 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_final.json 2> gpurun_out/r04/bench_final.err || { tail -20 gpurun_out/r04/bench_final.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_final.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value', d['value'], 'vpc', d['value_per_class']['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'avg_launch_ms', r['avg_launch_ms'])
print({k:(v.get('avg_ms'),v.get('frac')) for k,v in r['kernels'].items() if isinstance(v,dict)}, 'ppx', d['ppx_eval_ms'])
print('classes', {k:(round(v['ms_per_step'],4)) for k,v in d['step_classes'].items() if isinstance(v,dict)})
print('state', {k:r['device_state'].get(k) for k in ('sclk_mhz','power_w','shader_clock_under_load_mhz')})
for n in ('C1','C2'):
    c=d['small_configs'][n]; print(n, c['ms_per_step'], {k:(v.get('avg_ms'),v.get('frac')) for k,v in c['roofline']['kernels'].items() if isinstance(v,dict)}, c['ppx_eval_ms'])
print('wg32', d['reference_default_wg']['value'])
c=d['large_configs']; print({k:(v.get('value'), v.get('ms_per_step'), v.get('ppx_eval_ms'), v['roofline']['frac']) for k,v in c.items()})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'cpp', {k:(v.get('edges_per_s') if isinstance(v,dict) else None) for k,v in d['cpp_dropin'].items()})
PY
grep "bench " gpurun_out/r04/bench_final.err | tail -3
