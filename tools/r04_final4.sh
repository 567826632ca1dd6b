#!/bin/bash
# closing run 4 (pi placement, adaptive settle, sustained window, replicated multi-GPU gradient): the whole GPU suite,
# smoke(), a 4-rank rehearsal of the multi-rank bench on one GPU (gloo: code path only), the driver's bench command
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gputests_final4.log 2>&1 || { tail -40 gpurun_out/r04/gputests_final4.log; exit 1; }
tail -3 gpurun_out/r04/gputests_final4.log
timeout -k 10 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
AMMSB_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 4 --steps 6 --warmup 2 --settle-s 0.2 --sustained-s 0 > gpurun_out/r04/rehearsal4c.json 2> gpurun_out/r04/rehearsal4c.err || { tail -30 gpurun_out/r04/rehearsal4c.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/rehearsal4c.json').read().strip().splitlines()[-1])
print('rehearsal4', d['n_gpus'], d['value'], d['ms_per_step'], d['config']['parallelism'], d['pi_placement'])
PY
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_final6.json 2> gpurun_out/r04/bench_final6.err || { tail -20 gpurun_out/r04/bench_final6.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_final6.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value', d['value'], 'vpc', d['value_per_class']['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'avg_launch_ms', r['avg_launch_ms'])
print({k:(v.get('avg_ms'),v.get('frac')) for k,v in r['kernels'].items() if isinstance(v,dict)}, 'ppx', d['ppx_eval_ms'])
su=d['sustained']; print('sustained', {k:su.get(k) for k in ('after_s','untimed_steps','value','value_per_class','update_phi_ms','frac')}, su['device_state'].get('power_w'), su['device_state'].get('shader_clock_under_load_mhz'))
print('settle', d['settle']['auto']); print('placement', d['pi_placement'])
print('state', {k:r['device_state'].get(k) for k in ('sclk_mhz','power_w','shader_clock_under_load_mhz')})
for n in ('C1','C2'):
    c=d['small_configs'][n]; print(n, c['ms_per_step'], {k:(v.get('avg_ms'),v.get('frac')) for k,v in c['roofline']['kernels'].items() if isinstance(v,dict)}, c['ppx_eval_ms'])
print('C5', {k:(v.get('value'), v.get('ms_per_step'), v.get('ppx_eval_ms'), v['roofline']['frac']) for k,v in d['large_configs'].items()}, 'wg32', d['reference_default_wg']['value'], d['reference_default_wg']['roofline']['kernels']['update_phi']['avg_ms'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'cpp', {k:(v.get('edges_per_s') if isinstance(v,dict) else None) for k,v in d['cpp_dropin'].items()})
PY
grep "bench " gpurun_out/r04/bench_final6.err | tail -2
