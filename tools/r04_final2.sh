#!/bin/bash
# closing run 2: launcher tests (2-rank rehearsal with the watchdog armed), a 4-rank rehearsal of the multi-rank bench
# on one GPU (gloo: code path only, never numbers), then the driver's bench command
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_bench_launcher.py tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/r04/launcher2.log 2>&1 || { tail -40 gpurun_out/r04/launcher2.log; exit 1; }
tail -3 gpurun_out/r04/launcher2.log
AMMSB_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 4 --steps 6 --warmup 2 --settle-s 0.2 > gpurun_out/r04/rehearsal4.json 2> gpurun_out/r04/rehearsal4.err || { tail -30 gpurun_out/r04/rehearsal4.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/rehearsal4.json').read().strip().splitlines()[-1])
print('rehearsal4', d['n_gpus'], d['value'], d['ms_per_step'], json.dumps(d['config'].get('phi_split'))[:1500])
PY
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_final2.json 2> gpurun_out/r04/bench_final2.err || { tail -20 gpurun_out/r04/bench_final2.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_final2.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value', d['value'], 'vpc', d['value_per_class']['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'avg_launch_ms', r['avg_launch_ms'])
print({k:(v.get('avg_ms'),v.get('frac')) for k,v in r['kernels'].items() if isinstance(v,dict)}, 'ppx', d['ppx_eval_ms'])
print('classes', {k:(round(v['ms_per_step'],4)) for k,v in d['step_classes'].items() if isinstance(v,dict)})
print('state', {k:r['device_state'].get(k) for k in ('sclk_mhz','power_w','shader_clock_under_load_mhz')}, d['settle']['steps'])
for n in ('C1','C2'):
    c=d['small_configs'][n]; print(n, c['ms_per_step'], {k:(v.get('avg_ms'),v.get('frac')) for k,v in c['roofline']['kernels'].items() if isinstance(v,dict)}, c['ppx_eval_ms'])
print('wg32', d['reference_default_wg']['value'], d['reference_default_wg']['roofline']['kernels']['update_phi']['avg_ms'])
c=d['large_configs']; print({k:(v.get('value'), v.get('ms_per_step'), v.get('ppx_eval_ms'), v['roofline']['frac']) for k,v in c.items()})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'cpp', {k:(v.get('edges_per_s') if isinstance(v,dict) else None) for k,v in d['cpp_dropin'].items()})
PY
grep "bench " gpurun_out/r04/bench_final2.err | tail -3
