#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gputests4.log 2>&1 || { tail -40 gpurun_out/r04/gputests4.log; exit 1; }
tail -3 gpurun_out/r04/gputests4.log
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench4.json 2> gpurun_out/r04/bench4.err || { tail -20 gpurun_out/r04/bench4.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench4.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['value_per_class']['value'], {k:(v.get('avg_ms'),v.get('frac')) for k,v in r['kernels'].items() if isinstance(v,dict)}, d['ppx_eval_ms'])
for n in ('C1','C2'):
    c=d['small_configs'][n]; print(n, c['ms_per_step'], c['step_classes']['nonlink']['ms_per_step'], {k:(v.get('avg_ms'),v.get('frac')) for k,v in c['roofline']['kernels'].items() if isinstance(v,dict)}, c['ppx_eval_ms'])
c=d['large_configs']; print({k:(v.get('value'), v.get('ms_per_step'), v.get('ppx_eval_ms')) for k,v in c.items()})
PY
