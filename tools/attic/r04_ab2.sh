#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 200 python tools/ppx_ab.py > gpurun_out/r04/ppx_ab.txt 2>&1 || { tail -20 gpurun_out/r04/ppx_ab.txt; exit 1; }
AMMSB_PPX_FOLD=0 timeout -k 10 200 python tools/ppx_ab.py >> gpurun_out/r04/ppx_ab.txt 2>&1 || exit 1
cat gpurun_out/r04/ppx_ab.txt
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "beta or grads or loop or golden or configs or learner or analytic or reference" > gpurun_out/r04/gputests2.log 2>&1 || { tail -40 gpurun_out/r04/gputests2.log; exit 1; }
tail -3 gpurun_out/r04/gputests2.log
timeout -k 10 200 tools/beta_trace.sh run 1024 65536 > gpurun_out/r04/beta_trace2.txt 2>&1 || { tail -20 gpurun_out/r04/beta_trace2.txt; exit 1; }
tail -12 gpurun_out/r04/beta_trace2.txt
timeout -k 10 400 python bench.py --steps 40 --warmup 10 --large none > gpurun_out/r04/bench2.json 2> gpurun_out/r04/bench2.err || { tail -20 gpurun_out/r04/bench2.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench2.json').read().strip().splitlines()[-1])
r=d['roofline']
print({k:(v.get('avg_ms'),v.get('frac')) for k,v in r['kernels'].items() if isinstance(v,dict)}, d['ppx_eval_ms'])
c=d['small_configs']['C2']; print('C2', c['ms_per_step'], {k:(v.get('avg_ms'),v.get('frac')) for k,v in c['roofline']['kernels'].items() if isinstance(v,dict)})
PY
