#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "beta or grads or loop or golden or configs or learner or analytic or reference or perplexity or ppx" > gpurun_out/r04/gputests3.log 2>&1 || { tail -40 gpurun_out/r04/gputests3.log; exit 1; }
tail -3 gpurun_out/r04/gputests3.log
timeout -k 10 200 tools/beta_trace.sh run 1024 65536 > gpurun_out/r04/beta_trace3.txt 2>&1 || { tail -20 gpurun_out/r04/beta_trace3.txt; exit 1; }
tail -8 gpurun_out/r04/beta_trace3.txt
for i in 1 2; do
timeout -k 10 400 python bench.py --steps 60 --warmup 10 --large none --no-cpu-baseline --cpp-dropin 0 > gpurun_out/r04/bench3_$i.json 2> gpurun_out/r04/bench3_$i.err || { tail -20 gpurun_out/r04/bench3_$i.err; exit 1; }
python - $i <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r04/bench3_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], {k:(v.get('avg_ms'),v.get('frac')) for k,v in r['kernels'].items() if isinstance(v,dict)}, d['ppx_eval_ms'])
c=d['small_configs']['C2']; print('C2', c['ms_per_step'], {k:(v.get('avg_ms'),v.get('frac')) for k,v in c['roofline']['kernels'].items() if isinstance(v,dict)}, c['ppx_eval_ms'])
c=d['small_configs']['C1']; print('C1', c['ms_per_step'], {k:(v.get('avg_ms'),v.get('frac')) for k,v in c['roofline']['kernels'].items() if isinstance(v,dict)})
print('wg32', d['reference_default_wg']['value'], {k:(v.get('avg_ms')) for k,v in d['reference_default_wg']['roofline']['kernels'].items() if isinstance(v,dict)})
PY
done
