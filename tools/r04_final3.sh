#!/bin/bash
# closing run 3: the whole GPU suite and smoke() on the tree with the replicated multi-GPU gradient, then a 4-rank
# rehearsal of the multi-rank bench on one GPU (gloo: code path only, never numbers)
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gputests_final3.log 2>&1 || { tail -40 gpurun_out/r04/gputests_final3.log; exit 1; }
tail -3 gpurun_out/r04/gputests_final3.log
timeout -k 10 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
AMMSB_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 4 --steps 6 --warmup 2 --settle-s 0.2 > gpurun_out/r04/rehearsal4b.json 2> gpurun_out/r04/rehearsal4b.err || { tail -30 gpurun_out/r04/rehearsal4b.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/rehearsal4b.json').read().strip().splitlines()[-1])
t=d['config']['phi_split']['trace']
print('rehearsal4', d['n_gpus'], d['value'], d['ms_per_step'], {k:t.get(k) for k in ('steps','update_pi_ms','grads_local_ms','grad_allgather_ms','gradient')})
PY
