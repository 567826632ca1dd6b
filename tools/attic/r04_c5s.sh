#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
for i in 1 2; do
timeout -k 10 500 python bench.py --workload C5s --extras 0 --no-cpu-baseline --cpp-dropin 0 --steps 40 --warmup 5 --sustained-s 0 > gpurun_out/r04/bench_c5s_$i.json 2> gpurun_out/r04/bench_c5s_$i.err || { tail -20 gpurun_out/r04/bench_c5s_$i.err; exit 1; }
python - <<PY
import json
d=json.loads(open('gpurun_out/r04/bench_c5s_$i.json').read().strip().splitlines()[-1])
print('C5s value', round(d['value']), 'update_phi', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'placement', d['pi_placement'])
PY
done
