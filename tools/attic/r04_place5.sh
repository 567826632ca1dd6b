#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
: > gpurun_out/r04/placement_ab2.txt
for c in 0 6 0 6 0 6; do
  AMMSB_PI_CANDIDATES=$c timeout -k 10 200 python tools/placement_learner.py 0 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04/placement_ab2.txt || { tail -30 gpurun_out/r04/placement_ab2.txt; exit 1; }
done
cat gpurun_out/r04/placement_ab2.txt
for c in 0 6 0 6; do
  AMMSB_PI_CANDIDATES=$c timeout -k 10 300 python bench.py --extras 0 --no-cpu-baseline --cpp-dropin 1 --sustained-s 2 > gpurun_out/r04/bench_place_$c.json 2> gpurun_out/r04/bench_place_$c.err || { tail -20 gpurun_out/r04/bench_place_$c.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04/bench_place_$c.json').read().strip().splitlines()[-1])
su=d['sustained']
print('candidates $c: value', round(d['value']), 'update_phi', d['roofline']['avg_launch_ms'], '| sustained', round(su['value']), su['update_phi_ms'], '| cpp', {k:(v.get('edges_per_s') if isinstance(v,dict) else None) for k,v in d['cpp_dropin'].items()}, '|', d['pi_placement'])
PY
  grep "pi placement" gpurun_out/r04/bench_place_$c.err | tail -3
done
