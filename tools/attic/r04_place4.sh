#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
: > gpurun_out/r04/placement_ab.txt
for c in 0 6 0 6 0 6 0 6; do
  AMMSB_PI_CANDIDATES=$c timeout -k 10 200 python tools/placement_learner.py 0 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04/placement_ab.txt || { tail -30 gpurun_out/r04/placement_ab.txt; exit 1; }
done
cat gpurun_out/r04/placement_ab.txt
