#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
: > gpurun_out/r04/placement_c5.txt
for s in "0 hold C5" "40 hold C5" "80 hold C5"; do
  timeout -k 10 380 python tools/placement_learner.py $s 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04/placement_c5.txt || { tail -30 gpurun_out/r04/placement_c5.txt; exit 1; }
  tail -1 gpurun_out/r04/placement_c5.txt
done
