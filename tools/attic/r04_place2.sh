#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
: > gpurun_out/r04/placement_learner.txt
for s in "0" "6 hold" "0" "6 hold" "6 free" "24 hold" "0"; do
  timeout -k 10 200 python tools/placement_learner.py $s 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04/placement_learner.txt || { tail -30 gpurun_out/r04/placement_learner.txt; exit 1; }
done
cat gpurun_out/r04/placement_learner.txt
