#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
: > gpurun_out/r04/placement_phi.txt
for i in 1 2 3; do
  timeout -k 10 300 python tools/placement_phi.py 5 12 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04/placement_phi.txt || { tail -30 gpurun_out/r04/placement_phi.txt; exit 1; }
done
cat gpurun_out/r04/placement_phi.txt
