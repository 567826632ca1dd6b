#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
: > gpurun_out/r04/placement_contig.txt
for i in 1 2 3; do
  timeout -k 10 300 python tools/placement_contig.py 3 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04/placement_contig.txt || { tail -30 gpurun_out/r04/placement_contig.txt; exit 1; }
done
cat gpurun_out/r04/placement_contig.txt
