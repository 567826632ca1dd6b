#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
: > gpurun_out/r04/phi_forms_ab.txt
for i in 1 2; do
  timeout -k 10 300 python tools/phi_forms_ab.py 12 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04/phi_forms_ab.txt || { tail -30 gpurun_out/r04/phi_forms_ab.txt; exit 1; }
done
cat gpurun_out/r04/phi_forms_ab.txt
