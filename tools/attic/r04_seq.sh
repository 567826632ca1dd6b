#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 300 python tools/phi_in_sequence.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/phi_in_sequence.txt || { tail -30 gpurun_out/r04/phi_in_sequence.txt; exit 1; }
cat gpurun_out/r04/phi_in_sequence.txt
