#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 300 python tools/phi_startup.py idle-first > gpurun_out/r04/phi_startup.txt 2>&1 || { tail -30 gpurun_out/r04/phi_startup.txt; exit 1; }
timeout -k 10 300 python tools/phi_startup.py load-first >> gpurun_out/r04/phi_startup.txt 2>&1 || { tail -30 gpurun_out/r04/phi_startup.txt; exit 1; }
grep -v "amdgpu.ids" gpurun_out/r04/phi_startup.txt
