#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 400 python tools/phi_timeline.py 14 10 > gpurun_out/r04/phi_timeline.txt 2>&1 || { tail -30 gpurun_out/r04/phi_timeline.txt; exit 1; }
grep "t in" gpurun_out/r04/phi_timeline.txt
