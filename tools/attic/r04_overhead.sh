#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
AMMSB_LOOP_HOSTPROF=1 timeout -k 10 400 python tools/run_overhead.py C3 > gpurun_out/r04/run_overhead.txt 2>&1 || { tail -30 gpurun_out/r04/run_overhead.txt; exit 1; }
cat gpurun_out/r04/run_overhead.txt | tail -30
