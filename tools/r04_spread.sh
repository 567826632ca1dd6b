#!/bin/bash
# Round 4, item 1: box info + update_phi spread experiment (3 processes production build, 3 processes trace build)
set -o pipefail
mkdir -p gpurun_out/r04
tools/box_info.sh > gpurun_out/r04/box_info.txt 2>&1
for i in 1 2 3; do
  timeout -k 10 240 python tools/phi_spread.py A,B32,Cpad,Dmemset,A >> gpurun_out/r04/spread_prod.jsonl 2>> gpurun_out/r04/spread.err || exit 1
  AMMSB_HIP_LIB=$PWD/tools/ab/trace/libammsb_hip_trace.so timeout -k 10 240 python tools/phi_spread.py A,A >> gpurun_out/r04/spread_trace.jsonl 2>> gpurun_out/r04/spread.err || exit 1
done
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench0.json 2> gpurun_out/r04/bench0.err
