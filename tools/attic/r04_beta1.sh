#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gputests1.log 2>&1 || { tail -40 gpurun_out/r04/gputests1.log; exit 1; }
tail -3 gpurun_out/r04/gputests1.log
timeout -k 10 200 tools/beta_trace.sh run 1024 65536 > gpurun_out/r04/beta_trace1.txt 2>&1 || { tail -20 gpurun_out/r04/beta_trace1.txt; exit 1; }
cat gpurun_out/r04/beta_trace1.txt | tail -25
timeout -k 10 400 python bench.py --steps 40 --warmup 10 > gpurun_out/r04/bench1.json 2> gpurun_out/r04/bench1.err || { tail -20 gpurun_out/r04/bench1.err; exit 1; }
