#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
AMMSB_BETA_PI_NT=1 timeout -k 10 600 python -m pytest tests/test_gpu_fused_pi_beta.py tests/test_gpu_graph_loop.py -x -q -m gpu > gpurun_out/r04/nt1.log 2>&1 || { tail -30 gpurun_out/r04/nt1.log; exit 1; }
tail -1 gpurun_out/r04/nt1.log
timeout -k 10 600 python -m pytest tests/test_gpu_fused_pi_beta.py tests/test_gpu_graph_loop.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r04/nt0.log 2>&1 || { tail -30 gpurun_out/r04/nt0.log; exit 1; }
tail -1 gpurun_out/r04/nt0.log
