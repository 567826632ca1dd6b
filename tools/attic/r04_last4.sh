#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests/test_gpu_pi_placement.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_generic_kernels.py -x -q -m gpu > gpurun_out/r04/last4.log 2>&1 || { tail -40 gpurun_out/r04/last4.log; exit 1; }
tail -2 gpurun_out/r04/last4.log
