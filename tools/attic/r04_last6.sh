#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests/test_gpu_graph_loop.py tests/test_gpu_loop_robustness.py tests/test_gpu_learner.py tests/test_cli.py tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/r04/last6.log 2>&1 || { tail -40 gpurun_out/r04/last6.log; exit 1; }
tail -2 gpurun_out/r04/last6.log
