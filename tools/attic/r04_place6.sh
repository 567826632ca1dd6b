#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests/test_gpu_pi_placement.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_bench_launcher.py tests/test_cli.py -x -q -m gpu > gpurun_out/r04/place_tests.log 2>&1 || { tail -60 gpurun_out/r04/place_tests.log; exit 1; }
tail -3 gpurun_out/r04/place_tests.log
