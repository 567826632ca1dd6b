#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests/test_gpu_fused_pi_beta.py tests/test_gpu_distributed.py tests/test_cpp_exchange.py tests/test_bench_launcher.py -x -q -m gpu > gpurun_out/r04/grads_tests.log 2>&1 || { tail -60 gpurun_out/r04/grads_tests.log; exit 1; }
tail -3 gpurun_out/r04/grads_tests.log
timeout -k 10 500 python tools/shard_host_cost.py 0.09 4 > gpurun_out/r04/shard_host_cost3.txt 2>&1 || { tail -30 gpurun_out/r04/shard_host_cost3.txt; exit 1; }
grep "host enqueue" gpurun_out/r04/shard_host_cost3.txt
