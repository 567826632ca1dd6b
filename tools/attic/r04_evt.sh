#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_learner.py tests/test_bench_launcher.py tests/test_gpu_graph_loop.py -x -q -m gpu > gpurun_out/r04/evt_tests.log 2>&1 || { tail -40 gpurun_out/r04/evt_tests.log; exit 1; }
tail -3 gpurun_out/r04/evt_tests.log
timeout -k 10 500 python tools/shard_host_cost.py 0.09 4 > gpurun_out/r04/shard_host_cost2.txt 2>&1 || { tail -30 gpurun_out/r04/shard_host_cost2.txt; exit 1; }
grep "host enqueue" gpurun_out/r04/shard_host_cost2.txt
