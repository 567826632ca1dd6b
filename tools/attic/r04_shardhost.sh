#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 500 python tools/shard_host_cost.py 0.09 4 > gpurun_out/r04/shard_host_cost.txt 2>&1 || { tail -30 gpurun_out/r04/shard_host_cost.txt; exit 1; }
grep -v "^$" gpurun_out/r04/shard_host_cost.txt | tail -60
