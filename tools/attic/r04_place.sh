#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 300 python tools/placement_probe.py 6 30 > gpurun_out/r04/placement_probe.txt 2>&1 || { tail -30 gpurun_out/r04/placement_probe.txt; exit 1; }
timeout -k 10 300 python tools/placement_probe.py 6 30 >> gpurun_out/r04/placement_probe.txt 2>&1 || { tail -30 gpurun_out/r04/placement_probe.txt; exit 1; }
cat gpurun_out/r04/placement_probe.txt
