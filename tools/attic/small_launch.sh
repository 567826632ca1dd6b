#!/bin/bash
# per-launch time of the step's kernels on a launch of a few dozen nodes (a link mini-batch), back to back
for cfg in "100000 256 64" "1000000 1024 64" "10000 32 32" "10000 32 64"; do
  set -- $cfg
  for noise in 1 0; do
    echo "== N=$1 K=$2 wg=$3 noise=$noise, 31 nodes"
    python tools/kbench.py --N $1 --K $2 --m 30 --set-edges 200000 --only phi,pi,beta --phi-wgs $3 --beta-wgs $3 --batch 200 --noise $noise 2>&1 | grep -v amdgpu
  done
done
