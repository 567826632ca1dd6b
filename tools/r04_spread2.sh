#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
for i in 1 2 3; do
  timeout -k 10 300 python tools/phi_spread.py A,V32,B32,B8,B2,V2,B16,A,B32 >> gpurun_out/r04/spread2_prod.jsonl 2>> gpurun_out/r04/spread2.err || exit 1
done
