#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 500 python tools/attic/ckpt_diff.py > gpurun_out/r04/ckpt_diff.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/ckpt_diff.txt | tail -30
