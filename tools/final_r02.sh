#!/bin/bash
# Round-2 closing run on one GPU box: full -m gpu suite, smoke, then the bench lines of every single-GPU workload
# (the default command last, with its CPU-baseline and C++ drop-in legs).  Outputs under gpurun_out/r02/.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1 || { tail $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
for wl in C1 C2; do
  timeout -k 10 400 python bench.py --workload $wl --steps 1000 --warmup 50 > $O/bench_$wl.log 2> $O/bench_$wl.err || { tail -5 $O/bench_$wl.err; exit 1; }
  timeout -k 10 400 python bench.py --workload $wl --loop eager --steps 1000 --warmup 50 --no-cpu-baseline --cpp-dropin 0 > $O/bench_${wl}_eager.log 2>/dev/null || exit 1
done
timeout -k 10 600 python bench.py --loop eager --no-cpu-baseline --cpp-dropin 0 > $O/bench_C3_eager.log 2>/dev/null || exit 1
timeout -k 10 900 python bench.py > $O/bench_C3.log 2> $O/bench_C3.err || { tail -5 $O/bench_C3.err; exit 1; }
tail -c 3000 $O/bench_C3.log
