#!/bin/bash
# Round-4 closing profile (pi placement in effect): rocprofv3 kernel trace + stats of the default bench command's
# workload (C3) and the bench line of the same process.  Summaries -> profiles/r04b_*.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04/profb
mkdir -p $O
COMMON="--no-cpu-baseline --cpp-dropin 0 --extras 0 --sustained-s 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o c3 -- python3 bench.py --steps 40 --warmup 5 $COMMON > $O/c3_bench.log 2> $O/c3_bench.err || exit 1
tail -1 $O/c3_bench.log | head -c 600
find $O -name "*stats*.csv" | head
