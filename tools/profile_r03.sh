#!/bin/bash
# Round-3 profiles on one GPU box (run from the repo root through gpurun): rocprofv3 kernel trace + stats of the
# default bench command (C3, captured-graph loop) and of C2, then the two PMC passes of C3.  Outputs under
# gpurun_out/r03/; tools/pmc_summary.py and the copy step below put the judged summaries under profiles/.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o c3 -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --cpp-dropin 0 --extras 0 > $O/c3_bench.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -o c2 -- python3 bench.py --workload C2 --steps 200 --warmup 20 --no-cpu-baseline --cpp-dropin 0 --extras 0 > $O/c2_bench.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c1 -o c1 -- python3 bench.py --workload C1 --steps 400 --warmup 40 --no-cpu-baseline --cpp-dropin 0 --extras 0 > $O/c1_bench.log 2>&1 || exit 1
# the counter passes run one kernel at a time: order the loop's chains with stream events, not polling kernels
export AMMSB_LOOP_HANDSHAKE=event   # (also the library's own default under --pmc)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --cpp-dropin 0 --extras 0 > $O/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --cpp-dropin 0 --extras 0 > $O/write.log 2>&1 || exit 1
find $O -name "*.csv" | head -20
# the same workload at the reference's default work-group sizes (32): kernel trace + stats
unset AMMSB_LOOP_HANDSHAKE
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3wg32 -o c3wg32 -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --cpp-dropin 0 --extras 0 --phi-wg 32 --beta-wg 32 --ppx-wg 32 > $O/c3wg32_bench.log 2>&1 || exit 1
