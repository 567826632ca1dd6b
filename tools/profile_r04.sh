#!/bin/bash
# Round-4 profiles on one GPU box (run from the repo root through gpurun): rocprofv3 kernel trace + stats of the
# default bench command's workload (C3), of C2 and of C5 (N = 10M, K = 4096), the two PMC passes of C3 and the SQ
# counters of its large kernels.  Raw output under gpurun_out/r04/prof; summaries are copied to profiles/r04_*.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04/prof
mkdir -p $O
COMMON="--no-cpu-baseline --cpp-dropin 0 --extras 0 --settle-s 0.5"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o c3 -- python3 bench.py --steps 40 --warmup 5 $COMMON > $O/c3_bench.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -o c2 -- python3 bench.py --workload C2 --steps 200 --warmup 20 $COMMON > $O/c2_bench.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o c5 -- python3 bench.py --workload C5 --steps 40 --warmup 6 $COMMON > $O/c5_bench.log 2>&1 || exit 1
# the counter passes run one kernel at a time: the loop orders its chains with stream events there (its own default under --pmc)
export AMMSB_LOOP_HANDSHAKE=event
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 bench.py --steps 20 --warmup 4 $COMMON > $O/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 bench.py --steps 20 --warmup 4 $COMMON > $O/write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O/sq -o p -- python3 bench.py --steps 30 --warmup 5 $COMMON > $O/sq.log 2>&1 || exit 1
unset AMMSB_LOOP_HANDSHAKE
find $O -name "*.csv" | head -40
