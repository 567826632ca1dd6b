#!/bin/bash
# kernel trace of the C1 loop (tools/host_prof.py) -> per-queue timeline window (tools/trace_view.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c1trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 tools/host_prof.py > $O/run.log 2>&1 || exit 1
f=$(find $O -name "*kernel_trace.csv" | head -1)
python3 tools/trace_view.py $f 30000 90 > $O/window.txt
cat $O/window.txt
grep "Run(" $O/run.log
