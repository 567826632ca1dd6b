#!/bin/bash
# C3-shaped run of the C++ command line (development aid): synthetic graph -> gzip data set -> ammsb_main.
set -e
cd "$(dirname "$0")/.."
D=${1:-/tmp/ammsb_c3}
mkdir -p $D
python - <<PY
import sys, time
sys.path.insert(0, '.')
import ammsb_pkg; ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import hostlib
t=time.time()
e = hostlib.generate_graph(1000000, 64, 32, seed=20260101)
hostlib.dump_dataset('$D/c3.bin.gz', 1000000, 0.01, e)
print('graph', e.size, 'edges,', round(time.time()-t,1), 's')
PY
./mcmc-ammsb-gpu_amd/ammsb_main --load-data 1 --load-file $D/c3.bin.gz -k 1024 -m 65536 -n 32 -x ${2:-200} -i 100 \
   --phi-wg 64 --beta-wg 64 --ppx-wg 64 --device-sampling ${3:-1} --async ${4:-0} 2>&1 | grep -v "^I   \|^I [a-z_]*:" | tail -25
