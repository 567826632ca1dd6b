#!/bin/bash
# Run the C++ command line on a synthetic graph (development aid): graph -> gzip data set -> ammsb_main.
#   tools/cli_bench.sh DIR N K M ITERS ASYNC [WG]
set -e
cd "$(dirname "$0")/.."
D=${1:-/tmp/ammsb_cli}; N=${2:-1000000}; K=${3:-1024}; M=${4:-65536}; IT=${5:-200}; AS=${6:-1}; WG=${7:-64}
mkdir -p $D
F=$D/g_$N.bin.gz
if [ ! -f $F ]; then
python - <<PY
import sys
sys.path.insert(0, '.')
import ammsb_pkg; ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import hostlib
e = hostlib.generate_graph($N, min($K, 64), 32, seed=20260101)
hostlib.dump_dataset('$F', $N, 0.01, e)
print('graph', e.size, 'edges')
PY
fi
./mcmc-ammsb-gpu_amd/ammsb_main --load-data 1 --load-file $F -k $K -m $M -n 32 -x $IT -i $IT \
   --phi-wg $WG --beta-wg $WG --ppx-wg $WG --device-sampling 1 --async $AS 2>&1 | grep -E "ppx\[|TOTAL|MINI-BATCH"
