#!/bin/bash
# In-kernel shader-clock stamps of update_phi_lds2_kernel (block 0, first node): where a wave's time goes.
# Build (CPU box): tools/phi_trace.sh build   -> tools/ab/trace/libammsb_hip_trace.so (-DAMMSB_PHI_TRACE)
# Run (GPU box):   tools/phi_trace.sh run K n nodes
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p tools/ab/trace
  for f in core phi beta ppx minibatch loop setbuild; do
    extra=""; [ $f = phi ] && extra="-DAMMSB_PHI_TRACE"
    if [ $f = phi ] || [ ! -f tools/ab/trace/ammsb_$f.o ] || [ mcmc-ammsb-gpu_amd/csrc/ammsb_$f.hip -nt tools/ab/trace/ammsb_$f.o ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $extra \
        -c mcmc-ammsb-gpu_amd/csrc/ammsb_$f.hip -o tools/ab/trace/ammsb_$f.o &
    fi
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/trace/libammsb_hip_trace.so tools/ab/trace/ammsb_*.o -lpthread
  exit 0
fi
shift
AMMSB_HIP_LIB=$PWD/tools/ab/trace/libammsb_hip_trace.so python - "$@" <<'PY'
import ctypes as C, os, sys
import numpy as np
ROOT = os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, ammsb_pkg
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import ops as hip
import oracle_lib as orc
orc.build()
from test_gpu_parity import Problem
K, n, nodes = [int(x) for x in sys.argv[1:4]]
N = max(4 * nodes, 20000)
pr = Problem(orc, hip, N, K, n, nodes, deg=8)
upd = hip.PhiUpdater(pr.ctx, pr.beta, pr.pi, pr.phi_sum, pr.dset, nodes, (42, 43), 64, streaming_only=True)
lib = pr.ctx.lib
lib.ammsb_debug_trace.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for rep in range(3):
    upd(pr.nodes, pr.nb, nodes)
    torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
assert lib.ammsb_debug_trace(buf, 256) == 0
t = np.array(buf[:], dtype=np.int64)
print(pr.ctx.kernel_names()["update_phi"], "nodes", nodes)
base = t[0]
names = {1: "neighbour ids staged", 2: "probes done", 3: "set-up done (loop starts)", 4: "row loop done", 5: "node stored"}
for k in (1, 2, 3):
    print("%-28s +%7d cycles" % (names[k], t[k] - base))
U = 4 if n % 4 == 0 else 2
prev = t[3]
for it in range(n // U):
    a, b = t[8 + 4 * it], t[8 + 4 * it + 1]
    print("  iter %2d: lgkm wait ends +%6d | rows landed +%6d (waited %5d) " % (it, a - prev, b - prev, b - a))
    prev = b
print("%-28s +%7d cycles (since loop start %d)" % (names[4], t[4] - base, t[4] - t[3]))
print("%-28s +%7d cycles" % (names[5], t[5] - base))
PY
