#!/bin/bash
# In-kernel shader-clock stamps of beta_grads_lds_kernel (slot 0): where a wave's time goes.
# Build (CPU box): tools/beta_trace.sh build; run (GPU box): tools/beta_trace.sh run K edges
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p tools/ab/btrace
  for f in core phi beta ppx minibatch loop setbuild; do
    extra=""; [ $f = beta ] && extra="-DAMMSB_BETA_TRACE"
    if [ $f = beta ] || [ ! -f tools/ab/btrace/ammsb_$f.o ] || [ mcmc-ammsb-gpu_amd/csrc/ammsb_$f.hip -nt tools/ab/btrace/ammsb_$f.o ] || [ mcmc-ammsb-gpu_amd/csrc/ammsb_dev.h -nt tools/ab/btrace/ammsb_$f.o ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $extra \
        -c mcmc-ammsb-gpu_amd/csrc/ammsb_$f.hip -o tools/ab/btrace/ammsb_$f.o &
    fi
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/btrace/libammsb_hip_btrace.so tools/ab/btrace/ammsb_*.o -lpthread
  exit 0
fi
shift
AMMSB_HIP_LIB=$PWD/tools/ab/btrace/libammsb_hip_btrace.so python - "$@" <<'PY'
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
K, ne = [int(x) for x in sys.argv[1:3]]
N = int(os.environ.get("BETA_TRACE_N", max(2 * ne, 20000)))
pr = Problem(orc, hip, N, K, 8, 16, deg=8)
rng = np.random.default_rng(1)
u = int(rng.integers(0, N))
vs = rng.permutation(N)[:ne].astype(np.uint64)
mbe = orc.make_edge(np.full(ne, u, dtype=np.uint64), vs)   # a node-stratified mini-batch: one shared end point
upd = hip.BetaUpdater(pr.ctx, pr.theta, pr.beta, pr.pi, pr.dset, (44, 45), 64)
dev = pr.ctx.from_numpy(mbe)
lib = pr.ctx.lib
lib.ammsb_debug_trace_beta.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for rep in range(3):
    upd.count_calls += 1
    upd.calculate_grads(dev, mbe.size)
    torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
assert lib.ammsb_debug_trace_beta(buf, 256) == 0
t = np.array(buf[:], dtype=np.int64)
print(pr.ctx.kernel_names()["beta_grads"], "edges", ne)
print("constants (theta reciprocals)  +%6d" % (t[1] - t[0]))
print("two key windows loaded+probed  +%6d" % (t[2] - t[1]))
print("first requests issued          +%6d" % (t[3] - t[2]))
# (the one-wave, separate-update_pi form reduces two edges per step and does not stamp slot 3 = "first requests")
steps = int((t[8:248:4] > 0).sum())
prev = t[2]
for k in range(min(steps, 16)):
    a, b, c, d = t[8 + 4 * k: 12 + 4 * k]
    tail = ("| probs + four sums +%5d" % (d - c)) if d > c else ""
    print("  step %2d: since the last stamp +%5d | requests + keys +%5d | wait for rows +%5d %s" % (k, a - prev, b - a, c - b, tail))
    prev = d if d > c else c
print("steps", steps, "| whole wave", t[4] - t[0], "cycles")
# per-block record of the last launch: is the launch one round of resident waves?
if hasattr(lib, "ammsb_debug_blocks_beta"):
    lib.ammsb_debug_blocks_beta.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    nb = min(ne, 2048)
    bb = (C.c_ulonglong * (5 * nb))()
    assert lib.ammsb_debug_blocks_beta(bb, nb) == 0
    q = np.frombuffer(bb, dtype=np.uint64).reshape(nb, 5).astype(np.int64)
    c0, c1, w0, w1, hw = (q[:, i] for i in range(5))
    T0 = w0.min()
    ok = (w1 - w0) > 100
    print("blocks %d | launch span %.1f us (first block start -> last block end, wall clock) | shader clock %.0f MHz" % (
        nb, (w1.max() - T0) / 100.0, float(np.median((c1 - c0)[ok] / ((w1 - w0)[ok] / 100.0)))))
    print("block start us: p0 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile((w0 - T0) / 100.0, [0, 50, 90, 99, 100])))
    print("block life  us: p0 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f | cycles p50 %d" % (tuple(np.percentile((w1 - w0) / 100.0, [0, 50, 90, 99, 100])) + (int(np.median(c1 - c0)),)))
    xcc = (hw >> 32) & 0xf
    hwid = hw & 0xffffffff
    cukey = xcc * 1000 + ((hwid >> 13) & 7) * 100 + ((hwid >> 12) & 1) * 50 + ((hwid >> 8) & 0xf)
    per_cu = np.bincount(np.unique(cukey, return_inverse=True)[1])
    print("distinct CUs %d | blocks per CU: min %d median %d max %d" % (per_cu.size, per_cu.min(), np.median(per_cu), per_cu.max()))
    edges_t = np.arange(0, (w1.max() - T0) + 200, 200)
    print("resident blocks every 2 us:", [int(((w0 - T0) <= e).sum() - ((w1 - T0) <= e).sum()) for e in edges_t])
PY
