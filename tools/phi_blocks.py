"""Per-block occupancy record of update_phi_lds2_kernel at C2's shape (needs the -DAMMSB_PHI_TRACE build:
tools/phi_trace.sh build; run with AMMSB_HIP_LIB=tools/ab/trace/libammsb_hip_trace.so).  Prints the shader clock rate,
how many blocks were resident over time, on how many distinct (XCC, SE, CU) and the per-block duration distribution."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, ammsb_pkg
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import ops as hip
import oracle_lib as orc
orc.build()
from test_gpu_parity import Problem
K, n, nodes = 256, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 8193
wg = int(sys.argv[2]) if len(sys.argv) > 2 else 64
pr = Problem(orc, hip, 100000, K, n, nodes, deg=8)
upd = hip.PhiUpdater(pr.ctx, pr.beta, pr.pi, pr.phi_sum, pr.dset, nodes, (42, 43), wg, streaming_only=True)
lib = pr.ctx.lib
lib.ammsb_debug_blocks.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for rep in range(4):
    upd(pr.nodes, pr.nb, nodes)
    torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
# the stamps of the LAST launch; update_pi between launches keeps pi changing as in the loop
nb = min(nodes, 16384)
buf = (C.c_ulonglong * (5 * nb))()
assert lib.ammsb_debug_blocks(buf, nb) == 0
t = np.array(buf[:], dtype=np.uint64).reshape(nb, 5)
print(pr.ctx.kernel_names()["update_phi"], "blocks", nb)
c0, c1, w0, w1, hw = (t[:, i].astype(np.int64) for i in range(5))
# (the shader-clock counters of different XCDs are not synchronised: rates are taken per block, spans on the wall clock)
span_w = (w1.max() - w0.min()) / 100.0  # us (100 MHz)
ok = (w1 - w0) > 200
print("kernel span %.1f us; shader clock %.2f GHz (median over blocks of cycles / wall time)" % (
    span_w, float(np.median((c1 - c0)[ok] / ((w1 - w0)[ok] / 100.0))) / 1e3))
dur = c1 - c0
print("block duration cycles: min %d p10 %d median %d p90 %d max %d" % (dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max()))
hwid = hw & 0xffffffff
xcc = (hw >> 32) & 0xf
cu = (hwid >> 8) & 0xf
sh = (hwid >> 12) & 0x1
se = (hwid >> 13) & 0x7
simd = (hwid >> 4) & 0x3
wave = hwid & 0xf
cukey = xcc * 1000 + se * 100 + sh * 50 + cu
print("distinct CUs used: %d, distinct (CU, SIMD, wave slot): %d" % (len(np.unique(cukey)), len(np.unique(cukey * 100 + simd * 16 + wave))))
# resident blocks over time (wall clock, 0.5 us bins)
T0 = w0.min()
edges = np.arange(0, (w1.max() - T0) + 50, 50)
res = [(int(((w0 - T0) <= e) .sum() - ((w1 - T0) <= e).sum())) for e in edges]
print("resident blocks every 0.5 us:", res)
starts = np.sort(w0 - T0) / 100.0
print("block start times us: first %.2f  1024th %.2f  2048th %.2f  4096th %.2f  last %.2f" % tuple(starts[[0, min(1023, nb - 1), min(2047, nb - 1), min(4095, nb - 1), nb - 1]]))
per_cu = np.array([((cukey == k) & ((w0 - T0) <= 1000) & ((w1 - T0) > 1000)).sum() for k in np.unique(cukey)])
print("resident blocks per CU at t = 10 us: min %d median %d max %d" % (per_cu.min(), np.median(per_cu), per_cu.max()))
