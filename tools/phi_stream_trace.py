"""stamps of update_phi_stream_kernel (block 0, its first three nodes); needs the -DAMMSB_PHI_TRACE build (tools/phi_trace.sh build)"""
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
K, n, nodes = 256, 32, 8193
pr = Problem(orc, hip, 100000, K, n, nodes, deg=8)
upd = hip.PhiUpdater(pr.ctx, pr.beta, pr.pi, pr.phi_sum, pr.dset, nodes, (42, 43), 64, streaming_only=True)
lib = pr.ctx.lib
lib.ammsb_debug_trace.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for rep in range(3):
    upd(pr.nodes, pr.nb, nodes)
    torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
assert lib.ammsb_debug_trace(buf, 256) == 0
t = np.array(buf[:], dtype=np.int64)
print(pr.ctx.kernel_names()["update_phi"])
print("first prologue (tables, beta, ids, probes, first request): +%d" % (t[1] - t[0]))
NIT = n // 4
end_prev = t[1]
for k in range(3):
    b = 8 + 32 * k
    if t[b] == 0:
        break
    print("node %d:" % k)
    for it in range(NIT):
        a0, a1 = t[b + 3 * it], t[b + 3 * it + 1]
        nxt = t[b + 3 * (it + 1)] if it + 1 < NIT else t[b + 3 * NIT]
        print("   step %d: to lgkm %5d | request+draw+wait %5d | stages+reduce %5d" % (it, a0 - end_prev, a1 - a0, nxt - a1))
        end_prev = nxt
    print("   node total %d" % (t[b + 3 * NIT] - (t[1] if k == 0 else t[8 + 32 * (k - 1) + 3 * NIT])))
