"""update_phi (C3 shape) over pi allocated the default way (torch -> hipMalloc) and as PHYSICALLY CONTIGUOUS device
memory (hipExtMallocWithFlags(hipDeviceMallocContiguous)), candidates interleaved, one process.
Usage: python tools/placement_contig.py [per_kind] [rounds]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.build()
import ammsb_pkg  # noqa: E402

ammsb_pkg.load()
import bench  # noqa: E402
from mcmc_ammsb_gpu_amd import hostlib, ops  # noqa: E402
from mcmc_ammsb_gpu_amd.learner import Config, Learner  # noqa: E402

per = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]


class Raw:
    def __init__(self, rows, cols, flags):
        self.ptr = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(self.ptr), rows * cols * 4, flags)
        if rc != 0:
            raise RuntimeError("hipExtMallocWithFlags(flags=%d) -> %d" % (flags, rc))
        self.__cuda_array_interface__ = {"shape": (rows, cols), "typestr": "<f4", "data": (self.ptr.value, False), "version": 2}

    def free(self):
        hip.hipFree(self.ptr)


N, K, m, n, deg, k_true = bench.WORKLOADS["C3"]
edges = hostlib.generate_graph(N, k_true, deg, seed=20260101)
ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
os.environ["AMMSB_PI_CANDIDATES"] = "0"
cfg = Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, strategy="NodeNonLink", phi_wg_size=64,
                               beta_wg_size=64, ppx_wg_size=64, device_sampling=True, graph_launch=False)
lrn = Learner(cfg, ds)
lrn.Run(3)
lrn.drain()
phi = lrn.phiUpdater
s = lrn.samples[lrn.phase]
lrn.futures[lrn.phase].result()
torch.cuda.synchronize()
pis, names, raws = [lrn.pi], ["learner's pi (torch)"], []
for c in range(per):
    p = ops.RowPartitionedMatrix(lrn.ctx, N, K)
    p.blocks[0].copy_(lrn.pi.blocks[0])
    pis.append(p)
    names.append("torch %d" % c)
    raw = Raw(N, K, 0x4)
    raws.append(raw)
    t = torch.as_tensor(raw, device="cuda")
    assert t.data_ptr() == raw.ptr.value
    t.copy_(lrn.pi.blocks[0])
    q = ops.RowPartitionedMatrix(lrn.ctx, 1, K)  # a shell: its storage is replaced by the raw allocation
    q.rows, q.rows_in_block = N, N
    q.blocks = [t]
    q.desc.blocks[0] = t.data_ptr()
    q.desc.rows_in_block = N
    q.desc.num_rows = N
    pis.append(q)
    names.append("contiguous %d" % c)
print("pi candidates at", [hex(p.blocks[0].data_ptr()) for p in pis], flush=True)
keep = phi.rand.seeds.clone()
orig = phi.pi


def launch():
    phi.update_phi(s.dev_nodes, s.neighbor_sampler.GetData(), s.n_nodes)


t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    for p in pis:
        phi.pi = p
        launch()
    torch.cuda.synchronize()
ms = [[] for _ in pis]
for r in range(rounds):
    for c, p in enumerate(pis):
        phi.pi = p
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(torch.cuda.current_stream())
        for _ in range(3):
            launch()
        e.record(torch.cuda.current_stream())
        e.synchronize()
        ms[c].append(a.elapsed_time(e) / 3)
phi.pi = orig
phi.rand.seeds.copy_(keep)
for c in range(len(pis)):
    v = sorted(ms[c])
    print("%-22s median %.4f ms  min %.4f  max %.4f" % (names[c], float(np.median(v)), v[0], v[-1]), flush=True)
lrn.close()
del pis
for r in raws:
    r.free()
