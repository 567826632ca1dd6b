"""The K = 1024 update_phi kernel forms against each other in ONE process over ONE allocation of pi (C3 mini-batch,
round-robin, 3 launches per turn): default two-slot ring, three-slot ring (lds3), two nodes per block (nb 2), ring 4.
Across processes such a comparison measures where pi landed, not the kernels (DESIGN_HISTORY.md R4.7).
Usage: python tools/phi_forms_ab.py [rounds]"""
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
from mcmc_ammsb_gpu_amd import hostlib  # noqa: E402
from mcmc_ammsb_gpu_amd.learner import Config, Learner  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N, K, m, n, deg, k_true = bench.WORKLOADS["C3"]
edges = hostlib.generate_graph(N, k_true, deg, seed=20260101)
ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
cfg = Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, strategy="NodeNonLink", phi_wg_size=64,
                               beta_wg_size=64, ppx_wg_size=64, device_sampling=True, graph_launch=False)
lrn = Learner(cfg, ds)
print("placement", lrn.pi_placement, flush=True)
lrn.Run(3)
lrn.drain()
phi = lrn.phiUpdater
s = lrn.samples[lrn.phase]
lrn.futures[lrn.phase].result()
torch.cuda.synchronize()
lib = lrn.ctx.lib
lib.ammsb_debug_phi_forms.argtypes = [C.c_int, C.c_int, C.c_int]
forms = [("default (two slots)", (0, 1, 0)), ("three slots (lds3)", (1, 1, 0)), ("two nodes per block", (0, 2, 0)),
         ("ring 4", (0, 1, 4))]
keep = phi.rand.seeds.clone()
want = None


def launch():
    phi.update_phi(s.dev_nodes, s.neighbor_sampler.GetData(), s.n_nodes)


# same bits from every form (same seeds, same call counter)
for name, f in forms:
    lib.ammsb_debug_phi_forms(*f)
    phi.rand.seeds.copy_(keep)
    launch()
    torch.cuda.synchronize()
    got = phi.phi_vec[:s.n_nodes].clone()
    if want is None:
        want = got
    print("%-22s %s  kernel %s" % (name, "bit-identical" if torch.equal(got, want) else "DIFFERS", lrn.ctx.kernel_names()["update_phi"]), flush=True)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    for name, f in forms:
        lib.ammsb_debug_phi_forms(*f)
        launch()
    torch.cuda.synchronize()
ms = [[] for _ in forms]
for r in range(rounds):
    for c, (name, f) in enumerate(forms):
        lib.ammsb_debug_phi_forms(*f)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(torch.cuda.current_stream())
        for _ in range(3):
            launch()
        e.record(torch.cuda.current_stream())
        e.synchronize()
        ms[c].append(a.elapsed_time(e) / 3)
lib.ammsb_debug_phi_forms(-1, -1, -1)
phi.rand.seeds.copy_(keep)
for c, (name, f) in enumerate(forms):
    v = sorted(ms[c])
    print("%-22s median %.4f ms  min %.4f  max %.4f" % (name, float(np.median(v)), v[0], v[-1]), flush=True)
lrn.close()
