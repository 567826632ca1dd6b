"""The real update_phi (C3 shape, 65 537 nodes x 32 neighbours, K = 1024) over several candidate allocations of pi in
ONE process: the learner's own pi and CANDS copies of it allocated afterwards, timed round-robin (HIP events, 3 launches
per turn).  If the launch time differs by candidate, where pi lands in HBM matters and a start-up tournament can pick.
Usage: python tools/placement_phi.py [cands] [rounds]"""
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

cands = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
N, K, m, n, deg, k_true = bench.WORKLOADS["C3"]
edges = hostlib.generate_graph(N, k_true, deg, seed=20260101)
ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
cfg = Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, strategy="NodeNonLink", phi_wg_size=64,
                               beta_wg_size=64, ppx_wg_size=64, device_sampling=True, graph_launch=False)
lrn = Learner(cfg, ds)
lrn.Run(3)
lrn.drain()
phi = lrn.phiUpdater
s = lrn.samples[lrn.phase]
lrn.futures[lrn.phase].result()
torch.cuda.synchronize()
pis = [lrn.pi]
spacers = []
for c in range(cands):
    if c % 2 == 1:
        spacers.append(torch.empty(((c + 3) * 91_000_003,), dtype=torch.uint8, device="cuda"))
    p = ops.RowPartitionedMatrix(lrn.ctx, N, K)
    p.blocks[0].copy_(lrn.pi.blocks[0])
    pis.append(p)
print("pi candidates at", [hex(p.blocks[0].data_ptr()) for p in pis], flush=True)
keep = phi.rand.seeds.clone()
orig = phi.pi


def launch():
    phi.update_phi(s.dev_nodes, s.neighbor_sampler.GetData(), s.n_nodes)


t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.0:  # warm clocks
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
    print("%s: median %.4f ms  min %.4f  max %.4f" % ("learner's pi " if c == 0 else "candidate %d  " % c, float(np.median(v)), v[0], v[-1]), flush=True)
lrn.close()
