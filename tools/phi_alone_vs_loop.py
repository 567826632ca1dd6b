"""Does the sampling chain that runs beside update_phi slow update_phi down?  ("the only evidence that
they do not slow update_phi itself is indirect".)  One process (so one of the two per-process launch speeds, DESIGN.md
5), the C3 learner: (a) update_phi's duration inside the descriptor loop, where the sampler chain of the mini-batch
two steps ahead runs concurrently on the second stream (in-kernel stamps, non-link steps); (b) the same kernel on
the same kind of mini-batch launched ALONE, nothing else on the device (HIP events), alternating a few times.

    python tools/phi_alone_vs_loop.py [workload]        (GPU box)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402
import ammsb_pkg  # noqa: E402
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import hostlib  # noqa: E402
from mcmc_ammsb_gpu_amd.learner import Config, Learner  # noqa: E402
import bench  # noqa: E402

W = sys.argv[1] if len(sys.argv) > 1 else "C3"
N, K, m, n, deg, k_true = bench.WORKLOADS[W]
edges = hostlib.generate_graph(N, k_true, deg, seed=20260101)
ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
wg = bench.pick_wg(K, 0, 16)
cfg = Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, strategy="Node", phi_wg_size=wg,
                               beta_wg_size=wg, ppx_wg_size=wg, device_sampling=True, graph_launch=True,
                               graph_timestamps=True)
lrn = Learner(cfg, ds)
lrn.Run(10)
lrn.drain()
for rnd in range(3):
    # (a) inside the loop
    steps = 60 if W == "C3" else 600
    lrn.step_log = []
    first = lrn.phiUpdater.count_calls + 1
    lrn.Run(steps)
    lrn.drain()
    st = lrn.loop.step_stamps(first, steps)
    ne = np.concatenate([r for r in lrn.step_log])
    lrn.step_log = None
    non = ne == m
    in_loop = (st[non, 1] - st[non, 0]) * 1e-6
    # (b) alone: the pending mini-batch if it is a non-link one, else draw until one is
    phi = lrn.phiUpdater
    s = lrn.samples[lrn.phase]
    tries = 0
    while s.n_edges != m and tries < 20:
        lrn.Run(1)
        lrn.drain()
        s = lrn.samples[lrn.phase]
        tries += 1
    keep = phi.rand.seeds.clone()
    calls = phi.count_calls
    alone = []
    for _ in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        phi.update_phi(s.dev_nodes, s.neighbor_sampler.GetData(), s.n_nodes)
        b.record()
        torch.cuda.synchronize()
        alone.append(a.elapsed_time(b))
    phi.rand.seeds.copy_(keep)
    phi.count_calls = calls
    torch.cuda.synchronize()
    print("%s round %d: update_phi in the loop (sampler chain beside it) %.4f ms mean / %.4f median over %d launches | "
          "alone %.4f ms mean / %.4f median / %.4f min over %d launches" % (
              W, rnd, in_loop.mean(), np.median(in_loop), in_loop.size, np.mean(alone[2:]), np.median(alone[2:]),
              min(alone), len(alone) - 2), flush=True)
lrn.close()
