"""update_phi (C3 or the workload named third, descriptor loop) with and without a spacer allocation made BEFORE the learner's buffers: does where
pi lands in HBM move the launch time?  One process per setting (tools/attic/r04_place2.sh alternates them on one box).
Usage: python tools/placement_learner.py SPACER_GB [hold|free] [workload]"""
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

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
mode = sys.argv[2] if len(sys.argv) > 2 else "hold"
wl = sys.argv[3] if len(sys.argv) > 3 else "C3"
N, K, m, n, deg, k_true = bench.WORKLOADS[wl]
edges = hostlib.generate_graph(N, k_true, deg, seed=20260101)
ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
spacer = None
if gb > 0:
    spacer = torch.empty((int(gb * (1 << 30)),), dtype=torch.uint8, device="cuda")
    spacer.zero_()
    torch.cuda.synchronize()
    if mode == "free":
        del spacer
        spacer = None
        torch.cuda.empty_cache()
wg = bench.pick_wg(K, 0, 16)
cfg = Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, strategy="Node", phi_wg_size=wg, beta_wg_size=wg,
                               ppx_wg_size=wg, device_sampling=True, graph_launch=True, graph_timestamps=True)
lrn = Learner(cfg, ds)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.5:
    lrn.Run(100)
    lrn.drain()
res = []
for w in range(4):
    first = lrn.phiUpdater.count_calls + 1
    lrn.step_log = []
    lrn.Run(60)
    lrn.drain()
    st = lrn.loop.step_stamps(first, 60)
    non = np.concatenate(lrn.step_log) == m
    res.append((st[non, 1] - st[non, 0]).mean() * 1e-6)
lrn.step_log = None
print(wl, "spacer %4.1f GB (%s): update_phi %s ms | pi blocks at %s | placement %s" % (
    gb, mode, " ".join("%.3f" % r for r in res), hex(lrn.pi.blocks[0].data_ptr()) if hasattr(lrn.pi, "blocks") else "?",
    lrn.pi_placement), flush=True)
lrn.close()
