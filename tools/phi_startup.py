"""Is update_phi's slow start a matter of LOAD (seconds of work) or of TIME since the process created its device
context?  C3, descriptor loop.  Windows of 5 + 20 steps (mean update_phi of the non-link steps, device stamps), with
either idle time or load between them.  Usage: python tools/phi_startup.py idle-first|load-first"""
import json
import os
import sys
import time

import numpy as np

T_PROC = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.build()
import ammsb_pkg  # noqa: E402

ammsb_pkg.load()
import bench  # noqa: E402
from mcmc_ammsb_gpu_amd import gpu_state, hostlib  # noqa: E402
from mcmc_ammsb_gpu_amd.learner import Config, Learner  # noqa: E402

order = sys.argv[1] if len(sys.argv) > 1 else "idle-first"
N, K, m, n, deg, k_true = bench.WORKLOADS["C3"]
edges = hostlib.generate_graph(N, k_true, deg, seed=20260101)
ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
cfg = Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, strategy="Node", phi_wg_size=64, beta_wg_size=64,
                               ppx_wg_size=64, device_sampling=True, graph_launch=True, graph_timestamps=True)
torch.cuda.init()
t_ctx = time.perf_counter()
lrn = Learner(cfg, ds)
dev = torch.cuda.current_device()


def window(tag):
    lrn.Run(5)
    lrn.drain()
    first = lrn.phiUpdater.count_calls + 1
    lrn.step_log = []
    lrn.Run(20)
    lrn.drain()
    st = lrn.loop.step_stamps(first, 20)
    ne = np.concatenate(lrn.step_log)
    lrn.step_log = None
    non = ne == m
    g = gpu_state.read(dev)
    print("%-26s t_ctx %5.1f s | update_phi %.3f ms (%d launches, min %.3f max %.3f) | power %s W sclk %s" % (
        tag, time.perf_counter() - t_ctx, (st[non, 1] - st[non, 0]).mean() * 1e-6, int(non.sum()),
        (st[non, 1] - st[non, 0]).min() * 1e-6, (st[non, 1] - st[non, 0]).max() * 1e-6, g.get("power_w"), g.get("sclk_mhz")),
        flush=True)


def load(seconds):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        lrn.Run(100)
        lrn.drain()


print(order, "learner ready %.1f s after the context" % (time.perf_counter() - t_ctx), flush=True)
window("first window")
lrn.HeldoutPerplexity()
window("after a perplexity call")
if order == "idle-first":
    time.sleep(8)
    window("after 8 s idle")
    load(4)
    window("after 4 s load")
else:
    load(4)
    window("after 4 s load")
    time.sleep(8)
    window("after 8 s idle")
load(4)
window("after 4 more s load")
time.sleep(20)
window("after 20 s idle")
lrn.close()
