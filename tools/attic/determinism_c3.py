#!/usr/bin/env python3
"""Run the C3 learner several times from the same seeds and compare the final state bit for bit (development aid:
a cross-stream race in the loop would show up here as a run-to-run difference)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ammsb_pkg  # noqa: E402

ammsb_pkg.load()
import torch  # noqa: E402
from mcmc_ammsb_gpu_amd import hostlib, ops  # noqa: E402
from mcmc_ammsb_gpu_amd.learner import Config, Learner  # noqa: E402

N, K, m, n = 1_000_000, 1024, 65536, 32
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
edges = hostlib.generate_graph(N, 64, 32, seed=20260101)
ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
ref = None
for r in range(runs):
    cfg = Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, strategy="Node", phi_wg_size=64,
                                   beta_wg_size=64, ppx_wg_size=64, device_sampling=True)
    lrn = Learner(cfg, ds)
    lrn.Run(iters)
    ppx = lrn.HeldoutPerplexity()
    theta = ops.to_numpy(lrn.theta).copy()
    chk = int(torch.sum(lrn.pi.blocks[0].view(torch.int32).to(torch.int64)).item())   # order-free checksum of pi's bits
    seeds = int(torch.sum(lrn.phiUpdater.rand.seeds).item())
    lrn.close()
    sig = (ppx, theta.tobytes(), chk, seeds, lrn.edges_done)
    print("run %d: ppx %.9f  pi checksum %d  edges %d" % (r, ppx, chk, lrn.edges_done), flush=True)
    if ref is None:
        ref = sig
    elif sig != ref:
        print("MISMATCH in run", r)
        sys.exit(1)
print("identical")
