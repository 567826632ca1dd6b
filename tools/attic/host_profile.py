#!/usr/bin/env python3
"""cProfile of the Python learner's enqueue path on a host-bound configuration (development aid)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ammsb_pkg  # noqa: E402

ammsb_pkg.load()
import torch  # noqa: E402
from mcmc_ammsb_gpu_amd import hostlib  # noqa: E402
from mcmc_ammsb_gpu_amd.learner import Config, Learner  # noqa: E402

N, K, m, n = 10000, 32, 1024, 32
edges = hostlib.generate_graph(N, 32, 32, seed=20260101)
ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
cfg = Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, strategy="Node", phi_wg_size=32,
                               beta_wg_size=32, ppx_wg_size=32, device_sampling=True)
lrn = Learner(cfg, ds)
lrn.Run(50)
lrn.drain()
pr = cProfile.Profile()
pr.enable()
lrn.Run(2000)
pr.disable()
lrn.drain()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
