"""debug aid: which step's non-link mini-batch comes up short (graph loop)?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ammsb_pkg
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import hostlib
from mcmc_ammsb_gpu_amd.learner import Config, Learner
N, K, m, n = 1_000_000, 1024, 65536, 32
edges = hostlib.generate_graph(N, 64, 32, seed=20260101)
ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
cfg = Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, strategy="Node", phi_wg_size=64,
                               beta_wg_size=64, ppx_wg_size=64, device_sampling=True, graph_launch=True)
lrn = Learner(cfg, ds)
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 1
smp = lrn.dev_sampler
for it in range(0, 60, chunk):
    lrn.Run(chunk)
    torch.cuda.synchronize()
    c = smp.count.cpu().numpy()
    p = lrn.samples[lrn.phase]
    print(it, "pending", p.choice, "count", c.tolist(), flush=True)
    if c[1]:
        nodes = p.dev_nodes[:m + 1].cpu().numpy()
        print("  unique nodes in pending buffer:", np.unique(nodes).size)
        smp.count[1].zero_()
