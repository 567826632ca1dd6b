"""microbenchmark of the neighbour sampler kernel: time vs nodes and vs draws per node (development aid)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ammsb_pkg
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import ops
N = 1_000_000
for n in (8, 16, 32):
    ctx = ops.Context(ops.make_params(N, 64, E=16 * N, num_node_sample=n))
    for nn in (33, 1025, 8193, 65537):
        nodes = ctx.from_numpy(np.random.default_rng(1).permutation(N)[:nn].astype(np.uint32))
        smp = ops.NeighborSampler(ctx, nn, (56, 57), 32)
        for _ in range(5):
            smp(nn, nodes)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        R = 50
        a.record()
        for _ in range(R):
            smp(nn, nodes)
        b.record()
        torch.cuda.synchronize()
        print("n=%2d nodes=%6d: %.2f us per launch" % (n, nn, a.elapsed_time(b) * 1e3 / R), flush=True)
    ctx.close()
