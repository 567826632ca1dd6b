"""development aid: beta gradient kernel time vs number of partial-row slots (AMMSB_BETA_SLOTS) at the C3 shape"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ammsb_pkg
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import hostlib, ops
N, K, m = 1_000_000, 1024, 65536
ctx = ops.Context(ops.make_params(N, K, E=16 * N, num_node_sample=32))
pi = ops.RowPartitionedMatrix(ctx, N, K)
phi = ctx.zeros((N,), torch.float32)
ops.RandomGammaAndNormalize(ctx, 1.0, 1.0, pi, phi)
theta = ctx.from_numpy(hostlib.theta_init(K))
beta = ctx.zeros((2 * K,), torch.float32)
ops.beta_from_theta(ctx, theta, beta)
rng = np.random.default_rng(1)
e = np.unique((rng.integers(0, N, 300000, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, N, 300000, dtype=np.uint64))
hs = hostlib.HostSet(e)
dset = ops.DeviceSet(ctx, hs.Serialize(), hs.BinsPerBucket(), hs.PrimeIdx())
u = np.uint64(12345)
v = rng.permutation(N)[:m].astype(np.uint64)
mb = ctx.from_numpy((np.minimum(u, v) << np.uint64(32)) | np.maximum(u, v))
bu = ops.BetaUpdater(ctx, theta, beta, pi, dset, (44, 45), 64)
for _ in range(5):
    bu.calculate_grads(mb, m)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    bu.calculate_grads(mb, m)
b.record()
torch.cuda.synchronize()
print("AMMSB_BETA_SLOTS=%s: %.1f us per gradient (rows + sum)" % (os.environ.get("AMMSB_BETA_SLOTS", "default"), a.elapsed_time(b) * 20))
