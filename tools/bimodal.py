"""development aid: is update_phi's two-speed behaviour at C3 (1.62 vs 1.70 ms per launch, per process) tied to where
pi lands in the address space?  Prints the block pointers and the per-launch time of one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ammsb_pkg
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import ops

N, K, n, m = 1_000_000, 1024, 32, 65536
pad = int(os.environ.get("PAD_MB", "0"))
junk = torch.empty(pad << 20, dtype=torch.uint8, device="cuda") if pad else None
p = ops.make_params(N, K, E=16 * N, num_node_sample=n)
ctx = ops.Context(p)
pi = ops.RowPartitionedMatrix(ctx, N, K)
phi_sum = ctx.zeros((N,), torch.float32)
ops.RandomGammaAndNormalize(ctx, 1.0, 1.0, pi, phi_sum)
rng = np.random.default_rng(0)
theta = ctx.from_numpy(rng.gamma(1.0, 1.0, 2 * K).astype(np.float32))
beta = ctx.zeros((2 * K,), torch.float32)
ops.beta_from_theta(ctx, theta, beta)
from mcmc_ammsb_gpu_amd import hostlib
u = rng.integers(0, N, 200000, dtype=np.uint64); v = rng.integers(0, N, 200000, dtype=np.uint64)
e = np.unique((np.minimum(u, v) << np.uint64(32)) | np.maximum(u, v))
hs = hostlib.HostSet(e)
dset = ops.DeviceSet(ctx, hs.Serialize(), hs.BinsPerBucket(), hs.PrimeIdx())
nn = m + 1
nodes = ctx.from_numpy(rng.permutation(N)[:nn].astype(np.uint32))
nbrs = ctx.from_numpy(rng.integers(0, N, size=(nn, n), dtype=np.uint32))
upd = ops.PhiUpdater(ctx, beta, pi, phi_sum, dset, nn, (42, 43), 64)
upd.count_calls = 1
for _ in range(3):
    upd.update_phi(nodes, nbrs, nn)
torch.cuda.synchronize()
ts = []
for _ in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); upd.update_phi(nodes, nbrs, nn); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ptrs = [hex(bk.data_ptr()) for bk in pi.blocks]
print("pad %d MB  pi blocks %s  phi_vec %s  update_phi median %.4f ms (min %.4f)" % (
    pad, ptrs, hex(upd.phi_vec.data_ptr()), float(np.median(ts)), min(ts)), flush=True)
