"""HeldoutPerplexity() call cost at C3's shape: kernel (HIP events) vs the whole call, result in pinned host memory
(one launch + one stream synchronisation) vs in device memory + a device-to-host copy; run once per AMMSB_PPX_FOLD
setting (the library reads it once per process)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import ammsb_pkg
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import hostlib, ops

N, K, H = 1_000_000, 1024, 163_148
rng = np.random.default_rng(0)
p = ops.make_params(N, K, E=16 * N, num_node_sample=32)
ctx = ops.Context(p)
pi = ops.RowPartitionedMatrix(ctx, N, K)
phi_sum = ctx.zeros((N,), torch.float32)
ops.RandomGammaAndNormalize(ctx, 1.0, 1.0, pi, phi_sum)
theta = ctx.from_numpy(rng.gamma(1.0, 1.0, 2 * K).astype(np.float32))
beta = ctx.zeros((2 * K,), torch.float32)
ops.beta_from_theta(ctx, theta, beta)
u = rng.integers(0, N, H, dtype=np.uint64); v = rng.integers(0, N, H, dtype=np.uint64)
e = np.unique((np.minimum(u, v) << np.uint64(32)) | np.maximum(u, v))
hs = hostlib.HostSet(e[: e.size // 2])
dset = ops.DeviceSet(ctx, hs.Serialize(), hs.BinsPerBucket(), hs.PrimeIdx())
calc = ops.PerplexityCalculator(ctx, beta, pi, ctx.from_numpy(e), dset, 64)
for _ in range(5):
    calc()
torch.cuda.synchronize()

def timed(fn, reps=30):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), float(np.min(ts))

def host_path():
    calc.count_calls += 1
    return calc.partial_host()

def dev_path():
    calc.count_calls += 1
    return calc.unpack(calc.partial())

def kernel_only():
    calc.count_calls += 1
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); calc.partial(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)

ks = [kernel_only() for _ in range(30)]
print("AMMSB_PPX_FOLD=%s kernel %s" % (os.environ.get("AMMSB_PPX_FOLD", "default(1)"), ctx.kernel_names()["perplexity"]))
print("  device time of the launch(es), HIP events: median %.4f ms min %.4f" % (np.median(ks), np.min(ks)))
print("  call, sums in pinned host memory:          median %.4f ms min %.4f" % timed(host_path))
print("  call, sums in device memory + .cpu():      median %.4f ms min %.4f" % timed(dev_path))
r1, r2 = host_path(), dev_path()
print("  agree:", r1[2:] == r2[2:], abs(r1[0] - r2[0]) / abs(r1[0]))
