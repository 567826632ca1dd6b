#!/usr/bin/env python3
"""Kernel micro-benchmark on synthetic C3-shaped state (no graph build): times update_phi / update_pi /
beta_grads / perplexity for several work-group sizes with HIP events.  Development tool."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ammsb_pkg  # noqa: E402

ammsb_pkg.load()
import torch  # noqa: E402
from mcmc_ammsb_gpu_amd import hostlib, ops  # noqa: E402


BATCH = 1  # --batch: launches between the two events (small kernels: the per-launch time in a back-to-back stream)


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(BATCH):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / BATCH)
    return float(np.median(ts)), float(np.min(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=1_000_000)
    ap.add_argument("--K", type=int, default=1024)
    ap.add_argument("--m", type=int, default=65536)
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--set-edges", type=int, default=2_000_000)
    ap.add_argument("--phi-wgs", default="64,128,256")
    ap.add_argument("--beta-wgs", default="64,128,256")
    ap.add_argument("--ppx-wgs", default="64,128,256")
    ap.add_argument("--noise", type=int, default=1)
    ap.add_argument("--only", default="phi,pi,beta,ppx")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--beta-random", type=int, default=0, help="1: both end points random (no shared row)")
    args = ap.parse_args()
    global BATCH
    BATCH = max(1, args.batch)
    N, K, m, n = args.N, args.K, args.m, args.n
    rng = np.random.default_rng(0)
    p = ops.make_params(N, K, E=16 * N, num_node_sample=n)
    ctx = ops.Context(p)
    pi = ops.RowPartitionedMatrix(ctx, N, K)
    phi_sum = ctx.zeros((N,), torch.float32)
    ops.RandomGammaAndNormalize(ctx, 1.0, 1.0, pi, phi_sum)
    theta = ctx.from_numpy(rng.gamma(1.0, 1.0, 2 * K).astype(np.float32))
    beta = ctx.zeros((2 * K,), torch.float32)
    ops.beta_from_theta(ctx, theta, beta)
    u = rng.integers(0, N, args.set_edges, dtype=np.uint64)
    v = rng.integers(0, N, args.set_edges, dtype=np.uint64)
    e = np.unique((np.minimum(u, v) << np.uint64(32)) | np.maximum(u, v))
    hs = hostlib.HostSet(e)
    dset = ops.DeviceSet(ctx, hs.Serialize(), hs.BinsPerBucket(), hs.PrimeIdx())
    nn = m + 1
    nodes = ctx.from_numpy(rng.permutation(N)[:nn].astype(np.uint32))
    nbrs = ctx.from_numpy(rng.integers(0, N, size=(nn, n), dtype=np.uint32))
    only = set(args.only.split(","))
    per_node = 4 * K * (n + 2) + 68 * n + 8
    if "phi" in only:
        for wg in [int(x) for x in args.phi_wgs.split(",")]:
            upd = ops.PhiUpdater(ctx, beta, pi, phi_sum, dset, nn, (42, 43), wg, phi_disable_noise=not args.noise)
            upd.count_calls = 1
            med, mn = timeit(lambda: upd.update_phi(nodes, nbrs, nn))
            print("update_phi wg=%4d: median %.3f ms (min %.3f)  %.0f GB/s  %.1f%% of 8 TB/s" %
                  (wg, med, mn, per_node * nn / med / 1e6, per_node * nn / med / 1e6 / 80), flush=True)
            if "pi" in only:
                med, mn = timeit(lambda: upd.update_pi(nodes, nn))
                print("update_pi  wg=%4d: median %.3f ms (min %.3f)  %.0f GB/s" %
                      (wg, med, mn, 8 * K * nn / med / 1e6), flush=True)
            del upd
    mb = (np.uint64(12345) << np.uint64(32)) | rng.permutation(N)[:m].astype(np.uint64)
    if args.beta_random:
        mb = (rng.integers(0, N, m, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, N, m, dtype=np.uint64)
    dev_edges = ctx.from_numpy(mb)
    if "beta" in only:
        for wg in [int(x) for x in args.beta_wgs.split(",")]:
            bu = ops.BetaUpdater(ctx, theta, beta, pi, dset, (44, 45), wg)
            med, mn = timeit(lambda: bu.calculate_grads(dev_edges, m))
            print("beta_grads wg=%4d: median %.3f ms (min %.3f)  %.0f GB/s (one row per edge + shared row)" %
                  (wg, med, mn, (4 * K + 72) * m / med / 1e6), flush=True)
    H = 160_000
    he = ctx.from_numpy(((rng.integers(0, N, H, dtype=np.uint64)) << np.uint64(32)) | rng.integers(0, N, H, dtype=np.uint64))
    if "ppx" in only:
        for wg in [int(x) for x in args.ppx_wgs.split(",")]:
            calc = ops.PerplexityCalculator(ctx, beta, pi, he, dset, wg)
            calc.count_calls = 1
            med, mn = timeit(lambda: calc.partial())
            print("perplexity wg=%4d: median %.3f ms (min %.3f)  %.0f GB/s" %
                  (wg, med, mn, (8 * K + 88) * H / med / 1e6), flush=True)


if __name__ == "__main__":
    main()
