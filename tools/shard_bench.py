#!/usr/bin/env python3
"""What one rank of an R-GPU job launches per non-link iteration for update_phi, timed on one GPU (development aid):
its own blocks chunk by chunk on the main stream plus the replicated groups on a side stream, against the full
launch.  No exchange is performed -- this isolates the compute side of learner._phi_sharded."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ammsb_pkg  # noqa: E402

ammsb_pkg.load()
import torch  # noqa: E402
from mcmc_ammsb_gpu_amd import hostlib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=int, default=8)
    ap.add_argument("--rho", type=float, default=0.117)
    ap.add_argument("--chunks", type=int, default=2)
    ap.add_argument("--side", type=int, default=1)
    a = ap.parse_args()
    N, K, n, nn = 1_000_000, 1024, 32, 65537
    rng = np.random.default_rng(0)
    ctx = ops.Context(ops.make_params(N, K, E=16 * N, num_node_sample=n))
    pi = ops.RowPartitionedMatrix(ctx, N, K)
    phi_sum = ctx.zeros((N,), torch.float32)
    ops.RandomGammaAndNormalize(ctx, 1.0, 1.0, pi, phi_sum)
    theta = ctx.from_numpy(hostlib.theta_init(K))
    beta = ctx.zeros((2 * K,), torch.float32)
    ops.beta_from_theta(ctx, theta, beta)
    e = np.unique(rng.integers(0, 2**40, 100000, dtype=np.uint64))
    hs = hostlib.HostSet(e)
    dset = ops.DeviceSet(ctx, hs.Serialize(), hs.BinsPerBucket(), hs.PrimeIdx())
    nodes = ctx.from_numpy(rng.permutation(N)[:nn].astype(np.uint32))
    nbrs = ctx.from_numpy(rng.integers(0, N, (nn, n), dtype=np.uint32))
    upd = ops.PhiUpdater(ctx, beta, pi, phi_sum, dset, nn + 64, (42, 43), 64)
    upd.count_calls = 1
    G = 65535
    g0 = int(a.rho * G)
    cc = (G - g0 + a.R * a.chunks - 1) // (a.R * a.chunks)
    side = torch.cuda.Stream()
    fork, join = torch.cuda.Event(), torch.cuda.Event()

    def full():
        upd.update_phi(nodes, nbrs, nn)

    def rank0():
        if a.side:
            fork.record()
            with ops.stream(side):
                fork.wait()
                upd.update_phi(nodes, nbrs, nn, 2, g0)
                join.record()
        for c in range(a.chunks):
            lo = g0 + c * a.R * cc
            upd.update_phi(nodes, nbrs, nn, lo, min(lo + cc, G))
        if a.side:
            join.wait()
        else:
            upd.update_phi(nodes, nbrs, nn, 2, g0)
    for name, fn in (("full launch", full), ("rank 0 of %d (rho %.3f, %d chunks, side stream %d)" % (a.R, a.rho, a.chunks, a.side), rank0)):
        ts = sorted(ops.elapsed_ms(fn) for _ in range(7))
        print("%-60s median %.3f ms" % (name, ts[3]))
    own = a.chunks * cc + g0
    print("groups this rank computes: %d of %d -> ideal %.3f of the full launch" % (own, G, own / G))


if __name__ == "__main__":
    main()
