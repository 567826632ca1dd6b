#!/usr/bin/env python3
"""Regression vectors for one small, fully specified iteration of the hot path -> tests/golden/iteration_K64.npz.

The reference ships no golden vectors for this path and cannot be built in this image, so these
are produced by oracle/ (the CPU restatement), NOT by the reference: they do not change the oracle's pin status.
What they do: (1) freeze the oracle -- tests/test_golden.py requires today's oracle build, on whatever host and
compiler, to reproduce them bit for bit; (2) give the HIP path a committed, oracle-independent target on the GPU box
(the -m gpu half of the same test file).

    python tests/golden/make_golden.py          # rewrites tests/golden/iteration_K64.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc  # noqa: E402

N, K, n, L = 500, 64, 8, 32
N_NODES, N_EDGES_MB, N_HELD = 24, 40, 60


def inputs():
    """Everything is derived from one numpy seed; returned as plain arrays (also stored in the fixture)."""
    rng = np.random.default_rng(20260101)
    edges = orc.random_graph_edges(rng, N, 3000)
    nodes = rng.permutation(N)[:N_NODES].astype(np.uint32)
    theta = rng.gamma(1.0, 1.0, size=2 * K).astype(np.float32)
    mb_edges = np.concatenate([edges[:N_EDGES_MB // 2],
                               orc.make_edge(rng.integers(0, N // 2, N_EDGES_MB // 2), rng.integers(N // 2, N, N_EDGES_MB // 2))])
    held = np.concatenate([edges[100:100 + N_HELD // 2],
                           orc.make_edge(rng.integers(0, N // 2, N_HELD // 2), rng.integers(N // 2, N, N_HELD // 2))])
    return dict(edges=edges, nodes=nodes, theta=theta, mb_edges=mb_edges.astype(np.uint64), held=held.astype(np.uint64))


def compute(inp):
    out = {}
    p = orc.make_params(N, K, n)
    # RNG: the first draws of stream {42, 43} and of stream {42 + 7, 43 + 7}
    for tag, (sx, sy) in (("a", (42, 43)), ("b", (49, 50))):
        s = orc.rng_init(1, sx, sy)
        u = np.zeros(16, dtype=np.uint64)
        orc.lib().orc_fill_rand(s, u, 16)
        out["rand_" + tag] = u
        s = orc.rng_init(1, sx, sy)
        f = np.zeros(64, dtype=np.float32)
        orc.lib().orc_fill_randn(s, f, 64)
        out["randn_" + tag] = f
        s = orc.rng_init(1, sx, sy)
        g = np.zeros(16, dtype=np.float32)
        orc.lib().orc_fill_gamma(s, 1.0, 1.0, g, 16)
        out["gamma_" + tag] = g
    # cuckoo set image + probes
    oset = orc.OracleSet(inp["edges"])
    out["set_slots"], out["set_shape"] = oset.slots, np.array([oset.num_bins, oset.prime_idx], dtype=np.uint64)
    probes = np.concatenate([inp["edges"][:50], inp["edges"][:50] ^ np.uint64(1 << 20)])
    out["set_probes"], out["set_hits"] = probes, oset.has(probes)
    # pi_0, phi_sum_0
    pi, phi_sum = orc.pi_init_gamma(N, K)
    out["pi0_rows"], out["phi_sum0"] = pi[:4].copy(), phi_sum.copy()
    # neighbour sampler {56, 57}
    ns_seeds = orc.rng_init(N_NODES * 2 * n, 56, 57)
    table, nbrs = orc.sample_neighbors(ns_seeds, inp["nodes"], N, n, 32)
    # make a third of the sampled pairs real links so that both y branches are taken
    out["neighbors"], out["ns_table"], out["ns_seeds_after"] = nbrs, table, ns_seeds.view(np.uint64)
    link_edges = orc.make_edge(np.repeat(inp["nodes"], 3), nbrs[:, :3].reshape(-1))
    oset2 = orc.OracleSet(np.unique(np.concatenate([inp["edges"], link_edges])))
    out["set2_slots"], out["set2_shape"] = oset2.slots, np.array([oset2.num_bins, oset2.prime_idx], dtype=np.uint64)
    # beta_0 from theta
    theta = inp["theta"].copy()
    beta = np.zeros_like(theta)
    orc.lib().orc_beta_from_theta(theta, beta, K)
    out["beta0"] = beta.copy()
    # update_phi (step 1, L lanes, noise on) and update_pi
    phi_seeds = orc.rng_init(N_NODES * L, 42, 43)
    phi_vec = orc.update_phi(p, beta, pi.reshape(-1), phi_sum, oset2, inp["nodes"], nbrs.reshape(-1).copy(), 1, phi_seeds,
                             L, 1, True)
    out["phi_vec"], out["phi_seeds_after"] = phi_vec, phi_seeds.view(np.uint64)
    orc.update_pi(p, pi.reshape(-1), phi_sum, phi_vec.reshape(-1).copy(), inp["nodes"], L, 1)
    out["pi1_rows"], out["phi_sum1"] = pi[inp["nodes"]].copy(), phi_sum[inp["nodes"]].copy()
    # beta gradient: reference order and float64 accumulation, then the theta step (scale 0.37)
    out["grads_ref_order"] = orc.beta_grads(p, theta, beta, pi.reshape(-1), oset2, inp["mb_edges"], L, 1, order=0)
    out["grads_f64"] = orc.beta_grads(p, theta, beta, pi.reshape(-1), oset2, inp["mb_edges"], L, 1, order=1)
    b_seeds = orc.rng_init(K, 44, 45)
    beta1 = orc.update_theta(p, theta, out["grads_ref_order"].copy(), 1, 0.37, b_seeds)
    out["theta1"], out["beta1"], out["beta_seeds_after"] = theta.copy(), beta1, b_seeds.view(np.uint64)
    # perplexity, two calls (running mean)
    hset = orc.OracleSet(inp["held"][:N_HELD // 2])
    out["hset_slots"], out["hset_shape"] = hset.slots, np.array([hset.num_bins, hset.prime_idx], dtype=np.uint64)
    state = np.zeros(inp["held"].size, dtype=np.float32)
    for call in (1, 2):
        sums, _ = orc.perplexity(p, beta1, pi.reshape(-1), hset, inp["held"], call, L, 1, state)
        out["ppx_state_%d" % call] = state.copy()
        out["ppx_sums_%d" % call] = np.array([sums.link_ll, sums.nonlink_ll, sums.link_cnt, sums.nonlink_cnt], dtype=np.float64)
    return out


def main():
    inp = inputs()
    out = compute(inp)
    path = os.path.join(ROOT, "tests", "golden", "iteration_K64.npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **{"in_" + k: v for k, v in inp.items()}, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
