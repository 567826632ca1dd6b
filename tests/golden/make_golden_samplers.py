#!/usr/bin/env python3
"""Golden vectors of the HOST mini-batch samplers ("host sampler outputs (edges, weight, node order) for
seeds 1..4 on a fixed small graph") -> tests/golden/host_samplers.npz.

Made by oracle/ammsb_oracle_samplers.c (the plain-C restatement of sample.cc:177-303 + learner.cc:162-173), NOT by
the reference, which cannot be built in this image.  They freeze three platform-dependent
things at once -- glibc's rand_r, libstdc++'s unordered_set iteration order and the prime table behind it -- so a
change of toolchain that would silently change every reference-exact trajectory shows up as a test failure
(tests/test_oracle_samplers.py checks the oracle AND libammsb_host.so against this file).

    python tests/golden/make_golden_samplers.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc  # noqa: E402

N, M = 600, 48
STRATEGIES = ["Node", "NodeLink", "NodeNonLink", "BFLink", "BFNonLink", "BF"]


def graph():
    rng = np.random.default_rng(20260102)
    edges = orc.random_graph_edges(rng, N, 2400)
    held, train = edges[:60].copy(), edges[60:].copy()
    return train, held


def compute():
    train, held = graph()
    tset, hset = orc.OracleSet(train), orc.OracleSet(held)
    out = {"in_training_edges": train, "in_heldout_links": held}
    E = train.size + held.size
    deg = np.bincount(np.concatenate([train >> np.uint64(32), train & np.uint64(0xFFFFFFFF)]).astype(np.int64), minlength=N)
    cap_e, cap_n = max(M, int(deg.max())), max(2 * M, 1 + int(deg.max()))
    for s in STRATEGIES:
        for seed0 in (1, 2, 3, 4):
            seed = seed0
            for it in range(3):
                e, v, w, seed = orc.host_sample(N, E, M, s, seed, train, tset, hset, cap_e, cap_n)
                key = "%s_s%d_i%d" % (s, seed0, it)
                out[key + "_edges"], out[key + "_nodes"] = e, v
                out[key + "_weight"] = np.array([w], dtype=np.float32)
                out[key + "_seed"] = np.array([seed], dtype=np.uint32)
    return out


if __name__ == "__main__":
    orc.build()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "host_samplers.npz"), **compute())
    print("written")
