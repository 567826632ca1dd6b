"""development aid: long runs of the captured-graph loop against the eager loop (bit-identical state after N steps)
and a longer graph-only soak (sampler shortfall check, finite state)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ammsb_pkg
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import hostlib, ops
from mcmc_ammsb_gpu_amd.learner import Config, Learner


def make(ds, K, m, graph):
    cfg = Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=32, strategy="Node", phi_wg_size=64 if K >= 256 else 32,
                                   beta_wg_size=64 if K >= 256 else 32, ppx_wg_size=64 if K >= 256 else 32,
                                   device_sampling=True, graph_launch=graph)
    return Learner(cfg, ds)


for name, (N, K, m, steps_cmp, steps_soak) in {"C1": (10_000, 32, 1024, 20000, 200000), "C2": (100_000, 256, 8192, 6000, 60000)}.items():
    edges = hostlib.generate_graph(N, min(K, 64), 32, seed=20260101)
    ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
    eager, graph = make(ds, K, m, False), make(ds, K, m, True)
    t0 = time.time(); eager.Run(steps_cmp); eager.drain(); te = time.time() - t0
    t0 = time.time()
    done = 0
    rng = np.random.default_rng(0)
    while done < steps_cmp:   # irregular Run() lengths: every ramp-up / hand-back path of the loop
        n = int(min(steps_cmp - done, rng.integers(1, 700)))
        graph.Run(n); done += n
    graph.drain(); tg = time.time() - t0
    same = (np.array_equal(eager.pi.host(), graph.pi.host()) and np.array_equal(ops.to_numpy(eager.theta), ops.to_numpy(graph.theta))
            and np.array_equal(eager.phiUpdater.rand.host(), graph.phiUpdater.rand.host())
            and np.array_equal(eager.dev_sampler.rand.host(), graph.dev_sampler.rand.host()))
    print("%s: %d steps eager %.2f s, graph %.2f s, bit-identical: %s, ppx %.4f / %.4f" % (
        name, steps_cmp, te, tg, same, eager.HeldoutPerplexity(), graph.HeldoutPerplexity()), flush=True)
    assert same
    t0 = time.time(); graph.Run(steps_soak); graph.drain(); ts = time.time() - t0
    pi = graph.pi.host()
    print("%s: soak %d steps in %.2f s (%.1f us/step), pi finite %s, rows sum to 1: %s, ppx %.4f" % (
        name, steps_soak, ts, ts / steps_soak * 1e6, bool(np.isfinite(pi).all()), bool(np.allclose(pi.sum(1), 1, atol=1e-4)),
        graph.HeldoutPerplexity()), flush=True)
    eager.close(); graph.close()
