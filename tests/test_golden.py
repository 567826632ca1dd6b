"""tests/golden/iteration_K64.npz -- one small, fully specified iteration of the hot path (made by
tests/golden/make_golden.py from oracle/, NOT from the reference: see that file's header).

  * CPU: today's oracle build reproduces every vector bit for bit (freezes the oracle across hosts/compilers).
  * GPU: the HIP path, driven through the C ABI from the fixture's INPUTS only, reproduces the integer vectors, the
    RNG draws, phi_vec, pi, phi_sum, theta, beta and the perplexity state bit for bit, and the gradient / the
    log-likelihood sums within 1e-5 (north_star tolerance; the sums are order-free)."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PATH = os.path.join(HERE, "golden", "iteration_K64.npz")


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a.view(np.uint64) if a.dtype == np.float64 else a


def test_oracle_reproduces_golden(orc):
    g = np.load(PATH)
    mod = _gen()
    inp = mod.inputs()
    for k, v in inp.items():
        assert np.array_equal(g["in_" + k], v), "input " + k   # numpy's Generator streams are stable by contract
    out = mod.compute({k: g["in_" + k] for k in inp})
    assert set(out) == {f for f in g.files if not f.startswith("in_")}
    for k, v in out.items():
        assert np.array_equal(_bits(g[k]), _bits(v)), k


@pytest.mark.gpu
def test_hip_reproduces_golden(orc):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import ammsb_pkg
    ammsb_pkg.load()
    from mcmc_ammsb_gpu_amd import ops
    g = np.load(PATH)
    mod = _gen()
    N, K, n, L = mod.N, mod.K, mod.n, mod.L
    ctx = ops.Context(ops.make_params(N, K, E=3000, num_node_sample=n))
    lib = ctx.lib

    def ptr(t):
        return C.c_void_p(t.data_ptr())
    # RNG draws
    for tag, (sx, sy) in (("a", (42, 43)), ("b", (49, 50))):
        rnd = ops.Random(ctx, 1, (sx, sy))
        out = ctx.empty((1, 64), torch.float32)
        ctx.check(lib.ammsb_randn_fill(ctx.handle, ptr(rnd.seeds), 1, 64, ptr(out), None))
        assert np.array_equal(ops.to_numpy(out).reshape(-1).view(np.uint32), g["randn_" + tag].view(np.uint32)), tag
    # cuckoo membership on the fixture's table image
    s1 = ops.DeviceSet(ctx, g["set_slots"], int(g["set_shape"][0]), int(g["set_shape"][1]))
    hits = ctx.zeros((g["set_probes"].size,), torch.uint8)
    ctx.check(lib.ammsb_set_has(ctx.handle, C.byref(s1.desc), ptr(ctx.from_numpy(g["set_probes"])), g["set_probes"].size,
                                ptr(hits), None))
    assert np.array_equal(ops.to_numpy(hits).astype(bool), g["set_hits"])
    # pi_0
    pi = ops.RowPartitionedMatrix(ctx, N, K)
    phi_sum = ctx.zeros((N,), torch.float32)
    ops.RandomGammaAndNormalize(ctx, 1.0, 1.0, pi, phi_sum)
    assert np.array_equal(pi.host()[:4].view(np.uint32), g["pi0_rows"].view(np.uint32))
    assert np.array_equal(ops.to_numpy(phi_sum).view(np.uint32), g["phi_sum0"].view(np.uint32))
    # neighbour sampler
    nodes = ctx.from_numpy(g["in_nodes"])
    ns = ops.NeighborSampler(ctx, g["in_nodes"].size, (56, 57), 32)
    ns(g["in_nodes"].size, nodes)
    torch.cuda.synchronize()
    assert np.array_equal(ops.to_numpy(ns.GetData()).view(np.uint32).reshape(-1, n), g["neighbors"])
    assert np.array_equal(ops.to_numpy(ns.hash).view(np.uint32).reshape(-1, 2 * n), g["ns_table"])
    assert np.array_equal(ns.rand.host().view(np.uint64), g["ns_seeds_after"])
    # update_phi / update_pi
    theta = ctx.from_numpy(g["in_theta"])
    beta = ctx.zeros((2 * K,), torch.float32)
    ops.beta_from_theta(ctx, theta, beta)
    assert np.array_equal(ops.to_numpy(beta).view(np.uint32), g["beta0"].view(np.uint32))
    s2 = ops.DeviceSet(ctx, g["set2_slots"], int(g["set2_shape"][0]), int(g["set2_shape"][1]))
    nn = g["in_nodes"].size
    upd = ops.PhiUpdater(ctx, beta, pi, phi_sum, s2, nn, (42, 43), L)
    upd(nodes, ns.GetData(), nn)
    torch.cuda.synchronize()
    assert np.array_equal(ops.to_numpy(upd.phi_vec)[:nn].view(np.uint32), g["phi_vec"].view(np.uint32))
    assert np.array_equal(upd.rand.host().view(np.uint64), g["phi_seeds_after"])
    rows = pi.host()[g["in_nodes"]]
    assert np.array_equal(rows.view(np.uint32), g["pi1_rows"].view(np.uint32))
    assert np.array_equal(ops.to_numpy(phi_sum)[g["in_nodes"]].view(np.uint32), g["phi_sum1"].view(np.uint32))
    # beta gradient (order-free: <= 1e-5 of the float64 accumulation) and the theta step on the fixture's gradient
    bu = ops.BetaUpdater(ctx, theta, beta, pi, s2, (44, 45), L)
    mb = ctx.from_numpy(g["in_mb_edges"])
    got = ops.to_numpy(bu.calculate_grads(mb, g["in_mb_edges"].size)).astype(np.float64)
    assert np.abs(got - g["grads_f64"]).max() <= 1e-5 * np.abs(g["grads_f64"]).max()
    bu.count_calls = 1
    bu.update_theta(0.37, ctx.from_numpy(g["grads_ref_order"]))
    torch.cuda.synchronize()
    assert np.array_equal(ops.to_numpy(theta).view(np.uint32), g["theta1"].view(np.uint32))
    assert np.array_equal(ops.to_numpy(beta).view(np.uint32), g["beta1"].view(np.uint32))
    assert np.array_equal(bu.rand.host().view(np.uint64), g["beta_seeds_after"])
    # perplexity: two calls
    hs = ops.DeviceSet(ctx, g["hset_slots"], int(g["hset_shape"][0]), int(g["hset_shape"][1]))
    calc = ops.PerplexityCalculator(ctx, beta, pi, ctx.from_numpy(g["in_held"]), hs, L)
    for call in (1, 2):
        calc()
        l0, l1, c0, c1 = calc.unpack(calc.sums)
        want = g["ppx_sums_%d" % call]
        assert (c0, c1) == (int(want[2]), int(want[3]))
        assert abs(l0 - want[0]) <= 1e-5 * abs(want[0]) and abs(l1 - want[1]) <= 1e-5 * abs(want[1])
        assert np.array_equal(ops.to_numpy(calc.ppx_per_edge).view(np.uint32), g["ppx_state_%d" % call].view(np.uint32))
