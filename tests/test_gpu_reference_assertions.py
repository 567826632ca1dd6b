"""The value-level assertions the REFERENCE's own tests make, with the HIP path on one side.

The reference compares its two kernel families -- per-thread (the CPU-device shape) against work-group -- and
accepts 2 % (phi / pi: wg-phi-test.cc:116-142), max(1e-5, 2 %) (theta_sum / theta: wg-beta-test.cc:105-140) and
5 % (i + 1) (perplexity: wg-perplexity-test.cc:86-108), on the shapes below.  Here the work-group side is the HIP
kernels and the per-thread side is the oracle's restatement of the per-thread kernels (phi.cc:78-152,
beta.cc:87-136, perplexity.cc:16-84): an independent formulation (different loop structure, different summation
order), so an error shared by the HIP kernels and the oracle's work-group restatement would show here.  Every case
is also held to the tolerance north_star states (1e-5).  Plus serialize-test.cc:90-134 at its own shape.
"""
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TIGHT = 1e-5  # BASELINE.json north_star


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import __graft_entry__ as ge
    ge.build()
    from mcmc_ammsb_gpu_amd import hostlib, learner, ops
    return ops, hostlib, learner, torch


class _OSet:
    def __init__(self, hs):
        self.slots, self.num_bins, self.prime_idx = hs.Serialize(), hs.BinsPerBucket(), hs.PrimeIdx()


def _random_edges(rng, N, count):  # GenerateRandomEdges of the reference tests: sorted unique canonical pairs
    u = rng.integers(0, N, count, dtype=np.uint64)
    v = rng.integers(0, N, count, dtype=np.uint64)
    return np.unique((np.minimum(u, v) << np.uint64(32)) | np.maximum(u, v))


def _near(a, b, rel, floor=0.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool((np.abs(a - b) <= np.maximum(floor, rel * np.abs(a))).all())


@pytest.mark.parametrize("wg", [32, 64, 128, 256])
def test_wg_phi_verify_modes(env, orc, wg):
    """wg-phi-test.cc:116-142: K = 512, N = 4096 = mini-batch (every vertex once), 4 neighbours, noise off."""
    ops, hostlib, learner, torch = env
    N, K, n = 4096, 512, 4
    rng = np.random.default_rng(42)
    edges = _random_edges(rng, N, 32 * N)
    hs = hostlib.HostSet(edges)
    oset = _OSet(hs)
    ctx = ops.Context(ops.make_params(N, K, E=edges.size, num_node_sample=n))
    po = orc.make_params(N, K, n)
    pi = ops.RowPartitionedMatrix(ctx, N, K)
    phi_sum = ctx.zeros((N,), torch.float32)
    ops.RandomGammaAndNormalize(ctx, 1.0, 1.0, pi, phi_sum)
    theta_h = hostlib.theta_init(K)
    beta_h = np.zeros_like(theta_h)
    orc.lib().orc_beta_from_theta(theta_h, beta_h, K)
    beta = ctx.from_numpy(beta_h)
    dset = ops.DeviceSet(ctx, oset.slots, oset.num_bins, oset.prime_idx)
    nodes_h = np.arange(N, dtype=np.uint32)
    nbrs_h = rng.integers(0, N, size=(N, n), dtype=np.uint32)  # rand() % N: a node may draw itself, as in the reference
    pi_h, phi_h = pi.host(), phi_sum.cpu().numpy().copy()
    upd = ops.PhiUpdater(ctx, beta, pi, phi_sum, dset, N, (42, 43), wg, phi_disable_noise=True)
    upd(ctx.from_numpy(nodes_h), ctx.from_numpy(nbrs_h), N)
    torch.cuda.synchronize()
    # the per-thread kernels (PHI_NODE_PER_THREAD), restated
    seeds = orc.rng_init(N * wg, 42, 43)
    want = orc.update_phi(po, beta_h, pi_h.reshape(-1), phi_h, oset, nodes_h, nbrs_h.reshape(-1), 1, seeds, wg, 0, False)
    orc.update_pi(po, pi_h.reshape(-1), phi_h, want.reshape(-1), nodes_h, wg, 0)
    got_pi, got_phi = pi.host(), phi_sum.cpu().numpy()
    assert _near(phi_h, got_phi, 0.02) and _near(pi_h, got_pi, 0.02)        # the reference's own bound
    # north_star's 1e-5: on phi_sum element-wise, on pi relative to the scale of its row (the two formulations add
    # the K terms of a row in different orders; an entry a thousand times smaller than its row's largest carries
    # the row's absolute rounding error, not its own)
    assert _near(phi_h, got_phi, TIGHT)
    assert (np.abs(pi_h.astype(np.float64) - got_pi) <= TIGHT * pi_h.max(axis=1, keepdims=True)).all()
    assert np.median(np.abs(pi_h.astype(np.float64) - got_pi) / pi_h) <= 1e-6
    ctx.close()


@pytest.mark.parametrize("wg", [32, 64, 128, 256, 512, 1024])
def test_wg_beta_verify_modes(env, orc, wg):
    """wg-beta-test.cc:105-140: K = 1024, N = 4096, 1024 random edges, scale 0.01; theta_sum and theta."""
    ops, hostlib, learner, torch = env
    N, K, n = 4096, 1024, 64
    rng = np.random.default_rng(7)
    edges = _random_edges(rng, N, 32 * N)
    hs = hostlib.HostSet(edges)
    oset = _OSet(hs)
    ctx = ops.Context(ops.make_params(N, K, E=edges.size, num_node_sample=n))
    po = orc.make_params(N, K, n)
    pi = ops.RowPartitionedMatrix(ctx, N, K)
    phi_sum = ctx.zeros((N,), torch.float32)
    ops.RandomGammaAndNormalize(ctx, 1.0, 1.0, pi, phi_sum)
    theta_h = hostlib.theta_init(K)
    beta_h = np.zeros_like(theta_h)
    orc.lib().orc_beta_from_theta(theta_h, beta_h, K)
    theta, beta = ctx.from_numpy(theta_h), ctx.from_numpy(beta_h)
    dset = ops.DeviceSet(ctx, oset.slots, oset.num_bins, oset.prime_idx)
    mb = _random_edges(rng, N, 1024)
    upd = ops.BetaUpdater(ctx, theta, beta, pi, dset, (44, 45), wg)
    # (wg = 32 is the reference's default beta_wg_size, main.cc:64: K / wg = 32 columns per work-item go through
    # beta_grads_gen_kernel, the generic form)
    upd(ctx.from_numpy(mb), mb.size, 0.01)
    torch.cuda.synchronize()
    # EDGE_PER_THREAD restated: per-thread partial rows summed serially, then the same theta step
    pi_h = pi.host()
    ts = np.zeros(K, dtype=np.float32)
    orc.lib().orc_sum_theta(theta_h, ts, K)
    g = orc.beta_grads(po, theta_h, beta_h, pi_h.reshape(-1), oset, mb, wg, 0, order=0)
    want_theta = theta_h.copy()
    orc.update_theta(po, want_theta, g, 1, 0.01, orc.rng_init(K, 44, 45))
    got_ts, got_theta = upd.GetThetaSum().cpu().numpy(), theta.cpu().numpy()
    assert _near(ts, got_ts, 0.02, 1e-5) and _near(want_theta, got_theta, 0.02, 1e-5)   # the reference's own bound
    assert np.array_equal(ts, got_ts)
    assert _near(want_theta, got_theta, TIGHT, 1e-9)
    assert _near(g, upd.GetGrads().cpu().numpy(), TIGHT, TIGHT * float(np.abs(g).max()))
    ctx.close()


@pytest.mark.parametrize("wg", [32, 64, 128, 256, 512, 1024])
def test_wg_perplexity_equal(env, orc, wg):
    """wg-perplexity-test.cc:86-108: K = N = 1024, 1024 edges that are all members of the set; three calls."""
    ops, hostlib, learner, torch = env
    N, K, n = 1024, 1024, 32
    rng = np.random.default_rng(11)
    edges = _random_edges(rng, N - 1, 1024)
    hs = hostlib.HostSet(edges)
    oset = _OSet(hs)
    ctx = ops.Context(ops.make_params(N, K, E=edges.size, num_node_sample=n))
    po = orc.make_params(N, K, n)
    pi_h = rng.gamma(1.0, 1.0, (N, K)).astype(np.float32)
    for r in range(N):   # PartitionedNormalizer (normalize.cc:34-52): WG_SUM order with 32 lanes, as at start-up
        orc.lib().orc_wg_normalize_f32(pi_h[r], K, 32)
    beta_h = rng.gamma(1.0, 1.0, 2 * K).astype(np.float32)
    b2 = beta_h.reshape(K, 2)
    s = (np.float32(0) + b2[:, 0]) + b2[:, 1]   # Normalizer(slice 2, wg 1)
    beta_h = (b2 / s[:, None]).astype(np.float32).reshape(-1)
    pi = ops.RowPartitionedMatrix(ctx, N, K)
    pi.load(pi_h)
    beta = ctx.from_numpy(beta_h)
    dset = ops.DeviceSet(ctx, oset.slots, oset.num_bins, oset.prime_idx)
    calc = ops.PerplexityCalculator(ctx, beta, pi, ctx.from_numpy(edges), dset, wg)
    state = np.zeros(edges.size, dtype=np.float32)
    for i in range(3):
        got = calc()
        sums, _ = orc.perplexity(po, beta_h, pi_h.reshape(-1), oset, edges, i + 1, wg, 0, state)   # EDGE_PER_THREAD
        want = -(sums.link_ll + sums.nonlink_ll) / (sums.link_cnt + sums.nonlink_cnt)
        assert abs(got - want) <= (i + 1) * 0.05 * abs(want)                                       # the reference's own bound
        assert abs(got - want) <= TIGHT * abs(want)
        assert _near(state, calc.ppx_per_edge.cpu().numpy(), TIGHT)
    ctx.close()


@pytest.mark.parametrize("device_sampling", [False, True])
def test_serialize_end_to_end_reference_shape(env, device_sampling):
    """serialize-test.cc:90-134 as written there: N = 1024, 1024 random edges, held-out ratio 0.1, every other field
    the Config() default (K = 32, mini-batch 32, 32 neighbours, work-groups 32), 10 + 10 iterations."""
    ops, hostlib, learner, torch = env
    rng = np.random.default_rng(1)
    N, iters = 1024, 10
    u = rng.integers(0, N, 1024, dtype=np.uint64)
    v = rng.integers(0, N, 1024, dtype=np.uint64)
    keep = u != v
    edges = np.unique((np.minimum(u, v) << np.uint64(32))[keep] | np.maximum(u, v)[keep])
    ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.1, rand_seed=1)

    def cfg():
        return learner.Config(heldout_ratio=0.1, ppx_interval=2 * iters - 1, device_sampling=device_sampling)
    out = io.BytesIO()
    l1 = learner.Learner(cfg(), ds)
    l1.Run(iters)
    assert l1.Serialize(out)
    l1.Run(iters)
    ppx = l1.HeldoutPerplexity()
    l1.close()
    l2 = learner.Learner(cfg(), ds)
    assert l2.Parse(io.BytesIO(out.getvalue()))
    l2.Run(iters)
    assert l2.HeldoutPerplexity() == ppx
    l2.close()


def test_training_perplexity_mode(env, orc):
    """MCMC_CALC_TRAIN_PPX (learner.cc:47-75, :204-212, :321-323) as Config.calc_train_ppx: the edge list follows the
    reference's construction, the value matches the oracle over that list, and the checkpoint carries the extra
    PerplexityCalculator record between BetaUpdater's and the held-out calculator's."""
    ops, hostlib, learner, torch = env
    N = 2000
    edges = hostlib.generate_graph(N, 8, 12, seed=3)
    ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.05, rand_seed=3)
    ratio = 0.05
    te = ds.train_ppx_edges(ratio, 1)
    links = int(np.float32(ratio) * ds.training_edges.size)
    non = int(links * ((N * (N - 1)) // 2) / float(ds.E))
    assert te.size == links + non and np.array_equal(te[:links], ds.training_edges[:links])
    rest = te[links:]
    assert ((rest >> np.uint64(32)) != (rest & np.uint64(0xFFFFFFFF))).all()
    assert not ds.training.Has(rest).any() and not ds.heldout.Has(rest).any()

    def cfg():
        return learner.Config.from_cli_defaults(K=64, mini_batch_size=128, num_node_sample=8, phi_wg_size=64,
                                                beta_wg_size=64, ppx_wg_size=64, calc_train_ppx=True,
                                                training_ppx_ratio=ratio)
    l1 = learner.Learner(cfg(), ds)
    got = l1.TrainingPerplexity()
    po = orc.make_params(N, 64, 8)
    state = np.zeros(te.size, dtype=np.float32)
    sums, _ = orc.perplexity(po, ops.to_numpy(l1.beta), l1.pi.host().reshape(-1), _OSet(ds.training), te, 1, 64, 1, state)
    want = float(np.exp(np.float32(-(sums.link_ll + sums.nonlink_ll) / (sums.link_cnt + sums.nonlink_cnt))))
    assert sums.link_cnt == links and sums.nonlink_cnt == non
    assert abs(got - want) <= TIGHT * want
    assert np.array_equal(ops.to_numpy(l1.trainingPerplexity.ppx_per_edge), state)
    l1.Run(15)
    out = io.BytesIO()
    assert l1.Serialize(out)
    l1.Run(15)
    a = (l1.TrainingPerplexity(), l1.HeldoutPerplexity())
    l1.close()
    l2 = learner.Learner(cfg(), ds)
    assert l2.Parse(io.BytesIO(out.getvalue()))
    l2.Run(15)
    assert (l2.TrainingPerplexity(), l2.HeldoutPerplexity()) == a
    l2.close()
    # a learner without the mode refuses the call and reads a stream without the record
    plain = learner.Learner(learner.Config.from_cli_defaults(K=64, mini_batch_size=128, num_node_sample=8, phi_wg_size=64,
                                                             beta_wg_size=64, ppx_wg_size=64), ds)
    with pytest.raises(ops.AmmsbError):
        plain.TrainingPerplexity()
    plain.close()
