"""Parity at BASELINE.json's full sizes (C3: N = 1M, K = 1024, mini-batch 65536, n = 32), where the
oracle cannot redo everything in seconds: oracle comparison on a sampled subset plus size-independent
properties (row normalisation, untouched-row checksums, shard invariance, count conservation).  Also a
C5-shaped case (K = 4096, > 2^32 elements in pi) for the 64-bit addressing the reference lacks."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(orc):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import __graft_entry__ as ge
    ge.build()
    from mcmc_ammsb_gpu_amd import hostlib, ops
    N, K, m, n = 1_000_000, 1024, 65536, 32
    rng = np.random.default_rng(2026)
    p = ops.make_params(N, K, E=16 * N, num_node_sample=n)
    ctx = ops.Context(p)
    pi = ops.RowPartitionedMatrix(ctx, N, K)
    phi_sum = ctx.zeros((N,), torch.float32)
    ops.RandomGammaAndNormalize(ctx, 1.0, 1.0, pi, phi_sum)
    theta_h = hostlib.theta_init(K)
    theta = ctx.from_numpy(theta_h)
    beta = ctx.zeros((2 * K,), torch.float32)
    ops.beta_from_theta(ctx, theta, beta)
    u = rng.integers(0, N, 2_000_000, dtype=np.uint64)
    v = rng.integers(0, N, 2_000_000, dtype=np.uint64)
    e = np.unique((np.minimum(u, v) << np.uint64(32)) | np.maximum(u, v))
    e = e[(e >> np.uint64(32)) != (e & np.uint64(0xFFFFFFFF))]
    hs = hostlib.HostSet(e)
    dset = ops.DeviceSet(ctx, hs.Serialize(), hs.BinsPerBucket(), hs.PrimeIdx())
    torch.cuda.synchronize()
    return dict(ops=ops, hostlib=hostlib, ctx=ctx, p=p, pi=pi, phi_sum=phi_sum, theta=theta, beta=beta, edges=e,
                hs=hs, dset=dset, rng=rng, N=N, K=K, m=m, n=n, torch=torch)


class _OSet:  # oracle-side view of the host library's table image
    def __init__(self, hs):
        self.slots, self.num_bins, self.prime_idx = hs.Serialize(), hs.BinsPerBucket(), hs.PrimeIdx()


def test_c3_update_phi_and_pi(orc, big):
    b = big
    ops, ctx, torch, rng = b["ops"], b["ctx"], b["torch"], b["rng"]
    N, K, m, n = b["N"], b["K"], b["m"], b["n"]
    nn = m + 1  # 65537 nodes: groups 0 and 1 handle two nodes each
    nodes_h = rng.permutation(N)[:nn].astype(np.uint32)
    nbrs_h = rng.integers(0, N, size=(nn, n), dtype=np.uint32)
    # give the first 64 nodes some true neighbours so both branches of y are exercised
    src = (b["edges"] >> np.uint64(32)).astype(np.uint32)
    dst = (b["edges"] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    for i in range(64):
        mine = dst[src == nodes_h[i]][:8]
        nbrs_h[i, :mine.size] = mine
    same = nbrs_h == nodes_h[:, None]
    nbrs_h[same] = (nbrs_h[same] + 1) % N
    nodes, nbrs = ctx.from_numpy(nodes_h), ctx.from_numpy(nbrs_h)
    L = 64
    upd = ops.PhiUpdater(ctx, b["beta"], b["pi"], b["phi_sum"], b["dset"], nn, (42, 43), L)
    upd.count_calls = 1
    upd.update_phi(nodes, nbrs, nn)
    torch.cuda.synchronize()
    pi_h = b["pi"].host()
    phi_sum_h = b["phi_sum"].cpu().numpy().copy()
    beta_h = b["beta"].cpu().numpy()
    got = upd.phi_vec[:nn].cpu().numpy()
    assert np.isfinite(got).all() and (got >= 1e-24).all()
    # oracle on a subset: the first 64 nodes, the two wrap-around nodes (second node of groups 0, 1) and 64 random ones.
    # Node i belongs to group i % 65535; to reproduce its stream state the oracle runs the group's earlier node too.
    po = orc.make_params(N, K, n)
    oset = _OSet(b["hs"])
    G = 65535
    pick = np.unique(np.concatenate([np.arange(64), [0, 1, G, G + 1], rng.integers(0, nn, 64)]))
    for i in pick:
        g = i % G
        chain = [g] if i < G else [g, i]          # nodes the group visits up to and including i
        seeds = orc.rng_init(L, 42 + g * L, 43 + g * L)  # stream g*L + l == {42 + g*L + l, 43 + g*L + l}
        for idx in chain:
            want = orc.update_phi(po, beta_h, pi_h.reshape(-1), phi_sum_h, oset, nodes_h[idx:idx + 1].copy(),
                                  nbrs_h[idx].copy(), 1, seeds, L, 1, True)
        assert np.array_equal(got[i].view(np.uint32), want[0].view(np.uint32)), "node index %d" % i
    # update_pi: rows normalised, phi_sum = row sum, every other row untouched
    before_rest = None
    touched = np.zeros(N, dtype=bool)
    touched[nodes_h] = True
    sample_rest = np.flatnonzero(~touched)[:: max(1, (N - nn) // 4096)]
    before_rest = pi_h[sample_rest].copy()
    upd.update_pi(nodes, nn)
    torch.cuda.synchronize()
    pi2 = b["pi"].host()
    ps2 = b["phi_sum"].cpu().numpy()
    s = got.astype(np.float64).sum(1)
    assert np.allclose(ps2[nodes_h], s, rtol=3e-6)
    assert np.allclose(pi2[nodes_h].astype(np.float64).sum(1), 1.0, atol=3e-6)
    assert np.array_equal(pi2[sample_rest], before_rest) and np.array_equal(ps2[~touched], phi_sum_h[~touched])
    sub = pick[:32]
    pi_o, ps_o = pi_h.copy()[nodes_h[sub]], None
    for k, i in enumerate(sub):  # exact normalisation on the subset (WG_SUM order L = 64)
        row = got[i].copy()
        sm = orc.lib().orc_wg_normalize_f32(row, K, L)
        assert np.array_equal(pi2[nodes_h[i]], row) and ps2[nodes_h[i]] == sm
    # shard invariance at full size: four group ranges reproduce the single launch bit for bit
    b["pi"].load(pi_h)
    b["phi_sum"].copy_(ctx.from_numpy(phi_sum_h))
    part = ops.PhiUpdater(ctx, b["beta"], b["pi"], b["phi_sum"], b["dset"], nn, (42, 43), L)
    part.count_calls = 1
    for lo, hi in ((0, 1), (1, 20000), (20000, 65534), (65534, 70000)):
        part.update_phi(nodes, nbrs, nn, lo, hi)
    torch.cuda.synchronize()
    assert torch.equal(part.phi_vec[:nn], upd.phi_vec[:nn])
    assert torch.equal(part.rand.seeds, upd.rand.seeds)


@pytest.mark.parametrize("beta_wg", [64, 128])  # 64: LDS-streamed kernel, 128: register kernel
def test_c3_beta_and_perplexity(orc, big, beta_wg):
    b = big
    ops, ctx, torch, rng = b["ops"], b["ctx"], b["torch"], b["rng"]
    N, K, m = b["N"], b["K"], b["m"]
    pi_h = b["pi"].host()
    beta_h, theta_h = b["beta"].cpu().numpy().copy(), b["theta"].cpu().numpy().copy()
    po = orc.make_params(N, K, b["n"])
    oset = _OSet(b["hs"])
    # one non-link-shaped batch (shared end point) with a few real links mixed in
    uu = np.uint64(rng.integers(0, N))
    vv = rng.permutation(N)[:m].astype(np.uint64)
    vv = vv[vv != uu]
    mb = np.concatenate([(np.minimum(uu, vv) << np.uint64(32)) | np.maximum(uu, vv), b["edges"][:m - vv.size + 64]])[:m]
    dev = ctx.from_numpy(mb)
    bu = ops.BetaUpdater(ctx, b["theta"].clone(), b["beta"].clone(), b["pi"], b["dset"], (44, 45), beta_wg)
    g = bu.calculate_grads(dev, m).cpu().numpy()
    exact = orc.beta_grads(po, theta_h, beta_h, pi_h.reshape(-1), oset, mb, beta_wg, 1, order=1)
    err = np.abs(g.astype(np.float64) - exact).max() / np.abs(exact).max()
    assert err <= 1e-5, err
    # linearity over edge shards (the multi-GPU contract) at full size
    acc = np.zeros(2 * K)
    for lo, hi in ((0, 8192), (8192, 8193), (8193, 40000), (40000, m)):
        acc += bu.calculate_grads(dev, m, lo, hi).cpu().numpy().astype(np.float64)
    assert np.abs(acc - g).max() / np.abs(g).max() <= 3e-6
    # perplexity over 160k held-out-shaped edges: counts conserve, state and sums match the oracle
    H = 160_000
    held = b["edges"][:H // 2]
    hs = b["hostlib"].HostSet(held)
    fake = (rng.integers(0, N, H // 2, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, N, H // 2, dtype=np.uint64)
    fake = fake[~hs.Has(fake)]
    he = np.concatenate([held, fake])
    dh = ops.DeviceSet(ctx, hs.Serialize(), hs.BinsPerBucket(), hs.PrimeIdx())
    calc = ops.PerplexityCalculator(ctx, b["beta"], b["pi"], ctx.from_numpy(he), dh, 64)
    state = np.zeros(he.size, dtype=np.float32)
    for call in (1, 2):
        got = calc()
        sums, _ = orc.perplexity(po, beta_h, pi_h.reshape(-1), _OSet(hs), he, call, 64, 1, state)
        l0, l1, c0, c1 = calc.unpack(calc.sums)
        assert (c0, c1) == (sums.link_cnt, sums.nonlink_cnt) == (held.size, fake.size)
        assert np.array_equal(calc.ppx_per_edge.cpu().numpy(), state)
        assert abs(l0 - sums.link_ll) <= 1e-10 * abs(sums.link_ll) and abs(l1 - sums.nonlink_ll) <= 1e-10 * abs(sums.nonlink_ll)
        want = -(sums.link_ll + sums.nonlink_ll) / he.size
        assert abs(got - want) <= 1e-5 * abs(want)


def test_c5_shape_64bit_rows(orc):
    """K = 4096 and N large enough that row * K exceeds 2^32 elements (the reference's uint row offset,
    partitioned-alloc.h:24-28, wraps there): rows near the end of pi must be addressed correctly."""
    import torch
    from mcmc_ammsb_gpu_amd import hostlib, ops
    N, K, n, L = 1_100_000, 4096, 8, 256   # 4.5e9 elements, 18 GB
    rng = np.random.default_rng(5)
    ctx = ops.Context(ops.make_params(N, K, E=16 * N, num_node_sample=n))
    pi = ops.RowPartitionedMatrix(ctx, N, K)
    phi_sum = ctx.zeros((N,), torch.float32)
    ops.RandomGammaAndNormalize(ctx, 1.0, 1.0, pi, phi_sum)
    # neighbours and nodes deliberately beyond row 2^32 / K = 1 048 576
    hi_rows = np.arange(1_048_576 - 8, N, dtype=np.uint32)
    nn = 256
    nodes_h = rng.choice(hi_rows, nn, replace=False).astype(np.uint32)
    nbrs_h = rng.choice(hi_rows, (nn, n)).astype(np.uint32)
    same = nbrs_h == nodes_h[:, None]
    nbrs_h[same] = nbrs_h[same] - 1
    e = (np.minimum(nodes_h[:64].astype(np.uint64), nbrs_h[:64, 0].astype(np.uint64)) << np.uint64(32)) | \
        np.maximum(nodes_h[:64].astype(np.uint64), nbrs_h[:64, 0].astype(np.uint64))
    e = np.unique(np.concatenate([e, rng.integers(0, 2**40, 5000, dtype=np.uint64)]))
    hs = hostlib.HostSet(e)
    dset = ops.DeviceSet(ctx, hs.Serialize(), hs.BinsPerBucket(), hs.PrimeIdx())
    theta_h = hostlib.theta_init(K)
    theta = ctx.from_numpy(theta_h)
    beta = ctx.zeros((2 * K,), torch.float32)
    ops.beta_from_theta(ctx, theta, beta)
    upd = ops.PhiUpdater(ctx, beta, pi, phi_sum, dset, nn, (42, 43), L)
    upd(ctx.from_numpy(nodes_h), ctx.from_numpy(nbrs_h), nn)
    torch.cuda.synchronize()
    # oracle needs only the rows involved: compact them into a small matrix with remapped ids
    rows = np.unique(np.concatenate([nodes_h, nbrs_h.ravel()]))
    remap = {int(r): k for k, r in enumerate(rows)}
    blk = pi.blocks[0]
    # rows as they were BEFORE update_pi overwrote the mini-batch nodes: regenerate pi_0 deterministically
    pi0 = ops.RowPartitionedMatrix(ctx, N, K)
    ps0 = ctx.zeros((N,), torch.float32)
    ops.RandomGammaAndNormalize(ctx, 1.0, 1.0, pi0, ps0)
    sub_pi = pi0.blocks[0][torch.from_numpy(rows.astype(np.int64)).to(ctx.device)].cpu().numpy()
    sub_ps = ps0[torch.from_numpy(rows.astype(np.int64)).to(ctx.device)].cpu().numpy()
    del pi0
    po = orc.make_params(N, K, n)
    lo_nodes = np.array([remap[int(x)] for x in nodes_h], dtype=np.uint32)
    lo_nbrs = np.array([[remap[int(x)] for x in r] for r in nbrs_h], dtype=np.uint32)
    # the cuckoo keys use the ORIGINAL ids: answer the probes up front and hand the oracle a set of remapped keys
    keys = (np.minimum(nodes_h[:, None].astype(np.uint64), nbrs_h.astype(np.uint64)) << np.uint64(32)) | \
        np.maximum(nodes_h[:, None].astype(np.uint64), nbrs_h.astype(np.uint64))
    link = hs.Has(keys.ravel()).reshape(keys.shape)
    lk = (np.minimum(lo_nodes[:, None].astype(np.uint64), lo_nbrs.astype(np.uint64)) << np.uint64(32)) | \
        np.maximum(lo_nodes[:, None].astype(np.uint64), lo_nbrs.astype(np.uint64))
    oset = orc.OracleSet(np.unique(np.concatenate([lk[link], [np.uint64(2**63)]])))
    assert link[:64, 0].all()
    seeds = orc.rng_init(nn * L, 42, 43)
    want = orc.update_phi(po, beta.cpu().numpy(), sub_pi.reshape(-1), sub_ps, oset, lo_nodes, lo_nbrs.reshape(-1), 1,
                          seeds, L, 1, True)
    assert np.array_equal(upd.phi_vec[:nn].cpu().numpy().view(np.uint32), want.view(np.uint32))
    got_rows = blk[torch.from_numpy(nodes_h.astype(np.int64)).to(ctx.device)].cpu().numpy()
    assert np.allclose(got_rows.astype(np.float64).sum(1), 1.0, atol=1e-5)
