"""-m gpu parity at the BASELINE.json workloads that had no test of their own in round 1:

  C2  configs[1]: N = 100k, K = 256, mini-batch 8192, n = 32 -- small enough for the oracle to redo a whole
      non-link and a whole link iteration, so every operator is compared in full (bit for bit where the
      contract is bit-exact), followed by six Learner iterations HIP-operators vs oracle-operators.
  C5  configs[4] at FULL size: N = 10M, K = 4096, average degree 64 (3.3e8-key cuckoo set, 164 GB pi in one
      allocation): one non-link update_phi / update_pi, one beta gradient, one perplexity pass, checked by
      size-independent properties and by the oracle on a subset (>= 64 nodes, 256 edges) that includes rows
      beyond 2^32 / K and keys from the full-size set.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FLOAT_TOL = 1e-5  # BASELINE.json north_star: 1e-5 relative on pi / beta / perplexity


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import __graft_entry__ as ge
    ge.build()
    from mcmc_ammsb_gpu_amd import hostlib, learner, ops
    return ops, hostlib, learner, torch


class _OSet:  # oracle-side view of a host-library table image
    def __init__(self, hs):
        self.slots, self.num_bins, self.prime_idx = hs.Serialize(), hs.BinsPerBucket(), hs.PrimeIdx()


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


# ------------------------------------------------------------------------------------------ C2

C2 = dict(N=100_000, K=256, m=8192, n=32, deg=32, k_true=64, L=64)


@pytest.fixture(scope="module")
def c2(env):
    ops, hostlib, learner, torch = env
    edges = hostlib.generate_graph(C2["N"], C2["k_true"], C2["deg"], seed=20260101)
    return hostlib.Dataset.robust(C2["N"], edges, heldout_ratio=0.01, rand_seed=1)


def test_c2_operators_full_compare(env, orc, c2):
    """Every operator of one non-link and one link iteration at C2, whole mini-batch, against the oracle."""
    ops, hostlib, learner, torch = env
    ds = c2
    N, K, m, n, L = C2["N"], C2["K"], C2["m"], C2["n"], C2["L"]
    p = ops.make_params(N, K, E=ds.E, num_node_sample=n)
    po = orc.make_params(N, K, n)
    ctx = ops.Context(p)
    pi = ops.RowPartitionedMatrix(ctx, N, K)
    phi_sum = ctx.zeros((N,), torch.float32)
    ops.RandomGammaAndNormalize(ctx, 1.0, 1.0, pi, phi_sum)
    pi_h, phi_h = orc.pi_init_gamma(N, K)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(pi.host()), _bits(pi_h)) and np.array_equal(_bits(phi_sum.cpu().numpy()), _bits(phi_h))
    theta_h = hostlib.theta_init(K)
    beta_h = np.zeros_like(theta_h)
    orc.lib().orc_beta_from_theta(theta_h, beta_h, K)
    theta, beta = ctx.from_numpy(theta_h), ctx.zeros((2 * K,), torch.float32)
    ops.beta_from_theta(ctx, theta, beta)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(beta.cpu().numpy()), _bits(beta_h))
    tset, hset = _OSet(ds.training), _OSet(ds.heldout)
    dts = ops.DeviceSet(ctx, tset.slots, tset.num_bins, tset.prime_idx)
    dhs = ops.DeviceSet(ctx, hset.slots, hset.num_bins, hset.prime_idx)
    max_nodes = ds.max_nodes(m)
    smp = ops.NeighborSampler(ctx, max_nodes, (56, 57), 32)
    s_seeds = orc.rng_init(max_nodes * 2 * n, 56, 57)
    phi = ops.PhiUpdater(ctx, beta, pi, phi_sum, dts, max_nodes, (42, 43), L)
    p_seeds = orc.rng_init(max_nodes * L, 42, 43)
    bu = ops.BetaUpdater(ctx, theta, beta, pi, dts, (44, 45), L)
    b_seeds = orc.rng_init(K, 44, 45)
    calc = ops.PerplexityCalculator(ctx, beta, pi, ctx.from_numpy(ds.heldout_edges), dhs, L)
    state = np.zeros(ds.heldout_edges.size, dtype=np.float32)
    seed = 1804289383
    for step, strategy in enumerate(("NodeNonLink", "NodeLink", "NodeNonLink"), start=1):
        mb, nodes_h, weight, seed = ds.sample(m, strategy, seed)   # the reference-exact host sampler
        ne, nv = mb.size, nodes_h.size
        assert (ne, nv) == ((m, m + 1) if strategy == "NodeNonLink" else (nv - 1, ne + 1))
        nodes, dev_edges = ctx.from_numpy(nodes_h), ctx.from_numpy(mb)
        # neighbour sampler: table image, packed result, stream states
        smp(nv, nodes)
        table, packed = orc.sample_neighbors(s_seeds, nodes_h, N, n, 32)
        torch.cuda.synchronize()
        assert np.array_equal(smp.GetData()[:nv].cpu().numpy().view(np.uint32), packed)
        assert np.array_equal(smp.GetHash()[:nv].cpu().numpy().view(np.uint32), table)
        assert np.array_equal(smp.rand.host(), s_seeds)
        # update_phi over the whole mini-batch: bit for bit, stream states included
        phi.count_calls += 1
        phi.update_phi(nodes, smp.GetData(), nv)
        want = orc.update_phi(po, beta_h, pi_h.reshape(-1), phi_h, tset, nodes_h, packed.reshape(-1), step, p_seeds,
                              L, 1, True)
        torch.cuda.synchronize()
        assert np.array_equal(_bits(phi.phi_vec[:nv].cpu().numpy()), _bits(want)), "update_phi step %d" % step
        assert np.array_equal(phi.rand.host(), p_seeds)
        # update_pi: pi rows and phi_sum bit for bit, nothing else touched
        phi.update_pi(nodes, nv)
        orc.update_pi(po, pi_h.reshape(-1), phi_h, want.reshape(-1), nodes_h, L, 1)
        torch.cuda.synchronize()
        assert np.array_equal(_bits(pi.host()), _bits(pi_h)), "update_pi step %d" % step
        assert np.array_equal(_bits(phi_sum.cpu().numpy()), _bits(phi_h))
        # beta gradient: the sum over edges (order-free contract), then the theta step bit for bit given it
        bu.count_calls += 1
        g = bu.calculate_grads(dev_edges, ne).cpu().numpy().copy()
        exact = orc.beta_grads(po, theta_h, beta_h, pi_h.reshape(-1), tset, mb, L, 1, order=1)
        ref = orc.beta_grads(po, theta_h, beta_h, pi_h.reshape(-1), tset, mb, L, 1, order=0)
        assert rel_err(g, exact) <= FLOAT_TOL
        assert rel_err(g, exact) <= max(2 * rel_err(ref, exact), 2e-7)
        ts = np.zeros(K, dtype=np.float32)
        orc.lib().orc_sum_theta(theta_h, ts, K)
        assert np.array_equal(_bits(bu.GetThetaSum().cpu().numpy()), _bits(ts))
        bu.update_theta(weight)
        beta_h = orc.update_theta(po, theta_h, g, step, weight, b_seeds)
        torch.cuda.synchronize()
        assert np.array_equal(_bits(theta.cpu().numpy()), _bits(theta_h))
        assert np.array_equal(_bits(beta.cpu().numpy()), _bits(beta_h))
        assert np.array_equal(bu.rand.host(), b_seeds)
        # perplexity over the whole held-out set: running-mean state bit for bit, sums, counts, value
        got = calc()
        sums, _ = orc.perplexity(po, beta_h, pi_h.reshape(-1), hset, ds.heldout_edges, step, L, 1, state)
        l0, l1, c0, c1 = calc.unpack(calc.sums)
        assert (c0, c1) == (sums.link_cnt, sums.nonlink_cnt) and c0 + c1 == ds.heldout_edges.size
        assert np.array_equal(_bits(calc.ppx_per_edge.cpu().numpy()), _bits(state))
        assert abs(l0 - sums.link_ll) <= 1e-10 * abs(sums.link_ll) and abs(l1 - sums.nonlink_ll) <= 1e-10 * abs(sums.nonlink_ll)
        want_ppx = -(sums.link_ll + sums.nonlink_ll) / (sums.link_cnt + sums.nonlink_cnt)
        assert abs(got - want_ppx) <= FLOAT_TOL * abs(want_ppx)
    ctx.close()


def test_c2_learner_matches_oracle_learner(env, orc, c2):
    """BASELINE configs[1]: six iterations of the same Learner code over the HIP operators and over the
    oracle-backed CPU operators, host rand_r mini-batches (the reference's stream)."""
    import oracle_ops
    ops, hostlib, learner, torch = env
    ds = c2

    def cfg():
        return learner.Config.from_cli_defaults(K=C2["K"], mini_batch_size=C2["m"], num_node_sample=C2["n"],
                                                strategy="Node", phi_wg_size=64, beta_wg_size=64, ppx_wg_size=64)
    dev = learner.Learner(cfg(), ds)
    cpu = learner.Learner(cfg(), ds, ops=oracle_ops)
    assert np.array_equal(dev.pi.host(), cpu.pi.host())
    assert np.array_equal(ops.to_numpy(dev.theta), cpu.theta.numpy())
    p_dev, p_cpu = dev.HeldoutPerplexity(), cpu.HeldoutPerplexity()
    assert abs(p_dev - p_cpu) <= FLOAT_TOL * p_cpu
    kinds, prev = set(), 0
    for it in range(1, 7):
        dev.Run(1)
        cpu.Run(1)
        dev.drain()
        kinds.add(dev.edges_done - prev == C2["m"])
        prev = dev.edges_done
        a, b = dev.pi.host(), cpu.pi.host()
        if it == 1:
            assert np.array_equal(a, b)          # phi / pi bit-identical given beta
            assert np.array_equal(ops.to_numpy(dev.phi), cpu.phi.numpy())
        else:                                    # beta differs in the last bits (order-free gradient sum)
            assert np.abs(a.astype(np.float64) - b).max() <= 2e-6
        t_dev, t_cpu = ops.to_numpy(dev.theta).astype(np.float64), cpu.theta.numpy().astype(np.float64)
        assert np.abs(t_dev - t_cpu).max() <= FLOAT_TOL * np.abs(t_cpu).max(), it
        b_dev, b_cpu = ops.to_numpy(dev.beta).astype(np.float64), cpu.beta.numpy().astype(np.float64)
        assert np.abs(b_dev - b_cpu).max() <= FLOAT_TOL * np.abs(b_cpu).max(), it
        assert dev.edges_done == cpu.edges_done  # identical mini-batches
    assert kinds == {True, False}, "six iterations should see both a link and a non-link mini-batch"
    p_dev, p_cpu = dev.HeldoutPerplexity(), cpu.HeldoutPerplexity()
    assert abs(p_dev - p_cpu) <= FLOAT_TOL * p_cpu
    dev.close(), cpu.close()


# ------------------------------------------------------------------------------------------ C5

C5 = dict(N=10_000_000, K=4096, m=65536, n=32, deg=64, k_true=64, L=256)


def _remapped_problem(orc, ops, torch, ctx, pi_block, phi_sum, rows_needed, key_pairs, host_set):
    """Compact the pi rows a subset needs into a small matrix with remapped ids (the oracle indexes pi by row id).
    key_pairs: [..., 2] original (a, b) vertex pairs whose membership the kernels will test; the oracle gets a set
    of the REMAPPED keys of exactly those pairs that are members of the full-size set."""
    rows = np.unique(np.asarray(rows_needed, dtype=np.int64))
    idx = torch.from_numpy(rows).to(ctx.device)
    sub_pi = pi_block[idx].cpu().numpy()
    sub_ps = phi_sum[idx].cpu().numpy()
    lut = {int(r): k for k, r in enumerate(rows)}
    remap = np.vectorize(lambda x: lut[int(x)], otypes=[np.uint32])
    kp = np.asarray(key_pairs, dtype=np.uint64).reshape(-1, 2)
    member = host_set.Has(orc.make_edge(kp[:, 0], kp[:, 1]))
    lk = orc.make_edge(remap(kp[:, 0]).astype(np.uint64), remap(kp[:, 1]).astype(np.uint64))
    oset = orc.OracleSet(np.unique(np.concatenate([lk[member], [np.uint64(2**63)]])))
    return rows, sub_pi, sub_ps, remap, oset, member


def test_c5_full_size(env, orc):
    ops, hostlib, learner, torch = env
    N, K, m, n, L = C5["N"], C5["K"], C5["m"], C5["n"], C5["L"]
    edges = hostlib.generate_graph(N, C5["k_true"], C5["deg"], seed=20260101)
    assert edges.size > 3.0e8
    ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
    del edges
    assert ds.training.Size() > 3.0e8                    # the 3.3e8-key cuckoo set
    cfg = learner.Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, strategy="Node",
                                           phi_wg_size=L, beta_wg_size=L, ppx_wg_size=L, device_sampling=True)
    lrn = learner.Learner(cfg, ds)
    ctx, pi, phi_sum = lrn.ctx, lrn.pi, lrn.phi
    assert len(pi.blocks) == 1 and pi.blocks[0].numel() == N * K > 2**35   # 164 GB in one allocation
    blk = pi.blocks[0]
    po = orc.make_params(N, K, n)
    beta_h, theta_h = ops.to_numpy(lrn.beta).copy(), ops.to_numpy(lrn.theta).copy()
    rng = np.random.default_rng(55)
    HI = 2**32 // K                                       # rows >= HI lie beyond the reference's 32-bit offsets

    # ---- one non-link mini-batch from the device sampler, neighbours from the neighbour sampler
    s = lrn.samples[0]
    ne, nv, weight = lrn.dev_sampler("NodeNonLink", s.dev_edges, s.dev_nodes)
    assert (ne, nv) == (m, m + 1)
    s.neighbor_sampler(nv, s.dev_nodes)
    torch.cuda.synchronize()
    assert int(lrn.dev_sampler.count[0].cpu()) >= m and int(lrn.dev_sampler.count[1].cpu()) == 0
    nodes_h = s.dev_nodes[:nv].cpu().numpy().view(np.uint32).copy()
    nbrs = s.neighbor_sampler.GetData()
    nbrs_h = nbrs[:nv].cpu().numpy().view(np.uint32).copy()
    assert np.unique(nodes_h).size == nv and (nodes_h >= HI).sum() > m // 2
    # give the first 48 nodes some TRUE neighbours (keys of the full-size set), so both branches of y run
    off, tgt = ds.training_csr()
    for i in range(48):
        mine = tgt[int(off[nodes_h[i]]):int(off[nodes_h[i] + 1])][:6]
        nbrs_h[i, :mine.size] = mine
    nbrs[:nv].copy_(ctx.from_numpy(nbrs_h))
    G = 65535
    pick = np.unique(np.concatenate([np.arange(48), [0, 1, G, G + 1], rng.integers(0, nv, 24)]))
    assert pick.size >= 64
    need = set(pick.tolist()) | {int(i % G) for i in pick}      # a group's earlier node advances its streams
    need = np.array(sorted(need))
    rows_needed = np.concatenate([nodes_h[need], nbrs_h[need].ravel()])
    assert (rows_needed >= HI).any()
    pairs = np.stack([np.repeat(nodes_h[need], n), nbrs_h[need].ravel()], 1)
    rows, sub_pi, sub_ps, remap, oset, member = _remapped_problem(orc, ops, torch, ctx, blk, phi_sum, rows_needed,
                                                                  pairs, ds.training)
    assert member.sum() >= 48                                   # real links were probed in the full-size set
    touched = np.zeros(N, dtype=bool)
    touched[nodes_h] = True
    rest = np.flatnonzero(~touched)[:: max(1, (N - nv) // 2048)]
    rest_t = torch.from_numpy(rest).to(ctx.device)
    rest_before, rest_ps = blk[rest_t].cpu().numpy(), phi_sum[rest_t].cpu().numpy()

    phi = lrn.phiUpdater
    phi.count_calls = 1
    phi.update_phi(s.dev_nodes, nbrs, nv)
    torch.cuda.synchronize()
    got = phi.phi_vec[:nv]
    assert bool(torch.isfinite(got).all()) and float(got.min()) >= 1e-24
    for i in pick:
        g = int(i % G)
        chain = [g] if i < G else [g, int(i)]
        seeds = orc.rng_init(L, 42 + g * L, 43 + g * L)
        for idx in chain:
            want = orc.update_phi(po, beta_h, sub_pi.reshape(-1), sub_ps, oset, remap(nodes_h[idx:idx + 1]),
                                  remap(nbrs_h[idx]), 1, seeds, L, 1, True)
        assert np.array_equal(_bits(got[int(i)].cpu().numpy()), _bits(want[0])), "node index %d" % i
    # shard invariance at full size (the multi-GPU contract): four group ranges == the single launch
    full = got.clone()
    seeds_after = phi.rand.seeds.clone()
    phi.rand.SetSeed(cfg.phi_seed)
    for lo, hi in ((0, 1), (1, 30000), (30000, 65534), (65534, 70000)):
        phi.update_phi(s.dev_nodes, nbrs, nv, lo, hi)
    torch.cuda.synchronize()
    assert torch.equal(phi.phi_vec[:nv], full) and torch.equal(phi.rand.seeds, seeds_after)
    del full

    # ---- the same launch at the reference's DEFAULT work-group size (32, main.cc:61): 128 columns per work-item at
    #      K = 4096 -- update_phi_gen_kernel at full C5 size (round 2 refused this shape), against the oracle at L = 32
    phi32 = ops.PhiUpdater(ctx, lrn.beta, pi, phi_sum, lrn.trainingSet, nv, (42, 43), 32)
    phi32.count_calls = 1
    phi32.update_phi(s.dev_nodes, nbrs, nv)
    torch.cuda.synchronize()
    assert "gen_kernel" in ctx.kernel_names()["update_phi"]
    got32 = phi32.phi_vec[:nv]
    assert bool(torch.isfinite(got32).all()) and float(got32.min()) >= 1e-24
    for i in pick[:40]:
        g = int(i % G)
        chain = [g] if i < G else [g, int(i)]
        seeds = orc.rng_init(32, 42 + g * 32, 43 + g * 32)
        for idx in chain:
            want = orc.update_phi(po, beta_h, sub_pi.reshape(-1), sub_ps, oset, remap(nodes_h[idx:idx + 1]),
                                  remap(nbrs_h[idx]), 1, seeds, 32, 1, True)
        assert np.array_equal(_bits(got32[int(i)].cpu().numpy()), _bits(want[0])), "wg 32, node index %d" % i
    del phi32, got32
    torch.cuda.empty_cache()

    # ---- update_pi: rows normalised in WG_SUM order, phi_sum = row sum, everything else untouched
    phi.update_pi(s.dev_nodes, nv)
    torch.cuda.synchronize()
    sub = pick[:32]
    for i in sub:
        row = got[int(i)].cpu().numpy().copy()
        sm = orc.lib().orc_wg_normalize_f32(row, K, L)
        r = int(nodes_h[i])
        assert np.array_equal(_bits(blk[r].cpu().numpy()), _bits(row)) and float(phi_sum[r]) == sm
    nd = torch.from_numpy(nodes_h.astype(np.int64)).to(ctx.device)
    sums = torch.zeros(nv, dtype=torch.float64, device=ctx.device)
    for a in range(0, nv, 8192):  # chunked: the gather of all 65537 rows at once would be another GB
        sums[a:a + 8192] = blk[nd[a:a + 8192]].double().sum(1)
    assert float((sums - 1.0).abs().max()) <= 1e-5
    assert np.array_equal(blk[rest_t].cpu().numpy(), rest_before) and np.array_equal(phi_sum[rest_t].cpu().numpy(), rest_ps)

    # ---- beta gradient over the mini-batch with 256 real links mixed in at the front
    mb = s.dev_edges[:ne].cpu().numpy().view(np.uint64).copy()
    mb[:256] = ds.training_edges[rng.integers(0, ds.training_edges.size, 256)]
    dev_mb = ctx.from_numpy(mb)
    bu = lrn.betaUpdater
    g_all = bu.calculate_grads(dev_mb, ne).cpu().numpy().astype(np.float64)
    assert np.isfinite(g_all).all()
    acc = np.zeros(2 * K)
    for lo, hi in ((0, 256), (256, 257), (257, 40000), (40000, m)):   # linearity over edge shards
        acc += bu.calculate_grads(dev_mb, ne, lo, hi).cpu().numpy().astype(np.float64)
    assert np.abs(acc - g_all).max() / np.abs(g_all).max() <= 3e-6
    # oracle on edges [0, 512): the 256 links (arbitrary rows of the 10M, keys of the full-size set) + 256 non-links
    sub_e = mb[:512]
    ea, eb = (sub_e >> np.uint64(32)).astype(np.uint32), (sub_e & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    assert (np.concatenate([ea, eb]) >= HI).any()
    rows, sub_pi, sub_ps, remap, oset, member = _remapped_problem(orc, ops, torch, ctx, blk, phi_sum,
                                                                  np.concatenate([ea, eb]), np.stack([ea, eb], 1),
                                                                  ds.training)
    assert member[:256].all() and not member[256:].any()
    g_sub = bu.calculate_grads(dev_mb, ne, 0, 512).cpu().numpy()
    lk = orc.make_edge(remap(ea).astype(np.uint64), remap(eb).astype(np.uint64))
    exact = orc.beta_grads(po, theta_h, beta_h, sub_pi.reshape(-1), oset, lk, L, 1, order=1)
    assert rel_err(g_sub, exact) <= FLOAT_TOL

    # ---- one perplexity pass over all 3.26 M held-out edges + the oracle on a 256-edge range
    calc = lrn.heldoutPerplexity
    H = calc.num_edges
    assert H == ds.heldout_edges.size > 3_000_000
    calc.count_calls = 1
    sums_t = calc.partial()
    l0, l1, c0, c1 = calc.unpack(sums_t)
    n_link = int(ds.heldout.Size())
    assert (c0, c1) == (n_link, H - n_link)              # every held-out edge counted once, on the right side
    per_edge = calc.ppx_per_edge.cpu().numpy()
    assert np.isfinite(per_edge).all() and (per_edge > 0).all() and (per_edge <= 1.0).all()
    assert np.isfinite(l0) and np.isfinite(l1) and l0 < 0 and l1 < 0
    lo = (H // 2) - 128                                   # straddles the link / non-link halves of the list
    he = ds.heldout_edges[lo:lo + 256]
    ha, hb = (he >> np.uint64(32)).astype(np.uint32), (he & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    rows, sub_pi, sub_ps, remap, oset, member = _remapped_problem(orc, ops, torch, ctx, blk, phi_sum,
                                                                  np.concatenate([ha, hb]), np.stack([ha, hb], 1),
                                                                  ds.heldout)
    lk = orc.make_edge(remap(ha).astype(np.uint64), remap(hb).astype(np.uint64))
    state = np.zeros(256, dtype=np.float32)
    osums, _ = orc.perplexity(po, beta_h, sub_pi.reshape(-1), oset, lk, 1, L, 1, state)
    assert np.array_equal(_bits(per_edge[lo:lo + 256]), _bits(state))
    # the same range as its own launch (edge_begin / edge_end): sums and counts of exactly those edges
    calc.ppx_per_edge[lo:lo + 256].zero_()
    r0, r1, k0, k1 = calc.unpack(calc.partial(lo, lo + 256))
    assert (k0, k1) == (osums.link_cnt, osums.nonlink_cnt) == (int(member.sum()), 256 - int(member.sum()))
    assert abs(r0 - osums.link_ll) <= 1e-10 * abs(osums.link_ll) and abs(r1 - osums.nonlink_ll) <= 1e-10 * abs(osums.nonlink_ll)
    # ---- the 3.2e8 training keys built into a table ON THE DEVICE (opt-in; the host build above took
    # most of this test's set-up time): exact membership on samples of members and of held-out / fake pairs
    import time
    keys_dev = ctx.from_numpy(ds.training_edges)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dset2 = ops.DeviceSet.build_on_device(ctx, keys_dev)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    print("device cuckoo build of %d keys: %.2f s (prime pair %d)" % (ds.training_edges.size, build_s, dset2.prime_idx))
    assert dset2.num_bins == ds.training.BinsPerBucket()
    sample = ds.training_edges[rng.integers(0, ds.training_edges.size, 2_000_000)]
    assert bool(dset2.Has(sample).all())
    assert np.array_equal(dset2.Has(ds.heldout_edges).cpu().numpy().astype(bool), ds.training.Has(ds.heldout_edges))
    assert int((dset2.data != -1).sum()) == ds.training_edges.size      # each key once, nothing else
    del dset2, keys_dev
    lrn.close()
