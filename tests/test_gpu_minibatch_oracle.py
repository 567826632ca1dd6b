"""Row f1: the device mini-batch sampler against (i) a serial restatement of its own algorithm
(oracle/ammsb_oracle_samplers.c, "The DEVICE mini-batch sampler"), bit for bit -- edges, node list, count, padding and the
advanced stream states, for Node / NodeLink / NodeNonLink at m = 1024 and 65536; and (ii) the reference-exact HOST sampler
(host/sample.cc == oracle restatement of sample.cc:249-303) on the same graph, as a two-sample test: mini-batch sizes,
weights, link share and the distribution of the non-link partners."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import __graft_entry__ as ge
    ge.build()
    from mcmc_ammsb_gpu_amd import hostlib, ops
    import oracle_lib as orc
    orc.build()
    return ops, hostlib, orc


def _setup(env, N, deg, m, seed=7, heldout_ratio=0.02):
    import torch
    ops, hostlib, orc = env
    edges = hostlib.generate_graph(N, 16, deg, seed=seed)
    ds = hostlib.Dataset.robust(N, edges, heldout_ratio=heldout_ratio, rand_seed=3)
    p = ops.make_params(ds.N, 32, E=ds.E, num_node_sample=8)
    ctx = ops.Context(p)
    ts = ops.DeviceSet(ctx, ds.training.Serialize(), ds.training.BinsPerBucket(), ds.training.PrimeIdx())
    hs = ops.DeviceSet(ctx, ds.heldout.Serialize(), ds.heldout.BinsPerBucket(), ds.heldout.PrimeIdx())
    off, tgt = ds.training_csr()
    he = np.ascontiguousarray(ds.heldout_edges, dtype=np.uint64)
    he = he[ds.heldout.Has(he)] if he.size else he
    ends = np.concatenate([he >> np.uint64(32), he & np.uint64(0xFFFFFFFF)]).astype(np.int64)
    hdeg = np.bincount(ends, minlength=ds.N)[:ds.N]
    smp = ops.DeviceMiniBatchSampler(ctx, off, tgt, ts, hs, m, seed=(1234, 5678), host_seed=99, heldout_degree=hdeg)
    e = ctx.empty((ds.max_edges(m),), torch.int64)
    v = ctx.empty((ds.max_nodes(m),), torch.int32)
    return ds, ctx, smp, off, tgt, e, v


class _Img:
    """a host cuckoo image as the oracle's set argument (slots, num_bins, prime_idx)"""

    def __init__(self, hset):
        self.slots = np.ascontiguousarray(hset.Serialize(), dtype=np.uint64)
        self.num_bins = int(hset.BinsPerBucket())
        self.prime_idx = int(hset.PrimeIdx())


@pytest.mark.parametrize("N,deg,m", [(20000, 16, 1024), (200000, 12, 65536)])
def test_device_sampler_equals_its_serial_statement(env, N, deg, m):
    import torch
    ops, hostlib, orc = env
    ds, ctx, smp, off, tgt, e, v = _setup(env, N, deg, m)
    tr, ho = _Img(ds.training), _Img(ds.heldout)
    seeds = orc.rng_init_mixed(smp.C, 1234, 5678)
    assert np.array_equal(smp.rand.host().view(np.uint64), seeds.view(np.uint64))  # ammsb_rng_init_mixed
    seen = {"link": 0, "non": 0}
    plan = ["Node"] * 10 + ["NodeNonLink"] * 3 + ["NodeLink"] * 3
    for it, strategy in enumerate(plan):
        choice = smp.choose(strategy)
        link, u, n, n_cand = choice
        ne, nv, w = smp.enqueue(choice, e, v)
        torch.cuda.synchronize()
        eh = e[:ne].cpu().numpy().view(np.uint64)
        vh = v[:nv].cpu().numpy().view(np.uint32)
        assert w == orc.device_minibatch_weight(link, ds.N, ds.E, m)
        if link:
            seen["link"] += 1
            we, wv, wn = orc.device_minibatch_link(off, tgt, u)
            assert (ne, nv) == (wn, wn + 1) and np.array_equal(eh, we) and np.array_equal(vh, wv)
        else:
            seen["non"] += 1
            we, wv, cnt = orc.device_minibatch_nonlink(seeds, n_cand, u, m, ds.N, tr, ho)
            assert (ne, nv) == (m, m + 1)
            assert np.array_equal(eh, we), "edges differ at mini-batch %d" % it
            assert np.array_equal(vh, wv)
            assert int(smp.count[0].cpu()) == cnt and cnt >= m
            # the candidate streams advanced exactly as stated (one draw for each of the call's candidates)
            assert np.array_equal(smp.rand.host().view(np.uint64), seeds.view(np.uint64))
    assert seen["link"] >= 3 and seen["non"] >= 3
    assert int(smp.count[1].cpu()) == 0


def test_device_sampler_shortfall_is_padded_as_stated_and_counted(env):
    """Too few candidate draws (the C ABI lets a caller ask for any multiple of 256 >= m): fewer than m distinct valid
    partners survive; the tail repeats earlier entries, count_out[0] < m and the sticky counter moves."""
    import torch
    ops, hostlib, orc = env
    m = 1024
    ds, ctx, smp, off, tgt, e, v = _setup(env, 4096, 8, m)
    tr, ho = _Img(ds.training), _Img(ds.heldout)
    seeds = orc.rng_init_mixed(smp.C, 1234, 5678)
    u = 17
    lib = ctx.lib
    ctx.check(lib.ammsb_minibatch_nonlink(ctx.handle, C.c_void_p(smp.rand.seeds.data_ptr()), m, smp.C, u, m,
                                          C.byref(smp.training_set.desc), C.byref(smp.heldout_set.desc),
                                          C.c_void_p(smp.workspace.data_ptr()), C.c_void_p(e.data_ptr()),
                                          C.c_void_p(v.data_ptr()), C.c_void_p(smp.count.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    we, wv, cnt = orc.device_minibatch_nonlink(seeds, m, u, m, ds.N, tr, ho)
    assert 0 < cnt < m  # 1024 draws out of 4096 values collide
    cnt_dev = smp.count.cpu().numpy()
    assert int(cnt_dev[0]) == cnt and int(cnt_dev[1]) == 1
    assert np.array_equal(e[:m].cpu().numpy().view(np.uint64), we)
    assert np.array_equal(v[:m + 1].cpu().numpy().view(np.uint32), wv)
    # the de-duplication table is empty again: the next call sees the state the statement assumes
    ctx.check(lib.ammsb_minibatch_nonlink(ctx.handle, C.c_void_p(smp.rand.seeds.data_ptr()), smp.C, smp.C, u, m,
                                          C.byref(smp.training_set.desc), C.byref(smp.heldout_set.desc),
                                          C.c_void_p(smp.workspace.data_ptr()), C.c_void_p(e.data_ptr()),
                                          C.c_void_p(v.data_ptr()), C.c_void_p(smp.count.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    we, wv, cnt = orc.device_minibatch_nonlink(seeds, smp.C, u, m, ds.N, tr, ho)
    assert cnt >= m and np.array_equal(e[:m].cpu().numpy().view(np.uint64), we)


def _ks_two_sample(a, b):
    """two-sample Kolmogorov-Smirnov statistic and its asymptotic p-value"""
    from scipy import stats
    r = stats.ks_2samp(a, b)
    return float(r.statistic), float(r.pvalue)


def test_device_sampler_matches_the_host_sampler_in_distribution(env):
    """sample.cc:249-303 (host, reference-exact rand_r stream) vs the device sampler on the same graph, 1200 mini-batches
    each: share of link batches (a fair coin, sample.cc:297), weights (sample.cc:268,292), |E_mb| of link batches (the
    degree of a uniformly drawn vertex that has edges), |E_mb| = m for non-link batches, and the non-link partners --
    uniform over the vertices that are valid partners of u -- compared by a two-sample KS test on the pooled partners."""
    import torch
    ops, hostlib, orc = env
    m, B = 256, 1200
    ds, ctx, smp, off, tgt, e, v = _setup(env, 20000, 16, m)
    # ---- host sampler (bit-identical to the plain-C restatement of the reference: tests/test_oracle_samplers.py)
    seed = 12345
    h_link_sizes, h_partners, h_links, h_w = [], [], 0, set()
    for _ in range(B):
        eh, vh, w, seed = ds.sample(m, "Node", seed)
        lo, hi = (eh >> np.uint64(32)).astype(np.int64), (eh & np.uint64(0xFFFFFFFF)).astype(np.int64)
        is_link = bool(ds.training.Has(eh[:1])[0])
        h_w.add((is_link, w))
        if is_link:
            h_links += 1
            h_link_sizes.append(eh.size)
        else:
            assert eh.size == m
            # the shared end point: the vertex every edge has (sample.cc:275-289 allows v == u; such an edge has lo == hi)
            cand = np.intersect1d(np.array([lo[0], hi[0]]), np.array([lo[1], hi[1]]))
            u = int(cand[0])
            h_partners.append(np.where(lo == u, hi, lo))
    # ---- device sampler
    d_link_sizes, d_partners, d_links, d_w = [], [], 0, set()
    for _ in range(B):
        ne, nv, w = smp("Node", e, v)
        torch.cuda.synchronize()
        vh = v[:nv].cpu().numpy().view(np.uint32)
        is_link = ne != m or bool(ds.training.Has(e[:1].cpu().numpy().view(np.uint64))[0])
        d_w.add((is_link, w))
        if is_link:
            d_links += 1
            d_link_sizes.append(ne)
        else:
            d_partners.append(vh[1:].astype(np.int64))
    smp.check()
    # the coin: both are Binomial(B, 1/2); |difference| of two independent ones has sd sqrt(B / 2)
    assert abs(h_links - B / 2) < 4.5 * np.sqrt(B / 4) and abs(d_links - B / 2) < 4.5 * np.sqrt(B / 4)
    assert abs(h_links - d_links) < 4.5 * np.sqrt(B / 2)
    # the weights are the reference's two constants, identical floats on both sides
    assert h_w == d_w == {(True, float(np.float32(ds.N))), (False, float(np.float32(2 * ds.E) / np.float32(m)))}
    # link batches: |E_mb| = deg(u) for u uniform among the vertices with edges -- same distribution
    st, pv = _ks_two_sample(np.array(h_link_sizes), np.array(d_link_sizes))
    assert pv > 1e-3, ("link batch sizes", st, pv)
    deg = np.diff(off.astype(np.int64))
    exp_mean = deg[deg > 0].mean()
    for sizes in (h_link_sizes, d_link_sizes):
        assert abs(np.mean(sizes) - exp_mean) < 5 * deg[deg > 0].std() / np.sqrt(len(sizes))
    # non-link partners: uniform over valid partners on both sides
    hp, dp = np.concatenate(h_partners), np.concatenate(d_partners)
    st, pv = _ks_two_sample(hp, dp)
    assert pv > 1e-3, ("non-link partners", st, pv)
    for part in (hp, dp):  # and each against the uniform law itself (the invalid partners are a 1e-3 fraction)
        from scipy import stats
        r = stats.kstest(part / float(ds.N), "uniform")
        assert r.statistic < 0.01, r
    # within one mini-batch the device sampler's partners are distinct and never u; the host's are distinct edges
    for part in d_partners[:50]:
        assert np.unique(part).size == part.size
