"""The captured-graph iteration (include/ammsb.h ammsb_loop) against the eager launch-by-launch loop: same
kernels, same arguments, same per-stream order => bit-identical trajectories.  Plus the device mini-batch
sampler's guarantees that the graph path relies on (per-vertex candidate counts, visible shortfalls)."""
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
    from mcmc_ammsb_gpu_amd import hostlib, learner, ops
    return ops, hostlib, learner, torch


@pytest.fixture(scope="module")
def small_ds(env):
    ops, hostlib, learner, torch = env
    N = 20000
    edges = hostlib.generate_graph(N, 16, 16, seed=7)
    return hostlib.Dataset.robust(N, edges, heldout_ratio=0.02, rand_seed=3)


def _state(ops, lrn):
    lrn.drain()
    return dict(pi=lrn.pi.host().copy(), phi=ops.to_numpy(lrn.phi).copy(), theta=ops.to_numpy(lrn.theta).copy(),
                beta=ops.to_numpy(lrn.beta).copy(), phi_seeds=lrn.phiUpdater.rand.host(),
                beta_seeds=lrn.betaUpdater.rand.host(), mb_seeds=lrn.dev_sampler.rand.host(),
                nbr_seeds=[s.neighbor_sampler.rand.host() for s in lrn.samples],
                step=lrn.stepCount, edges=lrn.edges_done, phase=lrn.phase,
                calls=(lrn.phiUpdater.count_calls, lrn.betaUpdater.count_calls))


def _same(a, b):
    for k in a:
        if isinstance(a[k], np.ndarray):
            assert np.array_equal(a[k], b[k]), k
        elif isinstance(a[k], list):
            assert all(np.array_equal(x, y) for x, y in zip(a[k], b[k])), k
        else:
            assert a[k] == b[k], (k, a[k], b[k])


@pytest.mark.parametrize("K,wg,m,strategy", [(64, 64, 256, "Node"),      # register kernels (one column per lane)
                                              (256, 64, 512, "Node"),     # LDS-streamed kernels <4, 1> (+ fused update_pi)
                                              (256, 64, 512, "NodeLink"),  # link batches only: device-read sizes
                                              (32, 32, 1024, "NodeNonLink"),
                                              (96, 64, 256, "Node"),      # K not a multiple of the work-group: column guards
                                              (128, 64, 256, "NodeLink"),  # two columns per lane, register kernels
                                              (512, 64, 300, "Node"),     # <8, 1>: update_pi folded into the gradient
                                              (256, 32, 512, "Node"),     # the reference's default wg: 32 virtual lanes per wave, fused
                                              (512, 32, 300, "Node"),
                                              (1024, 32, 200, "Node"),    # <16, 1, ..., 32>, separate update_pi
                                              (2048, 32, 150, "Node"),    # phi <32, 1, 2, 1, 32>, generic gradient kernel
                                              (1024, 64, 200, "Node")])   # <16, 1>: the C3 kernels (fusion opt-in, below)
def test_graph_loop_equals_eager_loop(env, small_ds, K, wg, m, strategy):
    ops, hostlib, learner, torch = env

    def make(graph):
        cfg = learner.Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=16, strategy=strategy,
                                               phi_wg_size=wg, beta_wg_size=wg, ppx_wg_size=wg, device_sampling=True,
                                               graph_launch=graph)
        return learner.Learner(cfg, small_ds)
    eager, graph = make(False), make(True)
    assert eager.loop is None and graph.loop is not None
    for n in (1, 2, 37):
        eager.Run(n)
        graph.Run(n)
        _same(_state(ops, eager), _state(ops, graph))
        assert eager.HeldoutPerplexity() == graph.HeldoutPerplexity()
    eager.close(), graph.close()


def test_loop_launch_forms_agree(env, small_ds, monkeypatch):
    """The loop's forms (read from the environment when the loop is created): chains launched directly from two host
    threads with the device-side hand-over (default), captured graphs replayed from two threads / from one, and
    stream events instead of polling kernels (the form for kernel-serialising profilers) -- one trajectory."""
    ops, hostlib, learner, torch = env

    def run(env_vars):
        for k in ("AMMSB_LOOP_LAUNCH", "AMMSB_LOOP_HANDSHAKE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env_vars.items():
            monkeypatch.setenv(k, v)
        cfg = learner.Config.from_cli_defaults(K=64, mini_batch_size=512, num_node_sample=32, strategy="Node",
                                               phi_wg_size=64, beta_wg_size=64, ppx_wg_size=64, device_sampling=True,
                                               graph_launch=True)
        lrn = learner.Learner(cfg, small_ds)
        for n in (150, 3, 64):   # 150 and 64 go through the two-thread path (>= 32 steps), 3 through the short one
            lrn.Run(n)
        st = _state(ops, lrn)
        ppx = lrn.HeldoutPerplexity()
        lrn.close()
        return st, ppx

    ref, ref_ppx = run({})
    for form in ({"AMMSB_LOOP_LAUNCH": "graph"}, {"AMMSB_LOOP_LAUNCH": "serial"}, {"AMMSB_LOOP_HANDSHAKE": "event"}):
        st, ppx = run(form)
        _same(ref, st)
        assert ppx == ref_ppx, form


@pytest.mark.parametrize("fuse,wg", [("", 64), ("", 32), ("1", 64), ("0", 64)])
def test_fused_update_pi_at_k1024(env, small_ds, monkeypatch, fuse, wg):
    """K = 1024 in the loop: by default update_pi is folded into the gradient kernel (since round 4; AMMSB_LOOP_FUSE_PI=1
    keeps the fusion to K <= 512, =0 turns it off; the flag is read once per process, so each setting runs in a child):
    the loop equals the eager loop bit for bit in every setting, at wg 64 and at the reference's default wg 32."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, __graft_entry__ as ge; ge.build()\n"
        "from mcmc_ammsb_gpu_amd import hostlib, learner, ops\n"
        "ds = hostlib.Dataset.robust(20000, hostlib.generate_graph(20000, 16, 16, seed=7), heldout_ratio=0.02, rand_seed=3)\n"
        "def make(g):\n"
        "    cfg = learner.Config.from_cli_defaults(K=1024, mini_batch_size=200, num_node_sample=16, strategy='Node', phi_wg_size=%d,\n"
        "                                           beta_wg_size=%d, ppx_wg_size=%d, device_sampling=True, graph_launch=g)\n" % (wg, wg, wg) +
        "    return learner.Learner(cfg, ds)\n"
        "a, b = make(False), make(True)\n"
        "for n in (1, 2, 37):\n"
        "    a.Run(n); b.Run(n); a.drain(); b.drain()\n"
        "    assert np.array_equal(a.pi.host(), b.pi.host()) and np.array_equal(ops.to_numpy(a.theta), ops.to_numpy(b.theta))\n"
        "    assert np.array_equal(ops.to_numpy(a.phi), ops.to_numpy(b.phi))\n"
        "print('OK', b.ctx.kernel_names()['beta_grads'])\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child_env = {k: v for k, v in os.environ.items() if k != "AMMSB_LOOP_FUSE_PI"}
    if fuse:
        child_env["AMMSB_LOOP_FUSE_PI"] = fuse
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=child_env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    # the kernel the loop's last gradient launch took: fused (third template argument true) exactly when asked for
    assert ("beta_grads_lds_kernel<16, 1, true, %d>" % wg in r.stdout) == (fuse == ""), r.stdout


def test_graph_and_eager_steps_interleave(env, small_ds):
    """Run() calls may alternate between the two forms on one learner (the pending mini-batch, the sample
    buffers' parity and every stream state carry over): eager 5 + graph 6 + eager 4 + graph 1 == eager 16."""
    ops, hostlib, learner, torch = env

    def make(graph):
        cfg = learner.Config.from_cli_defaults(K=128, mini_batch_size=384, num_node_sample=8, strategy="Node",
                                               phi_wg_size=64, beta_wg_size=64, ppx_wg_size=64, device_sampling=True,
                                               graph_launch=graph)
        return learner.Learner(cfg, small_ds)
    ref, mix = make(False), make(True)
    ref.Run(16)
    loop = mix.loop
    for n, use_graph in ((5, False), (6, True), (4, False), (1, True)):
        mix.loop = loop if use_graph else None
        mix.Run(n)
    mix.loop = loop
    _same(_state(ops, ref), _state(ops, mix))
    assert ref.HeldoutPerplexity() == mix.HeldoutPerplexity()
    ref.close(), mix.close()


def test_graph_loop_many_steps_and_chunks(env, small_ds):
    """More steps than one descriptor upload holds (1024) and than one Run() chunk (512)."""
    ops, hostlib, learner, torch = env

    def make(graph):
        cfg = learner.Config.from_cli_defaults(K=32, mini_batch_size=128, num_node_sample=8, strategy="Node",
                                               phi_wg_size=32, beta_wg_size=32, ppx_wg_size=32, device_sampling=True,
                                               graph_launch=graph)
        return learner.Learner(cfg, small_ds)
    eager, graph = make(False), make(True)
    graph.GRAPH_CHUNK = 1500  # one ammsb_loop_run call spanning two descriptor uploads
    eager.Run(2100)
    graph.Run(2100)
    _same(_state(ops, eager), _state(ops, graph))
    eager.close(), graph.close()


def test_graph_loop_timestamps(env, small_ds):
    ops, hostlib, learner, torch = env
    cfg = learner.Config.from_cli_defaults(K=256, mini_batch_size=2048, num_node_sample=16, strategy="NodeNonLink",
                                           phi_wg_size=64, beta_wg_size=64, ppx_wg_size=64, device_sampling=True,
                                           graph_launch=True, graph_timestamps=True)
    lrn = learner.Learner(cfg, small_ds)
    lrn.Run(20)
    b, e = lrn.loop.timestamps(1, 20)
    d = e - b
    assert (d > 0).all() and (d < 5e6).all()          # update_phi of 2049 nodes: microseconds, not milliseconds
    assert (b[1:] > e[:-1]).all()                     # steps run one after the other
    lrn.close()


def _sampler(env, ds, m):
    ops, hostlib, learner, torch = env
    p = ops.make_params(ds.N, 32, E=ds.E, num_node_sample=8)
    ctx = ops.Context(p)
    ts = ops.DeviceSet(ctx, ds.training.Serialize(), ds.training.BinsPerBucket(), ds.training.PrimeIdx())
    hs = ops.DeviceSet(ctx, ds.heldout.Serialize(), ds.heldout.BinsPerBucket(), ds.heldout.PrimeIdx())
    off, tgt = ds.training_csr()
    he = ds.heldout_edges[ds.heldout.Has(ds.heldout_edges)]
    hdeg = np.bincount(np.concatenate([he >> np.uint64(32), he & np.uint64(0xFFFFFFFF)]).astype(np.int64),
                       minlength=ds.N)
    smp = ops.DeviceMiniBatchSampler(ctx, off, tgt, ts, hs, m, seed=(1234, 5678), host_seed=5, heldout_degree=hdeg)
    e = ctx.empty((max(ds.max_edges(m), m),), torch.int64)
    v = ctx.empty((ds.max_nodes(m),), torch.int32)
    return ctx, smp, e, v


def test_hub_vertex_gets_enough_candidates(env):
    """a vertex with more neighbours than the sampler's margin (0.08 m + 1024) used to come up short
    and the mini-batch was silently padded with duplicates.  The candidate count now follows the vertex."""
    ops, hostlib, learner, torch = env
    N, m, hub = 40000, 2048, 17
    rng = np.random.default_rng(3)
    spokes = rng.permutation(N)[:6000]
    spokes = spokes[spokes != hub].astype(np.uint64)                   # degree ~6000 >> 0.08 * 2048 + 1024 = 1188
    star = (np.minimum(spokes, np.uint64(hub)) << np.uint64(32)) | np.maximum(spokes, np.uint64(hub))
    edges = np.unique(np.concatenate([star, hostlib.generate_graph(N, 16, 8, seed=11)]))
    rng.shuffle(edges)
    ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.02, rand_seed=3)
    ctx, smp, e, v = _sampler(env, ds, m)
    assert smp.excluded[hub] > 0.08 * m + 1024
    small = smp._candidates_for(1)
    mine = smp._candidates_for(smp.excluded[hub])
    assert small < mine <= smp.C
    ne, nv, w = smp.enqueue((0, hub, 0, mine), e, v)
    torch.cuda.synchronize()
    smp.check()
    assert int(smp.count[0]) >= m
    eh = e[:m].cpu().numpy().view(np.uint64)
    vh = v[:m + 1].cpu().numpy().view(np.uint32)
    assert vh[0] == hub and np.unique(vh).size == m + 1 and np.unique(eh).size == m   # no duplicate nodes
    assert not ds.training.Has(eh).any() and not ds.heldout.Has(eh).any()
    # with the old fixed count the same vertex does come up short -- and that is now an error, not a silent pad
    smp.enqueue((0, hub, 0, small), e, v)
    torch.cuda.synchronize()
    if int(smp.count[0]) < m:
        with pytest.raises(ops.AmmsbError):
            smp.check()
    smp.check()  # the sticky counter is cleared by the failed check
    ctx.close()


def test_short_minibatch_is_reported(env, small_ds):
    ops, hostlib, learner, torch = env
    m = 2048
    ctx, smp, e, v = _sampler(env, small_ds, m)
    smp.enqueue((0, 5, 0, m), e, v)      # m candidates can never give m distinct valid partners
    torch.cuda.synchronize()
    assert int(smp.count[0]) < m and int(smp.count[1]) == 1
    with pytest.raises(ops.AmmsbError):
        smp.check()
    ctx.close()


def test_device_sampler_statistics(env, small_ds):
    """The device sampler is distribution-equivalent to sample.cc:249-303, not stream-equivalent: check the
    distribution.  Non-link partners uniform over the valid vertices (256-bucket chi-square and a KS distance on the
    ids), the Node strategy's coin fair, link vertices uniform over the vertices that have an edge."""
    ops, hostlib, learner, torch = env
    ds, m = small_ds, 2048
    ctx, smp, e, v = _sampler(env, ds, m)
    N = ds.N
    ids = []
    for _ in range(24):
        smp.enqueue(smp.choose("NodeNonLink"), e, v)
        torch.cuda.synchronize()
        ids.append(v[1:m + 1].cpu().numpy().view(np.uint32).astype(np.int64))
    smp.check()
    ids = np.concatenate(ids)                                   # 49 152 partner ids
    hist = np.bincount(ids * 256 // N, minlength=256).astype(np.float64)
    exp = ids.size / 256.0
    chi2 = ((hist - exp) ** 2 / exp).sum()
    assert chi2 < 255 + 5 * np.sqrt(2 * 255), chi2              # chi-square(255): mean 255, sd 22.6
    srt = np.sort(ids) / float(N)
    ks = np.abs(srt - (np.arange(ids.size) + 0.5) / ids.size).max()
    assert ks < 1.95 / np.sqrt(ids.size), ks                    # KS critical value at alpha ~ 0.001
    # the coin and the vertex of 4000 Node-strategy choices
    picks = [smp.choose("Node") for _ in range(4000)]
    links = np.array([p[0] for p in picks])
    assert abs(links.mean() - 0.5) < 4.5 * 0.5 / np.sqrt(links.size)
    us = np.array([p[1] for p in picks if p[0] == 0])
    hu = np.bincount(us * 16 // N, minlength=16).astype(np.float64)
    assert ((hu - us.size / 16.0) ** 2 / (us.size / 16.0)).sum() < 15 + 5 * np.sqrt(30)
    deg = smp.degree
    assert all(deg[p[1]] > 0 and p[2] == deg[p[1]] for p in picks if p[0] == 1)   # sampleNodeLink's retry rule
    # non-link choices carry the candidate count of their vertex
    assert all(p[3] == smp._candidates_for(smp.excluded[p[1]]) for p in picks if p[0] == 0)
    ctx.close()
