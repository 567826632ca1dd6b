"""GPU tests of the callers either side of the kernels: device mini-batch sampler, Learner loop."""
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
    return ops, hostlib, learner


@pytest.fixture(scope="module")
def small_ds(env):
    ops, hostlib, learner = env
    N = 20000
    edges = hostlib.generate_graph(N, 16, 16, seed=7)
    return hostlib.Dataset.robust(N, edges, heldout_ratio=0.02, rand_seed=3)


def test_device_minibatch_sampler(env, small_ds):
    import torch
    ops, hostlib, learner = env
    ds = small_ds
    m = 2048
    p = ops.make_params(ds.N, 32, E=ds.E, num_node_sample=8)
    ctx = ops.Context(p)
    ts = ops.DeviceSet(ctx, ds.training.Serialize(), ds.training.BinsPerBucket(), ds.training.PrimeIdx())
    hs = ops.DeviceSet(ctx, ds.heldout.Serialize(), ds.heldout.BinsPerBucket(), ds.heldout.PrimeIdx())
    off, tgt = ds.training_csr()

    def make():
        return ops.DeviceMiniBatchSampler(ctx, off, tgt, ts, hs, m, seed=(1234, 5678), host_seed=99)

    a, b = make(), make()
    e = ctx.empty((ds.max_edges(m),), torch.int64)
    v = ctx.empty((ds.max_nodes(m),), torch.int32)
    e2, v2 = torch.empty_like(e), torch.empty_like(v)
    saw = set()
    for it in range(12):
        ne, nv, w = a("Node", e, v)
        ne2, nv2, w2 = b("Node", e2, v2)
        torch.cuda.synchronize()
        assert (ne, nv, w) == (ne2, nv2, w2)
        eh = e[:ne].cpu().numpy().view(np.uint64)
        vh = v[:nv].cpu().numpy().view(np.uint32)
        assert np.array_equal(eh, e2[:ne].cpu().numpy().view(np.uint64))  # deterministic in (seeds, u)
        assert np.array_equal(vh, v2[:nv].cpu().numpy().view(np.uint32))
        u = vh[0]
        lo, hi = eh >> np.uint64(32), eh & np.uint64(0xFFFFFFFF)
        assert (lo < hi).all() and np.unique(eh).size == ne
        assert ((lo == u) | (hi == u)).all()                      # all edges share the end point u
        other = np.where(lo == u, hi, lo).astype(np.uint32)
        assert np.array_equal(other, vh[1:]) and np.unique(vh).size == nv  # node list = {u} + distinct v
        link = ds.training.Has(eh)
        if link.all():
            saw.add("link")
            assert ne == off[u + 1] - off[u] and w == float(ds.N)
            assert sorted(other.tolist()) == sorted(tgt[off[u]:off[u + 1]].tolist())
        else:
            saw.add("non")
            assert ne == m and nv == m + 1 and not link.any() and not ds.heldout.Has(eh).any()
            assert w == float(np.float32(2 * ds.E) / np.float32(m))
            assert int(a.count[0].cpu()) >= m and int(a.count[1].cpu()) == 0  # enough distinct candidates survived
    assert saw == {"link", "non"}
    # uniformity of the non-link partner (chi-square on 16 buckets over several draws)
    hist = np.zeros(16)
    for it in range(8):
        a("NodeNonLink", e, v)
        torch.cuda.synchronize()
        hist += np.bincount(v[1:m + 1].cpu().numpy().view(np.uint32) * 16 // ds.N, minlength=16)
    exp = hist.sum() / 16
    assert ((hist - exp) ** 2 / exp).sum() < 60.0


@pytest.mark.parametrize("device_sampling", [False, True])
def test_learner_runs_and_learns(env, small_ds, device_sampling):
    import torch
    ops, hostlib, learner = env
    cfg = learner.Config.from_cli_defaults(K=32, mini_batch_size=512, num_node_sample=16, phi_wg_size=32,
                                           beta_wg_size=32, ppx_wg_size=32, device_sampling=device_sampling)
    lrn = learner.Learner(cfg, small_ds)
    p0 = lrn.HeldoutPerplexity()
    lrn.Run(300)
    p1 = lrn.HeldoutPerplexity()
    lrn.drain()
    pi = lrn.pi.host()
    assert np.isfinite(pi).all() and np.allclose(pi.sum(1), 1.0, atol=1e-4)
    beta = lrn.beta.cpu().numpy()
    assert np.isfinite(beta).all() and np.allclose(beta[0::2] + beta[1::2], 1.0, atol=1e-6)
    assert np.isfinite(p0) and np.isfinite(p1) and p1 < p0, (p0, p1)  # the sampler actually learns
    assert lrn.stepCount == 301 and lrn.phiUpdater.count_calls == 300 and lrn.betaUpdater.count_calls == 300
    lrn.close()


def test_learner_is_deterministic(env, small_ds):
    # serialize-test.cc:121-132's contract minus the file round trip: same seeds => bit-identical run
    ops, hostlib, learner = env
    out = []
    for _ in range(2):
        cfg = learner.Config.from_cli_defaults(K=64, mini_batch_size=256, num_node_sample=8, phi_wg_size=64,
                                               beta_wg_size=64, ppx_wg_size=64)
        lrn = learner.Learner(cfg, small_ds)
        lrn.Run(10)
        a = lrn.HeldoutPerplexity()
        lrn.Run(10)
        b = lrn.HeldoutPerplexity()
        lrn.drain()
        out.append((a, b, lrn.pi.host().copy(), lrn.theta.cpu().numpy().copy()))
        lrn.close()
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3])


def test_cpp_dropin_learner(env):
    """tests/cpp/learner_test.cc: a caller written against the reference's C++ API (Config, Graph,
    GenerateSetsFromEdges, clcuda::Queue, Learner::Run / HeldoutPerplexity) runs on the HIP path."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mcmc-ammsb-gpu_amd", "learner_test")
    assert os.path.exists(exe), "build() did not produce learner_test"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout


@pytest.mark.parametrize("device_sampling", [False, True])
def test_checkpoint_end_to_end(env, small_ds, device_sampling):
    """serialize-test.cc:90-134 on the HIP path: run, Serialize, run == fresh learner, Parse, run."""
    import io
    ops, hostlib, learner = env
    iters = 25

    def cfg():
        return learner.Config.from_cli_defaults(K=64, mini_batch_size=512, num_node_sample=16, heldout_ratio=0.02,
                                                phi_wg_size=64, beta_wg_size=64, ppx_wg_size=64,
                                                device_sampling=device_sampling)
    out = io.BytesIO()
    l1 = learner.Learner(cfg(), small_ds)
    l1.Run(iters)
    assert l1.Serialize(out)
    l1.Run(iters)
    ppx = l1.HeldoutPerplexity()
    pi1 = l1.pi.host()
    theta1 = ops.to_numpy(l1.theta)
    l1.close()
    l2 = learner.Learner(cfg(), small_ds)
    assert l2.Parse(io.BytesIO(out.getvalue()))
    l2.Run(iters)
    assert l2.HeldoutPerplexity() == ppx
    assert np.array_equal(l2.pi.host(), pi1) and np.array_equal(ops.to_numpy(l2.theta), theta1)
    l2.close()


def test_checkpoint_cpp_python_round_trip(env, tmp_path):
    """One stream format for both hosts: the C++ Learner's checkpoint resumes in the Python learner and
    the other way round, with bit-identical perplexities after 40 more iterations."""
    import os
    import subprocess
    ops, hostlib, learner = env
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mcmc-ammsb-gpu_amd", "learner_test")
    d = str(tmp_path)
    out = subprocess.run([exe, "ckpt", d], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr
    edges = np.fromfile(os.path.join(d, "edges.bin"), dtype=np.uint64)
    ds = hostlib.Dataset(20000, edges, heldout_ratio=0.05, rand_seed=12345)

    def cfg():  # tests/cpp/learner_test.cc FillConfig
        return learner.Config.from_cli_defaults(K=64, mini_batch_size=256, num_node_sample=16, heldout_ratio=0.05,
                                                phi_wg_size=64, beta_wg_size=64, ppx_wg_size=64)

    def bits(path):
        return int(open(os.path.join(d, path)).read())

    def f32bits(x):
        return int(np.float32(x).view(np.uint32))
    l = learner.Learner(cfg(), ds)
    with open(os.path.join(d, "cpp.ckpt"), "rb") as f:
        assert l.Parse(f)
        assert f.read() == b""  # every byte of the C++ stream is consumed
    l.Run(40)
    assert f32bits(l.HeldoutPerplexity()) == bits("cpp_ppx.txt")
    l.close()
    # the other direction
    l = learner.Learner(cfg(), ds)
    l.Run(40)
    with open(os.path.join(d, "py.ckpt"), "wb") as f:
        l.Serialize(f)
    l.Run(40)
    want = f32bits(l.HeldoutPerplexity())
    l.close()
    # byte-level: same state at the same step => same stream, except the wall-clock fields
    a = open(os.path.join(d, "cpp.ckpt"), "rb").read()
    b = open(os.path.join(d, "py.ckpt"), "rb").read()
    assert abs(len(a) - len(b)) <= 64
    out = subprocess.run([exe, "resume", d], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr
    assert bits("cpp_from_py_ppx.txt") == want


@pytest.mark.parametrize("async_launch", [False, True, "graph"])
def test_cpp_checkpoint_with_device_sampling(env, tmp_path, async_launch):
    """serialize-test.cc:90-134 for the C++ Learner with Config::device_sampling: the trailing extension record
    (host generator, batch sizes, candidate streams) makes the resumed run bit-identical."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mcmc-ammsb-gpu_amd", "learner_test")
    out = subprocess.run([exe, "ckpt", str(tmp_path)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, AMMSB_TEST_DEVICE_SAMPLING="1", **({"AMMSB_TEST_ASYNC": "1"} if async_launch else {}),
                                  **({"AMMSB_TEST_GRAPH": "1"} if async_launch == "graph" else {})))
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_c1_learner_matches_oracle_learner(env, orc):
    """BASELINE configs[0] (N = 10k, K = 32, mini-batch 1024, n = 32): the whole iteration sequence -- host rand_r
    mini-batches, neighbour sampling, update_phi, update_pi, beta gradient, theta step, perplexity -- run by the same
    Learner once over the HIP operators and once over the oracle-backed CPU operators (tests/oracle_ops.py)."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_ops
    ops, hostlib, learner = env
    N, K, m, n = 10000, 32, 1024, 32
    edges = hostlib.generate_graph(N, 32, 32, seed=20260101)
    ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)

    def cfg():
        return learner.Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, strategy="Node",
                                                phi_wg_size=32, beta_wg_size=32, ppx_wg_size=32)
    dev = learner.Learner(cfg(), ds)
    cpu = learner.Learner(cfg(), ds, ops=oracle_ops)
    assert np.array_equal(dev.pi.host(), cpu.pi.host())                      # pi_0 from the device gamma streams
    assert np.array_equal(ops.to_numpy(dev.theta), cpu.theta.numpy())
    p_dev, p_cpu = dev.HeldoutPerplexity(), cpu.HeldoutPerplexity()
    assert abs(p_dev - p_cpu) <= 1e-5 * p_cpu
    for it in range(1, 7):
        dev.Run(1)
        cpu.Run(1)
        dev.drain()
        a, b = dev.pi.host(), cpu.pi.host()
        if it == 1:
            assert np.array_equal(a, b)                                       # phi / pi are bit-identical given beta
            assert np.array_equal(ops.to_numpy(dev.phi), cpu.phi.numpy())
        else:   # beta differs in the last bits (order-free gradient sum), and so does everything after it
            assert np.abs(a.astype(np.float64) - b).max() <= 2e-6
        t_dev, t_cpu = ops.to_numpy(dev.theta).astype(np.float64), cpu.theta.numpy().astype(np.float64)
        assert np.abs(t_dev - t_cpu).max() <= 1e-5 * np.abs(t_cpu).max(), it
        assert dev.edges_done == cpu.edges_done                               # identical mini-batches
    p_dev, p_cpu = dev.HeldoutPerplexity(), cpu.HeldoutPerplexity()
    assert abs(p_dev - p_cpu) <= 1e-5 * p_cpu
    dev.close(), cpu.close()
