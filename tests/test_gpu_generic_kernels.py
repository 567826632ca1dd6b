"""The generic (any K, any work-group size) kernels against the specialised ones and the oracle.

The reference loops K_PER_THREAD = ceil(K / wg) columns per work-item generically (phi.cc:214-302,
beta.cc:145-233, perplexity.cc:93-157) and defaults every work-group size to 32 (main.cc:61-64), so
(K = 1024, wg = 32) and (K = 4096, wg = 32) are inputs it accepts.  The shapes that only the generic
kernels take are part of test_gpu_parity.py's case lists; here the generic form is forced
(AMMSB_{PHI,BETA,PPX}_FORM=g, read once per process, hence the child process) onto shapes the specialised
kernels also run, and both are held to the oracle bit for bit -- update_phi, update_pi, the partial-row
sum of the gradient, the per-edge perplexity state.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
import numpy as np
sys.path[:0] = [%(root)r, %(tests)r]
import torch
import ammsb_pkg
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import ops as hip
import oracle_lib as orc
orc.build()
from test_gpu_parity import Problem

def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)

# ---- update_phi / update_pi: (N, K, n, nodes, L)
for N, K, n, n_nodes, L in [(2048, 64, 8, 300, 64), (2048, 64, 8, 300, 32), (2048, 256, 32, 100, 64),
                            (2048, 256, 32, 100, 128), (2048, 256, 7, 100, 256), (1024, 1000, 5, 64, 64),
                            (1024, 1024, 32, 40, 64), (1024, 1024, 32, 40, 256), (512, 2048, 8, 20, 128),
                            (512, 2048, 8, 20, 512), (600, 100, 6, 50, 16),
                            (70000, 64, 4, 65535 + 700, 32)]:   # groups with a second node: the noise pre-pass walks both
    for noise in (False, True):
        pr = Problem(orc, hip, N, K, n, n_nodes)
        upd = hip.PhiUpdater(pr.ctx, pr.beta, pr.pi, pr.phi_sum, pr.dset, n_nodes, (42, 43), L,
                             phi_disable_noise=not noise)
        seeds = orc.rng_init(n_nodes * L, 42, 43)
        pi_h, phi_sum_h = pr.pi_h.copy(), pr.phi_sum_h.copy()
        for step in (1, 2):
            upd(pr.nodes, pr.nb, n_nodes)
            pr.sync()
            want = orc.update_phi(pr.p_orc, pr.beta_h, pi_h.reshape(-1), phi_sum_h, pr.oset, pr.nodes_h,
                                  pr.nb_h.reshape(-1), step, seeds, L, 1, noise)
            got = upd.phi_vec.cpu().numpy()[:n_nodes]
            assert np.array_equal(bits(got), bits(want)), ("phi_vec", N, K, n, L, noise, step)
            assert np.array_equal(upd.rand.host(), seeds), ("streams", K, L)
            orc.update_pi(pr.p_orc, pi_h.reshape(-1), phi_sum_h, want.reshape(-1), pr.nodes_h, L, 1)
            assert np.array_equal(bits(pr.pi.host()), bits(pi_h)), ("pi", K, L)
            assert np.array_equal(pr.phi_sum.cpu().numpy(), phi_sum_h), ("phi_sum", K, L)
        pr.ctx.close()
print("phi ok")

# ---- beta gradient: the partial rows are those of the specialised kernels, so the summed gradient must equal the
#      one a default-form process computes (passed in through a file) bit for bit, and agree with the oracle
ref = np.load(sys.argv[1])
for idx, (N, K, n_edges, L) in enumerate([(2048, 64, 500, 64), (2048, 64, 500, 32), (2048, 256, 3000, 128),
                                          (1024, 1000, 300, 256), (4096, 1024, 1024, 64), (1024, 2048, 300, 128)]):
    pr = Problem(orc, hip, N, K, 8, 16)
    rng = pr.rng
    non = orc.make_edge(rng.integers(0, N, n_edges), rng.integers(0, N, n_edges))
    mbe = np.concatenate([pr.edges[: n_edges // 3], non[: n_edges - n_edges // 3]]).astype(np.uint64)
    rng.shuffle(mbe)
    upd = hip.BetaUpdater(pr.ctx, pr.theta, pr.beta, pr.pi, pr.dset, (44, 45), L)
    upd.count_calls += 1
    g = upd.calculate_grads(pr.ctx.from_numpy(mbe), mbe.size).cpu().numpy().copy()
    exact = orc.beta_grads(pr.p_orc, pr.theta_h, pr.beta_h, pr.pi_h.reshape(-1), pr.oset, mbe, L, 1, order=1)
    err = np.abs(g.astype(np.float64) - exact).max() / np.abs(exact).max()
    assert err <= 1e-5, ("beta grads", K, L, err)
    assert np.array_equal(bits(g), bits(ref["g%%d" %% idx])), ("beta grads differ from the specialised kernel", K, L)
    ts = np.zeros(K, dtype=np.float32)
    orc.lib().orc_sum_theta(pr.theta_h, ts, K)
    assert np.array_equal(bits(upd.GetThetaSum().cpu().numpy()), bits(ts))
    pr.ctx.close()
print("beta ok")

# ---- perplexity
for N, K, L in [(1024, 1024, 64), (1024, 1024, 256), (2048, 96, 32), (2048, 1000, 128), (512, 2048, 64)]:
    pr = Problem(orc, hip, N, K, 8, 16)
    held = pr.edges[:611]
    hset = orc.OracleSet(held)
    fake = orc.make_edge(pr.rng.integers(0, N, 600), pr.rng.integers(0, N, 600))
    he = np.concatenate([held, fake[~hset.has(fake)]]).astype(np.uint64)
    dh = hip.DeviceSet(pr.ctx, hset.slots, hset.num_bins, hset.prime_idx)
    calc = hip.PerplexityCalculator(pr.ctx, pr.beta, pr.pi, pr.ctx.from_numpy(he), dh, L)
    state = np.zeros(he.size, dtype=np.float32)
    for call in (1, 2):
        got = calc()
        pr.sync()
        sums, _ = orc.perplexity(pr.p_orc, pr.beta_h, pr.pi_h.reshape(-1), hset, he, call, L, 1, state)
        assert np.array_equal(bits(calc.ppx_per_edge.cpu().numpy()), bits(state)), ("ppx state", K, L, call)
        l0, l1, c0, c1 = calc.unpack(calc.sums)
        assert (c0, c1) == (sums.link_cnt, sums.nonlink_cnt)
        assert abs(l0 - sums.link_ll) <= 1e-12 * abs(sums.link_ll) and abs(l1 - sums.nonlink_ll) <= 1e-12 * abs(sums.nonlink_ll)
    pr.ctx.close()
print("ppx ok")
"""

BETA_SHAPES = [(2048, 64, 500, 64), (2048, 64, 500, 32), (2048, 256, 3000, 128), (1024, 1000, 300, 256),
               (4096, 1024, 1024, 64), (1024, 2048, 300, 128)]


def test_generic_forms_match_specialised_and_oracle(orc, tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no fallback path exists)")
    import ammsb_pkg
    ammsb_pkg.load()
    from mcmc_ammsb_gpu_amd import ops as hip
    from test_gpu_parity import Problem
    # the specialised kernels' gradients (this process: default dispatch)
    ref = {}
    for idx, (N, K, n_edges, L) in enumerate(BETA_SHAPES):
        pr = Problem(orc, hip, N, K, 8, 16)
        rng = pr.rng
        non = orc.make_edge(rng.integers(0, N, n_edges), rng.integers(0, N, n_edges))
        mbe = np.concatenate([pr.edges[: n_edges // 3], non[: n_edges - n_edges // 3]]).astype(np.uint64)
        rng.shuffle(mbe)
        upd = hip.BetaUpdater(pr.ctx, pr.theta, pr.beta, pr.pi, pr.dset, (44, 45), L)
        upd.count_calls += 1
        ref["g%d" % idx] = upd.calculate_grads(pr.ctx.from_numpy(mbe), mbe.size).cpu().numpy().copy()
        pr.ctx.close()
    path = str(tmp_path / "beta_ref.npz")
    np.savez(path, **ref)
    env = dict(os.environ, AMMSB_PHI_FORM="g", AMMSB_BETA_FORM="g", AMMSB_PPX_FORM="g")
    script = CHILD % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}
    out = subprocess.run([sys.executable, "-c", script, path], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "phi ok" in out.stdout and "beta ok" in out.stdout and "ppx ok" in out.stdout


PAIR_CHILD = r"""
import sys
import numpy as np
sys.path[:0] = [%(root)r, %(tests)r]
import torch
import ammsb_pkg
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import ops as hip
import oracle_lib as orc
orc.build()
from test_gpu_parity import Problem

def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)

for N, K, n, n_nodes, L in %(cases)s:
    for noise in (False, True):
        pr = Problem(orc, hip, N, K, n, n_nodes, deg=4 if N > 10000 else 16)
        upd = hip.PhiUpdater(pr.ctx, pr.beta, pr.pi, pr.phi_sum, pr.dset, n_nodes, (42, 43), L,
                             phi_disable_noise=not noise, streaming_only=True)
        seeds = orc.rng_init(n_nodes * L, 42, 43)
        pi_h, phi_sum_h = pr.pi_h.copy(), pr.phi_sum_h.copy()
        for step in (1, 2):
            upd(pr.nodes, pr.nb, n_nodes)
            pr.sync()
            assert %(kernel)r in pr.ctx.kernel_names()["update_phi"], pr.ctx.kernel_names()
            want = orc.update_phi(pr.p_orc, pr.beta_h, pi_h.reshape(-1), phi_sum_h, pr.oset, pr.nodes_h,
                                  pr.nb_h.reshape(-1), step, seeds, L, 1, noise)
            got = upd.phi_vec.cpu().numpy()[:n_nodes]
            assert np.array_equal(bits(got), bits(want)), ("phi_vec", N, K, n, L, noise, step)
            assert np.array_equal(upd.rand.host(), seeds), ("streams", K, L)
            orc.update_pi(pr.p_orc, pi_h.reshape(-1), phi_sum_h, want.reshape(-1), pr.nodes_h, L, 1)
            if N > 10000:
                break
            assert np.array_equal(bits(pr.pi.host()), bits(pi_h)), ("pi", K, L)
        pr.ctx.close()
print("pair ok")
"""


def test_two_nodes_per_wave_update_phi(orc):
    """update_phi_pair_kernel (opt-in, AMMSB_PHI_PAIR=1): each half of a wave owns a node; work-group sizes 32 and 64,
    odd node counts (an idle half), more nodes than groups (a half with a second node)."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no fallback path exists)")
    env = dict(os.environ, AMMSB_PHI_PAIR="1")
    cases = [(2048, 256, 32, 201, 64), (2048, 256, 32, 200, 32), (2048, 256, 6, 100, 64),
             (2048, 512, 32, 101, 64), (2048, 512, 8, 100, 32), (70000, 256, 4, 65535 + 700, 64)]
    script = PAIR_CHILD % {"root": ROOT, "tests": os.path.join(ROOT, "tests"), "cases": repr(cases), "kernel": "pair_kernel"}
    out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "pair ok" in out.stdout


def test_persistent_streaming_update_phi(orc):
    """update_phi_stream_kernel (opt-in, AMMSB_PHI_STREAM=1; K = 256, launches of more than 1024 groups): persistent
    one-wave blocks walk several virtual groups and fetch the next node's prologue through LDS-DMA while the current
    node's rows are reduced.  Work-group sizes 32 and 64, n = 16 / 20 / 32, more nodes than groups (a group's second
    node follows its first in the same block)."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no fallback path exists)")
    env = dict(os.environ, AMMSB_PHI_STREAM="1")
    cases = [(20000, 256, 32, 5000, 64), (20000, 256, 16, 3001, 32), (20000, 256, 20, 2000, 64),
             (70000, 256, 16, 65535 + 900, 64)]
    script = PAIR_CHILD % {"root": ROOT, "tests": os.path.join(ROOT, "tests"), "cases": repr(cases), "kernel": "stream_kernel"}
    out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "pair ok" in out.stdout


def test_persistent_grid_update_phi(orc):
    """update_phi_lds2_kernel with a persistent grid (opt-in, AMMSB_PHI_PERSIST=2: as many blocks as the chip holds, block b
    takes virtual groups b, b + gridDim.x, ...): the groups keep their streams and nodes whichever block runs them."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no fallback path exists)")
    env = dict(os.environ, AMMSB_PHI_PERSIST="2")
    cases = [(20000, 256, 32, 9000, 64), (20000, 256, 16, 8193, 32), (70000, 256, 16, 65535 + 900, 64)]
    script = PAIR_CHILD % {"root": ROOT, "tests": os.path.join(ROOT, "tests"), "cases": repr(cases), "kernel": "lds2_kernel"}
    out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "pair ok" in out.stdout


@pytest.mark.gpu
def test_three_slot_update_phi_kernel_is_bit_identical():
    """update_phi_lds3_kernel (K = 1024, one wave per node, three ring slots in the LDS of two slots + the noise buffer:
    the lane's normals travel through the node's output row; opt-in, AMMSB_PHI_LDS3=1, read once per process): every
    K = 1024 update_phi parity case, the extreme-value cases, the > 65 535-node and group-shard cases and C3 at full size
    run against the oracle with the kernel forced on, in a child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AMMSB_PHI_LDS3="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_gpu_parity.py"), os.path.join(root, "tests", "test_gpu_fullsize.py"),
                        "-k", "update_phi or c3_update_phi"], cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
    # and the forced kernel is the one that ran for the C3 row shape
    code = ("import numpy as np, torch, __graft_entry__ as ge; ge.build()\n"
            "import sys; sys.path.insert(0, 'tests')\n"
            "import oracle_lib as orc; orc.build()\n"
            "from mcmc_ammsb_gpu_amd import ops as hip\n"
            "from test_gpu_parity import Problem\n"
            "pr = Problem(orc, hip, 4096, 1024, 32, 300)\n"
            "upd = hip.PhiUpdater(pr.ctx, pr.beta, pr.pi, pr.phi_sum, pr.dset, 300, (42, 43), 64, streaming_only=True)\n"
            "upd(pr.nodes, pr.nb, 300); torch.cuda.synchronize()\n"
            "print(pr.ctx.kernel_names()['update_phi'])\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "update_phi_lds3_kernel<16, 64, true>" in r.stdout, r.stdout + r.stderr
