"""Known answers derived from the reference's kernel TEXT by hand, not from any implementation: for uniform
memberships (every pi row = 1/K, K a power of two) and constant community strengths the formulas of phi.cc:214-302,
beta.cc:87-136 / :51-82 and perplexity.cc:16-65 collapse to closed forms.

  update_phi   probs_k is the same for every k, so probs_k / probs_sum = 1/K exactly and every gradient term is
               (1/K) / ((1/K) S) - 1/S = 0: phi*_k = | S/K + eps_t/2 (alpha - S/K) + sqrt(eps_t S/K) |  (noise off: PHI_RANDN = 1)
  gradient     link edge:      f = (b/K^2) / (b/K + eps (1 - 1/K)),       g[2k] -= f / theta_sum,  g[2k+1] += f (1/theta_1 - 1/theta_sum)
               non-link edge:  f = ((1-b)/K^2) / ((1-b)/K + (1-eps)(1 - 1/K)), g[2k] += f (1/theta_0 - 1/theta_sum), g[2k+1] -= f / theta_sum
  theta step   theta* = | theta + eps_t/2 (eta - theta + scale g) + sqrt(eps_t theta) |   (noise off)
  perplexity   link: s = b/K;  non-link: s = (1-b)/K + (1 - 1/K)(1 - eps);  value = exp(-mean log s)

These pin the structure of each formula (signs, the N/n factor, the eps terms, which theta component a link feeds)
independently of oracle/ -- the CPU test holds the oracle to them, the GPU test the HIP kernels.  They do not pin the
summation orders (every order gives the same answer here); the bit-level comparisons elsewhere do that.
"""
import numpy as np
import pytest

N, K, n, S = 512, 64, 8, 8.0
T0, T1 = 3.0, 1.0            # theta pair -> beta_k = 0.25
B = T1 / (T0 + T1)
A_, B_, C_, EPS, ALPHA = 0.0315, 1024.0, 0.5, 1e-7, 1.0 / K


def eps_t(step):
    return A_ * (1.0 + step / B_) ** (-C_)    # learner.cc:41-43


def make_inputs():
    rng = np.random.default_rng(5)
    u = rng.integers(0, N, 4000, dtype=np.uint64)
    v = rng.integers(0, N, 4000, dtype=np.uint64)
    keep = u != v
    edges = np.unique((np.minimum(u, v) << np.uint64(32))[keep] | np.maximum(u, v)[keep])
    nodes = rng.permutation(N)[:100].astype(np.uint32)
    nbrs = rng.integers(0, N, (nodes.size, n), dtype=np.uint32)
    same = nbrs == nodes[:, None]
    nbrs[same] = (nbrs[same] + 1) % N
    # some true neighbours so that both branches of y are taken
    src, dst = (edges >> np.uint64(32)).astype(np.uint32), (edges & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    for i in range(nodes.size):
        mine = dst[src == nodes[i]][:3]
        nbrs[i, :mine.size] = mine
    pi = np.full((N, K), 1.0 / K, dtype=np.float32)
    phi_sum = np.full(N, S, dtype=np.float32)
    theta = np.tile(np.array([T0, T1], dtype=np.float32), K)
    beta = np.tile(np.array([T0 / (T0 + T1), B], dtype=np.float32), K)
    mb = np.concatenate([edges[:300], (rng.integers(0, N // 2, 200, dtype=np.uint64) << np.uint64(32)) | rng.integers(N // 2, N, 200, dtype=np.uint64)])
    return edges, nodes, nbrs, pi, phi_sum, theta, beta, mb


def expected_phi(step):
    e = eps_t(step)
    pk = S / K
    return abs(pk + e / 2 * (ALPHA - pk) + np.sqrt(e * pk))


def expected_grads(n_link, n_non):
    ts = T0 + T1
    fl = (B / K**2) / (B / K + EPS * (1 - 1.0 / K))
    fn = ((1 - B) / K**2) / ((1 - B) / K + (1 - EPS) * (1 - 1.0 / K))
    g0 = n_link * fl * (0.0 - 1 / ts) + n_non * fn * (1 / T0 - 1 / ts)
    g1 = n_link * fl * (1 / T1 - 1 / ts) + n_non * fn * (0.0 - 1 / ts)
    return g0, g1


def expected_theta(step, scale, g0, g1):
    e = eps_t(step)
    return (abs(T0 + e / 2 * (1.0 - T0 + scale * g0) + np.sqrt(e * T0)),
            abs(T1 + e / 2 * (1.0 - T1 + scale * g1) + np.sqrt(e * T1)))


def expected_ppx(n_link, n_non):
    sl = B / K
    sn = (1 - B) / K + (1 - 1.0 / K) * (1 - EPS)
    return n_link * np.log(sl), n_non * np.log(sn)


def check(phi_vec, pi_after, phi_sum_after, nodes, grads, theta_after, sums, n_link, n_non, scale):
    want = expected_phi(1)
    assert np.allclose(phi_vec, want, rtol=2e-6, atol=0), (phi_vec.min(), phi_vec.max(), want)
    assert np.allclose(pi_after[nodes], 1.0 / K, rtol=1e-6) and np.allclose(phi_sum_after[nodes], K * want, rtol=2e-6)
    g0, g1 = expected_grads(n_link, n_non)
    assert np.allclose(grads[0::2], g0, rtol=1e-5) and np.allclose(grads[1::2], g1, rtol=1e-5), (grads[:2], g0, g1)
    t0, t1 = expected_theta(1, scale, g0, g1)
    assert np.allclose(theta_after[0::2], t0, rtol=1e-5) and np.allclose(theta_after[1::2], t1, rtol=1e-5)
    ll, ln_, cl, cn = sums
    wl, wn = expected_ppx(cl, cn)
    assert cl + cn > 0 and abs(ll - wl) <= 1e-5 * abs(wl) + 1e-12 and abs(ln_ - wn) <= 1e-5 * abs(wn) + 1e-12


def test_oracle_meets_the_closed_forms(orc):
    edges, nodes, nbrs, pi, phi_sum, theta, beta, mb = make_inputs()
    p = orc.make_params(N, K, n)
    oset = orc.OracleSet(edges)
    L = 32
    seeds = orc.rng_init(nodes.size * L, 42, 43)
    pv = orc.update_phi(p, beta, pi.reshape(-1), phi_sum, oset, nodes, nbrs.reshape(-1), 1, seeds, L, 1, False)
    pi2, ps2 = pi.copy(), phi_sum.copy()
    orc.update_pi(p, pi2.reshape(-1), ps2, pv.reshape(-1), nodes, L, 1)
    g = orc.beta_grads(p, theta, beta, pi.reshape(-1), oset, mb, L, 1, order=1)
    n_link = int(oset.has(mb).sum())
    th = theta.copy()
    orc.update_theta(p, th, g, 1, 0.5, orc.rng_init(K, 44, 45), noise_on=False)
    held = mb
    sums, _ = orc.perplexity(p, beta, pi.reshape(-1), oset, held, 1, L, 1, np.zeros(held.size, np.float32))
    check(pv, pi2, ps2, nodes, g, th, (sums.link_ll, sums.nonlink_ll, sums.link_cnt, sums.nonlink_cnt), n_link,
          mb.size - n_link, 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("L", [32, 64])
def test_hip_kernels_meet_the_closed_forms(orc, L):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import __graft_entry__ as ge
    ge.build()
    from mcmc_ammsb_gpu_amd import hostlib, ops
    edges, nodes, nbrs, pi_h, phi_h, theta_h, beta_h, mb = make_inputs()
    ctx = ops.Context(ops.make_params(N, K, E=edges.size, num_node_sample=n))
    hs = hostlib.HostSet(edges)
    dset = ops.DeviceSet(ctx, hs.Serialize(), hs.BinsPerBucket(), hs.PrimeIdx())
    pi = ops.RowPartitionedMatrix(ctx, N, K)
    pi.load(pi_h)
    phi_sum = ctx.from_numpy(phi_h)
    theta, beta = ctx.from_numpy(theta_h), ctx.from_numpy(beta_h)
    upd = ops.PhiUpdater(ctx, beta, pi, phi_sum, dset, nodes.size, (42, 43), L, phi_disable_noise=True)
    calc = ops.PerplexityCalculator(ctx, beta, pi, ctx.from_numpy(mb), dset, L)
    calc()                                   # before pi changes
    sums = calc.unpack(calc.sums)
    bu = ops.BetaUpdater(ctx, theta, beta, pi, dset, (44, 45), L, disable_noise=True)
    bu.count_calls = 1
    g = bu.calculate_grads(ctx.from_numpy(mb), mb.size).cpu().numpy().copy()
    bu.update_theta(0.5)
    upd(ctx.from_numpy(nodes), ctx.from_numpy(nbrs), nodes.size)
    torch.cuda.synchronize()
    n_link = int(hs.Has(mb).sum())
    check(upd.phi_vec[:nodes.size].cpu().numpy(), pi.host(), phi_sum.cpu().numpy(), nodes, g, theta.cpu().numpy(), sums,
          n_link, mb.size - n_link, 0.5)
    ctx.close()
