"""Opt-in parallel cuckoo build on the device: exact membership against the serial host build
(cuckoo.cc:117-161 restated in host/cuckoo.cc), image allowed to differ, kernels indifferent to which image they probe."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import __graft_entry__ as ge
    ge.build()
    from mcmc_ammsb_gpu_amd import hostlib, ops
    return ops, hostlib, torch


@pytest.mark.parametrize("n", [1, 7, 1000, 2_000_000])
def test_device_build_has_exactly_the_keys(env, n):
    ops, hostlib, torch = env
    rng = np.random.default_rng(n)
    N = 1 << 22
    u = rng.integers(0, N, int(n * 1.2) + 8, dtype=np.uint64)
    v = rng.integers(0, N, u.size, dtype=np.uint64)
    keys = np.unique((np.minimum(u, v) << np.uint64(32)) | np.maximum(u, v))[:n]
    rng.shuffle(keys)
    ctx = ops.Context(ops.make_params(N, 32, E=keys.size, num_node_sample=8))
    dev = ops.DeviceSet.build_on_device(ctx, keys)
    host = hostlib.HostSet(keys)                       # the serial reference-order build
    assert dev.num_bins == host.BinsPerBucket() and 0 <= dev.prime_idx < 4
    image = dev.data.cpu().numpy().view(np.uint64)
    assert image.size == host.Capacity()
    held = image[image != EMPTY]
    assert np.array_equal(np.sort(held), np.sort(keys))   # every key exactly once, nothing else
    # every key sits in one of ITS two bins: the device lookup finds all of them and agrees with the host set elsewhere
    assert bool(dev.Has(keys).cpu().numpy().all())
    probe = (rng.integers(0, N, 500_000, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, N, 500_000, dtype=np.uint64)
    assert np.array_equal(dev.Has(probe).cpu().numpy().astype(bool), host.Has(probe))
    ctx.close()


def test_device_build_repeated_small_sets(env):
    """The build is a race by design (claim / swap); a lost key needs a particular interleaving (round 2: a displaced
    key met its own stale copy in L1 and was dropped, once in a few hundred 1000-key builds).  300 builds, every image
    checked for exactly the key set."""
    ops, hostlib, torch = env
    rng = np.random.default_rng(77)
    N = 1 << 20
    ctx = ops.Context(ops.make_params(N, 32, E=1000, num_node_sample=8))
    degenerate = 0
    for it in range(300):
        n = int(rng.integers(200, 3000))
        keys = np.unique((rng.integers(0, N, n, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, N, n, dtype=np.uint64))
        try:
            dev = ops.DeviceSet.build_on_device(ctx, keys)
        except ops.AmmsbError:
            # the reference's hash pairs degenerate for some table sizes: then the serial
            # build fails too, and that is the only failure allowed
            with pytest.raises(ops.AmmsbError):
                hostlib.HostSet(keys)
            degenerate += 1
            continue
        image = dev.data.cpu().numpy().view(np.uint64)
        assert np.array_equal(np.sort(image[image != EMPTY]), keys), it
    assert degenerate < 60
    ctx.close()


def test_kernels_do_not_care_which_image_they_probe(env, orc):
    """update_phi over the host-built image and over the device-built image of the same key set: bit-identical."""
    ops, hostlib, torch = env
    N, K, n, nn, L = 50_000, 256, 16, 3000, 64
    rng = np.random.default_rng(3)
    edges = hostlib.generate_graph(N, 16, 24, seed=9)
    ctx = ops.Context(ops.make_params(N, K, E=edges.size, num_node_sample=n))
    hs = hostlib.HostSet(edges)
    set_h = ops.DeviceSet(ctx, hs.Serialize(), hs.BinsPerBucket(), hs.PrimeIdx())
    set_d = ops.DeviceSet.build_on_device(ctx, edges)
    assert not np.array_equal(set_h.data.cpu().numpy(), set_d.data.cpu().numpy()) or set_h.prime_idx == set_d.prime_idx
    pi = ops.RowPartitionedMatrix(ctx, N, K)
    phi_sum = ctx.zeros((N,), torch.float32)
    ops.RandomGammaAndNormalize(ctx, 1.0, 1.0, pi, phi_sum)
    beta = ctx.zeros((2 * K,), torch.float32)
    ops.beta_from_theta(ctx, ctx.from_numpy(hostlib.theta_init(K)), beta)
    nodes_h = rng.permutation(N)[:nn].astype(np.uint32)
    nbrs_h = rng.integers(0, N, size=(nn, n), dtype=np.uint32)
    src, dst = (edges >> np.uint64(32)).astype(np.uint32), (edges & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    order = np.argsort(src, kind="stable")
    first = np.searchsorted(src[order], nodes_h)
    for i in range(nn):            # some true neighbours, so that the link branch is taken
        j = first[i]
        if j < src.size and src[order][j] == nodes_h[i]:
            nbrs_h[i, 0] = dst[order][j]
    nodes, nbrs = ctx.from_numpy(nodes_h), ctx.from_numpy(nbrs_h)
    out = []
    for s in (set_h, set_d):
        upd = ops.PhiUpdater(ctx, beta, pi, phi_sum, s, nn, (42, 43), L)
        upd.count_calls = 1
        upd.update_phi(nodes, nbrs, nn)
        torch.cuda.synchronize()
        out.append(upd.phi_vec[:nn].clone())
    assert torch.equal(out[0], out[1])
    link = set_h.Has(ctx.from_numpy((np.minimum(nodes_h, nbrs_h[:, 0]).astype(np.uint64) << np.uint64(32)) |
                                    np.maximum(nodes_h, nbrs_h[:, 0]).astype(np.uint64)))
    assert int(link.sum()) > 100
    ctx.close()
