"""ammsb_update_pi_beta_grads (include/ammsb.h): update_pi and the beta gradient of a node-stratified mini-batch as
one launch, against the two separate calls -- pi rows, phi_sum and the gradient bit for bit, for every shape the entry
accepts; and its refusals.  (The same kernels inside the descriptor loop are compared with the eager loop in
test_gpu_graph_loop.py; the multi-GPU schedule that uses this entry in test_gpu_distributed.py.)"""
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
    N = 20000
    edges = hostlib.generate_graph(N, 16, 16, seed=7)
    return ops, learner, hostlib.Dataset.robust(N, edges, heldout_ratio=0.02, rand_seed=3)


def _make(learner, ds, K, wg, m, strategy):
    cfg = learner.Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=16, strategy=strategy,
                                           phi_wg_size=wg, beta_wg_size=wg, ppx_wg_size=wg, device_sampling=True,
                                           graph_launch=False)
    return learner.Learner(cfg, ds)


def _half_step(lrn, fused):
    """update_phi of the pending mini-batch, then update_pi + gradient -- separately or as the one launch."""
    s = lrn.samples[lrn.phase]
    lrn.futures[lrn.phase].result()
    lrn.ops.wait_event(s.ready)
    phi, beta = lrn.phiUpdater, lrn.betaUpdater
    phi.count_calls += 1
    phi.update_phi(s.dev_nodes, s.neighbor_sampler.GetData(), s.n_nodes)
    assert s.n_nodes == s.n_edges + 1
    if fused:
        g = beta.update_pi_and_grads(phi, s.dev_nodes, s.dev_edges, s.n_edges)
    else:
        phi.update_pi(s.dev_nodes, s.n_nodes)
        g = beta.calculate_grads(s.dev_edges, s.n_edges)
    lrn.ops.synchronize()
    return lrn.pi.host().copy(), lrn.ops.to_numpy(lrn.phi).copy(), lrn.ops.to_numpy(g).copy(), s.n_edges


@pytest.mark.parametrize("K,wg,m,strategy", [(32, 32, 1024, "NodeNonLink"),   # register kernels, one column per lane
                                              (64, 32, 300, "Node"),           # ... two
                                              (128, 64, 257, "Node"),
                                              (96, 64, 256, "NodeLink"),       # column guards; link mini-batches
                                              (256, 64, 512, "NodeNonLink"),   # LDS-streamed <4, 1, true, 64>
                                              (256, 32, 512, "Node"),
                                              (512, 64, 300, "NodeNonLink"),
                                              (512, 32, 300, "NodeLink"),
                                              (1024, 64, 200, "NodeNonLink"),  # the C3 kernel
                                              (1024, 32, 200, "Node"),
                                              (1024, 64, 5000, "NodeNonLink")])  # more edges than slots: several trips per slot
def test_one_launch_equals_update_pi_then_gradient(env, K, wg, m, strategy):
    ops, learner, ds = env
    a, b = _make(learner, ds, K, wg, m, strategy), _make(learner, ds, K, wg, m, strategy)
    assert b.betaUpdater.can_fuse_update_pi(b.phiUpdater)
    for lrn in (a, b):
        lrn.Run(3)  # identical states: same seeds, same launches
        lrn.drain()
    for rep in range(2):
        pa, sa, ga, na = _half_step(a, fused=False)
        pb, sb, gb, nb = _half_step(b, fused=True)
        assert na == nb and na > 0
        assert np.array_equal(pa, pb), "pi"
        assert np.array_equal(sa, sb), "phi_sum"
        assert np.array_equal(ga, gb), "gradient"
        assert np.isfinite(ga).all() and np.abs(ga).max() > 0
    assert "true" in b.ctx.kernel_names()["beta_grads"]  # the fused instantiation is what ran
    a.close(), b.close()


def test_shapes_the_entry_refuses(env):
    ops, learner, ds = env
    lrn = _make(learner, ds, 2048, 64, 128, "Node")   # 32 columns per lane: no fused form
    assert not lrn.betaUpdater.can_fuse_update_pi(lrn.phiUpdater)
    lrn.Run(1)
    lrn.drain()
    s = lrn.samples[lrn.phase]
    with pytest.raises(Exception, match="not a shape the fused kernels take"):
        lrn.betaUpdater.update_pi_and_grads(lrn.phiUpdater, s.dev_nodes, s.dev_edges, s.n_edges)
    lrn.close()
    cfg = learner.Config.from_cli_defaults(K=256, mini_batch_size=128, num_node_sample=16, strategy="Node", phi_wg_size=64,
                                           beta_wg_size=32, ppx_wg_size=64, device_sampling=True, graph_launch=False)
    lrn = learner.Learner(cfg, ds)   # update_pi's WG_SUM is over phi_wg lanes: the two sizes must agree
    assert not lrn.betaUpdater.can_fuse_update_pi(lrn.phiUpdater)
    lrn.close()
