"""N > 1 on the device: two ranks share the one GPU of the test box (gloo carries the exchange through host
memory, see ops._via_host), so what runs here is the real sharded path -- the HIP kernels launched over group /
edge ranges, chunked phi_vec exchange with tail rows, rank-ordered gradient sum, perplexity slices -- against a
single-process run of the same learner.  RCCL itself needs two GPUs and is exercised by bench.py --gpus N."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, outdir, case, chunks, device_sampling, rep, grads="sharded"):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import ammsb_pkg
    ammsb_pkg.load()
    from mcmc_ammsb_gpu_amd import hostlib, learner, ops
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    N, K, m, n, iters = case
    edges = hostlib.generate_graph(N, 8, 12, seed=5)
    ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.05, rand_seed=7)
    cfg = learner.Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, phi_wg_size=64, beta_wg_size=64,
                                           ppx_wg_size=64, strategy="Node", phi_chunks=chunks,
                                           device_sampling=device_sampling, phi_replicate=rep, beta_grads=grads,
                                           beta_shard_min_edges=0 if grads == "sharded" else 4096)
    lrn = learner.Learner(cfg, ds, rank=rank, world_size=world, group=None)
    if world > 1 and grads != "sharded":
        assert lrn.grads_mode == "replicated" and lrn.grads_fused == bool(device_sampling)
    p0 = lrn.HeldoutPerplexity()
    lrn.Run(1)
    pi1 = lrn.pi.host()
    lrn.Run(iters - 1)
    p1 = lrn.HeldoutPerplexity()
    lrn.drain()
    np.savez(os.path.join(outdir, "w%d_r%d.npz" % (world, rank)), pi1=pi1, pi=lrn.pi.host(),
             theta=ops.to_numpy(lrn.theta), beta=ops.to_numpy(lrn.beta), ppx=np.array([p0, p1]),
             edges=np.array([lrn.edges_done]), seeds=lrn.phiUpdater.rand.host().view(np.uint64),
             split=np.array([lrn.g_rep, lrn.cc]))
    lrn.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("case,chunks,device_sampling,rep,grads", [
    ((3000, 256, 256, 8, 6), 4, False, 0.0, "sharded"),      # K = 256: LDS-streamed phi / beta kernels; first block only -> broadcasts
    ((150000, 64, 70000, 2, 3), 3, False, 0.0, "sharded"),   # > 65535 mini-batch nodes: tail rows, 3 overlapped chunks
    ((150000, 64, 16384, 4, 4), 2, True, 0.25, "sharded"),   # device-side mini-batch sampler; a quarter of the groups replicated
    ((150000, 64, 70000, 2, 3), 2, False, "auto", "sharded"),  # split calibrated at start-up from a timed launch and exchange
    # every rank the whole gradient (Config.beta_grads): no collective for it, and the run is the single-GPU one BIT FOR
    # BIT -- with update_pi folded into the gradient launch for device-sampled mini-batches ("auto" picks that) ...
    ((150000, 64, 16384, 4, 6), 2, True, 0.25, "auto"),
    ((30000, 256, 4096, 8, 6), 2, True, 0.1, "auto"),        # ... at K = 256 (the LDS-streamed fused kernel) ...
    ((3000, 256, 256, 8, 6), 4, False, 0.0, "replicated"),   # ... and as separate launches for host-sampled ones
], ids=["small", "tail-3chunks", "device-sampling", "auto-split", "replicated-fused-k64", "replicated-fused-k256",
        "replicated-host-sampled"])
def test_world2_on_one_gpu(tmp_path, case, chunks, device_sampling, rep, grads):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import torch.multiprocessing as mp
    import __graft_entry__ as ge
    ge.build()
    out = str(tmp_path)
    ctx = mp.get_context("spawn")
    single = ctx.Process(target=_run, args=(0, 1, 0, out, case, chunks, device_sampling, rep, grads))
    single.start()
    single.join(600)
    assert single.exitcode == 0
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, out, case, chunks, device_sampling, rep, grads)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    one = np.load(os.path.join(out, "w1_r0.npz"))
    r0 = np.load(os.path.join(out, "w2_r0.npz"))
    r1 = np.load(os.path.join(out, "w2_r1.npz"))
    for k in ("pi", "theta", "beta", "ppx", "edges"):       # replicas stay identical
        assert np.array_equal(r0[k], r1[k]), k
    assert np.array_equal(r0["pi1"], one["pi1"])             # phi / pi do not depend on the split over ranks
    assert r0["edges"][0] == one["edges"][0] and r0["ppx"][0] == one["ppx"][0]
    if grads != "sharded":  # the whole gradient on every rank: nothing is summed in another order
        for k in ("pi", "theta", "beta"):
            assert np.array_equal(r0[k], one[k]), k
        assert r0["ppx"][1] == one["ppx"][1] or abs(r0["ppx"][1] - one["ppx"][1]) <= 1e-6 * one["ppx"][1]
    assert np.allclose(r0["theta"], one["theta"], rtol=2e-5, atol=1e-7)   # gradient summed in another association
    # pi after several iterations: same trajectory up to the rounding of beta (entries are O(1/K); tiny ones sit
    # next to the 1e-24 clamp and are compared absolutely)
    d = np.abs(r0["pi"].astype(np.float64) - one["pi"])
    assert d.max() <= 2e-6 and (d <= 1e-3 * np.abs(one["pi"]) + 1e-7).mean() > 0.999, (d.max(), d.mean())
    assert abs(r0["ppx"][1] - one["ppx"][1]) <= 1e-5 * one["ppx"][1]
    L = 64
    g_rep, cc = (int(x) for x in r0["split"])
    assert np.array_equal(r0["split"], r1["split"])
    s0, s1, s = r0["seeds"].reshape(-1, 2), r1["seeds"].reshape(-1, 2), one["seeds"].reshape(-1, 2)
    lim = min(s.shape[0], s0.shape[0], 65535 * L)
    g = np.arange(lim) // L
    owner = np.where(g < g_rep, -1, ((g - g_rep) // cc) % 2)   # -1: replicated, advanced by both ranks
    assert np.array_equal(s0[:lim][owner != 1], s[:lim][owner != 1])
    assert np.array_equal(s1[:lim][owner != 0], s[:lim][owner != 0])


def _run_nccl_single(outdir, rep):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import ammsb_pkg
    ammsb_pkg.load()
    from mcmc_ammsb_gpu_amd import hostlib, learner, ops
    torch.cuda.set_device(0)
    N, K, m, n, iters = 150000, 64, 70000, 2, 3
    edges = hostlib.generate_graph(N, 8, 12, seed=5)
    ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.05, rand_seed=7)

    def cfg(force):
        return learner.Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, phi_wg_size=64, beta_wg_size=64,
                                                ppx_wg_size=64, strategy="Node", phi_chunks=3, phi_replicate=rep,
                                                force_exchange=force)
    ref = learner.Learner(cfg(False), ds)
    ref.Run(iters)
    want = (ref.pi.host(), ops.to_numpy(ref.theta), ref.HeldoutPerplexity())
    ref.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    lrn = learner.Learner(cfg(True), ds, rank=0, world_size=1)
    lrn.Run(iters)
    got = (lrn.pi.host(), ops.to_numpy(lrn.theta), lrn.HeldoutPerplexity())
    lrn.close()
    dist.barrier()
    dist.destroy_process_group()
    ok = np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2]
    open(os.path.join(outdir, "nccl_ok_%s" % rep), "w").write("1" if ok else "0")


@pytest.mark.parametrize("rep", [0.0, "auto"])
def test_rccl_calls_with_one_rank(tmp_path, rep):
    """The exchange code path on the real backend: a one-rank RCCL group runs the chunked in-place
    all_gather_into_tensor / broadcast / small all-gathers (and, with "auto", the start-up calibration) exactly as a
    multi-GPU job issues them; with one rank they move nothing, so the run must equal the plain single-GPU learner
    bit for bit.  What this cannot show is more than one GPU (the test boxes have one)."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import torch.multiprocessing as mp
    import __graft_entry__ as ge
    ge.build()
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_run_nccl_single, args=(str(tmp_path), rep))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    assert open(os.path.join(str(tmp_path), "nccl_ok_%s" % rep)).read() == "1"


def _run_ckpt_gpu(rank, world, port, outdir):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import io
    import torch
    import torch.distributed as dist
    import ammsb_pkg
    ammsb_pkg.load()
    from mcmc_ammsb_gpu_amd import hostlib, learner, ops
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    N, K, m, n = 150000, 64, 70000, 2
    edges = hostlib.generate_graph(N, 8, 12, seed=5)
    ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.05, rand_seed=7)

    def cfg():
        return learner.Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, phi_wg_size=64, beta_wg_size=64,
                                                ppx_wg_size=64, strategy="Node", phi_chunks=2, phi_replicate=0.1,
                                                device_sampling=True)
    a = learner.Learner(cfg(), ds, rank=rank, world_size=world)
    a.Run(2)
    a.HeldoutPerplexity()
    buf = io.BytesIO()
    a.Serialize(buf)
    a.Run(2)
    want = (a.pi.host(), ops.to_numpy(a.theta), a.HeldoutPerplexity())
    a.close()
    shared = [buf.getvalue() if rank == 0 else None]
    dist.broadcast_object_list(shared, src=0)
    b = learner.Learner(cfg(), ds, rank=rank, world_size=world)
    b.Parse(io.BytesIO(shared[0]))
    b.Run(2)
    got = (b.pi.host(), ops.to_numpy(b.theta), b.HeldoutPerplexity())
    b.close()
    ok = np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2]
    open(os.path.join(outdir, "ckpt_ok_r%d" % rank), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_world2_checkpoint_on_one_gpu(tmp_path):
    """Multi-rank checkpoint with the HIP kernels and the device sampler: both ranks restore from rank 0's stream
    and continue bit-identically to the uninterrupted two-rank run."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import torch.multiprocessing as mp
    import __graft_entry__ as ge
    ge.build()
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_run_ckpt_gpu, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    for r in range(2):
        assert open(os.path.join(str(tmp_path), "ckpt_ok_r%d" % r)).read() == "1"
