"""N > 1 path on CPU: world size 2, gloo backend.  The product's learner.py (sharding by virtual-group
range / edge range, phi_vec all-gather + tail broadcasts, rank-ordered gradient sum, perplexity scalar
gather) runs unchanged; only the operator set is swapped for the oracle-backed one in
tests/oracle_ops.py, so the collective logic is what is under test."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, outdir, case, chunks=4, rep="auto", grads="sharded"):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    import oracle_lib
    oracle_lib.lib().orc_set_num_threads(2)
    import ammsb_pkg
    ammsb_pkg.load()
    from mcmc_ammsb_gpu_amd import hostlib, learner
    import oracle_ops
    group = None
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    N, K, m, n, iters = case
    edges = hostlib.generate_graph(N, 8, 12, seed=5)
    ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.05, rand_seed=7)
    cfg = learner.Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, phi_wg_size=32, beta_wg_size=32,
                                           ppx_wg_size=32, strategy="Node", sample_parallel=True, phi_chunks=chunks,
                                           phi_replicate=rep, beta_grads=grads,
                                           # (sharded: every mini-batch, link batches included, takes the collective)
                                           beta_shard_min_edges=0 if grads == "sharded" else 4096)
    lrn = learner.Learner(cfg, ds, ops=oracle_ops, rank=rank, world_size=world, group=group)
    p0 = lrn.HeldoutPerplexity()
    if world > 1:
        lrn.shard_trace = []  # the step trace bench.py's N > 1 record is made of (marks only: the trajectory is untouched)
    lrn.Run(1)
    pi1 = lrn.pi.host()
    lrn.Run(iters - 1)
    p1 = lrn.HeldoutPerplexity()
    lrn.drain()
    if world > 1:
        import json
        rep_ = lrn.shard_report()
        with open(os.path.join(outdir, "w%d_r%d_trace.json" % (world, rank)), "w") as f:
            json.dump({"report": rep_, "steps_traced": len(lrn.shard_trace)}, f)
    np.savez(os.path.join(outdir, "w%d_r%d.npz" % (world, rank)), pi1=pi1, pi=lrn.pi.host(), phi=lrn.phi.numpy(),
             theta=lrn.theta.numpy(), beta=lrn.beta.numpy(), ppx=np.array([p0, p1]), edges=np.array([lrn.edges_done]),
             seeds=lrn.phiUpdater.rand.host().view(np.uint64), ppx_state=lrn.heldoutPerplexity.ppx_per_edge.numpy(),
             split=np.array([lrn.g_rep, lrn.cc]))
    lrn.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("case,chunks,rep", [
    ((3000, 32, 256, 8, 6), 4, "auto"),      # link + non-link batches; only the first block is live -> broadcasts
    ((150000, 32, 70000, 2, 2), 1, 0.0),     # > 65535 mini-batch nodes: both ranks own groups, tail rows beyond group 65534
    ((150000, 32, 70000, 2, 2), 3, 0.0),     # same with the exchange cut into 3 overlapped chunks
    ((150000, 32, 70000, 2, 2), 2, 0.3),     # 30 % of the groups replicated (they own the tail rows), the rest exchanged
    ((3000, 32, 256, 8, 6), 2, 0.002),       # replicated prefix shorter than a link batch: both kinds of group in one step
    ((150000, 32, 70000, 2, 2), 2, 0.00003), # 1 replicated group, 4465 tail rows: replicated AND exchanged tail rows
], ids=["small", "tail-1chunk", "tail-3chunks", "replicate-30pct", "replicate-tiny", "mixed-tail"])
@pytest.mark.parametrize("grads", ["sharded", "replicated"])
def test_world2_matches_single_process(tmp_path, case, chunks, rep, grads):
    if grads == "replicated" and (chunks, rep) not in ((4, "auto"), (2, 0.3)):
        pytest.skip("the replicated gradient is covered on two of the splits")
    import torch.multiprocessing as mp
    import __graft_entry__ as ge
    ge.build()
    out = str(tmp_path)
    _run(0, 1, 0, out, case)
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_run, args=(r, 2, port, out, case, chunks, rep, grads)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    one = np.load(os.path.join(out, "w1_r0.npz"))
    r0 = np.load(os.path.join(out, "w2_r0.npz"))
    r1 = np.load(os.path.join(out, "w2_r1.npz"))
    if grads == "replicated":  # every rank the whole gradient (Config.beta_grads): the single-process run bit for bit
        for k in ("pi", "phi", "theta", "beta"):
            assert np.array_equal(r0[k], one[k]), k
    # replicas stay identical
    for k in ("pi", "phi", "theta", "beta", "ppx", "edges"):
        assert np.array_equal(r0[k], r1[k]), k
    # the step trace (what makes the first multi-GPU bench line diagnosable): every joint of the sharded step is there
    import json
    for r in range(2):
        tr = json.load(open(os.path.join(out, "w2_r%d_trace.json" % r)))
        assert tr["steps_traced"] == case[4]
        rp = tr["report"]
        if rp["steps"]:  # (non-link steps of the run)
            for key in ("step_ms", "phi_phase_ms", "update_pi_ms", "grads_local_ms", "grad_allgather_ms", "update_theta_ms",
                        "exchange_ms", "exchange_chunks", "how"):
                assert key in rp, key
            assert rp["step_ms"] > 0 and rp["phi_phase_ms"] <= rp["step_ms"] * 1.001
            parts = rp["phi_phase_ms"] + rp["update_pi_ms"] + rp["grads_local_ms"] + rp["grad_allgather_ms"] + rp["update_theta_ms"]
            assert abs(parts - rp["step_ms"]) <= 1e-6 * max(1.0, rp["step_ms"]) + 1e-3
            if rp["exchange_chunks"]:
                assert all(c["exchange_ms"] > 0 and c["received_bytes"] >= 0 for c in rp["exchange_chunks"])
                assert "overlap_ms" in rp and "allgather_GBps_received" in rp and "phi_local_ms" in rp
    # first iteration: phi/pi do not depend on how groups are split over ranks -> bit-identical to 1 process
    assert np.array_equal(r0["pi1"], one["pi1"])
    assert r0["edges"][0] == one["edges"][0]
    assert r0["ppx"][0] == one["ppx"][0]
    # later iterations see beta from a gradient summed in a different association (slices, rank order)
    assert np.allclose(r0["theta"], one["theta"], rtol=2e-5, atol=1e-7)
    assert np.allclose(r0["pi"], one["pi"], rtol=5e-4, atol=1e-7)
    assert abs(r0["ppx"][1] - one["ppx"][1]) <= 1e-5 * one["ppx"][1]
    # stream ownership is fixed (block b of Cc groups belongs to rank b % 2): each rank advanced exactly
    # its own blocks' streams, and their union equals the single-process state
    L = 32
    g_rep, cc = (int(x) for x in r0["split"])
    assert np.array_equal(r0["split"], r1["split"])
    s0, s1, s = r0["seeds"].reshape(-1, 2), r1["seeds"].reshape(-1, 2), one["seeds"].reshape(-1, 2)
    lim = min(s.shape[0], s0.shape[0], 65535 * L)
    g = np.arange(lim) // L
    owner = np.where(g < g_rep, -1, ((g - g_rep) // cc) % 2)   # -1: replicated, advanced by both ranks
    assert np.array_equal(s0[:lim][owner != 1], s[:lim][owner != 1])
    assert np.array_equal(s1[:lim][owner != 0], s[:lim][owner != 0])
    # perplexity state: each rank owns a contiguous slice of the held-out edges
    H = one["ppx_state"].size
    per = (H + 1) // 2
    assert np.allclose(r0["ppx_state"][:per], one["ppx_state"][:per], rtol=1e-4)
    assert np.allclose(r1["ppx_state"][per:], one["ppx_state"][per:], rtol=1e-4)


@pytest.mark.parametrize("case,chunks,rep", [
    ((150000, 32, 70000, 2, 2), 2, 0.12),    # C4's rank count and its expected split: 12 % of the groups replicated
    ((3000, 32, 256, 8, 4), 1, 0.0),         # small batches: every launch fits rank 0's block (broadcast path), 7 idle owners
], ids=["c4-shape-split", "small"])
def test_world8_matches_single_process(tmp_path, case, chunks, rep):
    """BASELINE's C4 shards over EIGHT ranks; no 8-GPU node is available to the builder, so the rank arithmetic of that
    world size (block ownership, in-place all-gather regions that reach past row G, tail rows parked by eight owners,
    rank-ordered gradient sum, eight perplexity slices) is rehearsed here over gloo with the oracle-backed operators."""
    import torch.multiprocessing as mp
    import __graft_entry__ as ge
    ge.build()
    out = str(tmp_path)
    W = 8
    _run(0, 1, 0, out, case)
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_run, args=(r, W, port, out, case, chunks, rep)) for r in range(W)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    one = np.load(os.path.join(out, "w1_r0.npz"))
    rs = [np.load(os.path.join(out, "w%d_r%d.npz" % (W, r))) for r in range(W)]
    for r in rs[1:]:   # replicas stay identical
        for k in ("pi", "phi", "theta", "beta", "ppx", "edges", "split"):
            assert np.array_equal(rs[0][k], r[k]), k
    assert np.array_equal(rs[0]["pi1"], one["pi1"])          # first iteration: independent of the split
    assert rs[0]["edges"][0] == one["edges"][0] and rs[0]["ppx"][0] == one["ppx"][0]
    assert np.allclose(rs[0]["theta"], one["theta"], rtol=5e-5, atol=1e-7)
    assert np.allclose(rs[0]["pi"], one["pi"], rtol=1e-3, atol=1e-7)
    assert abs(rs[0]["ppx"][1] - one["ppx"][1]) <= 2e-5 * one["ppx"][1]
    L = 32
    g_rep, cc = (int(x) for x in rs[0]["split"])
    s = one["seeds"].reshape(-1, 2)
    lim = min(s.shape[0], 65535 * L)
    g = np.arange(lim) // L
    owner = np.where(g < g_rep, -1, ((g - g_rep) // cc) % W)   # -1: replicated, advanced by every rank
    for r in range(W):
        sr = rs[r]["seeds"].reshape(-1, 2)
        mine = (owner == r) | (owner == -1)
        assert np.array_equal(sr[:lim][mine], s[:lim][mine]), "rank %d's own streams" % r
    H = one["ppx_state"].size
    per = (H + W - 1) // W
    for r in range(W):
        lo, hi = min(r * per, H), min((r + 1) * per, H)
        assert np.allclose(rs[r]["ppx_state"][lo:hi], one["ppx_state"][lo:hi], rtol=1e-4)


def _run_ckpt(rank, world, port, outdir):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import io
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    import oracle_lib
    oracle_lib.lib().orc_set_num_threads(2)
    import ammsb_pkg
    ammsb_pkg.load()
    from mcmc_ammsb_gpu_amd import hostlib, learner
    import oracle_ops
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    N, K, m, n = 150000, 32, 70000, 2
    edges = hostlib.generate_graph(N, 8, 12, seed=5)
    ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.05, rand_seed=7)

    def cfg():
        return learner.Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, phi_wg_size=32, beta_wg_size=32,
                                                ppx_wg_size=32, strategy="Node", phi_chunks=2, phi_replicate=0.1)
    a = learner.Learner(cfg(), ds, ops=oracle_ops, rank=rank, world_size=world)
    a.Run(2)
    a.HeldoutPerplexity()
    buf = io.BytesIO()
    a.Serialize(buf)                       # collective: owners hand out their streams / perplexity slices first
    a.Run(2)
    want = (a.pi.host(), a.theta.numpy().copy(), a.HeldoutPerplexity())
    a.close()
    shared = [buf.getvalue() if rank == 0 else None]
    dist.broadcast_object_list(shared, src=0)   # every rank restores from RANK 0's file
    b = learner.Learner(cfg(), ds, ops=oracle_ops, rank=rank, world_size=world)
    b.Parse(io.BytesIO(shared[0]))
    b.Run(2)
    got = (b.pi.host(), b.theta.numpy().copy(), b.HeldoutPerplexity())
    b.close()
    ok = np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2]
    with open(os.path.join(outdir, "ckpt_r%d.bin" % rank), "wb") as f:
        f.write(buf.getvalue())
    open(os.path.join(outdir, "ckpt_ok_r%d" % rank), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_world2_checkpoint_is_complete(tmp_path):
    """A multi-rank checkpoint must carry the phi streams and perplexity means of EVERY rank's blocks: the two ranks
    write the same bytes, and a fresh two-rank learner restored from them continues bit-identically."""
    import torch.multiprocessing as mp
    import __graft_entry__ as ge
    ge.build()
    out = str(tmp_path)
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_run_ckpt, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    for r in range(2):
        assert open(os.path.join(out, "ckpt_ok_r%d" % r)).read() == "1"
    a = open(os.path.join(out, "ckpt_r0.bin"), "rb").read()
    b = open(os.path.join(out, "ckpt_r1.bin"), "rb").read()
    assert len(a) == len(b) and sum(x != y for x, y in zip(a, b)) <= 16   # only the wall-clock fields may differ


def _run_p2p(rank, world, port, outdir):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    import ammsb_pkg
    ammsb_pkg.load()
    from mcmc_ammsb_gpu_amd import ops
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ok = True
    for chunk, cols in ((1, 4), (7, 33), (513, 64)):
        region = torch.full((world * chunk, cols), -1.0)
        region[rank * chunk:(rank + 1) * chunk] = torch.arange(chunk * cols, dtype=torch.float32).reshape(chunk, cols) + 1000.0 * rank
        ops.wait_work(ops.p2p_all_gather_rows(dist, region, chunk, rank, world, None))
        for r in range(world):
            want = torch.arange(chunk * cols, dtype=torch.float32).reshape(chunk, cols) + 1000.0 * r
            ok = ok and bool(torch.equal(region[r * chunk:(r + 1) * chunk], want))
    np.save(os.path.join(outdir, "p2p_%d.npy" % rank), np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_direct_all_gather_pairs_up(tmp_path, world):
    """ops.p2p_all_gather_rows (the direct form of the phi_vec exchange: one batch of point-to-point sends and
    receives per rank) over gloo with host tensors: every rank ends with every chunk, for 2 and 3 ranks."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_run_p2p, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert bool(np.load(os.path.join(str(tmp_path), "p2p_%d.npy" % r))[0])
