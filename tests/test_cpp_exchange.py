"""mcmc::Exchange (include/mcmc/exchange.h), the C++ learner's multi-GPU transport, and the sharded mcmc::Learner
on top of it.  CPU: the rendezvous and the host-memory collectives with 3 ranks.  GPU: the device collectives
("host" transport with 3 ranks on the one card; "rccl" with the single rank RCCL accepts there), then
ammsb_main --exchange host with 2 and 3 ranks against a single rank on the same data: pi, phi_sum, every RNG
stream and the running perplexity means bit-identical, theta / beta within the gradient sum's re-association."""
import os
import re
import socket
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mcmc-ammsb-gpu_amd")
# (tools/run_asan.sh points these at sanitizer builds)
XT = os.environ.get("AMMSB_XT_EXE") or os.path.join(PKG, "exchange_test")
EXE = os.environ.get("AMMSB_MAIN_EXE") or os.path.join(PKG, "ammsb_main")


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    assert os.path.exists(XT) and os.path.exists(EXE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ranks(argv, world, timeout=600, per_rank_args=None):
    """Start `world` copies of argv, one per rank, all on device 0; returns the CompletedProcess-like results."""
    port, xport = _free_port(), _free_port()   # the launcher hands the rendezvous its own free port (not MASTER_PORT + 1)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), AMMSB_EXCHANGE_PORT=str(xport), HSA_ENABLE_IPC_MODE_LEGACY="0")
        extra = per_rank_args(r) if per_rank_args else []
        procs.append(subprocess.Popen(argv + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    out = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=timeout)
            out.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return out


def test_rendezvous_and_host_collectives(built):
    res = _ranks([XT, "host", "hostmem"], 3, timeout=120)
    for r, (rc, o, e) in enumerate(res):
        assert rc == 0, e
        assert o.strip() == "OK %d of 3 host" % r


def test_single_rank_needs_no_peer(built):
    env = dict(os.environ, RANK="0", WORLD_SIZE="1")
    r = subprocess.run([XT, "host", "hostmem"], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "OK 0 of 1 host"


def test_bad_rank_is_an_error(built):
    env = dict(os.environ, RANK="3", WORLD_SIZE="2")
    r = subprocess.run([XT, "host", "hostmem"], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "bad RANK / WORLD_SIZE" in r.stderr
    env = dict(os.environ, RANK="0", WORLD_SIZE="1")
    r = subprocess.run([XT, "carrier-pigeon", "hostmem"], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "unknown exchange kind" in r.stderr


def test_rendezvous_gives_up_on_a_missing_peer_and_ignores_strays(built):
    """Rank 0 of a 2-rank job whose peer never starts must fail after the time-out, not hang; a stray
    connection to the rendezvous port (not one of the job's ranks) is dropped, not fatal."""
    xport = _free_port()
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               AMMSB_EXCHANGE_PORT=str(xport), AMMSB_EXCHANGE_TIMEOUT_S="3")
    p = subprocess.Popen([XT, "host", "hostmem"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    import time
    for _ in range(50):  # a stray: connects, sends a nonsense rank
        try:
            with socket.create_connection(("127.0.0.1", xport), timeout=1) as c:
                c.sendall((12345).to_bytes(4, "little"))
            break
        except OSError:
            time.sleep(0.1)
    o, e = p.communicate(timeout=60)
    assert p.returncode != 0 and "rendezvous timed out: 1 rank(s) never connected" in e, (o, e)
    # a real peer after a stray is accepted
    xport = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                   AMMSB_EXCHANGE_PORT=str(xport))
        procs.append(subprocess.Popen([XT, "host", "hostmem"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        if r == 0:
            for _ in range(50):
                try:
                    with socket.create_connection(("127.0.0.1", xport), timeout=1) as c:
                        c.sendall((-7).to_bytes(4, "little", signed=True))
                    break
                except OSError:
                    time.sleep(0.1)
    for r, p in enumerate(procs):
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0 and o.startswith("OK"), (r, o, e)


@pytest.mark.gpu
def test_device_collectives_host_transport(built):
    for rc, o, e in _ranks([XT, "host", "device"], 3, timeout=300):
        assert rc == 0 and o.startswith("OK"), e


@pytest.mark.gpu
def test_device_collectives_rccl_single_rank(built):
    # RCCL refuses two ranks on one device; the one-rank communicator still goes through init and all three calls
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([XT, "rccl", "device"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "OK 0 of 1 rccl", r.stderr  # after RCCL's banner
    # the direct form of the in-place all-gather (grouped ncclSend / ncclRecv): with one rank an empty group
    r = subprocess.run([XT, "rccl", "device"], env=dict(env, AMMSB_EXCHANGE_FORM="p2p"), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "OK 0 of 1 rccl", r.stderr


# ---------------------------------------------------------------- the sharded learner


def _records(path):
    """The checkpoint as its list of length-prefixed records (serialize.h:13-24)."""
    recs = []
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        (n,) = struct.unpack_from("<Q", data, pos)
        recs.append(data[pos + 8:pos + 8 + n])
        pos += 8 + n
    assert pos == len(data)
    return recs


def _ppx_lines(stderr):
    return [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"ppx\[(\d+)\] = ([0-9.eE+-]+)", stderr)]


def _graph_file(path, N, deg, seed):
    rng = np.random.default_rng(seed)
    K = 16
    comm = rng.integers(0, K, N)
    src = rng.integers(0, N, N * deg // 2)
    same = rng.random(src.size) < 0.8
    order = np.argsort(comm, kind="stable")
    starts = np.searchsorted(comm[order], np.arange(K + 1))
    pick = rng.random(src.size)
    c = comm[src]
    dst_same = order[(starts[c] + (pick * (starts[c + 1] - starts[c])).astype(np.int64)).clip(0, N - 1)]
    dst = np.where(same, dst_same, rng.integers(0, N, src.size))
    keep = src != dst
    with open(path, "w") as f:
        f.write("# synthetic\n# Nodes: %d\n#\n#\n" % N)
        np.savetxt(f, np.stack([src[keep], dst[keep]], 1), fmt="%d", delimiter="\t")


def _heldout_edges(d):
    import math
    import ammsb_pkg
    ammsb_pkg.load()
    from mcmc_ammsb_gpu_amd import hostlib
    N, ratio, edges = hostlib.load_dataset(d)
    return 2 * (edges.size - math.ceil((1 - ratio / 2) * edges.size))  # links + as many fake pairs (data.cc:87-126)


SHARDED_GRADS = ("--beta-grads", "0", "--beta-shard-min-edges", "0")  # every mini-batch's gradient cut over the ranks


def _compare(single_ck, rank_ck, K, H, exact=False):
    ref = _records(single_ck)
    if exact:  # every rank computed the whole gradient: the checkpoint is the single rank's, byte for byte
        got = _records(rank_ck)
        assert len(got) == len(ref)
        for i, (a, b) in enumerate(zip(ref, got)):
            if len(a) >= 200:
                assert a == b, "record %d (%d bytes) differs" % (i, len(a))
            elif i < 2:
                assert a[-8 * K:] == b[-8 * K:], "theta / beta differ"
        return
    for ck in [rank_ck]:
        got = _records(ck)
        assert len(got) == len(ref)
        # the held-out calculator's running means: the one buffer of H floats
        i_ppx = [i for i, a in enumerate(ref) if 4 * H < len(a) <= 4 * H + 12]
        assert len(i_ppx) == 1
        i_ppx = i_ppx[0]
        for i, (a, b) in enumerate(zip(ref, got)):
            assert len(a) == len(b), i
            if i < 2:  # beta, theta: [2K] floats after the record header
                fa = np.frombuffer(a[-8 * K:], dtype=np.float32)
                fb = np.frombuffer(b[-8 * K:], dtype=np.float32)
                np.testing.assert_allclose(fb, fa, rtol=2e-4, atol=1e-7)
            elif i == i_ppx:  # the running means: call 1 saw beta_0 (exact), call 2 the re-associated beta_1
                fa = np.frombuffer(a[-4 * H:], dtype=np.float32)
                fb = np.frombuffer(b[-4 * H:], dtype=np.float32)
                np.testing.assert_allclose(fb, fa, rtol=2e-4)
            elif len(a) >= 200:  # every buffer: pi blocks, phi_sum, RNG streams, running means, the pending samples
                assert a == b, "record %d (%d bytes) differs" % (i, len(a))

def _dataset(tmp_path, N, deg, flags, seed):
    g, d = str(tmp_path / "g.txt"), str(tmp_path / "g.bin.gz")
    _graph_file(g, N, deg, seed=seed)  # seeds whose held-out cuckoo set builds (the reference's Set can fail, cuckoo.cc:117-129)
    r = subprocess.run([EXE, "-f", g, "--dump-data", "1", "--dump-file", d] + flags, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return d


def _run_case(tmp_path, d, world, flags, iters, interval, ppx_rtol, tag, timeout=900,
              split=("--phi-replicate", "0", "--phi-chunks", "1") + SHARDED_GRADS):
    # (`split`: the ranks' sharding flags -- default: every group exchanged in one chunk and the gradient cut over the
    # ranks, the plain schedule)
    common = [EXE, "--load-data", "1", "--load-file", d] + flags + ["-x", str(iters), "-i", str(interval)]
    ck1 = str(tmp_path / (tag + "single.ckpt"))
    one = subprocess.run(common + ["--checkpoint-out", ck1], capture_output=True, text=True, timeout=timeout)
    assert one.returncode == 0, one.stderr
    res = _ranks(common + ["--exchange", "host"] + list(split), world, timeout=timeout,
                 per_rank_args=lambda r: ["--checkpoint-out", str(tmp_path / ("%srank%d.ckpt" % (tag, r)))])
    ppx1 = _ppx_lines(one.stderr)
    assert len(ppx1) == iters // interval + 1
    first = None
    for rc, o, e in res:
        assert rc == 0, e
        ppx = _ppx_lines(e)
        assert [s for s, _ in ppx] == [s for s, _ in ppx1]
        np.testing.assert_allclose([p for _, p in ppx], [p for _, p in ppx1], rtol=ppx_rtol)
        first = first or ppx
        assert ppx == first  # every rank reports the same values
    return ck1, str(tmp_path / (tag + "rank0.ckpt"))  # the other ranks write to /dev/null: rank 0's is the checkpoint


# One iteration is exact in everything but theta / beta: phi and pi read beta_0, which every run shares, and the
# gradient's per-rank partials re-associate one float sum.  From the second iteration on beta differs in its last
# bits and the trajectories drift apart like any two roundings of the same chain; those runs compare perplexities.

@pytest.mark.gpu
def test_two_ranks_small_minibatches_host_sampler(built, tmp_path):
    """Every launch fits rank 0's block (the broadcast path); reference loop with the host samplers."""
    K = 32
    flags = ["-k", str(K), "-m", "64", "-n", "16", "-r", "0.05", "--phi-wg", "64", "--beta-wg", "64", "--ppx-wg", "64"]
    d = _dataset(tmp_path, 3000, 12, flags, seed=12)
    _compare(*_run_case(tmp_path, d, 2, flags, iters=1, interval=1, ppx_rtol=1e-5, tag="a"), K, _heldout_edges(d))
    _run_case(tmp_path, d, 2, flags, iters=150, interval=50, ppx_rtol=3e-3, tag="b")


@pytest.mark.gpu
def test_three_ranks_full_blocks_and_tail_rows(built, tmp_path):
    """m = 65536 non-link mini-batches: 65537 nodes > MAX_GROUPS, so the all-gather covers every block and the two
    tail rows are parked and handed out; device sampling, enqueue-only loop."""
    K = 32
    flags = ["-k", str(K), "-m", "65536", "-n", "32", "-r", "0.02", "--phi-wg", "32", "--beta-wg", "32", "--ppx-wg", "32",
             "--device-sampling", "1", "--async", "1"]
    d = _dataset(tmp_path, 150000, 10, flags, seed=11)
    _compare(*_run_case(tmp_path, d, 3, flags + ["-s", "NodeNonLink"], iters=1, interval=1, ppx_rtol=1e-5, tag="a"), K, _heldout_edges(d))
    _run_case(tmp_path, d, 3, flags + ["-s", "Node"], iters=8, interval=4, ppx_rtol=3e-3, tag="b")


@pytest.mark.gpu
@pytest.mark.parametrize("split,tag", [
    (("--phi-replicate", "0.3", "--phi-chunks", "2") + SHARDED_GRADS, "hybrid"),      # 30 % of the groups replicated, two overlapped chunks per rank
    (("--phi-replicate", "0.00003", "--phi-chunks", "3") + SHARDED_GRADS, "mixed"),   # ONE replicated group: replicated and exchanged tail rows
    (("--phi-replicate", "-1") + SHARDED_GRADS, "auto"),                              # measured at start-up (whatever it comes out as)
])
def test_hybrid_split_and_chunked_exchange(built, tmp_path, split, tag):
    """The tuned schedule of learner.py behind the C++ API (Config::phi_replicate / phi_chunks): replicated prefix,
    world * chunks blocks exchanged on their own stream beside the next block's update_phi, tail rows of both kinds.
    One iteration: pi, phi_sum, every RNG stream and the pending samples byte-identical to a single rank's; the
    checkpoint (a collective) carries the owners' streams whatever the split."""
    K = 32
    flags = ["-k", str(K), "-m", "65536", "-n", "32", "-r", "0.02", "--phi-wg", "32", "--beta-wg", "32", "--ppx-wg", "32",
             "--device-sampling", "1", "--async", "1"]
    d = _dataset(tmp_path, 150000, 10, flags, seed=11)
    _compare(*_run_case(tmp_path, d, 3, flags + ["-s", "NodeNonLink"], iters=1, interval=1, ppx_rtol=1e-5, tag=tag + "a",
                        split=split), K, _heldout_edges(d))
    _run_case(tmp_path, d, 2, flags + ["-s", "Node"], iters=6, interval=3, ppx_rtol=3e-3, tag=tag + "b", split=split)


@pytest.mark.gpu
@pytest.mark.parametrize("K,wg,extra,tag", [
    (32, 32, ("--device-sampling", "1", "--async", "1"), "fused"),   # update_pi folded into the gradient launch (default: auto)
    (256, 64, ("--device-sampling", "1", "--async", "1"), "fused256"),
    (32, 32, ("--beta-grads", "1"), "host"),                         # host-sampled mini-batches: separate launches
])
def test_replicated_gradient_is_the_single_rank_run_byte_for_byte(built, tmp_path, K, wg, extra, tag):
    """Config::beta_grads: every rank computes the whole gradient (no collective for it) -- after six iterations of
    link and non-link mini-batches the two-rank checkpoint, theta and beta included, is the single rank's."""
    m = "16384" if K == 32 else "4096"
    flags = ["-k", str(K), "-m", m, "-n", "16", "-r", "0.02", "--phi-wg", str(wg), "--beta-wg", str(wg), "--ppx-wg", str(wg)]
    flags += list(extra)
    d = _dataset(tmp_path, 60000, 10, flags, seed=11)
    ck = _run_case(tmp_path, d, 2, flags + ["-s", "Node"], iters=6, interval=3, ppx_rtol=1e-6, tag=tag,
                   split=("--phi-replicate", "0.3", "--phi-chunks", "2"))
    _compare(*ck, K, _heldout_edges(d), exact=True)
