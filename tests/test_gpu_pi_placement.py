"""Learner._place_pi / mcmc::Learner::PlacePi: at start-up several allocations of pi are timed under update_phi and the
fastest is kept (where the table lands in HBM moves the launch by up to 10 %, profiles/README.md round 4).  It must not
change a single bit of the run: every candidate holds the same pi, the streams and call counter are restored."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import __graft_entry__ as ge
    ge.build()
    from mcmc_ammsb_gpu_amd import hostlib, learner, ops
    N = 270000  # pi = 1.1 GB at K = 1024: the smallest table the placement looks at
    edges = hostlib.generate_graph(N, 16, 12, seed=3)
    return ops, learner, hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=5)


@pytest.mark.parametrize("graph", [False, True])
def test_placement_keeps_the_trajectory_bit_for_bit(env, graph):
    ops, learner, ds = env

    def run(cands):
        cfg = learner.Config.from_cli_defaults(K=1024, mini_batch_size=2048, num_node_sample=16, strategy="Node",
                                               phi_wg_size=64, beta_wg_size=64, ppx_wg_size=64, device_sampling=True,
                                               graph_launch=graph, pi_placement_candidates=cands)
        lrn = learner.Learner(cfg, ds)
        place = lrn.pi_placement
        p0 = lrn.HeldoutPerplexity()
        lrn.Run(6)
        lrn.drain()
        out = (lrn.pi.host().copy(), ops.to_numpy(lrn.phi).copy(), ops.to_numpy(lrn.theta).copy(),
               lrn.phiUpdater.rand.host().copy(), p0, lrn.HeldoutPerplexity(), lrn.phiUpdater.count_calls)
        lrn.close()
        return place, out
    none, a = run(0)
    place, b = run(4)
    assert none is None
    assert place["candidates"] == 4 and len(place["update_phi_ms"]) == 4 and 0 <= place["kept"] < 4
    assert place["kept_ms"] == min(place["update_phi_ms"]) and all(t > 0 for t in place["update_phi_ms"])
    for x, y in zip(a, b):
        if isinstance(x, np.ndarray):
            assert np.array_equal(x, y)
        else:
            assert x == y


def test_small_tables_are_left_where_they_are(env):
    ops, learner, ds = env
    cfg = learner.Config.from_cli_defaults(K=256, mini_batch_size=512, num_node_sample=16, strategy="Node", phi_wg_size=64,
                                           beta_wg_size=64, ppx_wg_size=64, device_sampling=True)
    lrn = learner.Learner(cfg, ds)   # 276 MB: a cache matter, not a placement one
    assert lrn.pi_placement is None
    lrn.close()


def test_cpp_learner_places_pi_and_keeps_the_checkpoint(env, tmp_path):
    """ammsb_main: the same run with and without the placement writes the same checkpoint, and says what it kept."""
    exe = os.path.join(ROOT, "mcmc-ammsb-gpu_amd", "ammsb_main")
    if not os.path.exists(exe):
        pytest.fail("ammsb_main is not built")
    from mcmc_ammsb_gpu_amd import hostlib
    f = str(tmp_path / "g.bin.gz")
    hostlib.dump_dataset(f, 270000, 0.01, hostlib.generate_graph(270000, 16, 12, seed=3))
    common = [exe, "--load-data", "1", "--load-file", f, "-k", "1024", "-m", "2048", "-n", "16", "-x", "4", "-i", "2",
              "--phi-wg", "64", "--beta-wg", "64", "--ppx-wg", "64", "--device-sampling", "1", "--async", "1", "--graph", "1"]
    outs = []
    for cands in ("0", "3"):
        ck = str(tmp_path / ("c%s.ckpt" % cands))
        r = subprocess.run(common + ["--pi-candidates", cands, "--checkpoint-out", ck], capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append((open(ck, "rb").read(), r.stderr))
    assert "pi placement" not in outs[0][1]
    assert "pi placement: update_phi over 3 allocations" in outs[1][1]
    # every buffer record (pi, phi_sum, theta, beta, RNG streams, pending samples) byte for byte; the short records
    # carry the operators' accumulated device times, which no two runs share (serialize.h:13-24: u64 length + payload)
    def records(data):
        recs, pos = [], 0
        while pos < len(data):
            (n,) = struct.unpack_from("<Q", data, pos)
            recs.append(data[pos + 8:pos + 8 + n])
            pos += 8 + n
        assert pos == len(data)
        return recs
    ra, rb = records(outs[0][0]), records(outs[1][0])
    assert len(ra) == len(rb) and sum(len(a) >= 200 for a in ra) >= 6
    for i, (a, b) in enumerate(zip(ra, rb)):
        assert len(a) == len(b), i
        if len(a) >= 200:
            assert a == b, "record %d (%d bytes) differs" % (i, len(a))
