"""CPU tests of the host library (libammsb_host.so): cuckoo set, graph, split, data-set files,
generator, host mini-batch samplers.  The cuckoo set is checked against the oracle's independent
restatement (identical table image) and the reference test's membership property."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def host():
    import __graft_entry__ as ge
    ge.build()
    from mcmc_ammsb_gpu_amd import hostlib
    return hostlib


def test_cuckoo_image_matches_oracle(host, orc):
    rng = np.random.default_rng(11)
    for n in (1, 7, 1000, 100_000):
        keys = np.unique(rng.integers(0, 2**64 - 1, size=n, dtype=np.uint64))
        rng.shuffle(keys)
        hs, os_ = host.HostSet(keys), orc.OracleSet(keys)
        assert hs.BinsPerBucket() == os_.num_bins and hs.PrimeIdx() == os_.prime_idx
        assert np.array_equal(hs.Serialize(), os_.slots)
        assert hs.Has(keys).all() and hs.Size() == keys.size
    e = orc.random_graph_edges(rng, 4096, 32 * 4096)  # structured (u<<32|v) keys, wg-phi-test.cc shape
    hs, os_ = host.HostSet(e), orc.OracleSet(e)
    assert np.array_equal(hs.Serialize(), os_.slots) and hs.PrimeIdx() == os_.prime_idx


def test_cuckoo_random_membership(host):
    # cuckoo-test.cc:29-43 at the reference's size: 2M random keys, half inserted
    rng = np.random.default_rng(1)
    keys = np.unique(rng.integers(0, 2**64 - 1, size=2 * 1024 * 1024, dtype=np.uint64))
    rng.shuffle(keys)
    in_len = 1 + (keys.size + 1) // 2
    s = host.HostSet(keys[:in_len])
    assert s.Has(keys[:in_len]).all() and not s.Has(keys[in_len:]).any()


def test_generator_and_split(host):
    N, deg = 20000, 32
    edges = host.generate_graph(N, 64, deg, seed=20260101)
    assert np.array_equal(edges, host.generate_graph(N, 64, deg, seed=20260101))  # deterministic
    u, v = edges >> np.uint64(32), edges & np.uint64(0xFFFFFFFF)
    assert (u < v).all() and v.max() < N and np.unique(edges).size == edges.size
    assert abs(2 * edges.size / N - deg) < 0.05 * deg
    ds = host.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
    edges = edges[: ds.E]
    # data.cc:84-88: training_len = ceil((1 - r/2) * E); held-out = the first E - training_len edges
    tl = int(np.ceil((1 - 0.01 / 2) * edges.size))
    hl = edges.size - tl
    assert ds.training_edges.size == tl and ds.heldout_edges.size == 2 * hl
    assert np.array_equal(ds.heldout_edges[:hl], edges[:hl]) and np.array_equal(ds.training_edges, edges[hl:])
    assert ds.training.Has(ds.training_edges).all() and ds.heldout.Has(ds.heldout_edges[:hl]).all()
    fake = ds.heldout_edges[hl:]
    assert not ds.training.Has(fake).any() and not ds.heldout.Has(fake).any() and np.unique(fake).size == hl
    fu, fv = fake >> np.uint64(32), fake & np.uint64(0xFFFFFFFF)
    assert (fu < fv).all()
    # Graph adjacency (data-test.cc:27-53): symmetric, insertion order, MaxFanOut
    off, tgt = ds.training_csr()
    degs = np.diff(off.astype(np.int64))
    assert degs.sum() == 2 * tl and degs.max() == ds.max_fan_out
    tu = (ds.training_edges >> np.uint64(32)).astype(np.int64)
    tv = (ds.training_edges & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert np.array_equal(np.bincount(np.concatenate([tu, tv]), minlength=N), degs)
    node = int(tu[0])
    mine = np.concatenate([tv[tu == node], tu[tv == node]])
    assert sorted(mine.tolist()) == sorted(tgt[off[node]:off[node + 1]].tolist())


def test_dataset_files_roundtrip(host, tmp_path):
    rng = np.random.default_rng(0)
    edges = np.unique(rng.integers(0, 2**40, size=5000, dtype=np.uint64))
    path = str(tmp_path / "d.gz")
    host.dump_dataset(path, 12345, 0.25, edges)
    N, r, e = host.load_dataset(path)
    assert N == 12345 and r == 0.25 and np.array_equal(e, edges)
    # byte layout of main.cc:110-124 (gzip'd): u64 N, f32 ratio, u64 count, u64 edges[]
    import gzip
    import struct
    raw = gzip.open(path).read()
    assert struct.unpack("<QfQ", raw[:20]) == (12345, 0.25, edges.size)
    assert np.array_equal(np.frombuffer(raw[20:], dtype=np.uint64), edges)
    # SNAP text: 4 header lines, duplicate + reversed pairs collapse, ids renumbered to [0, N)
    txt = tmp_path / "g.txt"
    txt.write_text("# a\n# b\n# c\n# d\n10 20\n20 10\n10 30\n30 40\n10 20\n")
    N2, e2 = host.load_snap(str(txt))
    assert N2 == 4 and e2.size == 3
    assert ((e2 >> np.uint64(32)) < 4).all() and ((e2 & np.uint64(0xFFFFFFFF)) < 4).all()


@pytest.mark.parametrize("strategy", ["Node", "NodeLink", "NodeNonLink", "BFLink", "BFNonLink", "BF"])
def test_host_samplers(host, strategy):
    N, m = 5000, 256
    edges = host.generate_graph(N, 16, 16, seed=3)
    ds = host.Dataset.robust(N, edges, 0.02, rand_seed=2)
    edges = edges[: ds.E]
    seed = 1234
    for _ in range(6):
        e, nodes, w, seed2 = ds.sample(m, strategy, seed)
        e_again, nodes_again, w_again, _ = ds.sample(m, strategy, seed)
        assert np.array_equal(e, e_again) and np.array_equal(nodes, nodes_again) and w == w_again
        seed = seed2
        u, v = e >> np.uint64(32), e & np.uint64(0xFFFFFFFF)
        assert (u <= v).all() and np.unique(e).size == e.size
        assert set(nodes.tolist()) == set(u.tolist()) | set(v.tolist()) and np.unique(nodes).size == nodes.size
        is_link = ds.training.Has(e)
        assert is_link.all() or not is_link.any()
        if is_link.all():
            if strategy in ("Node", "NodeLink"):
                assert w == np.float32(N)                       # sample.cc:268
                hub = np.intersect1d(u, v) if e.size > 1 else u[:1]
                cand = [x for x in set(u.tolist()) | set(v.tolist()) if ((u == x) | (v == x)).all()]
                assert cand, "all link edges share one end point"
            else:
                assert e.size == m and w == np.float32(edges.size) / m  # sample.cc:238
        else:
            assert e.size == m
            assert not ds.heldout.Has(e).any() or strategy.startswith("BF")
            if strategy in ("Node", "NodeNonLink"):
                assert w == np.float32(2 * edges.size) / np.float32(m)  # sample.cc:292
            else:
                assert abs(w - (N * (N - 1) / 2.0 - edges.size) / m) <= 1e-6 * w  # sample.cc:206-207
        assert e.size <= ds.max_edges(m) and nodes.size <= ds.max_nodes(m)
