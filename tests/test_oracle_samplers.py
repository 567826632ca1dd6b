"""a9 (host mini-batch samplers): oracle/ammsb_oracle_samplers.c -- a plain-C restatement of sample.cc:177-303 and
learner.cc:162-173, including the libstdc++ unordered_set iteration order they depend on -- against (1) the real
container and (2) the product's host samplers (libammsb_host.so), bit for bit, for seeds 1..4 and all six
strategies ("host sampler outputs (edges, weight, node order) for seeds 1..4")."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def real(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("uset") / "libuset.so")
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "cpp", "uset_order.cc")])
    L = C.CDLL(so)
    L.real_uset_order_u64.restype = C.c_uint64
    L.real_uset_order_u64.argtypes = [np.ctypeslib.ndpointer(np.uint64), C.c_uint64, np.ctypeslib.ndpointer(np.uint64)]
    L.real_uset_order_u32.restype = C.c_uint64
    L.real_uset_order_u32.argtypes = [np.ctypeslib.ndpointer(np.uint32), C.c_uint64, np.ctypeslib.ndpointer(np.uint32)]
    return L


def test_unordered_set_order_matches_libstdcxx(orc, real):
    rng = np.random.default_rng(7)
    cases = [np.arange(0, dtype=np.uint64), np.array([5], dtype=np.uint64), np.arange(1, 14, dtype=np.uint64)]
    for n in (2, 11, 12, 13, 14, 29, 30, 97, 98, 1000, 5087, 65536, 200003):   # around every early rehash
        cases.append(rng.integers(0, 2**20, n, dtype=np.uint64))                 # vertex-like keys, duplicates
        cases.append((rng.integers(0, 2**20, n, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, 2**20, n, dtype=np.uint64))
    cases.append(np.repeat(rng.integers(0, 50, 400, dtype=np.uint64), 3))
    for keys in cases:
        keys = np.ascontiguousarray(keys)
        want = np.zeros(max(keys.size, 1), dtype=np.uint64)
        n = real.real_uset_order_u64(keys, keys.size, want)
        got = orc.uset_order(keys)
        assert got.size == n and np.array_equal(got, want[:n]), keys.size
        if keys.size and keys.max() < 2**32:   # std::unordered_set<Vertex>: same hash values, same order
            k32 = keys.astype(np.uint32)
            w32 = np.zeros(max(k32.size, 1), dtype=np.uint32)
            n32 = real.real_uset_order_u32(k32, k32.size, w32)
            assert n32 == n and np.array_equal(w32[:n32].astype(np.uint64), got)


class _OSet:
    def __init__(self, hs):
        self.slots, self.num_bins, self.prime_idx = hs.Serialize(), hs.BinsPerBucket(), hs.PrimeIdx()


@pytest.fixture(scope="module")
def dataset():
    import __graft_entry__ as ge
    ge.build()
    from mcmc_ammsb_gpu_amd import hostlib
    N = 3000
    edges = hostlib.generate_graph(N, 8, 12, seed=5)
    return hostlib.Dataset.robust(N, edges, heldout_ratio=0.05, rand_seed=3)


@pytest.mark.parametrize("strategy", ["Node", "NodeLink", "NodeNonLink", "BFLink", "BFNonLink", "BF"])
def test_host_samplers_match_the_oracle(orc, dataset, strategy):
    ds = dataset
    m = 256
    tset, hset = _OSet(ds.training), _OSet(ds.heldout)
    for seed0 in (1, 2, 3, 4):
        a, b = seed0, seed0
        for it in range(6):   # the seed stream carries over from one mini-batch to the next (Sample::seed)
            e1, v1, w1, a = ds.sample(m, strategy, a)
            e2, v2, w2, b = orc.host_sample(ds.N, ds.E, m, strategy, b, ds.training_edges, tset, hset,
                                            ds.max_edges(m), ds.max_nodes(m))
            assert a == b, "rand_r stream diverged"
            assert np.array_equal(e1, e2), "edge order"
            assert np.array_equal(v1, v2), "node order"
            assert np.float32(w1).tobytes() == np.float32(w2).tobytes(), "weight"


def test_host_sample_reports_overflow_instead_of_writing(dataset):
    """the C entry point takes the capacities and refuses a mini-batch that does not fit."""
    from mcmc_ammsb_gpu_amd import hostlib
    ds = dataset
    e = np.zeros(8, dtype=np.uint64)
    v = np.zeros(8, dtype=np.uint32)
    ne, nv, w, s = C.c_uint64(), C.c_uint64(), C.c_float(), C.c_uint(1)
    rc = ds.lib.ammsb_host_sample(ds._h, ds.N, ds.E, 256, hostlib.STRATEGIES["NodeNonLink"], C.byref(s), e, e.size,
                                  C.byref(ne), v, v.size, C.byref(nv), C.byref(w))
    assert rc == -2 and ne.value == 256 and nv.value >= 256
    assert not e.any() and not v.any()   # nothing was copied


def test_golden_host_sampler_vectors(orc):
    """tests/golden/host_samplers.npz (made by the oracle, see make_golden_samplers.py): today's oracle build -- on
    whatever glibc / libstdc++ this host has -- reproduces every vector (edges, node order, weight, rand_r seed) bit
    for bit.  test_host_samplers_match_the_oracle ties the product's host library to the same oracle."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mgs", os.path.join(HERE, "golden", "make_golden_samplers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = np.load(os.path.join(HERE, "golden", "host_samplers.npz"))
    now = mod.compute()
    assert set(now) == set(g.files)
    for k in g.files:
        assert np.array_equal(g[k], now[k]), k


def test_device_sampler_statement_properties(orc, dataset):
    """The serial statement of the DEVICE sampler (row f1; the -m gpu suite compares the kernels with it bit for bit):
    checked here against its own definition with numpy -- draw = rand(stream j) mod N, validity, first occurrence among
    the valid candidates, candidate order, count, padding, weights."""
    ds = dataset
    tr, ho = _OSet(ds.training), _OSet(ds.heldout)
    N, m, C_ = ds.N, 256, 1024
    seeds = orc.rng_init_mixed(C_, 1234, 5678)
    s0 = seeds.copy()
    # the mixed seeding: SplitMix64 finaliser of sx + 2 i / sy + 2 i + 1
    def mix(z):
        z = (z + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return z ^ (z >> 31)
    for i in (0, 1, 77, C_ - 1):
        assert int(s0["x"][i]) == mix(1234 + 2 * i) and int(s0["y"][i]) == mix(5678 + 2 * i + 1)
    u = 5
    e, v, cnt = orc.device_minibatch_nonlink(seeds, C_, u, m, N, tr, ho)
    # the draws, from an independent evaluation of xorshift128+ (random.cl.inc:13-25) on the initial states
    M = 2**64 - 1
    draws = []
    for i in range(C_):
        s1, s0_ = int(s0["x"][i]), int(s0["y"][i])
        x = s0_
        s1 ^= (s1 << 23) & M
        y = s1 ^ s0_ ^ (s1 >> 17) ^ (s0_ >> 26)
        draws.append(((y + s0_) & M) % N)
        assert int(seeds["x"][i]) == x and int(seeds["y"][i]) == y  # every candidate stream advanced by one draw
    draws = np.array(draws, dtype=np.uint64)
    keys = orc.make_edge(np.full(C_, u, dtype=np.uint64), draws)
    valid = (draws != u) & ~ds.training.Has(keys) & ~ds.heldout.Has(keys)
    seen, kept = set(), []
    for j in range(C_):
        if valid[j] and int(draws[j]) not in seen:
            seen.add(int(draws[j]))
            kept.append(j)
    assert cnt == len(kept) >= m
    assert np.array_equal(v[1:], draws[kept[:m]].astype(np.uint32)) and v[0] == u
    assert np.array_equal(e, keys[kept[:m]])
    # shortfall: fewer candidates than needed -> count < m, the tail repeats the head
    seeds2 = orc.rng_init_mixed(C_, 9, 10)
    e2, v2, cnt2 = orc.device_minibatch_nonlink(seeds2, 64, u, m, N, tr, ho)
    assert 0 < cnt2 <= 64 < m
    assert np.array_equal(e2[cnt2:], e2[np.arange(cnt2, m) % cnt2]) and np.array_equal(v2[1 + cnt2:], v2[1 + np.arange(cnt2, m) % cnt2])
    # link half and weights (sample.cc:252-268, :292)
    off, tgt = ds.training_csr()
    uu = int(np.flatnonzero(np.diff(off.astype(np.int64)) > 2)[0])
    le, lv, ln = orc.device_minibatch_link(off, tgt, uu)
    nb = tgt[int(off[uu]):int(off[uu + 1])]
    assert ln == nb.size and lv[0] == uu and np.array_equal(lv[1:], nb)
    assert np.array_equal(le, orc.make_edge(np.full(ln, uu, dtype=np.uint64), nb.astype(np.uint64))) and ds.training.Has(le).all()
    assert orc.device_minibatch_weight(1, N, ds.E, m) == float(np.float32(N))
    assert orc.device_minibatch_weight(0, N, ds.E, m) == float(np.float32(2 * ds.E) / np.float32(m))
