"""Pins the CPU oracle against the known-answer checks of the reference's own tests.

Each test names the reference test it mirrors (file:line).  These are the only
fixtures the reference holds for the hot path (it ships no golden vectors).
"""
import ctypes as C

import numpy as np
import pytest

LENGTHS = [1, 2, 3, 4, 5, 6, 7, 11, 31, 32, 33, 47, 48, 49, 63, 64, 65, 127, 128, 1023, 1024, 11331]
WGS = [2, 4, 16, 32, 64, 96, 113]


@pytest.mark.parametrize("wg", WGS)
def test_wg_sum_exact_integers(orc, wg):
    # wg-sum-test.cc:22-48 WgSumParameterizedTest.VaryLength
    rng = np.random.default_rng(wg)
    for v in LENGTHS:
        host = np.arange(1, v + 1, dtype=np.uint32)
        rng.shuffle(host)
        assert orc.lib().orc_wg_sum_u32(host, v, wg) == v * (v + 1) // 2


def test_wg_sum_rows_1024(orc):
    # wg-sum-test.cc:53-78 CustomSumPerformance: 1024 rows of 1..1024, wg 32
    row = np.arange(1, 1025, dtype=np.uint32)
    assert orc.lib().orc_wg_sum_u32(row, 1024, 32) == 1024 * 1025 // 2


def _ulp_diff(a, b):
    ai = np.asarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    bi = np.asarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.abs(ai - bi)


@pytest.mark.parametrize("wg", WGS)
def test_wg_normalize_4ulp(orc, wg):
    # wg-normalize-test.cc:24-48 VaryLength: ASSERT_FLOAT_EQ((i+1)/sum, host[i]) == 4 ULP
    for v in LENGTHS:
        host = np.arange(1, v + 1, dtype=np.float32)
        s = orc.lib().orc_wg_normalize_f32(host, v, wg)
        total = np.float32((v * (v + 1)) / 2.0)
        want = (np.arange(1, v + 1, dtype=np.float32) / total).astype(np.float32)
        assert _ulp_diff(host, want).max() <= 4
        assert _ulp_diff(np.float32(s), total) <= 4


def test_wg_normalize_partitioned_shape(orc):
    # wg-normalize-test.cc:135-168 PartitionedNormalizerClassTest: rows of 1..1000, wg 32, g_sum
    cols = 1000
    row = np.arange(1, cols + 1, dtype=np.float32)
    s = orc.lib().orc_wg_normalize_f32(row, cols, 32)
    total = np.float32(cols * (cols + 1) // 2)
    assert _ulp_diff(np.float32(s), total) <= 4
    want = (np.arange(1, cols + 1, dtype=np.float32) / total).astype(np.float32)
    assert _ulp_diff(row, want).max() <= 4


def test_wg_sort_matches_sort(orc):
    # wg-sort-test.cc:23-44: 256 random uints == std::sort
    rng = np.random.default_rng(7)
    for n in (2, 64, 256, 1024):
        a = rng.integers(0, 2**32, size=n, dtype=np.uint32)
        a[::7] = a[0]  # duplicates exercise the index tie-break
        out = np.zeros_like(a)
        orc.lib().orc_wg_sort_u32(a, out, n)
        assert np.array_equal(out, np.sort(a))
        f = rng.standard_normal(n).astype(np.float32)
        fo = np.zeros_like(f)
        orc.lib().orc_wg_sort_f32(f, fo, n)
        assert np.array_equal(fo, np.sort(f))


def test_cuckoo_random_membership(orc):
    # cuckoo-test.cc:29-43 CuckooSetTest.RandomMembership (scaled from 2M to 200k keys for CPU time)
    rng = np.random.default_rng(1)
    keys = np.unique(rng.integers(0, 2**64 - 1, size=200_000, dtype=np.uint64))
    rng.shuffle(keys)
    in_len = 1 + (keys.size + 1) // 2
    s = orc.OracleSet(keys[:in_len])
    assert s.num_bins == orc.lib().orc_set_num_bins(in_len)
    assert s.has(keys[:in_len]).all()
    assert not s.has(keys[in_len:]).any()
    # table image: every inserted key appears exactly once, the rest is KEY_INVALID (cuckoo.cc:91)
    live = s.slots[s.slots != np.uint64(2**64 - 1)]
    assert live.size == in_len and np.array_equal(np.sort(live), np.sort(keys[:in_len]))


def test_cuckoo_edge_keys(orc):
    # same through MakeEdge-shaped keys (u<<32|v), the form every kernel looks up (learner.cc:22-27)
    rng = np.random.default_rng(2)
    e = orc.random_graph_edges(rng, 4096, 32 * 4096)
    s = orc.OracleSet(e[: e.size // 2])
    assert s.has(e[: e.size // 2]).all() and not s.has(e[e.size // 2:]).any()


def test_rng_seed_layout(orc):
    # random-test.cc:60-63: seed[i] = {42+i, 43+i}
    seeds = orc.rng_init(1000, 42, 43)
    i = np.arange(1000, dtype=np.uint64)
    assert np.array_equal(seeds["x"], 42 + i) and np.array_equal(seeds["y"], 43 + i)


def test_rpm_addressing(orc):
    # test-partitioned-alloc.cc:14-37,39-93: 1000x1000, rows_in_block = 1000/11 -> 11+1 blocks
    rows, cols, rib = 1000, 1000, 1000 // 11
    nblocks = rows // rib + (1 if rows % rib else 0)
    assert nblocks == 12
    blk, off = C.c_uint32(), C.c_uint64()
    for r in (0, 1, rib - 1, rib, 5 * rib + 3, 999):
        orc.lib().orc_rpm_locate(rib, cols, r, C.byref(blk), C.byref(off))
        assert blk.value == r // rib and off.value == (r % rib) * cols
    # 64-bit offsets: the reference's uint product wraps at rows_in_block*cols >= 2^32
    orc.lib().orc_rpm_locate(2**21, 4096, 2**21 - 1, C.byref(blk), C.byref(off))
    assert off.value == (2**21 - 1) * 4096 > 2**32


def test_neighbor_sampler_invariants(orc):
    # wg-sample-test.cc:22-72 WgBTest.AA: N=12000, n=20; packed == non-empty slots in table order,
    # no duplicate ids, exactly capacity-n empties (marker N), never the node itself.
    N, n, wg = 12000, 20, 32
    rng = np.random.default_rng(3)
    nodes = rng.integers(0, N, size=4096, dtype=np.uint32)
    seeds = orc.rng_init(2 * 4096 * 2 * n, 56, 57)
    for _ in range(3):
        table, packed = orc.sample_neighbors(seeds, nodes, N, n, wg)
        for j in range(nodes.size):
            live = table[j][table[j] != N]
            assert live.size == n and np.array_equal(live, packed[j])
            assert np.unique(live).size == n
            assert (table[j] == N).sum() == n  # capacity 2n - n
            assert nodes[j] not in live and live.max() < N
