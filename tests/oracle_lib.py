"""ctypes binding of oracle/libammsb_oracle.so -- the CPU checker.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")

SEED_DT = np.dtype([("x", np.uint64), ("y", np.uint64)])
seedp = np.ctypeslib.ndpointer(SEED_DT, flags="C_CONTIGUOUS")


class Params(C.Structure):
    _fields_ = [("N", C.c_uint64), ("K", C.c_uint64), ("n_neighbors", C.c_uint32),
                ("alpha", C.c_float), ("a", C.c_float), ("b", C.c_float), ("c", C.c_float),
                ("epsilon", C.c_float), ("eta0", C.c_float), ("eta1", C.c_float)]


class SetT(C.Structure):
    _fields_ = [("slots", C.POINTER(C.c_uint64)), ("num_bins", C.c_uint64),
                ("prime_idx", C.c_uint32), ("count", C.c_uint64)]


class PpxSums(C.Structure):
    _fields_ = [("link_ll", C.c_double), ("nonlink_ll", C.c_double),
                ("link_cnt", C.c_uint64), ("nonlink_cnt", C.c_uint64)]


def build(path=None, archflags=None):
    """(Re)build the oracle shared object; returns its path."""
    out = path or os.path.join(ORACLE_DIR, "libammsb_oracle.so")
    cmd = ["make", "-s", "-C", ORACLE_DIR, "OUT=" + out]
    if archflags:
        cmd.append("ARCHFLAGS=" + archflags)
    subprocess.check_call(cmd)
    return out


def lib(path=None):
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    so = path or os.environ.get("AMMSB_ORACLE_LIB") or os.path.join(ORACLE_DIR, "libammsb_oracle.so")  # override: sanitizer build
    if not os.path.exists(so):
        build(so)
    L = C.CDLL(so)
    sig = {
        "orc_quantize_param": (C.c_float, [C.c_float]),
        "orc_eps_t": (C.c_float, [C.POINTER(Params), C.c_uint32]),
        "orc_rng_init": (None, [seedp, C.c_uint64, C.c_uint64, C.c_uint64]),
        "orc_fill_rand": (None, [seedp, u64p, C.c_uint64]),
        "orc_fill_random": (None, [seedp, f32p, C.c_uint64]),
        "orc_fill_randn": (None, [seedp, f32p, C.c_uint64]),
        "orc_fill_gamma": (None, [seedp, C.c_float, C.c_float, f32p, C.c_uint64]),
        "orc_randint": (C.c_int32, [seedp, C.c_int32, C.c_int32]),
        "orc_set_num_bins": (C.c_uint64, [C.c_uint64]),
        "orc_set_build": (C.c_int, [C.POINTER(SetT), u64p, C.c_uint64]),
        "orc_set_free": (None, [C.POINTER(SetT)]),
        "orc_set_has_many": (None, [u64p, C.c_uint64, C.c_uint32, u64p, C.c_uint64, u8p]),
        "orc_rpm_locate": (None, [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_uint64)]),
        "orc_wg_sum_f32": (C.c_float, [f32p, C.c_uint32, C.c_uint32]),
        "orc_wg_sum_u32": (C.c_uint32, [u32p, C.c_uint32, C.c_uint32]),
        "orc_wg_normalize_f32": (C.c_float, [f32p, C.c_uint32, C.c_uint32]),
        "orc_wg_sort_u32": (None, [u32p, u32p, C.c_uint32]),
        "orc_wg_sort_f32": (None, [f32p, f32p, C.c_uint32]),
        "orc_pi_init_gamma": (None, [f32p, f32p, C.c_uint64, C.c_uint64, C.c_float, C.c_float,
                                     C.c_uint64, C.c_uint64]),
        "orc_sample_neighbors": (None, [seedp, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                        u32p, u32p]),
        "orc_update_phi": (None, [C.POINTER(Params), f32p, f32p, f32p, u64p, C.c_uint64, C.c_uint32,
                                  u32p, u32p, C.c_uint32, C.c_uint32, seedp, C.c_uint32, C.c_int,
                                  C.c_int, f32p]),
        "orc_update_pi": (None, [C.POINTER(Params), f32p, f32p, f32p, u32p, C.c_uint32, C.c_uint32,
                                 C.c_int]),
        "orc_sum_theta": (None, [f32p, f32p, C.c_uint64]),
        "orc_beta_grads": (None, [C.POINTER(Params), f32p, f32p, f32p, f32p, u64p, C.c_uint64,
                                  C.c_uint32, u64p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, f32p]),
        "orc_update_theta": (None, [C.POINTER(Params), f32p, f32p, C.c_uint32, C.c_float, seedp,
                                    C.c_int]),
        "orc_beta_from_theta": (None, [f32p, f32p, C.c_uint64]),
        "orc_perplexity": (None, [C.POINTER(Params), f32p, f32p, u64p, C.c_uint64, C.c_uint32, u64p,
                                  C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, f32p, C.c_void_p,
                                  C.POINTER(PpxSums)]),
        "orc_ppx_value": (C.c_double, [C.POINTER(PpxSums)]),
        "orc_num_threads": (C.c_int, []),
        "orc_set_num_threads": (None, [C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _LIB = L
    return L


# ---------------------------------------------------------------- helpers

def make_params(N, K, n, alpha=None, a=0.0315, b=1024.0, c=0.5, epsilon=1e-7, eta0=1.0, eta1=1.0):
    """Kernel constants as the reference bakes them (config.cc:57-83): floats go through "%e"."""
    q = lib().orc_quantize_param
    if alpha is None:
        alpha = np.float32(1.0) / np.float32(K)  # main.cc:153
    return Params(N, K, n, q(alpha), q(a), q(b), q(c), q(epsilon), q(eta0), q(eta1))


def rng_init(n, sx, sy):
    seeds = np.zeros(n, dtype=SEED_DT)
    lib().orc_rng_init(seeds, n, sx, sy)
    return seeds


class OracleSet:
    """Host cuckoo set (cuckoo.cc:92-220) built by the oracle."""

    def __init__(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        self._s = SetT()
        rc = lib().orc_set_build(C.byref(self._s), keys, keys.size)
        if rc != 0:
            raise RuntimeError("orc_set_build failed rc=%d" % rc)
        self.num_bins = int(self._s.num_bins)
        self.prime_idx = int(self._s.prime_idx)
        cap = 2 * self.num_bins * 4
        self.slots = np.ctypeslib.as_array(self._s.slots, shape=(cap,)).copy()
        lib().orc_set_free(C.byref(self._s))

    def has(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.zeros(keys.size, dtype=np.uint8)
        lib().orc_set_has_many(self.slots, self.num_bins, self.prime_idx, keys, keys.size, out)
        return out.astype(bool)


def make_edge(u, v):
    u = np.asarray(u, dtype=np.uint64)
    v = np.asarray(v, dtype=np.uint64)
    lo, hi = np.minimum(u, v), np.maximum(u, v)
    return (lo << np.uint64(32)) | hi


def random_graph_edges(rng, N, n_edges):
    """Unique, canonical (u<v), shuffled random edges -- the shape the reference tests use
    (wg-phi-test.cc:26-39)."""
    u = rng.integers(0, N, size=int(n_edges * 1.1) + 16, dtype=np.uint64)
    v = rng.integers(0, N, size=u.size, dtype=np.uint64)
    keep = u != v
    e = np.unique(make_edge(u[keep], v[keep]))
    rng.shuffle(e)
    return e[:n_edges].copy()


def update_phi(p, beta, pi, phi_sum, oset, nodes, neighbors, step, seeds, L, mode_wg, noise_on):
    out = np.zeros((nodes.size, p.K), dtype=np.float32)
    lib().orc_update_phi(C.byref(p), beta, pi, phi_sum, oset.slots, oset.num_bins, oset.prime_idx,
                         nodes, neighbors, nodes.size, step, seeds, L, int(mode_wg), int(noise_on), out)
    return out


def update_pi(p, pi, phi_sum, phi_vec, nodes, L, mode_wg):
    lib().orc_update_pi(C.byref(p), pi, phi_sum, phi_vec, nodes, nodes.size, L, int(mode_wg))


def beta_grads(p, theta, beta, pi, oset, edges, L, mode_wg, order=0):
    theta_sum = np.zeros(p.K, dtype=np.float32)
    lib().orc_sum_theta(theta, theta_sum, p.K)
    grads = np.zeros(2 * p.K, dtype=np.float32)
    lib().orc_beta_grads(C.byref(p), theta, theta_sum, beta, pi, oset.slots, oset.num_bins,
                         oset.prime_idx, edges, edges.size, L, int(mode_wg), int(order), grads)
    return grads


def update_theta(p, theta, grads, step, scale, seeds, noise_on=True):
    lib().orc_update_theta(C.byref(p), theta, grads, step, np.float32(scale), seeds, int(noise_on))
    beta = np.zeros_like(theta)
    lib().orc_beta_from_theta(theta, beta, p.K)
    return beta


def perplexity(p, beta, pi, oset, edges, call_count, L, mode_wg, ppx_per_edge, want_ll=False):
    sums = PpxSums()
    ll = np.zeros(edges.size, dtype=np.float32) if want_ll else None
    lib().orc_perplexity(C.byref(p), beta, pi, oset.slots, oset.num_bins, oset.prime_idx, edges,
                         edges.size, call_count, L, int(mode_wg), ppx_per_edge,
                         ll.ctypes.data if want_ll else None, C.byref(sums))
    return sums, ll


def pi_init_gamma(N, K, eta0=1.0, eta1=1.0, sx=11, sy=113):
    pi = np.zeros((N, K), dtype=np.float32)
    phi_sum = np.zeros(N, dtype=np.float32)
    lib().orc_pi_init_gamma(pi.reshape(-1), phi_sum, N, K, eta0, eta1, sx, sy)
    return pi, phi_sum


def sample_neighbors(seeds, nodes, N, n, wg):
    table = np.zeros((nodes.size, 2 * n), dtype=np.uint32)
    packed = np.zeros((nodes.size, n), dtype=np.uint32)
    lib().orc_sample_neighbors(seeds, nodes, nodes.size, N, n, wg, table.reshape(-1), packed.reshape(-1))
    return table, packed


def _bind_samplers(L):
    L.orc_uset_order.restype = C.c_uint64
    L.orc_uset_order.argtypes = [u64p, C.c_uint64, u64p]
    L.orc_host_sample.restype = C.c_int
    L.orc_host_sample.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint), u64p, C.c_uint64,
                                  u64p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32,
                                  u64p, C.c_uint64, C.POINTER(C.c_uint64), u32p, C.c_uint64, C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_float)]
    L.orc_rng_init_mixed.restype = None
    L.orc_rng_init_mixed.argtypes = [np.ctypeslib.ndpointer(SEED_DT, flags="C"), C.c_uint64, C.c_uint64, C.c_uint64]
    L.orc_device_minibatch_nonlink.restype = C.c_int
    L.orc_device_minibatch_nonlink.argtypes = [np.ctypeslib.ndpointer(SEED_DT, flags="C"), C.c_uint32, C.c_uint32, C.c_uint32,
                                               C.c_uint64, u64p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32,
                                               u64p, u32p, u32p]
    L.orc_device_minibatch_link.restype = C.c_int
    L.orc_device_minibatch_link.argtypes = [u64p, u32p, C.c_uint32, u64p, u32p, u32p]
    L.orc_device_minibatch_weight.restype = C.c_float
    L.orc_device_minibatch_weight.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint32]
    return L


def rng_init_mixed(n, sx, sy):
    """ammsb_rng_init_mixed restated: SplitMix64-scrambled stream states."""
    L = _bind_samplers(lib())
    seeds = np.zeros(n, dtype=SEED_DT)
    L.orc_rng_init_mixed(seeds, n, sx, sy)
    return seeds


def device_minibatch_nonlink(seeds, n_candidates, u, m, N, tset, hset):
    """The device sampler's non-link half restated (oracle/ammsb_oracle_samplers.c): advances seeds[:n_candidates] in
    place; returns (edges [m], nodes [m + 1], distinct valid candidates found)."""
    L = _bind_samplers(lib())
    e = np.zeros(m, dtype=np.uint64)
    v = np.zeros(m + 1, dtype=np.uint32)
    cnt = np.zeros(1, dtype=np.uint32)
    ho = hset.slots.ctypes.data if hset is not None else None
    rc = L.orc_device_minibatch_nonlink(seeds, n_candidates, u, m, N, tset.slots, tset.num_bins, tset.prime_idx, ho,
                                        hset.num_bins if hset is not None else 0,
                                        hset.prime_idx if hset is not None else 0, e, v, cnt)
    if rc != 0:
        raise RuntimeError("orc_device_minibatch_nonlink rc=%d" % rc)
    return e, v, int(cnt[0])


def device_minibatch_link(offsets, targets, u):
    L = _bind_samplers(lib())
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    targets = np.ascontiguousarray(targets, dtype=np.uint32)
    n = int(offsets[u + 1] - offsets[u])
    e = np.zeros(max(n, 1), dtype=np.uint64)
    v = np.zeros(n + 1, dtype=np.uint32)
    cnt = np.zeros(1, dtype=np.uint32)
    rc = L.orc_device_minibatch_link(offsets, targets, u, e, v, cnt)
    if rc != 0:
        raise RuntimeError("orc_device_minibatch_link rc=%d" % rc)
    return e[:n], v, int(cnt[0])


def device_minibatch_weight(link, N, E, m):
    return float(_bind_samplers(lib()).orc_device_minibatch_weight(int(bool(link)), N, E, m))


def uset_order(keys):
    """Iteration order of libstdc++'s std::unordered_set<uint64_t> after inserting `keys` in order (restated)."""
    L = _bind_samplers(lib())
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    out = np.zeros(max(keys.size, 1), dtype=np.uint64)
    n = L.orc_uset_order(keys, keys.size, out)
    return out[:n].copy()


STRATEGIES = {"Node": 0, "NodeLink": 1, "NodeNonLink": 2, "BFLink": 3, "BFNonLink": 4, "BF": 5}


def host_sample(N, E, mini_batch, strategy, seed, training_edges, tset, hset, edges_cap, nodes_cap):
    """DoSample's host half (sample.cc:177-303 + learner.cc:162-173) restated: (edges, nodes, weight, seed')."""
    L = _bind_samplers(lib())
    te = np.ascontiguousarray(training_edges, dtype=np.uint64)
    e = np.zeros(max(edges_cap, 1), dtype=np.uint64)
    v = np.zeros(max(nodes_cap, 1), dtype=np.uint32)
    ne, nv, w, s = C.c_uint64(), C.c_uint64(), C.c_float(), C.c_uint(seed)
    ho = hset.slots.ctypes.data if hset is not None else None
    rc = L.orc_host_sample(N, E, mini_batch, STRATEGIES[strategy], C.byref(s), te, te.size, tset.slots, tset.num_bins,
                           tset.prime_idx, ho, hset.num_bins if hset is not None else 0,
                           hset.prime_idx if hset is not None else 0, e, edges_cap, C.byref(ne), v, nodes_cap,
                           C.byref(nv), C.byref(w))
    if rc != 0:
        raise RuntimeError("orc_host_sample rc=%d (needs %d edges, %d nodes)" % (rc, ne.value, nv.value))
    return e[:ne.value].copy(), v[:nv.value].copy(), float(w.value), int(s.value)
