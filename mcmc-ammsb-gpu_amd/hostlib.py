"""ctypes binding of libammsb_host.so (include/ammsb_host.h): host cuckoo sets, graphs, data-set
files, the synthetic a-MMSB generator and the reference-exact host mini-batch samplers."""
import ctypes as C
import os

import numpy as np

from ._capi import AmmsbError, load as _load_hip

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMMSB_HOST_LIB") or os.path.join(_HERE, "libammsb_host.so")  # override: sanitizer build

STRATEGIES = {"Node": 0, "NodeLink": 1, "NodeNonLink": 2, "BFLink": 3, "BFNonLink": 4, "BF": 5}

_vp, _u64, _u32 = C.c_void_p, C.c_uint64, C.c_uint32
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")

SIGNATURES = {
    "ammsb_host_set_create": (_vp, [_u64p, _u64]),
    "ammsb_host_set_destroy": (None, [_vp]),
    "ammsb_host_set_bins": (_u64, [_vp]),
    "ammsb_host_set_prime_idx": (_u32, [_vp]),
    "ammsb_host_set_size": (_u64, [_vp]),
    "ammsb_host_set_data": (C.POINTER(C.c_uint64), [_vp]),
    "ammsb_host_set_has": (C.c_int, [_vp, _u64p, _u64, _u8p]),
    "ammsb_host_generate_graph": (C.c_int64, [_u64, _u32, C.c_double, _u64, C.POINTER(C.POINTER(C.c_uint64))]),
    "ammsb_host_free": (None, [_vp]),
    "ammsb_host_load_snap": (C.c_int64, [C.c_char_p, C.POINTER(_u64), C.POINTER(C.POINTER(C.c_uint64))]),
    "ammsb_host_dump_dataset": (C.c_int, [C.c_char_p, _u64, C.c_float, _u64p, _u64]),
    "ammsb_host_load_dataset": (C.c_int64, [C.c_char_p, C.POINTER(_u64), C.POINTER(C.c_float),
                                            C.POINTER(C.POINTER(C.c_uint64))]),
    "ammsb_host_dataset_create": (_vp, [_u64, _u64p, _u64, C.c_double, C.c_uint]),
    "ammsb_host_dataset_destroy": (None, [_vp]),
    "ammsb_host_dataset_num_training": (_u64, [_vp]),
    "ammsb_host_dataset_num_heldout": (_u64, [_vp]),
    "ammsb_host_dataset_training_edges": (C.POINTER(C.c_uint64), [_vp]),
    "ammsb_host_dataset_heldout_edges": (C.POINTER(C.c_uint64), [_vp]),
    "ammsb_host_dataset_training_set": (_vp, [_vp]),
    "ammsb_host_dataset_heldout_set": (_vp, [_vp]),
    "ammsb_host_dataset_max_fan_out": (_u64, [_vp]),
    "ammsb_host_dataset_training_csr": (C.c_int, [_vp, _u64p, _u32p]),
    "ammsb_host_theta_init": (C.c_int, [_u64, C.c_float, C.c_float,
                                        np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")]),
    "ammsb_host_train_ppx_edges": (C.c_int64, [_vp, _u64, _u64, C.c_float, C.c_uint, C.POINTER(C.POINTER(C.c_uint64))]),
    "ammsb_host_sample": (C.c_int, [_vp, _u64, _u64, _u64, C.c_int, C.POINTER(C.c_uint), _u64p, _u64,
                                    C.POINTER(_u64), _u32p, _u64, C.POINTER(_u64), C.POINTER(C.c_float)]),
}

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AmmsbError("%s not found: run __graft_entry__.build()" % LIB_PATH)
        _load_hip()  # libammsb_host.so resolves ammsb_params_quantize from libammsb_hip.so
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _take(ptr, n):
    """Copy a malloc'd u64 array into numpy and free it."""
    out = np.ctypeslib.as_array(ptr, shape=(max(int(n), 1),))[:int(n)].copy()
    load().ammsb_host_free(C.cast(ptr, C.c_void_p))
    return out


class HostSet:
    """mcmc::cuckoo::Set (cuckoo.h:16-67)."""

    def __init__(self, keys=None, _handle=None, _owner=None):
        self.lib = load()
        if _handle is not None:
            self._h, self._owned, self._owner = _handle, False, _owner
        else:
            keys = np.ascontiguousarray(keys, dtype=np.uint64)
            self._h = self.lib.ammsb_host_set_create(keys, keys.size)
            self._owned, self._owner = True, None
            if not self._h:
                raise AmmsbError("Failed to insert into set (all prime pairs exhausted)")

    def BinsPerBucket(self):
        return int(self.lib.ammsb_host_set_bins(self._h))

    def PrimeIdx(self):
        return int(self.lib.ammsb_host_set_prime_idx(self._h))

    def Size(self):
        return int(self.lib.ammsb_host_set_size(self._h))

    def Capacity(self):
        return 2 * 4 * self.BinsPerBucket()

    def Serialize(self):
        return np.ctypeslib.as_array(self.lib.ammsb_host_set_data(self._h), shape=(self.Capacity(),)).copy()

    def Has(self, keys):
        keys = np.ascontiguousarray(np.atleast_1d(keys), dtype=np.uint64)
        out = np.zeros(keys.size, dtype=np.uint8)
        self.lib.ammsb_host_set_has(self._h, keys, keys.size, out)
        return out.astype(bool)

    def __del__(self):
        try:
            if self._owned and self._h:
                self.lib.ammsb_host_set_destroy(self._h)
        except Exception:
            pass


def generate_graph(N, K_true, avg_degree, seed=20260101):
    """Synthetic a-MMSB graph (mcmc::GenerateSyntheticGraph): unique canonical shuffled edges."""
    p = C.POINTER(C.c_uint64)()
    n = load().ammsb_host_generate_graph(N, K_true, float(avg_degree), seed, C.byref(p))
    if n < 0:
        raise AmmsbError("graph generation failed")
    return _take(p, n)


def load_snap(path):
    p, N = C.POINTER(C.c_uint64)(), _u64()
    n = load().ammsb_host_load_snap(path.encode(), C.byref(N), C.byref(p))
    if n < 0:
        raise AmmsbError("cannot read %s" % path)
    return int(N.value), _take(p, n)


def dump_dataset(path, N, heldout_ratio, edges):
    edges = np.ascontiguousarray(edges, dtype=np.uint64)
    if load().ammsb_host_dump_dataset(path.encode(), N, heldout_ratio, edges, edges.size) != 0:
        raise AmmsbError("cannot write %s" % path)


def load_dataset(path):
    p, N, r = C.POINTER(C.c_uint64)(), _u64(), C.c_float()
    n = load().ammsb_host_load_dataset(path.encode(), C.byref(N), C.byref(r), C.byref(p))
    if n < 0:
        raise AmmsbError("cannot read %s" % path)
    return int(N.value), float(r.value), _take(p, n)


def theta_init(K, eta0=1.0, eta1=1.0):
    """theta_0 exactly as Learner::Learner draws it (learner.cc:150-153)."""
    out = np.zeros(2 * K, dtype=np.float32)
    if load().ammsb_host_theta_init(K, eta0, eta1, out) != 0:
        raise AmmsbError("theta init failed")
    return out


class Dataset:
    """The data half of mcmc::Config after main.cc:102-154: training / held-out edges, both cuckoo
    sets, the training graph."""

    def __init__(self, N, edges, heldout_ratio=0.01, rand_seed=1):
        self.lib = load()
        edges = np.ascontiguousarray(edges, dtype=np.uint64)
        self.N, self.E = int(N), int(edges.size)
        self.heldout_ratio = heldout_ratio
        self._h = self.lib.ammsb_host_dataset_create(self.N, edges, edges.size, heldout_ratio, rand_seed)
        if not self._h:
            raise AmmsbError("Failed to generate training/heldout sets")
        nt = self.lib.ammsb_host_dataset_num_training(self._h)
        nh = self.lib.ammsb_host_dataset_num_heldout(self._h)
        self.training_edges = np.ctypeslib.as_array(self.lib.ammsb_host_dataset_training_edges(self._h),
                                                    shape=(max(nt, 1),))[:nt].copy()
        self.heldout_edges = np.ctypeslib.as_array(self.lib.ammsb_host_dataset_heldout_edges(self._h),
                                                   shape=(max(nh, 1),))[:nh].copy()
        self.training = HostSet(_handle=self.lib.ammsb_host_dataset_training_set(self._h), _owner=self)
        hs = self.lib.ammsb_host_dataset_heldout_set(self._h)
        self.heldout = HostSet(_handle=hs, _owner=self) if hs else None
        self.max_fan_out = int(self.lib.ammsb_host_dataset_max_fan_out(self._h))

    @classmethod
    def robust(cls, N, edges, heldout_ratio=0.01, rand_seed=1, attempts=64):
        """The reference's cuckoo hash pair degenerates for some table sizes (both hashes reduce to
        the key's low bits when the bin count has small factors), in which case its data-set
        preparation fails (data.cc:92-95).  The harness keeps the reference behaviour in __init__ and,
        for generated graphs only, retries here with a few trailing edges dropped so that the
        held-out set gets a different bin count."""
        edges = np.ascontiguousarray(edges, dtype=np.uint64)
        step = max(1, int(np.ceil(2.0 / max(heldout_ratio, 1e-9))))
        last = None
        for t in range(attempts):
            try:
                return cls(N, edges[: edges.size - t * step], heldout_ratio, rand_seed)
            except AmmsbError as e:
                last = e
        raise last

    def training_csr(self):
        off = np.zeros(self.N + 1, dtype=np.uint64)
        tgt = np.zeros(2 * self.training_edges.size, dtype=np.uint32)
        self.lib.ammsb_host_dataset_training_csr(self._h, off, tgt)
        return off, tgt

    def train_ppx_edges(self, ratio=0.01, seed=1):
        """MakeEdgesForTrainingPerplexity (learner.cc:47-75)."""
        p = C.POINTER(C.c_uint64)()
        n = self.lib.ammsb_host_train_ppx_edges(self._h, self.N, self.E, ratio, seed, C.byref(p))
        if n < 0:
            raise AmmsbError("training perplexity edge list: too many edges for one launch (lower the ratio)")
        return _take(p, n)

    def max_nodes(self, mini_batch):
        return max(2 * mini_batch, 1 + self.max_fan_out)  # phi.cc:620-622

    def max_edges(self, mini_batch):
        return max(mini_batch, self.max_fan_out)  # sample.cc:129

    def sample(self, mini_batch, strategy, seed):
        """One host mini-batch: returns (edges, nodes, weight, new_seed) -- DoSample's host half
        (learner.cc:175-185)."""
        e = np.zeros(self.max_edges(mini_batch), dtype=np.uint64)
        v = np.zeros(self.max_nodes(mini_batch), dtype=np.uint32)
        ne, nv, w, s = _u64(), _u64(), C.c_float(), C.c_uint(seed)
        rc = self.lib.ammsb_host_sample(self._h, self.N, self.E, mini_batch, STRATEGIES[strategy], C.byref(s),
                                        e, e.size, C.byref(ne), v, v.size, C.byref(nv), C.byref(w))
        if rc == -2:  # learner.cc:184-189: the mini-batch does not fit the device buffers sized by the reference rule
            raise AmmsbError("%d | %d (edges), %d | %d (nodes)" % (ne.value, e.size, nv.value, v.size))
        if rc != 0:
            raise AmmsbError("host sampler failed")
        return e[:ne.value].copy(), v[:nv.value].copy(), float(w.value), int(s.value)

    def __del__(self):
        try:
            if self._h:
                self.lib.ammsb_host_dataset_destroy(self._h)
        except Exception:
            pass
