"""Host-side mirror of the reference's operator interface on top of the C ABI.

Same names, argument meaning and error behaviour as the C++ functors of the reference, so that the
parity tests read like the reference's own tests:

    PhiUpdater            mcmc/phi.h:10-61      operator()(nodes, neighbors, n)
    BetaUpdater           mcmc/beta.h:15-72     operator()(edges, n, scale), GetThetaSum(), GetGrads()
    PerplexityCalculator  mcmc/perplexity.h:23-114   operator()()
    NeighborSampler       mcmc/sample.h:16-49   operator()(n, nodes), GetData(), GetHash()
    Random                mcmc/random.h:22-47   OpenClRandom (seed array)
    DeviceSet             mcmc/cuckoo.h:69-86   OpenClSet
    RowPartitionedMatrix  mcmc/partitioned-alloc.h:73-140

torch is used for device memory and streams only.  Every method enqueues on the current torch HIP
stream and returns without synchronising (the reference calls queue.Finish() after each launch; a
caller that wants that behaviour calls torch.cuda.synchronize()).
"""
import ctypes as C
import os
import threading

import numpy as np
import torch

from . import _capi
from ._capi import AmmsbError, NOISE_OFF, PHI_STREAMING, Params, PpxSums, Rpm, SetDesc, check

SEED_DT = np.dtype([("x", np.uint64), ("y", np.uint64)])


# torch.cuda.current_stream() costs ~3 us and every launch needs the handle (a dozen per iteration, a third of
# the host time of a step).  The streams this module switches to itself (`stream(s)` below) are tracked per
# thread, and a caller that owns a loop can pin the ambient stream for its duration (`pin_current_stream`);
# anything else falls back to asking torch.
_tls = threading.local()


def _stream():
    stack = getattr(_tls, "stack", None)
    if stack:
        return stack[-1]
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _torch_stream():
    """The torch.cuda.Stream object of the same cache (events are recorded on / waited for by stream OBJECTS: an
    Event.record() without one asks torch for the current stream, ~8 us, six times per iteration of the multi-GPU
    schedule)."""
    stack = getattr(_tls, "tstack", None)
    if stack:
        return stack[-1]
    return torch.cuda.current_stream()


class pin_current_stream:
    """with pin_current_stream(): ...  -- the stream that is current on entry is used for every launch of this
    thread inside the block (unless `stream(s)` switches), without asking torch again."""

    def __enter__(self):
        if not hasattr(_tls, "stack"):
            _tls.stack, _tls.tstack = [], []
        cur = torch.cuda.current_stream()
        _tls.stack.append(C.c_void_p(cur.cuda_stream))
        _tls.tstack.append(cur)
        return self

    def __exit__(self, *exc):
        _tls.stack.pop()
        _tls.tstack.pop()
        return False


class _StreamScope:
    """torch.cuda.stream(s) plus the handle cache above."""

    def __init__(self, s):
        self.s = s
        self.tc = torch.cuda.stream(s)

    def __enter__(self):
        self.tc.__enter__()
        if not hasattr(_tls, "stack"):
            _tls.stack, _tls.tstack = [], []
        _tls.stack.append(C.c_void_p(self.s.cuda_stream))
        _tls.tstack.append(self.s)
        return self

    def __exit__(self, *exc):
        _tls.stack.pop()
        _tls.tstack.pop()
        return self.tc.__exit__(*exc)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def make_params(N, K, E=0, num_node_sample=32, alpha=0.0, a=0.0315, b=1024.0, c=0.5, epsilon=1e-7,
                eta0=1.0, eta1=1.0, quantize=True):
    """mcmc::Config numeric fields (config.h:25-102) with main.cc's rule alpha == 0 -> 1/K
    (main.cc:153).  quantize applies MakeCompileFlags' "%e" round trip (config.cc:57-83)."""
    if alpha == 0:
        alpha = float(np.float32(1.0) / np.float32(K))
    p = Params(N, K, E, num_node_sample, alpha, a, b, c, epsilon, eta0, eta1)
    if quantize:
        check(_capi.load().ammsb_params_quantize(C.byref(p)))
    return p


class Context:
    """One per device; owns the C-ABI context (replaces clcuda::Queue + the JIT-built programs)."""

    def __init__(self, params, device=None):
        if not torch.cuda.is_available():
            raise AmmsbError("no HIP device visible: the MI355X path has no CPU fallback")
        self.lib = _capi.load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.params = params
        self._ctx = C.c_void_p()
        check(self.lib.ammsb_ctx_create(self.device.index, C.byref(params), C.byref(self._ctx)))

    @property
    def handle(self):
        return self._ctx

    def check(self, rc):
        check(rc, self._ctx)

    KERNELS = ("update_phi", "update_pi", "beta_grads", "perplexity", "update_phi_small")

    def kernel_names(self):
        """Names of the kernels the last update_phi / update_pi / gradient / perplexity calls on this context
        dispatched to, spelled as in a rocprofv3 kernel trace (ammsb_last_kernel_name).  "update_phi" is the form large
        launches take, "update_phi_small" the several-waves-per-node form of launches of at most 512 groups (link
        mini-batches); a context that has only ever made small launches reports that one under both keys."""
        names = {k: self.lib.ammsb_last_kernel_name(self.handle, i).decode() for i, k in enumerate(self.KERNELS)}
        if not names["update_phi"]:
            names["update_phi"] = names["update_phi_small"]
        return names

    def close(self):
        if self._ctx:
            self.lib.ammsb_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- small allocation helpers (device memory comes from torch's caching allocator)
    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def from_numpy(self, arr):
        a = np.ascontiguousarray(arr)
        if a.dtype == np.uint64:
            a = a.view(np.int64)
        elif a.dtype == np.uint32:
            a = a.view(np.int32)
        elif a.dtype == SEED_DT:
            a = a.view(np.int64).reshape(-1, 2)
        return torch.from_numpy(a).to(self.device)


def to_numpy(t, dtype=None):
    a = t.detach().cpu().numpy()
    return a.view(dtype) if dtype is not None else a


class Random:
    """OpenClRandom: `size` xorshift128+ streams, stream i seeded {sx+i, sy+i} (random.cc:31-43)."""

    def __init__(self, ctx, size, seed, mixed=False):
        self.ctx = ctx
        self.size = int(size)
        self.mixed = bool(mixed)  # SplitMix64-scrambled states (streams the reference does not have)
        self.seeds = ctx.empty((self.size, 2), torch.int64)
        self.SetSeed(seed)

    def SetSeed(self, seed):
        sx, sy = seed
        fn = self.ctx.lib.ammsb_rng_init_mixed if self.mixed else self.ctx.lib.ammsb_rng_init
        self.ctx.check(fn(self.ctx.handle, _ptr(self.seeds), self.size, int(sx), int(sy), _stream()))

    def GetSeeds(self):
        return self.seeds

    def host(self):
        return to_numpy(self.seeds).view(np.uint64).reshape(-1, 2).copy().view(SEED_DT).reshape(-1)

    def load(self, seeds):
        self.seeds.copy_(self.ctx.from_numpy(seeds))


class DeviceSet:
    """OpenClSet: device image of a host cuckoo Set (cuckoo.cc:222-239).  `slots` is the array
    Set::Serialize() returns; num_bins = BinsPerBucket(); prime_idx = PrimeIdx()."""

    def __init__(self, ctx, slots, num_bins, prime_idx):
        slots = np.ascontiguousarray(slots, dtype=np.uint64)
        if slots.size != 2 * int(num_bins) * 4:
            raise AmmsbError("set image has %d slots, expected 2*%d*4" % (slots.size, num_bins))
        self.ctx = ctx
        self.data = ctx.from_numpy(slots)
        self.num_bins = int(num_bins)
        self.prime_idx = int(prime_idx)
        self.desc = SetDesc(self.data.data_ptr(), self.num_bins, self.prime_idx)

    @classmethod
    def build_on_device(cls, ctx, keys):
        """Opt-in parallel construction on the GPU (ammsb_set_build): same layout and hash pairs, exact membership,
        a different (interleaving-dependent) image than the host's rand_r random-walk build, which stays the
        default and is what parity runs use.  `keys`: distinct u64 keys, a device tensor or a host array."""
        k = keys if torch.is_tensor(keys) else ctx.from_numpy(np.ascontiguousarray(keys, dtype=np.uint64))
        n = int(k.numel())
        self = cls.__new__(cls)
        self.ctx = ctx
        self.num_bins = int(ctx.lib.ammsb_set_num_bins(n))
        self.data = ctx.empty((2 * self.num_bins * 4,), torch.int64)
        scratch = ctx.zeros((1,), torch.int32)
        pidx = C.c_uint32(0)
        ctx.check(ctx.lib.ammsb_set_build(ctx.handle, _ptr(k), n, _ptr(self.data), self.num_bins, C.byref(pidx),
                                          _ptr(scratch), _stream()))
        self.prime_idx = int(pidx.value)
        self.desc = SetDesc(self.data.data_ptr(), self.num_bins, self.prime_idx)
        return self

    def Has(self, keys):
        """The `find` kernel of cuckoo-test.cc:45-53 over a device or host key array."""
        k = keys if torch.is_tensor(keys) else self.ctx.from_numpy(np.asarray(keys, dtype=np.uint64))
        out = self.ctx.empty((k.numel(),), torch.uint8)
        self.ctx.check(self.ctx.lib.ammsb_set_has(self.ctx.handle, C.byref(self.desc), _ptr(k), k.numel(),
                                                  _ptr(out), _stream()))
        return out


class RowPartitionedMatrix:
    """RowPartitionedMatrix<float> (partitioned-alloc.h:73-140): rows split into <= 32 blocks of
    rows_in_block rows.  On a 288 GB MI355X one block holds any configuration of interest, so
    rows_in_block defaults to all rows; a smaller value reproduces the reference's multi-block
    layout (tests use 11+1 blocks)."""

    def __init__(self, ctx, rows, cols, rows_in_block=0, dtype=torch.float32):
        self.ctx = ctx
        self.rows, self.cols = int(rows), int(cols)
        self.rows_in_block = int(rows_in_block) if rows_in_block else self.rows
        nfull, rem = divmod(self.rows, self.rows_in_block)
        sizes = [self.rows_in_block] * nfull + ([rem] if rem else [])
        if len(sizes) > _capi.RPM_MAX_BLOCKS:
            raise AmmsbError("more than 32 blocks")
        self.blocks = [ctx.empty((s, self.cols), dtype) for s in sizes]
        self.desc = Rpm()
        for i, b in enumerate(self.blocks):
            self.desc.blocks[i] = b.data_ptr()
        self.desc.rows_in_block = self.rows_in_block
        self.desc.num_rows = self.rows
        self.desc.num_cols = self.cols
        self.desc.num_blocks = len(self.blocks)

    def adopt(self, other):
        """Take over `other`'s storage (same shape): every holder of this object then works on the new blocks."""
        if (other.rows, other.cols, other.rows_in_block) != (self.rows, self.cols, self.rows_in_block):
            raise AmmsbError("adopt: shapes differ")
        self.blocks, self.desc = other.blocks, other.desc

    def Rows(self):
        return self.rows

    def Cols(self):
        return self.cols

    def RowsPerBlock(self):
        return self.rows_in_block

    def Blocks(self):
        return self.blocks

    def load(self, host):
        host = np.ascontiguousarray(host).reshape(self.rows, self.cols)
        r = 0
        for b in self.blocks:
            b.copy_(self.ctx.from_numpy(host[r:r + b.shape[0]]))
            r += b.shape[0]

    def host(self):
        return np.concatenate([to_numpy(b) for b in self.blocks], axis=0)

    def gather_rows(self, rows):
        """rows (device int64 tensor) -> [len, cols] tensor; single-block fast path."""
        if len(self.blocks) == 1:
            return self.blocks[0][rows]
        return torch.cat(self.blocks, 0)[rows]


def RandomGammaAndNormalize(ctx, eta0, eta1, pi, phi_sum, seed=(11, 113)):
    """random.cc:159-167: N*32 streams seeded {11,113}; pi rows ~ Gamma, normalised; phi_sum = sums."""
    rnd = Random(ctx, pi.Rows() * 32, seed)
    ctx.check(ctx.lib.ammsb_pi_init_gamma(ctx.handle, C.byref(pi.desc), _ptr(phi_sum), eta0, eta1,
                                          _ptr(rnd.seeds), _stream()))
    return rnd


class NeighborSampler:
    """sample.h:16-49.  max_nodes = max(2*mini_batch, 1+MaxFanOut) sizes the buffers (sample.cc:88-97)."""

    def __init__(self, ctx, max_nodes, neighbor_seed=(56, 57), wg=32):
        self.ctx = ctx
        self.n = ctx.params.num_node_sample
        self.capacity = 2 * self.n
        self.local = int(wg)
        self.hash = ctx.empty((max_nodes, self.capacity), torch.int32)
        self.data = ctx.empty((max_nodes, self.n), torch.int32)
        self.rand = Random(ctx, max_nodes * self.capacity, neighbor_seed)
        self.max_nodes = int(max_nodes)

    def __call__(self, num_samples, nodes):
        if num_samples > self.max_nodes:
            raise AmmsbError("%d samples > buffer for %d" % (num_samples, self.max_nodes))
        self.ctx.check(self.ctx.lib.ammsb_sample_neighbors(
            self.ctx.handle, _ptr(self.rand.seeds), _ptr(nodes), int(num_samples), self.local,
            _ptr(self.hash), _ptr(self.data), _stream()))

    def GetHash(self):
        return self.hash

    def GetData(self):
        return self.data

    def HashCapacityPerSample(self):
        return self.capacity

    def DataSizePerSample(self):
        return self.n


class PhiUpdater:
    """phi.h:10-61 / phi.cc:608-763.  mode is always the work-group form on this hardware."""

    def __init__(self, ctx, beta, pi, phi, training_set, max_nodes, phi_seed=(42, 43), phi_wg_size=64,
                 phi_disable_noise=False, streaming_only=False):
        self.ctx, self.beta, self.pi, self.phi, self.set = ctx, beta, pi, phi, training_set
        self.local = int(phi_wg_size)
        # streaming_only: small launches too go through the one-wave-per-node kernels (AMMSB_PHI_STREAMING)
        self.flags = (NOISE_OFF if phi_disable_noise else 0) | (PHI_STREAMING if streaming_only else 0)
        self.max_nodes = int(max_nodes)
        self.phi_vec = ctx.empty((self.max_nodes, ctx.params.K), torch.float32)
        # phi.cc:625-629: max(2m, 1+maxdeg) * wg streams
        self.rand = Random(ctx, self.max_nodes * self.local, phi_seed)
        self.count_calls = 0

    def update_phi(self, nodes, neighbors, n, group_begin=0, group_end=0xFFFFFFFF):
        if n == 0:
            raise AmmsbError("mini-batch nodes size = 0!")  # phi.cc:732
        if n > self.max_nodes:
            raise AmmsbError("grads too small")  # phi.cc:734-737 analogue
        c = self.ctx
        c.check(c.lib.ammsb_update_phi(c.handle, _ptr(self.beta), C.byref(self.pi.desc), _ptr(self.phi),
                                       C.byref(self.set.desc), _ptr(nodes), _ptr(neighbors), int(n),
                                       self.count_calls, _ptr(self.rand.seeds), self.local, self.flags,
                                       int(group_begin), int(group_end), _ptr(self.phi_vec), _stream()))

    def update_pi(self, nodes, n, phi_vec=None):
        c = self.ctx
        pv = self.phi_vec if phi_vec is None else phi_vec
        c.check(c.lib.ammsb_update_pi(c.handle, C.byref(self.pi.desc), _ptr(self.phi), _ptr(pv),
                                      _ptr(nodes), int(n), self.local, _stream()))

    def __call__(self, mini_batch_nodes, neighbors, num_mini_batch_nodes):
        self.count_calls += 1  # phi.cc:739
        self.update_phi(mini_batch_nodes, neighbors, num_mini_batch_nodes)
        self.update_pi(mini_batch_nodes, num_mini_batch_nodes)


class BetaUpdater:
    """beta.h:15-72 / beta.cc:236-384 (EDGE_PER_WORKGROUP)."""

    def __init__(self, ctx, theta, beta, pi, training_set, beta_seed=(44, 45), beta_wg_size=256,
                 disable_noise=False):
        self.ctx, self.theta, self.beta, self.pi, self.set = ctx, theta, beta, pi, training_set
        self.local = int(beta_wg_size)
        self.flags = NOISE_OFF if disable_noise else 0
        self.rand = Random(ctx, ctx.params.K, beta_seed)  # beta.cc:251-252
        self.grads = ctx.zeros((2 * ctx.params.K,), torch.float32)
        self.count_calls = 0

    def calculate_grads(self, edges, num_edges, edge_begin=0, edge_end=0xFFFFFFFF, out=None):
        c = self.ctx
        g = self.grads if out is None else out
        c.check(c.lib.ammsb_beta_grads(c.handle, _ptr(self.theta), _ptr(self.beta), C.byref(self.pi.desc),
                                       C.byref(self.set.desc), _ptr(edges), int(num_edges), int(edge_begin),
                                       int(min(edge_end, num_edges)), self.local, _ptr(g), _stream()))
        return g

    def can_fuse_update_pi(self, phi_updater):
        """Whether update_pi_and_grads takes this context's (K, work-group sizes)."""
        c = self.ctx
        return bool(c.lib.ammsb_can_fuse_pi_beta(c.handle, int(phi_updater.local), self.local))

    def update_pi_and_grads(self, phi_updater, nodes, edges, num_edges, out=None):
        """phi_updater.update_pi(nodes, num_edges + 1) and calculate_grads(edges, num_edges) as one launch
        (ammsb_update_pi_beta_grads): node-stratified mini-batches only -- edge t = (nodes[0], nodes[t + 1])."""
        c = self.ctx
        g = self.grads if out is None else out
        c.check(c.lib.ammsb_update_pi_beta_grads(c.handle, _ptr(self.theta), _ptr(self.beta), C.byref(self.pi.desc),
                                                 _ptr(phi_updater.phi), _ptr(phi_updater.phi_vec), _ptr(nodes),
                                                 C.byref(self.set.desc), _ptr(edges), int(num_edges), self.local,
                                                 _ptr(g), _stream()))
        return g

    def update_theta(self, scale, grads=None):
        c = self.ctx
        g = self.grads if grads is None else grads
        c.check(c.lib.ammsb_update_theta(c.handle, _ptr(self.theta), _ptr(self.beta), _ptr(g),
                                         self.count_calls, float(scale), _ptr(self.rand.seeds), self.flags,
                                         _stream()))

    def __call__(self, edges, num_edges, scale):
        self.count_calls += 1  # beta.cc:336
        self.calculate_grads(edges, num_edges)
        self.update_theta(scale)

    def GetGrads(self):
        return self.grads

    def GetThetaSum(self):
        """theta_sum of the last gradient launch (beta.h:27), as a device tensor [K]."""
        c = self.ctx
        out = c.empty((c.params.K,), torch.float32)
        c.check(c.lib.ammsb_theta_sum(c.handle, _ptr(out), _stream()))
        return out


def sum_rows(ctx, rows, out):
    """out[c] = rows[0, c] + rows[1, c] + ... in ascending row order (the multi-GPU gradient reduction)."""
    ctx.check(ctx.lib.ammsb_sum_rows_f32(ctx.handle, _ptr(rows), int(rows.shape[0]), int(rows.shape[1]), _ptr(out),
                                         _stream()))
    return out


def beta_from_theta(ctx, theta, beta):
    ctx.check(ctx.lib.ammsb_beta_from_theta(ctx.handle, _ptr(theta), _ptr(beta), _stream()))


class PerplexityCalculator:
    """perplexity.h:23-114 / perplexity.cc:184-274 (EDGE_PER_WORKGROUP)."""

    def __init__(self, ctx, beta, pi, edges, edge_set, ppx_wg_size=64):
        self.ctx, self.beta, self.pi, self.edges, self.set = ctx, beta, pi, edges, edge_set
        self.local = int(ppx_wg_size)
        self.num_edges = int(edges.numel())
        self.ppx_per_edge = ctx.zeros((max(self.num_edges, 1),), torch.float32)  # perplexity.cc:204-205
        self.dev_sums = ctx.zeros((4,), torch.int64)  # ammsb_ppx_sums
        # the same 32 bytes in host-mapped pinned memory: the kernel writes its result where the host reads it, so a
        # call is one launch and one stream synchronisation (no device-to-host copy)
        self.host_sums = torch.zeros((4,), dtype=torch.int64, pin_memory=True)
        self.sums = self.dev_sums  # where the last pass put its result (device tensor, or host_sums after operator())
        self.count_calls = 0

    def partial(self, edge_begin=0, edge_end=0xFFFFFFFF, out=None):
        """Enqueue one pass over edges [edge_begin, edge_end); returns the sums tensor (device memory unless `out`)."""
        c = self.ctx
        out = self.dev_sums if out is None else out
        self.sums = out
        c.check(c.lib.ammsb_perplexity(c.handle, _ptr(self.beta), C.byref(self.pi.desc), C.byref(self.set.desc),
                                       _ptr(self.edges), self.num_edges, int(edge_begin),
                                       int(min(edge_end, self.num_edges)), self.count_calls, self.local,
                                       _ptr(self.ppx_per_edge), _ptr(out), _stream()))
        return out

    def partial_host(self, edge_begin=0, edge_end=0xFFFFFFFF):
        """One pass with the sums written straight into pinned host memory; waits for the stream and returns
        (link_ll, nonlink_ll, link_cnt, nonlink_cnt)."""
        self.partial(edge_begin, edge_end, out=self.host_sums)
        torch.cuda.current_stream().synchronize()
        raw = self.host_sums.numpy()
        ll = raw[:2].view(np.float64)
        cnt = raw[2:].view(np.uint64)
        return float(ll[0]), float(ll[1]), int(cnt[0]), int(cnt[1])

    @staticmethod
    def unpack(sums):
        raw = to_numpy(sums)
        ll = raw[:2].view(np.float64)
        cnt = raw[2:].view(np.uint64)
        return float(ll[0]), float(ll[1]), int(cnt[0]), int(cnt[1])

    @staticmethod
    def value(link_ll, nonlink_ll, link_cnt, nonlink_cnt):
        avg = 0.0
        if link_cnt + nonlink_cnt != 0:  # perplexity.cc:264-268
            avg = (link_ll + nonlink_ll) / (link_cnt + nonlink_cnt)
        return -avg

    def __call__(self):
        self.count_calls += 1  # perplexity.cc:252
        return self.value(*self.partial_host())


MB_CHOICE_DT = np.dtype([("link", np.uint32), ("u", np.uint32), ("n", np.uint32), ("n_candidates", np.uint32)])


class DeviceMiniBatchSampler:
    """Device-side replacement for sampleNode + ExtractNodesFromMiniBatch (sample.cc:249-303,
    learner.cc:162-173).  The coin flip and the choice of u stay on the host (a numpy Generator), so
    the sizes of the mini-batch are known without reading anything back; the m distinct non-links /
    the edges of u are produced on the device straight into the caller's edge and node buffers.

    A mini-batch is first CHOSEN (`choose`: link?, u, deg(u), candidate draws) and then ENQUEUED
    (`enqueue`, eager) or handed to the captured-graph loop (GraphLoop.run), so both paths consume the
    same host stream of choices.  The number of candidate draws of a non-link batch follows the vertex:
    enough for m + 1 + deg_training(u) + deg_heldout(u) distinct values (every invalid partner of u
    could be drawn) plus the 8 % + 1024 margin, so a hub vertex cannot come up short.  A shortfall by
    sheer bad luck is still detected: the device keeps a sticky counter that `check()` reads at the
    learner's synchronisation points."""

    BLOCK = 64  # choices drawn from the host generator at a time

    def __init__(self, ctx, csr_offsets, csr_targets, training_set, heldout_set, mini_batch, seed=(1234, 5678),
                 host_seed=20260101, heldout_degree=None):
        self.ctx = ctx
        self.m = int(mini_batch)
        self.N, self.E = int(ctx.params.N), int(ctx.params.E)
        self.training_set, self.heldout_set = training_set, heldout_set
        off = np.ascontiguousarray(csr_offsets, dtype=np.uint64)
        self.degree = np.diff(off.astype(np.int64))
        if not (self.degree > 0).any():
            raise AmmsbError("training graph has no edges")
        self.max_fan_out = int(self.degree.max())
        self.excluded = self.degree + 1  # invalid partners of u: itself and its neighbours in both graphs
        if heldout_degree is not None:
            self.excluded = self.excluded + np.asarray(heldout_degree, dtype=np.int64)
        self.offsets = ctx.from_numpy(off)
        self.targets = ctx.from_numpy(np.ascontiguousarray(csr_targets, dtype=np.uint32))
        self._cand = {}
        self.C = self._candidates_for(int(self.excluded.max()))  # capacity: streams and workspace
        if self.C == 0:
            raise AmmsbError("device sampling needs N >= 2 * mini_batch with room for the largest degree "
                             "(N=%d, m=%d, max excluded=%d)" % (self.N, self.m, int(self.excluded.max())))
        self.rand = Random(ctx, self.C, seed, mixed=True)
        # 0xFF once: every call leaves the de-duplication table empty again (include/ammsb.h)
        self.workspace = torch.full((int(ctx.lib.ammsb_minibatch_workspace_bytes(self.C)),), 255, dtype=torch.uint8,
                                    device=ctx.device)
        self.count = ctx.zeros((2,), torch.int32)  # [0] distinct candidates of the last call, [1] sticky shortfalls
        self.host_rng = np.random.default_rng(host_seed)
        self.queue = []  # choices drawn ahead, BLOCK at a time: (coin, u) pairs not yet consumed
        self.w_link = float(np.float32(self.N))                               # sample.cc:268
        self.w_nonlink = float(np.float32(2 * self.E) / np.float32(self.m))   # sample.cc:292
        # The candidate streams, the workspace and the counter are shared by successive calls, which the learner
        # issues on two alternating streams: each call waits for the previous one's kernels before it starts.
        self._done = torch.cuda.Event()
        self._done_valid = False

    def _candidates_for(self, excluded):
        key = (int(excluded) + 31) // 32 * 32  # few distinct values
        c = self._cand.get(key)
        if c is None:
            c = self._cand[key] = int(self.ctx.lib.ammsb_minibatch_candidates_for(self.N, self.m, key))
        return c

    def _refill(self):
        coins = self.host_rng.integers(0, 2, size=self.BLOCK)  # rand_r(seed) % 2, sample.cc:297
        us = self.host_rng.integers(0, self.N, size=self.BLOCK)
        self.queue = list(zip(coins.tolist(), us.tolist()))[::-1]

    def choose(self, strategy):
        """The next mini-batch: (link, u, n, n_candidates)."""
        link = {"Node": None, "NodeLink": True, "NodeNonLink": False}.get(strategy, "bad")
        if link == "bad":
            raise AmmsbError("device sampling implements Node / NodeLink / NodeNonLink only")
        while True:
            if not self.queue:
                self._refill()
            coin, u = self.queue.pop()
            is_link = bool(coin) if link is None else link
            if is_link:
                n = int(self.degree[u])
                if n == 0:  # sampleNodeLink retries until the vertex has an edge (sample.cc:254-263)
                    while True:
                        if not self.queue:
                            self._refill()
                        u = self.queue.pop()[1]
                        n = int(self.degree[u])
                        if n > 0:
                            break
                return (1, u, n, 0)
            return (0, u, 0, self._candidates_for(self.excluded[u]))

    def choose_many(self, strategy, n):
        """n consecutive choices as a structured array (link, u, n, n_candidates) -- the same host stream and the same
        results as n calls of choose(), drawn a block at a time and assembled with numpy (the graph loop hands
        hundreds of choices to the library per call; a Python loop per choice would be most of its host time)."""
        out = np.zeros(n, dtype=MB_CHOICE_DT)
        link_mode = {"Node": None, "NodeLink": True, "NodeNonLink": False}.get(strategy, "bad")
        if link_mode == "bad":
            raise AmmsbError("device sampling implements Node / NodeLink / NodeNonLink only")
        done = 0
        while done < n:
            if not self.queue:
                self._refill()
            take = min(len(self.queue), n - done)
            blk = np.array(self.queue[len(self.queue) - take:][::-1], dtype=np.int64)   # pop order
            coins, us = blk[:, 0], blk[:, 1]
            is_link = coins.astype(bool) if link_mode is None else np.full(take, link_mode)
            bad = is_link & (self.degree[us] == 0)
            if bad.any():   # a link choice on a vertex without edges consumes further queue entries: scalar path
                first = int(np.flatnonzero(bad)[0])
                take = first
            if take:
                del self.queue[len(self.queue) - take:]
                seg = out[done:done + take]
                lk = is_link[:take]
                seg["link"] = lk
                seg["u"] = us[:take]
                seg["n"] = np.where(lk, self.degree[us[:take]], 0)
                seg["n_candidates"] = np.where(lk, 0, self._cand_table()[(self.excluded[us[:take]] + 31) // 32])
                done += take
            if bad.any() and done < n:
                out[done] = self.choose(strategy)
                done += 1
        return out

    def _cand_table(self):
        """n_candidates by excluded-count bucket (32 per bucket), for choose_many."""
        t = getattr(self, "_cand_tab", None)
        if t is None:
            top = (int(self.excluded.max()) + 31) // 32
            t = self._cand_tab = np.array([self._candidates_for(32 * b) if b else self._candidates_for(1)
                                           for b in range(top + 1)], dtype=np.int64)
        return t

    def sizes(self, choice):
        """(n_edges, n_nodes, weight) of a choice."""
        link, u, n, _ = choice
        if link:
            return n, n + 1, self.w_link
        return self.m, self.m + 1, self.w_nonlink

    def enqueue(self, choice, dev_edges, dev_nodes):
        """Eager form: enqueue the kernels of one chosen mini-batch on the current stream."""
        c = self.ctx
        if self._done_valid:
            self._done.wait()
        link, u, n, n_cand = choice
        if link:
            c.check(c.lib.ammsb_minibatch_link(c.handle, _ptr(self.offsets), _ptr(self.targets), u, n,
                                               _ptr(dev_edges), _ptr(dev_nodes), _stream()))
        else:
            hs = C.byref(self.heldout_set.desc) if self.heldout_set is not None else None
            c.check(c.lib.ammsb_minibatch_nonlink(c.handle, _ptr(self.rand.seeds), n_cand, self.C, u, self.m,
                                                  C.byref(self.training_set.desc), hs, _ptr(self.workspace),
                                                  _ptr(dev_edges), _ptr(dev_nodes), _ptr(self.count), _stream()))
        self._done.record()
        self._done_valid = True
        return self.sizes(choice)

    def __call__(self, strategy, dev_edges, dev_nodes):
        """Choose and enqueue one mini-batch; returns (n_edges, n_nodes, weight)."""
        return self.enqueue(self.choose(strategy), dev_edges, dev_nodes)

    def mark_used(self):
        """The shared sampler state was just used by work queued on the current stream (a graph run)."""
        self._done.record()
        self._done_valid = True

    def check(self):
        """Raise if any non-link mini-batch since the last check found fewer than m distinct valid partners
        (its tail then repeats earlier entries: not a valid sample).  Synchronises; call at sync points."""
        short = int(self.count[1].item())
        if short:
            self.count[1].zero_()
            raise AmmsbError("device mini-batch sampler: %d mini-batch(es) found fewer than %d distinct non-links "
                             "(last count %d)" % (short, self.m, int(self.count[0].item())))

    def state(self):
        return dict(rng=self.host_rng.bit_generator.state, queue=[[int(a), int(b)] for a, b in self.queue])

    def load_state(self, st):
        self.host_rng.bit_generator.state = st["rng"]
        self.queue = [(int(a), int(b)) for a, b in st.get("queue", [])]


class GraphLoop:
    """ammsb_loop (include/ammsb.h): whole iterations replayed as captured hipGraphs over a Learner's buffers."""

    def __init__(self, ctx, theta, beta, pi, phi_sum, training_set, heldout_set, phi, beta_upd, samples, sampler,
                 timestamps=False):
        self.ctx = ctx
        cfg = _capi.LoopConfig()
        cfg.theta, cfg.beta = theta.data_ptr(), beta.data_ptr()
        cfg.pi = C.pointer(pi.desc)
        cfg.phi_sum = phi_sum.data_ptr()
        cfg.training_set = C.pointer(training_set.desc)
        cfg.heldout_set = C.pointer(heldout_set.desc) if heldout_set is not None else None
        cfg.phi_seeds, cfg.phi_vec = phi.rand.seeds.data_ptr(), phi.phi_vec.data_ptr()
        cfg.phi_wg, cfg.phi_flags = phi.local, phi.flags
        cfg.beta_seeds, cfg.grads = beta_upd.rand.seeds.data_ptr(), beta_upd.grads.data_ptr()
        cfg.beta_wg, cfg.beta_flags = beta_upd.local, beta_upd.flags
        for i, s in enumerate(samples):
            ns = s.neighbor_sampler
            cfg.edges[i], cfg.nodes[i] = s.dev_edges.data_ptr(), s.dev_nodes.data_ptr()
            cfg.neighbors[i], cfg.nbr_table[i] = ns.data.data_ptr(), ns.hash.data_ptr()
            cfg.nbr_seeds[i] = ns.rand.seeds.data_ptr()
        cfg.nbr_wg = samples[0].neighbor_sampler.local
        cfg.csr_offsets, cfg.csr_targets = sampler.offsets.data_ptr(), sampler.targets.data_ptr()
        cfg.mb_seeds, cfg.mb_candidates = sampler.rand.seeds.data_ptr(), sampler.C
        cfg.mb_workspace, cfg.mb_count = sampler.workspace.data_ptr(), sampler.count.data_ptr()
        cfg.mini_batch, cfg.max_fan_out = sampler.m, sampler.max_fan_out
        cfg.max_nodes = min(int(s.dev_nodes.numel()) for s in samples)
        cfg.max_edges = min(int(s.dev_edges.numel()) for s in samples)
        cfg.flags = _capi.LOOP_TIMESTAMPS if timestamps else 0
        self.timestamps_on = bool(timestamps)
        self._keep = (theta, beta, pi, phi_sum, training_set, heldout_set, phi, beta_upd, samples, sampler)
        self._h = C.c_void_p()
        torch.cuda.synchronize()  # the captured kernels' buffers are initialised before anything replays
        ctx.check(ctx.lib.ammsb_loop_create(ctx.handle, C.byref(cfg), C.byref(self._h)))

    def run(self, pending, nxt, first_step, parity):
        """nxt: structured array of MB_CHOICE_DT (DeviceMiniBatchSampler.choose_many) or a list of 4-tuples."""
        if not isinstance(nxt, np.ndarray):
            nxt = np.array([tuple(int(x) for x in ch) for ch in nxt], dtype=MB_CHOICE_DT)
        nxt = np.ascontiguousarray(nxt)
        n = int(nxt.shape[0])
        pend = _capi.MbChoice(*[int(x) for x in pending])
        self.ctx.check(self.ctx.lib.ammsb_loop_run(self._h, C.byref(pend),
                                                   C.cast(nxt.ctypes.data, C.POINTER(_capi.MbChoice)), n,
                                                   int(first_step), int(parity), _stream()))

    def check(self):
        """Synchronises the loop's streams.  If a device-side wait between the two chains gave up, the library has
        skipped what came after it, switched to the stream-event hand-over and re-run the missing steps by the time
        this returns (ammsb_loop_check; status() counts the fallback); raises only if that recovery failed."""
        n = C.c_uint32(0)
        self.ctx.check(self.ctx.lib.ammsb_loop_check(self._h, C.byref(n)))
        if n.value:
            raise AmmsbError("graph loop: %d device-side wait(s) timed out and the run could not be resumed on the "
                             "event hand-over" % n.value)

    def status(self):
        """(event_handover, fallbacks): whether the chains are ordered with stream events, and how many runs had to be
        finished that way after a device-side wait gave up."""
        ev, fb = C.c_uint32(0), C.c_uint32(0)
        self.ctx.check(self.ctx.lib.ammsb_loop_status(self._h, C.byref(ev), C.byref(fb)))
        return bool(ev.value), int(fb.value)

    def timestamps(self, first_step, n):
        """(begin_ns, end_ns) arrays of update_phi for steps first_step .. first_step + n - 1.  Synchronises."""
        b, e = np.zeros(n, dtype=np.float64), np.zeros(n, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        self.ctx.check(self.ctx.lib.ammsb_loop_timestamps(self._h, int(first_step), int(n), b.ctypes.data_as(dp),
                                                          e.ctypes.data_as(dp)))
        return b, e

    STAMP_SLOTS = 8  # AMMSB_LOOP_STAMP_SLOTS
    STAMP_NAMES = ("update_phi", "update_pi", "beta_grads", "sum_theta", "released", "next_batch")

    def step_stamps(self, first_step, n):
        """[n, 8] device times in ns of steps first_step .. first_step + n - 1 (include/ammsb.h,
        ammsb_loop_step_stamps): columns 0..3 = start of update_phi / update_pi / the partial-row kernel / the
        partial-row sum + theta step, 4 = step released, 5 = next mini-batch available.  Synchronises."""
        out = np.zeros((int(n), self.STAMP_SLOTS), dtype=np.float64)
        self.ctx.check(self.ctx.lib.ammsb_loop_step_stamps(self._h, int(first_step), int(n),
                                                           out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def close(self):
        if self._h:
            self.ctx.lib.ammsb_loop_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- stream / event / collective plumbing used by learner.py (torch is the transport only)

def new_stream(ctx):
    """Stream of a Sample (sampling + neighbour kernels of the NEXT mini-batch).  Default priority on
    purpose: a same-box A/B of high priority showed no gain, and at default priority the chain already finishes
    inside the update_phi launch it overlaps."""
    return torch.cuda.Stream(device=ctx.device)


def stream(s):
    return _StreamScope(s)


def new_event():
    return torch.cuda.Event()


def record_event(ev):
    ev.record(_torch_stream())  # on the current stream


def sync_event(ev):
    ev.synchronize()  # the HOST waits (wait_event makes the current stream wait)


def wait_event(ev):
    ev.wait(_torch_stream())  # the current stream waits for the event


def synchronize():
    torch.cuda.synchronize()


def timing_mark():
    """A timing event recorded on torch's current stream (the multi-GPU step trace, Learner.shard_trace)."""
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(_torch_stream())
    return ev


def mark_elapsed_ms(a, b):
    """Device time between two timing_mark()s (both must have completed: call synchronize() first)."""
    return a.elapsed_time(b)


def elapsed_ms(fn):
    """Device time of fn() on the current stream (HIP events), for start-up calibration."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record(torch.cuda.current_stream())
    fn()
    b.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return a.elapsed_time(b)


def pinned(shape, dtype):
    return torch.empty(shape, dtype=dtype, pin_memory=True)


class ClockProbe:
    """ammsb_clock_probe: the shader clock every XCD holds while something else runs.  launch() enqueues the probe
    blocks on the probe's own stream (beside whatever the caller has in flight); read() waits for them and returns
    {"mhz_per_xcd": [...], "mhz": median, "blocks": n}."""

    def __init__(self, ctx, n_blocks=64):
        self.ctx, self.n = ctx, int(n_blocks)
        self.buf = ctx.zeros((self.n, 3), torch.int64)
        self.stream = torch.cuda.Stream(device=ctx.device)
        self.done = torch.cuda.Event()

    def launch(self, spin_us=200):
        c = self.ctx
        c.check(c.lib.ammsb_clock_probe(c.handle, _ptr(self.buf), self.n, int(spin_us), C.c_void_p(self.stream.cuda_stream)))
        self.done.record(self.stream)

    def read(self):
        self.done.synchronize()
        t = self.buf.cpu().numpy().astype(np.float64)
        ok = t[:, 1] > 0
        mhz = t[ok, 0] / t[ok, 1] * 100.0
        per = [round(float(np.median(mhz[t[ok, 2] == x])), 1) if (t[ok, 2] == x).any() else None for x in range(8)]
        return {"mhz_per_xcd": per, "mhz": round(float(np.median(mhz)), 1) if mhz.size else None, "blocks": int(ok.sum())}


def _via_host(dist, group):
    """gloo has no all_gather_into_tensor and no device transport: when the process group is gloo (several
    ranks sharing one GPU in tests, or a box without RCCL) the payload is staged through host memory.  Only
    the exchange changes; every kernel still runs on the device."""
    return dist.get_backend(group) == "gloo"


def all_gather_rows_async(dist, region, chunk, rank, world, group, mode="collective"):
    """In-place all-gather of the `world` equal row chunks of `region` (rank r owns chunk r), issued
    asynchronously: RCCL runs it on its own stream behind the work already queued on the current one,
    so the next phi launch overlaps it.  The in-place form (send buffer = own slot of the receive
    buffer) moves each chunk once.
    mode "collective": one all_gather_into_tensor.  mode "p2p": the direct form -- this rank's chunk sent to each
    of the world - 1 peers and their chunks received, as ONE batch of point-to-point operations (RCCL runs the
    2 (world - 1) transfers of a batch concurrently, one per xGMI link on a fully connected node, where a ring
    all-gather is bound by a single link's rate).  Same result; Learner._calibrate_split times both."""
    if _via_host(dist, group):
        mine = region[rank * chunk:(rank + 1) * chunk].cpu()
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        for r in range(world):
            if r != rank:
                region[r * chunk:(r + 1) * chunk].copy_(parts[r])
        return None
    if mode == "p2p" and world > 1:
        return p2p_all_gather_rows(dist, region, chunk, rank, world, group)
    return dist.all_gather_into_tensor(region, region[rank * chunk:(rank + 1) * chunk], group=group, async_op=True)


def p2p_all_gather_rows(dist, region, chunk, rank, world, group):
    """The direct form of the in-place all-gather: one batch of 2 (world - 1) point-to-point operations.  Returns
    the batch's work handles (wait_work takes the list)."""
    mine = region[rank * chunk:(rank + 1) * chunk]
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    batch = []
    for d in range(1, world):  # rank r sends to r + d and receives from r - d in round d: every link busy in every round
        to, frm = (rank + d) % world, (rank - d) % world
        batch.append(dist.P2POp(dist.isend, mine, peer(to), group))
        batch.append(dist.P2POp(dist.irecv, region[frm * chunk:(frm + 1) * chunk], peer(frm), group))
    return dist.batch_isend_irecv(batch)


def broadcast_async(dist, rows, src, group):
    if _via_host(dist, group):
        host = rows.cpu()
        dist.broadcast(host, src=src, group=group)
        if dist.get_rank(group) != src:
            rows.copy_(host)
        return None
    return dist.broadcast(rows, src=src, group=group, async_op=True)


def wait_work(work):
    if work is None:
        return
    for w in (work if isinstance(work, (list, tuple)) else [work]):
        w.wait()  # makes the current stream wait for the collective; does not block the host


def all_gather_flat(dist, out, local, rank, world, group):
    """out[r] = rank r's `local` (tiny payloads: [2K] gradient partials, 4 perplexity scalars)."""
    if _via_host(dist, group):
        mine = local.reshape(-1).cpu()
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        out.copy_(torch.stack(parts).reshape(out.shape))
        return
    dist.all_gather_into_tensor(out.view(-1), local.reshape(-1), group=group)
