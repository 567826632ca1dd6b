"""CPU stand-in for mcmc_ammsb_gpu_amd.ops, backed by the oracle, for tests only.

Same class and function names as the product's ops module so that learner.py (the sharding and
exchange logic under test) runs unchanged on CPU tensors with the gloo backend.  Never imported by
the product.
"""
import ctypes as C

import numpy as np
import torch

import oracle_lib as orc

SEED_DT = orc.SEED_DT
AmmsbError = RuntimeError


def make_params(N, K, E=0, num_node_sample=32, alpha=0.0, a=0.0315, b=1024.0, c=0.5, epsilon=1e-7,
                eta0=1.0, eta1=1.0, quantize=True):
    if alpha == 0:
        alpha = float(np.float32(1.0) / np.float32(K))
    p = orc.make_params(N, K, num_node_sample, alpha=np.float32(alpha), a=a, b=b, c=c, epsilon=epsilon,
                        eta0=eta0, eta1=eta1)
    p.E = E  # python attribute only; the oracle struct has no E
    return p


class Context:
    def __init__(self, params, device=None):
        self.params = params
        self.device = torch.device("cpu")
        self.lib = None

    def empty(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    def from_numpy(self, arr):
        a = np.ascontiguousarray(arr)
        if a.dtype == np.uint64:
            a = a.view(np.int64)
        elif a.dtype == np.uint32:
            a = a.view(np.int32)
        elif a.dtype == SEED_DT:
            a = a.view(np.int64).reshape(-1, 2)
        return torch.from_numpy(a.copy())

    def close(self):
        pass


def _u32(t):
    return t.numpy().view(np.uint32)


def _u64(t):
    return t.numpy().view(np.uint64)


def _seeds(t):
    return t.numpy().view(np.uint64).reshape(-1, 2).view(SEED_DT).reshape(-1)


class Random:
    def __init__(self, ctx, size, seed):
        self.ctx, self.size = ctx, int(size)
        self.seeds = torch.zeros((self.size, 2), dtype=torch.int64)
        self.SetSeed(seed)

    def SetSeed(self, seed):
        orc.lib().orc_rng_init(_seeds(self.seeds), self.size, int(seed[0]), int(seed[1]))

    def host(self):
        return _seeds(self.seeds).copy()


class DeviceSet:
    def __init__(self, ctx, slots, num_bins, prime_idx):
        self.slots = np.ascontiguousarray(slots, dtype=np.uint64)
        self.num_bins, self.prime_idx = int(num_bins), int(prime_idx)


class RowPartitionedMatrix:
    def __init__(self, ctx, rows, cols, rows_in_block=0, dtype=torch.float32):
        self.rows, self.cols = int(rows), int(cols)
        self.blocks = [torch.zeros((self.rows, self.cols), dtype=dtype)]

    def Rows(self):
        return self.rows

    def Cols(self):
        return self.cols

    def RowsPerBlock(self):
        return self.rows

    def Blocks(self):
        return self.blocks

    def host(self):
        return self.blocks[0].numpy().copy()

    def flat(self):
        return self.blocks[0].numpy().reshape(-1)


def RandomGammaAndNormalize(ctx, eta0, eta1, pi, phi_sum, seed=(11, 113)):
    orc.lib().orc_pi_init_gamma(pi.flat(), phi_sum.numpy(), pi.rows, pi.cols, eta0, eta1, seed[0], seed[1])


def beta_from_theta(ctx, theta, beta):
    orc.lib().orc_beta_from_theta(theta.numpy(), beta.numpy(), ctx.params.K)


class NeighborSampler:
    def __init__(self, ctx, max_nodes, neighbor_seed=(56, 57), wg=32):
        self.ctx, self.n, self.local = ctx, ctx.params.n_neighbors, int(wg)
        self.hash = torch.zeros((max_nodes, 2 * self.n), dtype=torch.int32)
        self.data = torch.zeros((max_nodes, self.n), dtype=torch.int32)
        self.rand = Random(ctx, max_nodes * 2 * self.n, neighbor_seed)

    def __call__(self, num_samples, nodes):
        ns = int(num_samples)
        orc.lib().orc_sample_neighbors(_seeds(self.rand.seeds), _u32(nodes)[:ns].copy(), ns, self.ctx.params.N,
                                       self.n, self.local, _u32(self.hash).reshape(-1), _u32(self.data).reshape(-1))

    def GetData(self):
        return self.data


class PhiUpdater:
    """Work-group form with the group range of the C ABI: groups outside [lo, hi) leave phi_vec rows
    and stream states untouched."""

    def __init__(self, ctx, beta, pi, phi, training_set, max_nodes, phi_seed=(42, 43), phi_wg_size=64,
                 phi_disable_noise=False):
        self.ctx, self.beta, self.pi, self.phi, self.set = ctx, beta, pi, phi, training_set
        self.local, self.noise = int(phi_wg_size), not phi_disable_noise
        self.phi_vec = torch.zeros((int(max_nodes), ctx.params.K), dtype=torch.float32)
        self.rand = Random(ctx, int(max_nodes) * self.local, phi_seed)
        self.count_calls = 0

    def update_phi(self, nodes, neighbors, n, group_begin=0, group_end=0xFFFFFFFF):
        p, L = self.ctx.params, self.local
        n = int(n)
        G = min(n, 65535)
        lo, hi = int(group_begin), min(int(group_end), G)
        nodes_h, nb_h = _u32(nodes)[:n], _u32(neighbors).reshape(-1, p.n_neighbors)[:n]
        # run the full oracle on copies, then keep only what the groups in [lo, hi) own
        seeds = _seeds(self.rand.seeds)
        trial = seeds.copy()
        out = orc.update_phi(p, self.beta.numpy(), self.pi.flat(), self.phi.numpy(), self.set, nodes_h.copy(),
                             nb_h.reshape(-1).copy(), self.count_calls, trial, L, 1, self.noise)
        idx = np.arange(n)
        mine = (idx % G >= lo) & (idx % G < hi)
        self.phi_vec.numpy()[:n][mine] = out[mine]
        seeds[lo * L:hi * L] = trial[lo * L:hi * L]

    def update_pi(self, nodes, n, phi_vec=None):
        n = int(n)
        pv = self.phi_vec if phi_vec is None else phi_vec
        orc.update_pi(self.ctx.params, self.pi.flat(), self.phi.numpy(), pv.numpy()[:n].reshape(-1).copy(),
                      _u32(nodes)[:n].copy(), self.local, 1)


class BetaUpdater:
    def __init__(self, ctx, theta, beta, pi, training_set, beta_seed=(44, 45), beta_wg_size=256, disable_noise=False):
        self.ctx, self.theta, self.beta, self.pi, self.set = ctx, theta, beta, pi, training_set
        self.local = int(beta_wg_size)
        self.rand = Random(ctx, ctx.params.K, beta_seed)
        self.grads = torch.zeros((2 * ctx.params.K,), dtype=torch.float32)
        self.count_calls = 0

    def calculate_grads(self, edges, num_edges, edge_begin=0, edge_end=0xFFFFFFFF, out=None):
        g = self.grads if out is None else out
        lo, hi = int(edge_begin), min(int(edge_end), int(num_edges))
        if lo >= hi:
            g.zero_()
            return g
        e = _u64(edges)[lo:hi].copy()
        g.numpy()[:] = orc.beta_grads(self.ctx.params, self.theta.numpy(), self.beta.numpy(), self.pi.flat(),
                                      self.set, e, self.local, 1, order=1)
        return g

    def update_theta(self, scale, grads=None):
        g = self.grads if grads is None else grads
        b = orc.update_theta(self.ctx.params, self.theta.numpy(), g.numpy().copy(), self.count_calls, scale,
                             _seeds(self.rand.seeds))
        self.beta.numpy()[:] = b


class PerplexityCalculator:
    def __init__(self, ctx, beta, pi, edges, edge_set, ppx_wg_size=64):
        self.ctx, self.beta, self.pi, self.edges, self.set = ctx, beta, pi, edges, edge_set
        self.local = int(ppx_wg_size)
        self.num_edges = int(edges.numel())
        self.ppx_per_edge = torch.zeros((max(self.num_edges, 1),), dtype=torch.float32)
        self.sums = torch.zeros((4,), dtype=torch.int64)
        self.count_calls = 0

    def partial(self, edge_begin=0, edge_end=0xFFFFFFFF):
        lo, hi = int(edge_begin), min(int(edge_end), self.num_edges)
        raw = self.sums.numpy()
        raw[:] = 0
        if lo < hi:
            state = self.ppx_per_edge.numpy()[lo:hi].copy()
            s, _ = orc.perplexity(self.ctx.params, self.beta.numpy(), self.pi.flat(), self.set,
                                  _u64(self.edges)[lo:hi].copy(), self.count_calls, self.local, 1, state)
            self.ppx_per_edge.numpy()[lo:hi] = state
            raw[:2] = np.array([s.link_ll, s.nonlink_ll], dtype=np.float64).view(np.int64)
            raw[2:] = np.array([s.link_cnt, s.nonlink_cnt], dtype=np.uint64).view(np.int64)
        return self.sums

    @staticmethod
    def unpack(sums):
        raw = sums.numpy()
        ll = raw[:2].copy().view(np.float64)
        cnt = raw[2:].copy().view(np.uint64)
        return float(ll[0]), float(ll[1]), int(cnt[0]), int(cnt[1])

    @staticmethod
    def value(link_ll, nonlink_ll, link_cnt, nonlink_cnt):
        return -((link_ll + nonlink_ll) / (link_cnt + nonlink_cnt)) if link_cnt + nonlink_cnt else 0.0


# ---- plumbing: no streams on the CPU; gloo collectives

class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def new_stream(ctx):
    return None


def pin_current_stream():
    return _Null()


def stream(s):
    return _Null()


def new_event():
    return None


def record_event(ev):
    pass


def wait_event(ev):
    pass


def sync_event(ev):
    pass


def synchronize():
    pass


def timing_mark():
    import time
    return time.perf_counter()


def mark_elapsed_ms(a, b):
    return (b - a) * 1e3


def pinned(shape, dtype):
    return torch.zeros(shape, dtype=dtype)


class _Done:
    def wait(self):
        pass


def all_gather_rows_async(dist, region, chunk, rank, world, group, mode="collective"):
    parts = [torch.zeros_like(region[:chunk]) for _ in range(world)]
    dist.all_gather(parts, region[rank * chunk:(rank + 1) * chunk].clone(), group=group)
    for r in range(world):
        region[r * chunk:(r + 1) * chunk] = parts[r]
    return _Done()


def broadcast_async(dist, rows, src, group):
    dist.broadcast(rows, src=src, group=group)
    return _Done()


def wait_work(work):
    work.wait()


def all_gather_flat(dist, out, local, rank, world, group):
    parts = [torch.zeros_like(local.reshape(-1)) for _ in range(world)]
    dist.all_gather(parts, local.reshape(-1).clone(), group=group)
    for r in range(world):
        out[r] = parts[r].reshape(out[r].shape)
