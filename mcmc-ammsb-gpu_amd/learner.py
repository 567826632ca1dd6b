"""mcmc::Learner on top of the operator mirror (ops.py): same state, same per-iteration order, same
double-buffered sampling as the reference (mcmc/learner.{h,cc}), plus the multi-GPU shard/exchange
the reference does not have.

    Learner(cfg, dataset)      learner.cc:77-156   allocate beta/theta/pi/phi, upload both sets, init
    Run(max_iters)             learner.cc:214-250  [join sample | launch next sample] phi, pi, beta
    HeldoutPerplexity()        learner.cc:196-203
    PrintStats()               learner.cc:252-299

Multi-GPU (one process per GPU, torch.distributed over RCCL): pi, phi_sum, beta/theta and both
cuckoo sets are replicated; the mini-batch is the same on every rank (same seeds).  Per iteration
    phi    the first g_rep virtual groups are REPLICATED (every rank computes them: same streams, same
           results, nothing to send); the other 65535 - g_rep are cut into blocks of
           Cc = ceil((65535 - g_rep) / (R * chunks)); block b belongs to rank b % R and is processed
           in chunk b // R -- a fixed ownership of RNG streams, so no stream state ever crosses ranks
           and results do not depend on R.  Chunk c's rows (one block per rank, contiguous) are
           all-gathered asynchronously while chunk c+1 and then the replicated groups are being
           computed; update_pi runs on every rank over all nodes once every chunk has arrived.
           Why replicate: every rank must end up with all 4K-byte rows, and a row costs about as much
           to receive over one xGMI link as to recompute (C3, 2 GPUs: 134 MB over a 77 GB/s link =
           1.75 ms against 0.85 ms to compute the same half), so the fastest split computes part of
           the "other" rows locally.  g_rep is fixed at start-up: a fraction given in the Config or
           ("auto") balanced from a timed update_phi and a timed all-gather;
    beta   rank r sums the gradient over its contiguous slice of the mini-batch edges; the R partial
           [2K] vectors are all-gathered and summed (one kernel, fixed association); every rank runs the identical
           update_theta (same streams) -> no broadcast;
    ppx    contiguous slices of the held-out edges; 4 scalars per rank are all-gathered.
"""
import concurrent.futures
import os
import time

import numpy as np
import torch

from ._capi import AmmsbError, MAX_GROUPS


class Config:
    """mcmc::Config (config.h:25-102): same field names and defaults.  Extras are marked (new)."""

    def __init__(self, **kw):
        self.heldout_ratio = 0.01
        self.alpha = 0.001
        self.a, self.b, self.c = 0.0315, 1024.0, 0.5
        self.epsilon = 1e-7
        self.eta0, self.eta1 = 1.0, 1.0
        self.K = 32
        self.mini_batch_size = 32
        self.num_node_sample = 32
        self.N = 0
        self.E = 0
        self.ppx_wg_size = 32
        self.ppx_interval = 100
        self.neighbor_sampler_wg_size = 32
        self.phi_wg_size = 32
        self.beta_wg_size = 32
        self.phi_seed = (42, 43)
        self.beta_seed = (113, 117)
        self.neighbor_seed = (3337, 54351)
        self.phi_disable_noise = False
        # MCMC_CALC_TRAIN_PPX (CMakeLists.txt:41, config.h:26-28, learner.cc:47-75) as a run-time switch
        self.calc_train_ppx = False
        self.training_ppx_ratio = 0.01
        self.training_ppx_seed = 1                   # (new) rand_r seed of the sampled non-links
        self.strategy = "Node"
        # kernel-variant knobs of the reference (config.h:60-66): accepted for source compatibility; every value
        # selects this build's one kernel family
        self.phi_mode = "PHI_NODE_PER_WORKGROUP_NAIVE"
        self.phi_probs_shared = self.phi_grads_shared = self.phi_pi_shared = True
        self.phi_vector_width = 1
        self.sum_grads_vector_width = 1
        self.sample_seeds = (1804289383, 846930886)  # (new) Sample::seed = rand() twice, sample.cc:132
        self.device_sampling = False                 # (new) draw mini-batches on the device
        self.device_sampling_seed = (1234, 5678)     # (new)
        self.sample_parallel = True                  # MCMC_SAMPLE_PARALLEL, CMakeLists.txt:42
        self.phi_chunks = 4                          # (new) multi-GPU: phi launches per iteration (exchange overlap)
        self.phi_replicate = "auto"                  # (new) multi-GPU: fraction of groups every rank computes itself
        self.phi_exchange = "auto"                   # (new) multi-GPU: "collective" (all-gather) | "p2p" (direct peer
                                                     # sends, one per link) | "auto" (time both at start-up)
        self.beta_grads = "auto"                     # (new) multi-GPU: "sharded" (edge slices + all-gather of the partial
                                                     # sums, rank-ordered) | "replicated" (every rank the whole
                                                     # gradient: no collective, theta bit-identical to one GPU's) |
                                                     # "auto": replicated where update_pi can be folded into the
                                                     # gradient launch (K <= 1024, device-sampled Node batches: the one
                                                     # launch costs less than update_pi + a slice + the all-gather)
        self.beta_shard_min_edges = 4096             # (new) multi-GPU: a mini-batch with at most this many edges is not
                                                     # cut over the ranks -- every rank computes its whole gradient (a link
                                                     # mini-batch has a few dozen edges: the all-gather of the partial
                                                     # sums would cost more than the gradient, and the result is then
                                                     # the single-GPU one bit for bit)
        self.pi_placement_candidates = 12            # (new) where pi lands in HBM moves update_phi's launch time by up to
                                                     # 10 % (profiles/README.md, round 4): at start-up this many
                                                     # allocations of pi are timed under update_phi and the fastest
                                                     # is kept (only when pi is >= 1 GB and the candidates fit in a
                                                     # third of the free HBM; 0 or 1: off; AMMSB_PI_CANDIDATES overrides)
        self.force_exchange = False                  # (new, tests) take the multi-rank code path even with one rank
        # (new) whole iterations as captured hipGraphs (include/ammsb.h ammsb_loop): "auto" = whenever it applies
        # (device sampling, one rank); False keeps the eager launch-by-launch loop (the parity form); results are
        # bit-identical either way
        self.graph_launch = "auto"
        self.graph_timestamps = False                # (new) keep device time stamps around update_phi (bench.py)
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError("unknown Config field %s" % k)
            setattr(self, k, v)

    @classmethod
    def from_cli_defaults(cls, **kw):
        """The reference CLI's defaults where they differ from the struct's (main.cc:43-81, :153)."""
        base = dict(alpha=0.0, beta_seed=(44, 45), neighbor_seed=(56, 57))
        base.update(kw)
        return cls(**base)


def _check_kernel_variant_knobs(cfg):
    """The reference's kernel-variant switches (config.h:60-66, phi.cc:608-700) -- same rules as the C++ host
    (host/operators.cc CheckedPhiConfig): the work-group modes are one kernel family here (same arithmetic, lane map and
    streams); PHI_NODE_PER_THREAD (one stream per node, phi.cc:124-152) is refused; phi_vector_width > 1 (Floatn column
    ownership, phi.cc:214-275) is NOT reproduced and says so."""
    import warnings
    mode = str(cfg.phi_mode)
    mode = {"THREAD": "PHI_NODE_PER_THREAD", "WG-NAIVE": "PHI_NODE_PER_WORKGROUP_NAIVE", "WG-SHARED": "PHI_NODE_PER_WORKGROUP_SHARED",
            "WG-GEN": "PHI_NODE_PER_WORKGROUP_CODE_GEN"}.get(mode.upper(), mode)  # the CLI's tokens, config.cc:118-131
    if mode == "PHI_NODE_PER_THREAD":
        raise AmmsbError("phi_mode PHI_NODE_PER_THREAD is the reference's CPU-device kernel (one RNG stream per node, "
                         "phi.cc:124-152): no MI355X form; use a PHI_NODE_PER_WORKGROUP_* mode")
    if mode not in ("PHI_NODE_PER_WORKGROUP_NAIVE", "PHI_NODE_PER_WORKGROUP_SHARED", "PHI_NODE_PER_WORKGROUP_CODE_GEN"):
        raise AmmsbError("Invalid phi mode: %s" % mode)  # config.cc:118-131
    if mode != "PHI_NODE_PER_WORKGROUP_NAIVE":
        warnings.warn("phi_mode %s: runs the PHI_NODE_PER_WORKGROUP_NAIVE kernel family (same arithmetic, lane-to-column "
                      "map and RNG streams)" % mode)
    if int(cfg.phi_vector_width) != 1:
        warnings.warn("phi_vector_width %d: NOT reproduced -- results follow phi_vector_width 1 (the reference's Floatn "
                      "column ownership, phi.cc:214-275, changes WG_SUM's association and the stream-to-column map)"
                      % int(cfg.phi_vector_width))
    if int(cfg.sum_grads_vector_width) != 1:
        warnings.warn("sum_grads_vector_width %d: accepted; it only widens sum_grads' loads (beta.cc:39-49), the sums "
                      "are the same" % int(cfg.sum_grads_vector_width))


class Sample:
    """sample.h:51-92: one of the two mini-batch buffers with its own queue (stream)."""

    def __init__(self, learner, seed):
        L = learner
        cfg, ds, ctx = L.cfg, L.dataset, L.ctx
        self.stream = L.ops.new_stream(ctx)
        self.max_edges = ds.max_edges(cfg.mini_batch_size)
        self.max_nodes = ds.max_nodes(cfg.mini_batch_size)
        self.dev_edges = ctx.zeros((self.max_edges,), torch.int64)   # zeros: whatever a kernel reads here is a valid id
        self.dev_nodes = ctx.zeros((self.max_nodes,), torch.int32)
        self.pin_edges = L.ops.pinned((self.max_edges,), torch.int64)
        self.pin_nodes = L.ops.pinned((self.max_nodes,), torch.int32)
        self.seed = int(seed)
        self.edges = np.zeros(0, dtype=np.uint64)
        self.nodes_vec = np.zeros(0, dtype=np.uint32)
        self.n_edges = self.n_nodes = 0
        self.neighbor_sampler = L.ops.NeighborSampler(ctx, self.max_nodes, cfg.neighbor_seed,
                                                      cfg.neighbor_sampler_wg_size)
        self.choice = None                   # device sampling: the (link, u, n, n_candidates) sitting in the buffers
        self.ready = L.ops.new_event()      # sampling finished (recorded on self.stream)
        self.consumed = L.ops.new_event()   # the iteration that used this sample finished (main stream)
        self.consumed_valid = False
        self.copied = L.ops.new_event()     # the host-to-device copies out of pin_edges / pin_nodes have executed
        self.copied_valid = False


class Learner:
    def __init__(self, cfg, dataset, ops=None, rank=0, world_size=1, group=None, ctx=None):
        if ops is None:
            from . import ops as _ops
            ops = _ops
        self.ops = ops
        self.cfg, self.dataset = cfg, dataset
        self.rank, self.world, self.group = int(rank), int(world_size), group
        self.sharded = self.world > 1 or bool(getattr(cfg, "force_exchange", False))
        cfg.N, cfg.E = dataset.N, dataset.E
        _check_kernel_variant_knobs(cfg)
        if cfg.alpha == 0:
            cfg.alpha = float(np.float32(1.0) / np.float32(cfg.K))  # main.cc:153
        self.params = ops.make_params(cfg.N, cfg.K, cfg.E, cfg.num_node_sample, cfg.alpha, cfg.a, cfg.b, cfg.c,
                                      cfg.epsilon, cfg.eta0, cfg.eta1)
        self.ctx = ctx if ctx is not None else ops.Context(self.params)
        c = self.ctx
        K, N = cfg.K, cfg.N
        # learner.cc:80-91
        self.beta = c.zeros((2 * K,), torch.float32)
        self.theta = c.zeros((2 * K,), torch.float32)
        self.pi = ops.RowPartitionedMatrix(c, N, K)
        self.phi = c.zeros((N,), torch.float32)
        ts, hs = dataset.training, dataset.heldout
        self.trainingSet = ops.DeviceSet(c, ts.Serialize(), ts.BinsPerBucket(), ts.PrimeIdx())
        self.heldoutSet = ops.DeviceSet(c, hs.Serialize(), hs.BinsPerBucket(), hs.PrimeIdx()) if hs else None
        self.heldoutEdges = c.from_numpy(dataset.heldout_edges)
        if self.heldoutSet is None:
            raise AmmsbError("held-out set is empty: raise heldout_ratio")
        max_nodes = dataset.max_nodes(cfg.mini_batch_size)
        self.nch = max(1, int(cfg.phi_chunks)) if self.sharded else 1
        self.xchg_mode = str(getattr(cfg, "phi_exchange", "auto"))  # "auto" is settled by _calibrate_split
        self.g_rep = 0
        self._set_split(0 if cfg.phi_replicate == "auto" else int(round(float(cfg.phi_replicate) * MAX_GROUPS)))
        # rows for any split: the exchanged blocks may reach past the last group by less than one block each
        phi_rows = max(max_nodes, MAX_GROUPS + self.world * max(1, int(cfg.phi_chunks)) if self.sharded else 0)
        # learner.cc:105-116
        self.heldoutPerplexity = ops.PerplexityCalculator(c, self.beta, self.pi, self.heldoutEdges,
                                                          self.heldoutSet, cfg.ppx_wg_size)
        self.trainingPerplexity = None
        if cfg.calc_train_ppx:  # learner.cc:92-104
            te = dataset.train_ppx_edges(cfg.training_ppx_ratio, cfg.training_ppx_seed)
            if te.size == 0:
                raise AmmsbError("training perplexity: no edges (raise training_ppx_ratio)")
            self.trainingPerplexityEdges = c.from_numpy(te)
            self.trainingPerplexity = ops.PerplexityCalculator(c, self.beta, self.pi, self.trainingPerplexityEdges,
                                                               self.trainingSet, cfg.ppx_wg_size)
        self.phiUpdater = ops.PhiUpdater(c, self.beta, self.pi, self.phi, self.trainingSet, phi_rows,
                                         cfg.phi_seed, cfg.phi_wg_size, cfg.phi_disable_noise)
        self.betaUpdater = ops.BetaUpdater(c, self.theta, self.beta, self.pi, self.trainingSet, cfg.beta_seed,
                                           cfg.beta_wg_size)
        self.stepCount = 1
        self.time = 0.0
        self.samplingTime = 0.0
        self.edges_done = 0
        self.step_log = None  # a list here collects every iteration's sizes (bench.py): (n_edges, n_nodes) per step
                              # from the eager loop, one int64 array of n_edges per call from the graph loop
        self.samples = [Sample(self, cfg.sample_seeds[0])]
        if cfg.sample_parallel:
            self.samples.append(Sample(self, cfg.sample_seeds[1]))
        self.phase = 0
        self.futures = [None, None]
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=1) if not cfg.device_sampling else None
        self.dev_sampler = None
        self.loop = None
        self._loop_dirty = False
        if cfg.device_sampling:
            off, tgt = dataset.training_csr()
            # held-out links of a vertex are invalid non-link partners too (sample.cc:283-285)
            he = np.ascontiguousarray(dataset.heldout_edges, dtype=np.uint64)
            he = he[dataset.heldout.Has(he)] if he.size else he
            ends = np.concatenate([he >> np.uint64(32), he & np.uint64(0xFFFFFFFF)]).astype(np.int64)
            hdeg = np.bincount(ends, minlength=N)[:N]
            self.dev_sampler = ops.DeviceMiniBatchSampler(c, off, tgt, self.trainingSet, self.heldoutSet,
                                                          cfg.mini_batch_size, cfg.device_sampling_seed,
                                                          heldout_degree=hdeg)
        # multi-GPU: how the beta gradient is computed (Config.beta_grads)
        self.grads_mode, self.grads_fused = "sharded", False
        if self.sharded:
            want = str(getattr(cfg, "beta_grads", "auto"))
            if want not in ("auto", "sharded", "replicated"):
                raise AmmsbError("beta_grads must be auto, sharded or replicated (got %r)" % (want,))
            fusable = (self.dev_sampler is not None and cfg.strategy in ("Node", "NodeLink", "NodeNonLink")
                       and hasattr(self.betaUpdater, "can_fuse_update_pi")
                       and self.betaUpdater.can_fuse_update_pi(self.phiUpdater))
            self.grads_mode = "replicated" if want == "replicated" or (want == "auto" and fusable) else "sharded"
            self.grads_fused = fusable
        if self.sharded:
            # the replicated groups run on their own stream next to the exchanged blocks: with 8 ranks a block is
            # ~1800 single-wave nodes, well under what the chip holds, and the two launches fill it together
            self.rep_stream = ops.new_stream(c)
            self.ev_fork, self.ev_join = ops.new_event(), ops.new_event()
            self.all_grads = c.zeros((self.world, 2 * K), torch.float32)
            self.grads_sum = c.zeros((2 * K,), torch.float32)
            self.all_sums = c.zeros((self.world, 4), torch.int64)
            self.tail_buf = c.zeros((max(max_nodes - MAX_GROUPS, 1), K), torch.float32)
        # learner.cc:150-155: theta_0 (host std::gamma) -> beta_0; pi_0 / phi_sum_0 (device gamma {11,113})
        from . import hostlib
        self.theta.copy_(c.from_numpy(hostlib.theta_init(K, cfg.eta0, cfg.eta1)))
        ops.beta_from_theta(c, self.theta, self.beta)
        ops.RandomGammaAndNormalize(c, cfg.eta0, cfg.eta1, self.pi, self.phi)
        self.pi_placement = None
        self._place_pi()
        ops.synchronize()
        if self.sharded and cfg.phi_replicate == "auto":
            self._calibrate_split()
        want_graph = cfg.graph_launch
        can_graph = (self.dev_sampler is not None and not self.sharded and len(self.samples) == 2
                     and hasattr(ops, "GraphLoop") and cfg.strategy in ("Node", "NodeLink", "NodeNonLink"))
        if want_graph is True and not can_graph:
            raise AmmsbError("graph_launch needs device sampling (Node strategies), sample_parallel and a single rank")
        if want_graph in (True, "auto") and can_graph:
            self.loop = ops.GraphLoop(c, self.theta, self.beta, self.pi, self.phi, self.trainingSet, self.heldoutSet,
                                      self.phiUpdater, self.betaUpdater, self.samples, self.dev_sampler,
                                      timestamps=bool(cfg.graph_timestamps))

    # ------------------------------------------------------------------ sampling (learner.cc:175-194)

    def _do_sample(self, sample):
        """DoSample: produce a mini-batch into `sample` on its own stream; returns the weight."""
        ops, cfg = self.ops, self.cfg
        with ops.stream(sample.stream):
            if sample.consumed_valid:
                ops.wait_event(sample.consumed)  # do not overwrite buffers a running iteration still reads
            if self.dev_sampler is not None:
                sample.choice = self.dev_sampler.choose(cfg.strategy)
                ne, nv, weight = self.dev_sampler.enqueue(sample.choice, sample.dev_edges, sample.dev_nodes)
            else:
                edges, nodes, weight, sample.seed = self.dataset.sample(cfg.mini_batch_size, cfg.strategy,
                                                                        sample.seed)
                if nodes.size == 0:
                    raise AmmsbError("mini-batch size = 0!")  # learner.cc:179
                ne, nv = edges.size, nodes.size
                if ne > sample.max_edges or nv > sample.max_nodes:
                    raise AmmsbError("%d | %d" % (ne, sample.max_edges))  # learner.cc:184-188
                sample.edges, sample.nodes_vec = edges, nodes
                # the staging buffers are read by an asynchronous copy: the one issued for this sample two iterations ago
                # must have executed before they are overwritten (the enqueue side can run ahead of the device, e.g.
                # while the first launches load their code: a mini-batch then silently became the next one's)
                if sample.copied_valid:
                    ops.sync_event(sample.copied)
                sample.pin_edges[:ne].copy_(torch.from_numpy(edges.view(np.int64)))
                sample.pin_nodes[:nv].copy_(torch.from_numpy(nodes.view(np.int32)))
                sample.dev_edges[:ne].copy_(sample.pin_edges[:ne], non_blocking=True)
                sample.dev_nodes[:nv].copy_(sample.pin_nodes[:nv], non_blocking=True)
                ops.record_event(sample.copied)
                sample.copied_valid = True
            sample.n_edges, sample.n_nodes = int(ne), int(nv)
            sample.neighbor_sampler(sample.n_nodes, sample.dev_nodes)
            ops.record_event(sample.ready)
        return weight

    def _launch_sample(self, idx):
        s = self.samples[idx]
        if self.pool is not None and self.cfg.sample_parallel:
            self.futures[idx] = self.pool.submit(self._do_sample, s)
        else:
            w = self._do_sample(s)
            f = concurrent.futures.Future()
            f.set_result(w)
            self.futures[idx] = f

    # ------------------------------------------------------------------ multi-GPU exchange

    def _dist(self):
        import torch.distributed as dist
        return dist

    # ------------------------------------------------------------------ where pi lands in HBM

    def _place_pi(self):
        """update_phi gathers 4 KiB rows of pi at random, and what the memory system delivers for that depends on
        WHERE the table sits: the same launch over copies of pi allocated one after the other in one process differs
        by up to 10 %, stably per allocation (tools/placement_phi.py, profiles/r04_placement_phi.txt) -- the
        "per-process level" of update_phi's launch time.  So, once pi is initialised: copy it into a few more
        allocations, time full-size update_phi launches over each (warm clocks first, then round-robin, so that
        anything time-dependent hits all alike), keep the fastest, release the rest.  Results do not depend on it
        (every candidate holds the same pi; the streams and the call counter are restored).  Every rank of a
        multi-GPU job places its own replica."""
        ops, cfg, phi = self.ops, self.cfg, self.phiUpdater
        want = int(os.environ.get("AMMSB_PI_CANDIDATES", getattr(cfg, "pi_placement_candidates", 0)) or 0)
        if want < 2 or not hasattr(ops, "elapsed_ms") or len(self.pi.blocks) != 1:
            return
        nbytes = 4 * cfg.N * cfg.K
        free = torch.cuda.mem_get_info(self.ctx.device)[0]
        if nbytes < (1 << 30):
            return  # a table of this size is a cache matter, not an HBM-placement one
        want = min(want, 1 + int(free // 3 // nbytes))  # the extra candidates take at most a third of the free memory
        if want < 2:
            self.pi_placement = {"candidates": 1, "why": "pi is %.0f GB: no room for a second candidate" % (nbytes / 1e9)}
            return
        c = self.ctx
        t_begin = time.perf_counter()
        n_nodes = min(self.samples[0].max_nodes, MAX_GROUPS)
        gen = torch.Generator(device="cpu").manual_seed(7)
        nodes = c.from_numpy(torch.randperm(cfg.N, generator=gen)[:n_nodes].numpy().astype(np.uint32))
        nbrs = c.from_numpy(torch.randint(0, cfg.N, (n_nodes, cfg.num_node_sample), generator=gen).numpy().astype(np.uint32))
        keep, calls = phi.rand.seeds.clone(), phi.count_calls
        phi.count_calls = 1
        cands = [self.pi] + [ops.RowPartitionedMatrix(c, cfg.N, cfg.K) for _ in range(want - 1)]
        for p in cands[1:]:
            p.blocks[0].copy_(self.pi.blocks[0])  # the initialised pi: whichever candidate is kept holds it
        mine = phi.pi
        times = [[] for _ in cands]

        def three(p):
            phi.pi = p
            for _ in range(3):
                phi.update_phi(nodes, nbrs, n_nodes)
        try:
            # clocks first: a launch right after idle runs 10-20 % long and tells candidates apart badly
            t_warm = time.perf_counter()
            while time.perf_counter() - t_warm < 0.4:
                for p in cands:
                    three(p)
                ops.synchronize()
            for rnd in range(3):  # round-robin: anything time-dependent hits all candidates alike
                for i, p in enumerate(cands):
                    times[i].append(ops.elapsed_ms(lambda: three(p)) / 3.0)
        finally:
            phi.pi = mine
            phi.rand.seeds.copy_(keep)
            phi.count_calls = calls
        med = [float(np.median(t)) for t in times]
        best = int(np.argmin(med))
        if best != 0:
            self.pi.adopt(cands[best])  # the objects holding self.pi see the new blocks; the first candidate is released
        self.pi_placement = {"candidates": len(cands), "update_phi_ms": [round(x, 4) for x in med], "kept": best,
                             "first_allocation_ms": round(med[0], 4), "kept_ms": round(med[best], 4)}
        del cands, p
        ops.synchronize()
        torch.cuda.empty_cache()
        self.pi_placement["seconds"] = round(time.perf_counter() - t_begin, 2)

    def _set_split(self, g_rep):
        """Fix the ownership map: groups [0, g_rep) replicated, the rest in R * chunks blocks of cc groups."""
        self.g_rep = max(0, min(int(g_rep), MAX_GROUPS))
        # keep a rank's block of a chunk near or above one chip-load of single-wave nodes (3 per SIMD = 3072)
        own = (MAX_GROUPS - self.g_rep) // max(self.world, 1)
        self.nch = max(1, min(max(1, int(self.cfg.phi_chunks)), own // 3072)) if self.sharded else 1
        blocks = self.world * self.nch
        self.cc = max(1, (MAX_GROUPS - self.g_rep + blocks - 1) // blocks)

    def _calibrate_split(self):
        """Choose g_rep so that computing (replicated + own) groups takes as long as receiving the others'.
        T = one update_phi over a full synthetic mini-batch (all groups), X = one all-gather of a full
        phi_vec; a fraction rho replicated costs T (rho + (1 - rho) / R) of compute and X (1 - rho) of
        exchange.  Stream states and pi are restored / untouched; every rank adopts rank 0's answer."""
        ops, phi, dist = self.ops, self.phiUpdater, self._dist()
        if not hasattr(ops, "elapsed_ms"):
            return  # operator sets without timers (the CPU stand-in of the tests) keep the configured split
        c, R, K = self.ctx, self.world, self.cfg.K
        n_nodes = min(self.samples[0].max_nodes, MAX_GROUPS)
        gen = torch.Generator(device="cpu").manual_seed(1)
        nodes = c.from_numpy(torch.randperm(self.cfg.N, generator=gen)[:n_nodes].numpy().astype(np.uint32))
        nbrs = c.from_numpy(torch.randint(0, self.cfg.N, (n_nodes, self.cfg.num_node_sample), generator=gen)
                            .numpy().astype(np.uint32))
        keep = phi.rand.seeds.clone()
        calls = phi.count_calls
        phi.count_calls = 1
        per = (n_nodes + R - 1) // R
        region = phi.phi_vec[:per * R] if phi.phi_vec.shape[0] >= per * R else phi.phi_vec[:(phi.phi_vec.shape[0] // R) * R]
        chunk = region.shape[0] // R

        def t_phi():
            phi.update_phi(nodes, nbrs, n_nodes)

        def t_xchg(mode):
            ops.wait_work(ops.all_gather_rows_async(dist, region, chunk, self.rank, R, self.group, mode=mode))
        T = min(ops.elapsed_ms(t_phi) for _ in range(3))
        scale = n_nodes / float(chunk * R)
        X = min(ops.elapsed_ms(lambda: t_xchg("collective")) for _ in range(3)) * scale
        x_p2p = None
        if self.xchg_mode == "auto" and R > 1 and not ops._via_host(dist, self.group):
            # The direct form (one send per peer and link): time it, and check its result against the collective's --
            # whole rows, on EVERY rank.  No exception is swallowed here: a rank that failed inside a batch of
            # point-to-point operations leaves its peers blocked in theirs, and the communicator in an undefined state;
            # that ends the run (loudly) instead of a later collective hanging.
            ops.synchronize()
            want = region.clone()
            x_p2p = min(ops.elapsed_ms(lambda: t_xchg("p2p")) for _ in range(3)) * scale
            ops.synchronize()
            ok = torch.equal(region, want)
            del want
            # one verdict for the whole job: usable only if it was right everywhere, and rank 0's timing decides
            v = torch.tensor([1.0 if ok else 0.0, x_p2p if self.rank == 0 else 0.0, X if self.rank == 0 else 0.0],
                             dtype=torch.float64).to(c.device)
            allv = torch.zeros((R, 3), dtype=torch.float64).to(c.device)
            ops.all_gather_flat(dist, allv, v, self.rank, R, self.group)
            allv = allv.cpu()
            all_ok = bool((allv[:, 0] > 0.5).all())
            x_p2p, X = float(allv[0, 1]), float(allv[0, 2])
            if not all_ok:
                x_p2p = None
        use_p2p = x_p2p is not None and x_p2p < 0.97 * X
        if self.xchg_mode == "auto":
            self.xchg_mode = "p2p" if use_p2p else "collective"   # the same inputs on every rank: the same choice
        if self.xchg_mode == "p2p" and x_p2p is not None:
            X = x_p2p
        phi.rand.seeds.copy_(keep)
        phi.count_calls = calls
        # T (rho + (1 - rho) / R) = X (1 - rho)
        rho = (X - T / R) / (T - T / R + X) if X > T / R else 0.0
        rho = min(max(rho, 0.0), 1.0)

        def cost(r):
            return max(T * (r + (1.0 - r) / R), X * (1.0 - r))
        if cost(1.0) <= cost(rho) * 1.02:  # links so slow that exchanging anything loses: every rank computes all groups
            rho = 1.0
        t = torch.tensor([rho], dtype=torch.float64)
        t = t.to(c.device)
        ops.wait_work(ops.broadcast_async(dist, t, 0, self.group))
        rho_f = float(t.item())
        self.calibration = {"phi_ms": T, "xchg_ms": X, "rho": rho_f, "exchange": self.xchg_mode,
                            "xchg_p2p_ms": x_p2p,
                            # what the split is expected to buy on update_phi alone: full launch / max(compute, exchange)
                            "predicted_phi_ms": cost(rho_f), "predicted_phi_speedup": T / cost(rho_f) if cost(rho_f) > 0 else None,
                            "pure_sharding_phi_speedup": T / cost(0.0) if cost(0.0) > 0 else None}
        self._set_split(int(float(t.item()) * MAX_GROUPS))
        ops.synchronize()

    # ---- step trace of the multi-GPU schedule (bench.py's N > 1 record; None = off, the default).  A list here
    # collects, for every iteration, timing marks on the main stream at the schedule's joints; shard_report() turns
    # them into per-step phi / exchange / overlap / update_pi / gradient-exchange times once the device has caught up.
    shard_trace = None

    def _mark(self, rec, name, **kw):
        if rec is not None:
            rec["marks"].append((name, self.ops.timing_mark(), kw))

    def shard_report(self):
        """Means over the traced NON-LINK steps (a mini-batch of more than half the configured size), in ms: what the
        first multi-GPU run needs to be read -- where a step's time went next to what the calibration predicted."""
        ops, tr = self.ops, self.shard_trace or []
        ops.synchronize()
        big = [r for r in tr if r["n_nodes"] > self.cfg.mini_batch_size // 2]
        if not big:
            return {"steps": 0}
        K = self.cfg.K
        acc, chunks = {}, {}

        def add(k, v):
            acc.setdefault(k, []).append(v)
        for r in big:
            m = r["marks"]
            t = {name: ev for name, ev, _ in m if not name.startswith(("phi_done", "xchg_done"))}
            el = lambda a, b: ops.mark_elapsed_ms(t[a], t[b])  # noqa: E731
            add("step_ms", el("begin", "end"))
            add("phi_phase_ms", el("begin", "phi_end"))
            add("update_pi_ms", el("phi_end", "pi_end"))
            add("grads_local_ms", el("pi_end", "grads_local"))
            add("grad_allgather_ms", el("grads_local", "grads_reduced"))
            add("update_theta_ms", el("grads_reduced", "end"))
            # own blocks: launched back to back on the main stream; the exchange of chunk c cannot start before its
            # block is computed and is known to be over when the main stream's wait for it returns
            pd = [(kw["chunk"], ev) for name, ev, kw in m if name.startswith("phi_done")]
            xd = [(kw["chunk"], ev, kw["bytes"]) for name, ev, kw in m if name.startswith("xchg_done")]
            if pd:
                add("phi_local_ms", ops.mark_elapsed_ms(t["begin"], pd[-1][1]))
            if "rep_done" in t:
                add("phi_replicated_ms", ops.mark_elapsed_ms(pd[-1][1] if pd else t["begin"], t["rep_done"]))
            exposed_from = t.get("rep_done", pd[-1][1] if pd else t["begin"])
            if xd:
                add("exchange_exposed_ms", max(0.0, ops.mark_elapsed_ms(exposed_from, xd[-1][1])))
                prev = None
                for (c, ev, nbytes) in xd:
                    start = dict(pd).get(c, t["begin"])
                    d0 = ops.mark_elapsed_ms(start, ev)
                    if prev is not None:
                        d0 = min(d0, ops.mark_elapsed_ms(prev, ev))
                    d0 = max(d0, 1e-6)
                    chunks.setdefault(c, {"ms": [], "bytes": nbytes})["ms"].append(d0)
                    prev = ev
        out = {k: float(np.mean(v)) for k, v in acc.items()}
        out["steps"] = len(big)
        out["gradient"] = ("replicated: every rank computes the whole gradient%s -- no collective; grads_local_ms is that "
                           "launch, grad_allgather_ms ~ 0" % (", update_pi folded into its launch (update_pi_ms ~ 0)"
                                                               if self.grads_fused else "")
                           if self.grads_mode == "replicated" else
                           "sharded: edge slices, all-gather of the R partial sums, added in rank order")
        per_chunk = []
        tot_ms = 0.0
        for c in sorted(chunks):
            ms = float(np.mean(chunks[c]["ms"]))
            tot_ms += ms
            per_chunk.append({"chunk": c, "exchange_ms": ms, "received_bytes": int(chunks[c]["bytes"]),
                              "GBps": chunks[c]["bytes"] / (ms * 1e-3) / 1e9})
        out["exchange_chunks"] = per_chunk
        out["exchange_ms"] = tot_ms
        if per_chunk:
            out["allgather_GBps_received"] = sum(c["received_bytes"] for c in per_chunk) / (tot_ms * 1e-3) / 1e9
            out["overlap_ms"] = max(0.0, tot_ms - out.get("exchange_exposed_ms", 0.0))
        out["how"] = ("timing marks on the main stream at the joints of the sharded step (after each own block's launch, "
                      "after the wait for each chunk's exchange, after the replicated groups, update_pi, the local "
                      "gradient, its rank-ordered reduction, the theta step); exchange_ms of a chunk = from its block's "
                      "end (or the previous chunk's exchange) to the end of its wait: an upper bound; overlap_ms = "
                      "exchange_ms - what was still exposed after the last compute of the phase")
        return out

    def _phi_sharded(self, s, n_nodes, rec=None):
        """update_phi over this rank's groups with the phi_vec exchange overlapped: own block of chunk 0,
        all-gather it (async), own block of chunk 1, ..., then the replicated groups while the last
        exchanges are still in flight."""
        ops, phi = self.ops, self.phiUpdater
        nodes, nbrs = s.dev_nodes, s.neighbor_sampler.GetData()
        if not self.sharded:
            phi.update_phi(nodes, nbrs, n_nodes)
            return
        dist = self._dist()
        pv = phi.phi_vec
        R, r, Cc, g0 = self.world, self.rank, self.cc, self.g_rep
        G = min(n_nodes, MAX_GROUPS)
        tail = n_nodes - G          # nodes i >= G are the second node of groups 0 .. tail-1; their rows are G + t
        rep_hi = min(g0, G)         # replicated groups live in this launch: [0, rep_hi)
        rep_tail = min(tail, rep_hi)
        # Replicated groups that own tail rows go first: the last all-gather region can reach past row G, and
        # every sender must already hold the final value of whatever it sends from there.
        if rep_tail > 0:
            phi.update_phi(nodes, nbrs, n_nodes, 0, rep_tail)
        # side stream only when a block is too small to fill the chip by itself (measured, tools/shard_bench.py:
        # 8 ranks 0.487 -> 0.448 ms, 2 ranks 1.25 -> 1.32 ms)
        forked = rep_hi > rep_tail and Cc < 4096
        if forked:  # replicated groups: side stream, ordered after everything queued on this one so far
            ops.record_event(self.ev_fork)
            with ops.stream(self.rep_stream):
                ops.wait_event(self.ev_fork)
                phi.update_phi(nodes, nbrs, n_nodes, rep_tail, rep_hi)
                ops.record_event(self.ev_join)
        works = []
        if G > g0:
            live_chunks = (G - g0 + R * Cc - 1) // (R * Cc)  # chunks that contain at least one live exchanged group
            for c in range(live_chunks):
                base = g0 + c * R * Cc
                lo = base + r * Cc
                hi = min(lo + Cc, G)
                if lo < hi:
                    phi.update_phi(nodes, nbrs, n_nodes, lo, hi)
                self._mark(rec, "phi_done%d" % c, chunk=c)
                if c == live_chunks - 1 and tail > g0:
                    # exchanged groups with a tail row: park each owner's rows before the region is overwritten
                    for b0 in range(g0, tail, Cc):
                        if ((b0 - g0) // Cc) % R == r:
                            b1 = min(b0 + Cc, tail)
                            self.tail_buf[b0:b1].copy_(pv[G + b0:G + b1])
                row_bytes = 4 * pv.shape[1]
                if G - base <= Cc:  # only rank 0's block is live in this chunk: a broadcast is enough
                    works.append((c, (G - base) * row_bytes if r != 0 else 0, ops.broadcast_async(dist, pv[base:G], 0, self.group)))
                else:
                    works.append((c, (R - 1) * Cc * row_bytes,
                                  ops.all_gather_rows_async(dist, pv[base:base + R * Cc], Cc, r, R, self.group,
                                                            mode="p2p" if self.xchg_mode == "p2p" else "collective")))
        if rep_hi > rep_tail and not forked:
            phi.update_phi(nodes, nbrs, n_nodes, rep_tail, rep_hi)  # overlaps the exchanges in flight
            self._mark(rec, "rep_done")
        for c, nbytes, w in works:
            ops.wait_work(w)
            self._mark(rec, "xchg_done%d" % c, chunk=c, bytes=nbytes)
        if forked:
            ops.wait_event(self.ev_join)
            self._mark(rec, "rep_joined")  # (side stream: the join comes after the exchange waits, it is not a compute joint)
        if tail > g0:
            for b0 in range(g0, tail, Cc):  # owners hand out their parked tail rows
                b1 = min(b0 + Cc, tail)
                ops.wait_work(ops.broadcast_async(dist, self.tail_buf[b0:b1], ((b0 - g0) // Cc) % R, self.group))
            pv[G + g0:G + tail].copy_(self.tail_buf[g0:tail])

    def _edge_range(self, n_edges):
        per = (n_edges + self.world - 1) // self.world
        return min(self.rank * per, n_edges), min((self.rank + 1) * per, n_edges)

    def _reduce_grads(self, local):
        if not self.sharded:
            return local
        dist = self._dist()
        self.ops.all_gather_flat(dist, self.all_grads, local, self.rank, self.world, self.group)
        # one reduction kernel over the R rows in rank order: a fixed association, identical on every rank
        if hasattr(self.ops, "sum_rows"):
            return self.ops.sum_rows(self.ctx, self.all_grads, self.grads_sum)
        return torch.sum(self.all_grads, dim=0)  # CPU stand-in of the tests

    # ------------------------------------------------------------------ the loop (learner.cc:214-250)

    def Run(self, max_iters, signaled=None):
        with self.ops.pin_current_stream():
            if self.loop is not None:
                self._run_graph(max_iters, signaled)
            else:
                self._run(max_iters, signaled)

    GRAPH_CHUNK = 512  # iterations enqueued between two looks at `signaled`

    def _run_graph(self, max_iters, signaled=None):
        """The same iterations as _run, enqueued as captured graphs (ammsb_loop): the host only chooses the
        mini-batches (link?, u) and hands them over; sizes, eps_t and weights travel in device descriptors."""
        ops, cfg, smp = self.ops, self.cfg, self.dev_sampler
        t1 = time.perf_counter()
        phi, beta = self.phiUpdater, self.betaUpdater
        if phi.count_calls != beta.count_calls:
            raise AmmsbError("graph_launch: phi and beta step counters differ (%d, %d)" % (phi.count_calls, beta.count_calls))
        if self.futures[self.phase] is None:
            self._launch_sample(self.phase)   # the first mini-batch is sampled eagerly, as in _run
        done = 0
        while done < max_iters and not (signaled is not None and signaled()):
            n = min(self.GRAPH_CHUNK, max_iters - done)
            self.futures[self.phase].result()
            s = self.samples[self.phase]
            ops.wait_event(s.ready)
            for o in self.samples:
                if o.consumed_valid:
                    ops.wait_event(o.consumed)
            if s.choice is None:
                raise AmmsbError("graph_launch: the pending mini-batch was not drawn by the device sampler")
            nxt = smp.choose_many(cfg.strategy, n)
            self.loop.run(s.choice, nxt, phi.count_calls + 1, self.phase)
            self._loop_dirty = True
            # mini-batch edges of the n steps just enqueued: the pending choice, then all but the last new one
            m = cfg.mini_batch_size
            ne = np.where(nxt["link"][:-1] != 0, nxt["n"][:-1], m).astype(np.int64)
            first_ne = smp.sizes(s.choice)[0]
            self.edges_done += int(first_ne + ne.sum())
            if self.step_log is not None:
                # one array of n_edges per call (n_nodes = n_edges + 1 on this path): per-step Python objects here cost
                # more host time than the launches of a small step
                self.step_log.append(np.concatenate(([first_ne], ne)).astype(np.int64))
            phi.count_calls += n
            beta.count_calls += n
            self.stepCount += n
            self.phase ^= n & 1
            # the new pending mini-batch sits in samples[phase]; everything queued so far orders the eager path
            p = self.samples[self.phase]
            p.choice = tuple(int(x) for x in nxt[-1])
            p.n_edges, p.n_nodes, w = smp.sizes(p.choice)
            f = concurrent.futures.Future()
            f.set_result(w)
            self.futures[self.phase] = f
            self.futures[1 - self.phase] = None
            ops.record_event(p.ready)
            for o in self.samples:
                ops.record_event(o.consumed)
                o.consumed_valid = True
            smp.mark_used()
            done += n
            # every 64 chunks the loop's record of what it has enqueued is released and a fallback is noticed
            # (ammsb_loop_check synchronises): a long Run() does not grow that record without bound
            self._chunks_since_check = getattr(self, "_chunks_since_check", 0) + 1
            if self._chunks_since_check >= 64:
                self._chunks_since_check = 0
                self._check_loop()
        self.time += time.perf_counter() - t1

    def _run(self, max_iters, signaled=None):
        ops, cfg = self.ops, self.cfg
        t1 = time.perf_counter()
        nsamples = len(self.samples)
        if nsamples == 2 and self.stepCount == 1 and self.futures[self.phase] is None:
            self._launch_sample(self.phase)
        it = 0
        while it < max_iters and not (signaled is not None and signaled()):
            ts0 = time.perf_counter()
            if nsamples == 2:
                weight = self.futures[self.phase].result()
                self._launch_sample(1 - self.phase)
            else:
                self._launch_sample(self.phase)
                weight = self.futures[self.phase].result()
            self.samplingTime += time.perf_counter() - ts0
            s = self.samples[self.phase]
            ops.wait_event(s.ready)  # main stream waits for the sample's stream
            n_nodes, n_edges = s.n_nodes, s.n_edges

            # phiUpdater_(nodes, neighbors, n)  -- phi.cc:728-763
            phi = self.phiUpdater
            phi.count_calls += 1
            if n_nodes == 0:
                raise AmmsbError("mini-batch nodes size = 0!")
            rec = None
            if self.shard_trace is not None and self.sharded:
                rec = {"n_nodes": n_nodes, "n_edges": n_edges, "marks": []}
                self.shard_trace.append(rec)
            self._mark(rec, "begin")
            self._phi_sharded(s, n_nodes, rec)
            self._mark(rec, "phi_end")
            beta = self.betaUpdater
            # multi-GPU: the gradient is cut over the ranks only in "sharded" mode and only for mini-batches worth a
            # collective; otherwise every rank computes all of it -- with update_pi folded into the same launch where
            # the shape and the mini-batch (edge t = (nodes[0], nodes[t + 1])) allow
            shard = (self.sharded and self.grads_mode == "sharded"
                     and n_edges > int(getattr(cfg, "beta_shard_min_edges", 0)))
            fuse = self.sharded and not shard and self.grads_fused and n_nodes == n_edges + 1
            if not fuse:
                phi.update_pi(s.dev_nodes, n_nodes)
            self._mark(rec, "pi_end")

            # betaUpdater_(edges, n, weight)  -- beta.cc:334-384
            beta.count_calls += 1
            if shard:
                e_lo, e_hi = self._edge_range(n_edges)
                local = beta.calculate_grads(s.dev_edges, n_edges, e_lo, e_hi)
                self._mark(rec, "grads_local")
                total = self._reduce_grads(local)
            elif fuse:
                total = beta.update_pi_and_grads(phi, s.dev_nodes, s.dev_edges, n_edges)
                self._mark(rec, "grads_local")
            else:
                total = beta.calculate_grads(s.dev_edges, n_edges, 0, n_edges)
                self._mark(rec, "grads_local")
            self._mark(rec, "grads_reduced")
            beta.update_theta(weight, total)
            self._mark(rec, "end")

            ops.record_event(s.consumed)
            s.consumed_valid = True
            self.edges_done += n_edges
            if self.step_log is not None:
                self.step_log.append((n_edges, n_nodes))
            if nsamples == 2:
                self.phase = 1 - self.phase
            it += 1
            self.stepCount += 1
        self.time += time.perf_counter() - t1

    def TrainingPerplexity(self):
        """learner.cc:204-212 (Config.calc_train_ppx)."""
        if self.trainingPerplexity is None:
            raise AmmsbError("TrainingPerplexity() needs Config.calc_train_ppx")
        return self._perplexity(self.trainingPerplexity)

    def HeldoutPerplexity(self):
        return self._perplexity(self.heldoutPerplexity)

    def _check_loop(self):
        """Before anything reads the model state after descriptor-loop runs: had a device-side wait given up, the
        steps behind it were skipped, and ammsb_loop_check re-runs them on the event hand-over (ops.GraphLoop.check);
        a checkpoint or a perplexity taken before that would be of a state some iterations short."""
        if self.loop is not None and self._loop_dirty:
            self.loop.check()
            self._loop_dirty = False

    @property
    def loop_fallbacks(self):
        """runs of the descriptor loop that had to be finished on the stream-event hand-over (0 in a healthy run)"""
        return self.loop.status()[1] if self.loop is not None else 0

    def _perplexity(self, calc):
        self._check_loop()
        t1 = time.perf_counter()
        calc.count_calls += 1
        H = calc.num_edges
        per = (H + self.world - 1) // self.world
        lo, hi = min(self.rank * per, H), min((self.rank + 1) * per, H)
        if self.sharded:
            sums = calc.partial(lo, hi)
            dist = self._dist()
            self.ops.all_gather_flat(dist, self.all_sums, sums, self.rank, self.world, self.group)
            parts = [calc.unpack(self.all_sums[r]) for r in range(self.world)]
            tot = [sum(p[i] for p in parts) for i in range(4)]
        else:
            tot = calc.partial_host(lo, hi) if hasattr(calc, "partial_host") else calc.unpack(calc.partial(lo, hi))
        ppx = float(np.exp(np.float32(calc.value(*tot))))  # learner.cc:202 std::exp(ppx)
        self.time += time.perf_counter() - t1
        return ppx

    # ------------------------------------------------------------------ checkpoint (learner.cc:301-363)

    def _sample_records(self, out, s):
        """Sample::Serialize (sample.h:62-75) + NeighborSampler::Serialize (sample.h:30-32)."""
        from . import checkpoint as ck
        ck.write_message(out, [(1, ck.BYTES, np.ascontiguousarray(s.edges, dtype=np.uint64).tobytes()),
                               (2, ck.BYTES, np.ascontiguousarray(s.nodes_vec, dtype=np.uint32).tobytes()),
                               (3, ck.VARINT, s.seed)])
        ck.write_buffer(out, s.dev_edges)
        ck.write_buffer(out, s.dev_nodes)
        ck.write_buffer(out, s.neighbor_sampler.rand.seeds)
        ck.write_buffer(out, s.neighbor_sampler.GetData())

    def _sample_parse(self, inp, s):
        from . import checkpoint as ck
        m = ck.read_message(inp, (1, 2, 3))
        ck.read_buffer(inp, s.dev_edges)
        ck.read_buffer(inp, s.dev_nodes)
        ck.read_buffer(inp, s.neighbor_sampler.rand.seeds)
        ck.read_buffer(inp, s.neighbor_sampler.GetData())
        s.edges = np.frombuffer(m[1], dtype=np.uint64).copy()
        s.nodes_vec = np.frombuffer(m[2], dtype=np.uint32).copy()
        s.seed = int(m[3])
        s.n_edges, s.n_nodes = s.edges.size, s.nodes_vec.size
        s.consumed_valid = False

    def _theta_sum(self):
        t = self.theta.reshape(-1, 2)
        return (t[:, 0] + t[:, 1]).contiguous()  # BetaUpdater::GetThetaSum(), beta.cc:20-28

    def _gather_sharded_state(self):
        """Multi-rank runs advance two pieces of state only at their owner: the phi streams of the exchanged group
        blocks and the running-mean perplexity of each rank's held-out slice.  Before a checkpoint every owner
        hands its part to the others (a collective: every rank must call Serialize), after which the state is
        complete and identical everywhere and any rank's file is the checkpoint."""
        if self.world <= 1:
            return
        ops, dist = self.ops, self._dist()
        self.drain()
        R, Cc, g0, L = self.world, self.cc, self.g_rep, self.cfg.phi_wg_size
        seeds = self.phiUpdater.rand.seeds  # [streams, 2] int64, stream = group * L + lane
        for b in range(R * self.nch):
            lo, hi = g0 + b * Cc, min(g0 + (b + 1) * Cc, MAX_GROUPS)
            if lo >= hi or lo * L >= seeds.shape[0]:
                break
            rows = seeds[lo * L:min(hi * L, seeds.shape[0])]
            ops.wait_work(ops.broadcast_async(dist, rows, b % R, self.group))
        for calc in ([self.trainingPerplexity] if self.trainingPerplexity is not None else []) + [self.heldoutPerplexity]:
            H = calc.num_edges
            per = (H + R - 1) // R
            for r in range(R):
                lo, hi = min(r * per, H), min((r + 1) * per, H)
                if lo < hi:
                    ops.wait_work(ops.broadcast_async(dist, calc.ppx_per_edge[lo:hi], r, self.group))
        ops.synchronize()

    def Serialize(self, out):
        """Write the learner state to the binary stream `out` in the reference's record order
        (learner.cc:316-329): beta, theta, pi, phi, PhiUpdater, BetaUpdater, held-out perplexity,
        LearnerProperties, Sample[0], Sample[1].  With device sampling a trailing extension record
        (ignored by the reference's Parse, which stops after the samples) carries the device sampler.
        Every rank holds the same replicated state; rank 0's file is the checkpoint."""
        from . import checkpoint as ck
        self._gather_sharded_state()
        two = len(self.samples) == 2
        if two and self.futures[self.phase] is None:
            self._launch_sample(self.phase)  # the reference's constructor has it in flight already
        weight = float(self.futures[self.phase].result()) if two else 0.0  # learner.cc:307-314
        self._check_loop()
        self.ops.synchronize()
        if self.dev_sampler is not None:
            self.dev_sampler.check()
        ck.write_buffer(out, self.beta)
        ck.write_buffer(out, self.theta)
        ck.write_rpm(out, self.pi)
        ck.write_buffer(out, self.phi)
        # PhiUpdater::Serialize (phi.cc:765-771)
        ck.write_buffer(out, self.phiUpdater.rand.seeds)
        ck.write_message(out, [(1, ck.VARINT, self.phiUpdater.count_calls), (2, ck.FIXED64, 0.0), (3, ck.FIXED64, 0.0)])
        # BetaUpdater::Serialize (beta.cc:386-397)
        ck.write_buffer(out, self.betaUpdater.rand.seeds)
        ck.write_buffer(out, self._theta_sum())  # derived: recomputed by every beta_grads launch
        ck.write_message(out, [(1, ck.VARINT, self.betaUpdater.count_calls)] + [(i, ck.FIXED64, 0.0) for i in range(2, 7)])
        # PerplexityCalculatorBase::Serialize (perplexity.cc:276-283); the training calculator first (learner.cc:321-324)
        for calc in ([self.trainingPerplexity] if self.trainingPerplexity is not None else []) + [self.heldoutPerplexity]:
            ck.write_message(out, [(1, ck.VARINT, calc.count_calls), (2, ck.FIXED64, 0.0), (3, ck.FIXED64, 0.0)])
            ck.write_buffer(out, calc.ppx_per_edge)
        ck.write_message(out, [(1, ck.VARINT, self.stepCount), (2, ck.VARINT, int(self.time * 1e9)),
                               (3, ck.VARINT, int(self.samplingTime * 1e9)), (4, ck.VARINT, self.phase),
                               (5, ck.FIXED64, weight)])
        for s in self.samples:
            self._sample_records(out, s)
        if self.dev_sampler is not None:
            import json
            st = dict(self.dev_sampler.state(), sizes=[[s.n_edges, s.n_nodes] for s in self.samples],
                      choices=[list(s.choice) if s.choice is not None else None for s in self.samples],
                      edges_done=self.edges_done)
            ck.write_message(out, [(1, ck.BYTES, b"AMMSB-DEVSAMPLER-1"), (2, ck.BYTES, json.dumps(st).encode())])
            ck.write_buffer(out, self.dev_sampler.rand.seeds)
        return True

    def Parse(self, inp):
        """Restore from a stream written by Serialize (or by the reference, for the same Config and
        buffer sizes).  Size mismatches raise CheckpointError (the reference returns false)."""
        from . import checkpoint as ck
        for f in self.futures:
            if f is not None:
                f.result()
        self.ops.synchronize()
        ck.read_buffer(inp, self.beta)
        ck.read_buffer(inp, self.theta)
        ck.read_rpm(inp, self.pi)
        ck.read_buffer(inp, self.phi)
        ck.read_buffer(inp, self.phiUpdater.rand.seeds)
        self.phiUpdater.count_calls = int(ck.read_message(inp, (1, 2, 3))[1])
        ck.read_buffer(inp, self.betaUpdater.rand.seeds)
        ck.read_buffer(inp, self._theta_sum())  # size check only
        self.betaUpdater.count_calls = int(ck.read_message(inp, (1, 2, 3, 4, 5, 6))[1])
        for calc in ([self.trainingPerplexity] if self.trainingPerplexity is not None else []) + [self.heldoutPerplexity]:
            calc.count_calls = int(ck.read_message(inp, (1, 2, 3))[1])
            ck.read_buffer(inp, calc.ppx_per_edge)
        props = ck.read_message(inp, (1, 2, 3, 4, 5))
        self.stepCount = int(props[1])
        self.time, self.samplingTime = props[2] * 1e-9, props[3] * 1e-9
        self.phase = int(props[4]) & 1
        for s in self.samples:
            self._sample_parse(inp, s)
        if self.dev_sampler is not None:
            import json
            m = ck.read_message(inp, (1, 2))
            if m[1] != b"AMMSB-DEVSAMPLER-1":
                raise ck.CheckpointError("checkpoint has no device-sampler record")
            st = json.loads(m[2].decode())
            self.dev_sampler.load_state(st)
            for s, (ne, nv), ch in zip(self.samples, st["sizes"], st.get("choices", [None, None])):
                s.n_edges, s.n_nodes = int(ne), int(nv)
                s.choice = tuple(int(x) for x in ch) if ch is not None else None
            self.edges_done = int(st["edges_done"])
            ck.read_buffer(inp, self.dev_sampler.rand.seeds)
        self.futures = [None, None]
        if len(self.samples) == 2:  # learner.cc:348-353
            f = concurrent.futures.Future()
            f.set_result(float(np.float32(props[5])))
            self.futures[self.phase] = f
        self.ops.synchronize()
        return True

    def drain(self):
        """Wait for the pending background sample and all device work (used before timing / teardown)."""
        for f in self.futures:
            if f is not None:
                f.result()
        self.ops.synchronize()
        if self.loop is not None:
            self.loop.check()
            self._loop_dirty = False
        if self.dev_sampler is not None:
            self.dev_sampler.check()  # a mini-batch that came up short is an error, never a silent duplicate

    def PrintStats(self, out=print):
        out("TOTAL    : %.6f" % self.time)
        out("SAMPLING : %.6f (%%%.2f)" % (self.samplingTime, 100 * self.samplingTime / max(self.time, 1e-12)))
        out("STEPS    : %d, mini-batch edges %d" % (self.stepCount - 1, self.edges_done))

    def close(self):
        self.drain()
        if self.loop is not None:
            self.loop.close()
            self.loop = None
        self._loop_dirty = False
        if self.pool is not None:
            self.pool.shutdown(wait=True)
