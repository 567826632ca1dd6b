#!/usr/bin/env python3
"""Headline benchmark: mini-batch edges/s of the SG-MCMC a-MMSB learner loop (+ perplexity-eval ms).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts the N ranks itself
                                                            as a child `python -m torch.distributed.run ...`)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], "C3"): synthetic a-MMSB graph N = 1M vertices, average degree 32,
K = 1024 communities, mini-batch m = 65536, n = 32 neighbour samples, held-out ratio 0.01, strategy
Node (a fair coin picks a link batch = all edges of one vertex, or a non-link batch = m non-links of
one vertex).  A step is one Learner iteration: mini-batch + neighbour sampling, update_phi, update_pi,
the beta gradient and the theta/beta update.  With N > 1 the same problem is sharded (strong scaling).

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

WORKLOADS = {
    # name: (N, K, m, n, avg_degree, K_true)
    "C1": (10_000, 32, 1024, 32, 32, 32),
    "C2": (100_000, 256, 8192, 32, 32, 64),
    "C3": (1_000_000, 1024, 65536, 32, 32, 64),
    # K = 4096 shapes (not the default; C5 takes minutes of host time to build its 320 M-edge graph and cuckoo sets)
    "C5s": (2_000_000, 4096, 65536, 32, 32, 64),   # 8.2e9 pi elements: beyond the reference's 32-bit row offsets
    "C5": (10_000_000, 4096, 65536, 32, 64, 64),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--exchange", default=os.environ.get("AMMSB_EXCHANGE", "auto"), choices=["auto", "collective", "p2p"],
                    help="multi-GPU phi_vec exchange: RCCL all-gather, direct peer sends (one per link), or time both")
    ap.add_argument("--phi-wg", type=int, default=0)
    ap.add_argument("--beta-wg", type=int, default=0)
    ap.add_argument("--ppx-wg", type=int, default=0)
    ap.add_argument("--host-sampling", action="store_true",
                    help="reference-exact host mini-batch sampler (rand_r) instead of the device sampler")
    ap.add_argument("--loop", default="auto", choices=["auto", "graph", "eager"],
                    help="iteration form: captured hipGraphs (auto: one rank + device sampling) or launch by launch")
    ap.add_argument("--cpp-dropin", type=int, default=1,
                    help="also time the C++ mcmc::Learner (ammsb_main, a child process) on the same graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="nodes/edges in the CPU-baseline sample")
    ap.add_argument("--ppx-calls", type=int, default=5)
    ap.add_argument("--extras", type=int, default=1,
                    help="one GPU: also time the workload at the reference's default work-group sizes (32) and, for C3, "
                         "short C1 and C2 runs (small_configs) and BASELINE's largest configuration C5 (large_configs)")
    ap.add_argument("--large", default=os.environ.get("AMMSB_BENCH_LARGE", "auto"), choices=["auto", "C5", "C5s", "none"],
                    help="large_configs of the default run: C5 (N=10M, K=4096: ~2 min of host set-up, 164 GB of pi), its "
                         "2M-vertex stand-in C5s, none; auto = C5 when HBM and host memory allow, else C5s")
    ap.add_argument("--watchdog-s", type=float, default=-1.0,
                    help="give up (exit 124, naming the phase and the rank on stderr and, from rank 0, as a JSON line) when "
                         "one phase of the run makes no progress for this long; -1: 300 s with more than one rank (a "
                         "collective one rank never joins blocks its peers for RCCL's own 10 minutes and says nothing "
                         "about where), off with one; 0: off")
    ap.add_argument("--sustained-s", type=float, default=3.0,
                    help="after the timed window: this many seconds of untimed iterations, then a SECOND window of the "
                         "same --steps, reported as `sustained` (what a long run sees once the package's power "
                         "controller has settled; `value` stays the contract's warm-up + steps window); 0: off")
    ap.add_argument("--settle-s", type=float, default=-8.0,
                    help="untimed iterations BEFORE the warm-up steps.  > 0: that many seconds.  < 0 (default -8): until "
                         "update_phi's launch time and the reported package power have stopped moving, at least 1 s, at most "
                         "that many seconds -- the clocks' ramp from idle and, on some boxes, several seconds in which the "
                         "power controller holds a higher shader clock and the launch runs 8 %% longer are start-up, not "
                         "the measurement (profiles/README.md, round 4).  0: off")
    return ap.parse_args()


def pick_wg(K, override, cap):
    if override:
        return override
    wg = 64
    while K // wg > cap and wg < 1024:  # keep <= cap columns per lane
        wg *= 2
    return wg


class Watchdog:
    """phase(name) says what the run is doing now; a phase that lasts longer than limit_s ends the process with the
    phase's name.  For the first runs on more than one GPU: a rank stuck in a collective is otherwise silent."""

    def __init__(self, limit_s, rank, world):
        import threading
        self.limit, self.rank, self.world = float(limit_s), rank, world
        self.name, self.since, self.lock = "start", time.monotonic(), threading.Lock()
        self.history = []
        if self.limit > 0:
            threading.Thread(target=self._watch, daemon=True).start()

    def phase(self, name):
        with self.lock:
            now = time.monotonic()
            self.history.append((self.name, round(now - self.since, 2)))
            self.name, self.since = name, now

    def _watch(self):
        while True:
            time.sleep(1.0)
            with self.lock:
                name, waited, hist = self.name, time.monotonic() - self.since, list(self.history[-8:])
            if name == "done":
                return
            if waited > self.limit:
                rec = {"error": "watchdog", "phase": name, "stuck_for_s": round(waited, 1), "rank": self.rank,
                       "world": self.world, "phases_before": hist}
                print("[bench] rank %d gave up: %s" % (self.rank, json.dumps(rec)), file=sys.stderr, flush=True)
                if self.rank == 0:
                    print(json.dumps(rec), flush=True)
                os._exit(124)


def loop_form(graphs):
    """Which form of ammsb_loop the library takes from the environment (csrc/ammsb_loop.hip reads the same variables)."""
    if not graphs:
        return "eager (one launch + host bookkeeping per kernel)"
    hs = os.environ.get("AMMSB_LOOP_HANDSHAKE")
    pmc = any(os.environ.get(v, "0") not in ("", "0", "False", "false")
              for v in ("ROCPROF_COUNTER_COLLECTION", "AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING", "CUDA_LAUNCH_BLOCKING"))
    if hs == "event" or (hs is None and pmc):
        return "device-descriptor loop (ammsb_loop): captured hipGraphs replayed from one host thread, stream-event hand-over"
    launch = {"graph": "captured hipGraphs replayed from two host threads",
              "serial": "captured hipGraphs replayed from one host thread"}.get(
                  os.environ.get("AMMSB_LOOP_LAUNCH", ""), "chains launched directly from two host threads")
    return "device-descriptor loop (ammsb_loop): %s, device-side hand-over" % launch


def host_cpu_info():
    """What the CPU baseline ran on: model name, logical CPUs visible, CPUs this process may use (affinity and
    cgroup quota -- a GPU box hands a container a share of a large host)."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    usable = affinity if quota is None else max(1, min(affinity, int(quota + 0.5)))
    return {"model": model, "nproc": os.cpu_count() or 1, "affinity": affinity, "cgroup_cpu_quota": quota,
            "usable": usable}


def cpu_baseline(args, lrn, cfg, ds, n_nodes_big):
    """The reference's CPU path (per-thread kernels, learner.cc:105-114) restated in oracle/, timed on
    the host cores of this box on a bounded slice of one non-link mini-batch: once on every CPU this process
    may use (OpenMP over nodes/edges), once on one thread."""
    import ctypes as C
    import oracle_lib as orc
    native = os.path.join(ROOT, "oracle", "libammsb_oracle_native.so")
    try:
        orc.build(native, "-march=native")
        lib_path = native
    except Exception:
        lib_path = None
    L = orc.lib(lib_path) if lib_path else orc.lib()
    info = host_cpu_info()
    cores = info["usable"]
    K, n = cfg.K, cfg.num_node_sample
    rng = np.random.default_rng(1)
    p = orc.make_params(cfg.N, K, n, alpha=np.float32(cfg.alpha))
    pi_h = lrn.pi.host()
    phi_h = lrn.phi.cpu().numpy().copy()
    beta_h = lrn.beta.cpu().numpy().copy()
    theta_h = lrn.theta.cpu().numpy().copy()
    tset = ds.training
    slots, bins, pidx = tset.Serialize(), tset.BinsPerBucket(), tset.PrimeIdx()

    def run(s, threads):
        """s nodes + s edges of a non-link mini-batch on `threads` OpenMP threads; returns seconds."""
        L.orc_set_num_threads(threads)
        u = int(rng.integers(0, cfg.N))
        vs = rng.permutation(cfg.N)[:s].astype(np.uint32)
        nbrs = rng.integers(0, cfg.N, size=(s, n), dtype=np.uint32)
        edges = orc.make_edge(np.full(s, u, dtype=np.uint64), vs.astype(np.uint64))
        seeds = orc.rng_init(s, 42, 43)
        phi_vec = np.zeros((s, K), dtype=np.float32)
        grads = np.zeros(2 * K, dtype=np.float32)
        theta_sum = np.zeros(K, dtype=np.float32)
        th = theta_h.copy()
        t0 = time.perf_counter()
        L.orc_update_phi(C.byref(p), beta_h, pi_h.reshape(-1), phi_h, slots, bins, pidx, vs, nbrs.reshape(-1), s, 1,
                         seeds, 32, 0, 1, phi_vec.reshape(-1))
        L.orc_update_pi(C.byref(p), pi_h.reshape(-1), phi_h, phi_vec.reshape(-1), vs, s, 32, 0)
        L.orc_sum_theta(th, theta_sum, K)
        L.orc_beta_grads(C.byref(p), th, theta_sum, beta_h, pi_h.reshape(-1), slots, bins, pidx, edges, s, 32, 0, 0, grads)
        L.orc_update_theta(C.byref(p), th, grads, 1, np.float32(1.0), orc.rng_init(K, 44, 45), 1)
        return time.perf_counter() - t0
    # calibrate on a small slice (one thread), then size both legs to ~8 s and ~4 s of wall time
    s0 = max(16, min(256, n_nodes_big - 1))
    per_item = run(s0, 1) / s0                       # seconds per (node, edge) pair on one thread
    s1 = args.cpu_sample or int(min(n_nodes_big - 1, max(64, 4.0 / per_item)))
    t1 = run(s1, 1)
    sN = args.cpu_sample or int(min(n_nodes_big - 1, max(64, 8.0 * cores / per_item)))
    tN = run(sN, cores)
    reps = 1
    while tN < 6.0 and reps < 64 and not args.cpu_sample:  # a whole mini-batch is too short on this box: repeat it
        tN += run(sN, cores)
        reps += 1
    sN_total = sN * reps
    elems = (n + 1) * K  # pi elements read per mini-batch node in update_phi
    return {"value": sN_total / tN, "unit": "mini-batch edges/s", "cores": int(cores), "kind": "port",
            "one_thread_value": s1 / t1, "cpu_model": info["model"], "nproc": info["nproc"],
            "affinity": info["affinity"], "cgroup_cpu_quota": info["cgroup_cpu_quota"],
            "omp": "OMP threads = CPUs usable by this process (min of affinity and cgroup quota), no explicit binding "
                   "(OMP_PROC_BIND unset), schedule(dynamic) over nodes / edges",
            "ns_per_pi_element_one_thread": round(t1 / (s1 * elems) * 1e9, 2),
            "sample": "%d x (%d of the %d nodes and %d of the %d edges of one non-link mini-batch) on %d threads in %.1f s "
                      "(phi+pi+beta, the reference's per-thread CPU kernels restated in oracle/, -O3 -march=native "
                      "-ffp-contract=off); one thread: %d nodes+edges in %.1f s"
                      % (reps, sN, n_nodes_big, sN, cfg.mini_batch_size, cores, tN, s1, t1)}


def cpp_dropin(args, hostlib, N, K, m, n, wg, edges, note):
    """north_star's host: the C++ mcmc::Learner behind the reference's command line (host/main.cc), run as a child
    process on the same graph (gzip data-set file, main.cc:110-124), for each loop form.  Its PrintStats line gives
    the mini-batch edge rate of Learner::Run alone (perplexity evaluations excluded)."""
    import re
    import tempfile
    exe = os.path.join(ROOT, "mcmc-ammsb-gpu_amd", "ammsb_main")
    if not os.path.exists(exe):
        return {"error": "ammsb_main not built"}
    out = {}
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "g.bin.gz")
        hostlib.dump_dataset(f, N, 0.01, edges)
        iters = max(args.steps, 1000)  # ~1 s per form: past the clocks' ramp from idle
        for name, flags in (("graph", ["--async", "1", "--graph", "1"]), ("async", ["--async", "1"]), ("sync", [])):
            cmd = [exe, "--load-data", "1", "--load-file", f, "-k", str(K), "-m", str(m), "-n", str(n), "-x", str(iters),
                   "-i", str(iters), "--phi-wg", str(wg), "--beta-wg", str(wg), "--ppx-wg", str(wg),
                   "--device-sampling", "1"] + flags
            t0 = time.perf_counter()
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            except subprocess.TimeoutExpired:
                out[name] = {"error": "timeout"}
                continue
            text = r.stdout + r.stderr
            mt = re.search(r"MINI-BATCH EDGES: (\d+) \(([0-9.eE+-]+) edges/s\)", text)
            tt = re.search(r"TOTAL\s*: ([0-9.eE+-]+)", text)
            px = re.findall(r"ppx\[\s*(\d+)\]\s*=?\s*([0-9.eE+-]+)", text)
            if r.returncode != 0 or not mt:
                out[name] = {"error": "rc=%d" % r.returncode, "tail": text[-300:]}
                continue
            out[name] = {"edges_per_s": float(mt.group(2)), "mini_batch_edges": int(mt.group(1)), "iterations": iters,
                         "total_s": float(tt.group(1)) if tt else None,
                         "perplexity": float(px[-1][1]) if px else None,
                         "process_s": round(time.perf_counter() - t0, 1)}
            note("cpp drop-in (%s): %.3e edges/s" % (name, out[name]["edges_per_s"]))
    out["command"] = "ammsb_main --load-data 1 --load-file <graph> -k %d -m %d -n %d --device-sampling 1 [--async 1 [--graph 1]]" % (K, m, n)
    return out


def launcher_command(gpus, argv, port=None):
    """The command `python bench.py --gpus N ...` turns itself into when it was started without a launcher:
    one rank per GPU under torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    if port is None:
        import socket
        with socket.socket() as so:  # a free port, so that back-to-back N = 2, 4, 8 runs never collide
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks_if_needed(args, argv):
    """`--gpus N` with N > 1 and no RANK in the environment: start the N ranks as a CHILD process (never an exec:
    nothing here has touched the GPU yet, and nothing will in this parent) and exit with its code.  Under a
    launcher, WORLD_SIZE must equal --gpus; a mismatch is an error, never a silent one-rank run."""
    if "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
        return
    if args.gpus <= 1:
        return
    cmd = launcher_command(args.gpus, argv)
    if os.environ.get("AMMSB_BENCH_SPAWN_DRYRUN"):
        print(json.dumps({"spawn": cmd}))
        raise SystemExit(0)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.call(cmd, env=env))


def phi_bytes_per_node(K, n):
    """algorithmic bytes of update_phi per mini-batch node."""
    return 4 * K * (n + 2) + 68 * n + 8


def measure(args, lrn, cfg, m, steps, warmup, world, dist, torch, workload, ppx_calls, settle_s=0.0):
    """ppx latency, then `warmup` untimed and EXACTLY `steps` timed iterations of `lrn` (barrier + synchronize on both
    sides, max over ranks).  Returns the numbers of one bench record."""
    from mcmc_ammsb_gpu_amd import gpu_state
    K, n = cfg.K, cfg.num_node_sample

    def sync():
        lrn.drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- settle: untimed iterations until the package's power controller has reached its steady state (the same
    # number of steps on every rank: decided by rank 0's clock, in whole chunks)
    phase = getattr(args, "phase", lambda name: None)
    settle_steps, settle_log = 0, None
    if settle_s != 0:
        # settle_s > 0: that many seconds.  settle_s < 0 ("auto"): until the package is in its steady state under this
        # load -- update_phi's launch time (device stamps) and the power the driver reports have both stopped moving
        # (five consecutive chunks within 1 % / 3 %), at least 1 s, at most -settle_s seconds.  On some boxes that is
        # immediate, on others the first seconds of load run 8 % slower at a HIGHER shader clock until the power
        # controller has found its state (profiles/README.md, round 4).
        phase("%s: settle iterations" % workload)
        chunk = 100 if m >= 32768 else (2000 if world == 1 else 200)
        t_s = time.perf_counter()
        hist = []
        dev_i = torch.cuda.current_device()
        while True:
            first_s = lrn.phiUpdater.count_calls + 1
            if settle_s < 0:
                lrn.step_log = []
            lrn.Run(chunk)
            sync()
            settle_steps += chunk
            t_now = time.perf_counter() - t_s
            if settle_s > 0:
                go = t_now < settle_s and settle_steps < 20000
            else:
                phi_ms = None
                if lrn.loop is not None:
                    try:
                        st_s = lrn.loop.step_stamps(first_s, chunk)
                        non = np.concatenate(lrn.step_log) == m
                        if non.any():
                            phi_ms = float((st_s[non, 1] - st_s[non, 0]).mean() * 1e-6)
                    except Exception:
                        phi_ms = None
                lrn.step_log = None
                hist.append((round(t_now, 2), phi_ms, gpu_state.read(dev_i).get("power_w")))
                last = hist[-5:]

                def steady(vals, tol):
                    vals = [v for v in vals if v]
                    return len(vals) < 5 or (max(vals) - min(vals)) <= tol * max(vals)
                stable = len(hist) >= 5 and steady([h[1] for h in last], 0.01) and steady([h[2] for h in last], 0.03)
                go = (t_now < 1.0 or not stable) and t_now < -settle_s and settle_steps < 40000
            if world > 1:
                flag = torch.tensor([1 if go else 0], dtype=torch.int32, device="cuda")
                dist.broadcast(flag, 0)
                go = bool(flag.item())
            if not go:
                break
        if settle_s < 0:
            settle_log = {"seconds": round(time.perf_counter() - t_s, 2),
                          "update_phi_ms_first_last": [next((h[1] for h in hist if h[1]), None), hist[-1][1]],
                          "power_w_first_last": [hist[0][2], hist[-1][2]], "chunks": len(hist)}

    # ---- perplexity latency (mean of ppx_calls, after one untimed call)
    ppx, ppx_ms = None, None
    phase("%s: HeldoutPerplexity() latency" % workload)
    if ppx_calls > 0:
        lrn.HeldoutPerplexity()
        sync()
        t0 = time.perf_counter()
        for _ in range(ppx_calls):
            ppx = lrn.HeldoutPerplexity()
        sync()
        ppx_ms = (time.perf_counter() - t0) * 1e3 / ppx_calls

    phase("%s: warm-up + timed steps" % workload)
    lrn.Run(warmup)
    sync()
    phi = lrn.phiUpdater
    # Device time of the dominant kernel (update_phi) inside the timed region, per launch.  Eager loop: HIP events
    # on the launch stream around each launch.  Descriptor loop: device wall-clock stamps written by the first block
    # of every kernel of the main chain (AMMSB_LOOP_TIMESTAMPS).
    ev = []
    orig = phi.update_phi

    def timed_update_phi(nodes, neighbors, nn, lo=0, hi=0xFFFFFFFF):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(torch.cuda.current_stream())
        orig(nodes, neighbors, nn, lo, hi)
        b.record(torch.cuda.current_stream())
        ev.append((a, b, nn, max(0, min(hi, min(nn, 65535)) - lo)))

    if lrn.loop is None:
        phi.update_phi = timed_update_phi
    lrn.step_log = []  # (n_edges, n_nodes) of every step of the timed region (both loops append)
    first_step = phi.count_calls + 1
    edges_before = lrn.edges_done
    sync()
    state_before = gpu_state.read(torch.cuda.current_device())
    prof = None
    if os.environ.get("AMMSB_BENCH_PROFILE"):  # development aid: cProfile of the enqueue side of the timed region
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    lrn.Run(steps)
    t_enq = time.perf_counter() - t0  # host side done enqueueing (the device may still be running)
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(14)
    sync()
    dt = time.perf_counter() - t0
    state_after = gpu_state.read(torch.cuda.current_device())
    phi.update_phi = orig
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    edges_done = lrn.edges_done - edges_before  # identical on every rank: the whole job's mini-batch edges
    step_log = []
    for rec in lrn.step_log:
        if isinstance(rec, np.ndarray):  # descriptor loop: n_edges of a whole call, n_nodes = n_edges + 1
            step_log.extend(zip(rec.tolist(), (rec + 1).tolist()))
        else:
            step_log.append(rec)
    lrn.step_log = None
    names = lrn.ctx.kernel_names()  # what the library dispatched to, spelled as rocprofv3 spells it

    # per-launch records of update_phi: (seconds, nodes in the mini-batch, nodes this rank's launch processed)
    step_classes, value_per_class, kernels = None, None, None
    if lrn.loop is not None:
        kept = min(steps, 8192)
        st = lrn.loop.step_stamps(first_step + steps - kept, kept)  # [kept, 8] ns
        log = step_log[-kept:]
        launches = [((st[i, 1] - st[i, 0]) * 1e-9, log[i][1], min(log[i][1], 65535)) for i in range(kept)]
        # a step's device time = from its update_phi start to the next step's (the sampling chain of the mini-batch two
        # steps ahead runs beside it); the last step of the window has no successor and is left out
        dur = (st[1:, 0] - st[:-1, 0]) * 1e-6
        non = np.array([log[i][0] == m for i in range(kept - 1)], dtype=bool)
        step_classes = {
            "nonlink": {"steps": int(non.sum()), "ms_per_step": float(dur[non].mean()) if non.any() else None,
                        "edges_per_step": m},
            "link": {"steps": int((~non).sum()), "ms_per_step": float(dur[~non].mean()) if (~non).any() else None,
                     "edges_per_step": float(np.mean([log[i][0] for i in range(kept - 1) if not non[i]])) if (~non).any() else None},
            "source": "device wall-clock stamps at the start of consecutive steps' update_phi",
        }
        sn, sl = step_classes["nonlink"], step_classes["link"]
        if sn["ms_per_step"] and sl["ms_per_step"]:
            # the Node strategy flips a fair coin per step (sample.cc:297): one non-link + one link step is the
            # expected pair, whatever the window's flips were
            value_per_class = {
                "value": (sn["edges_per_step"] + sl["edges_per_step"]) / ((sn["ms_per_step"] + sl["ms_per_step"]) * 1e-3),
                "unit": "edges/s",
                "how": "(m + mean link-batch edges) / (non-link step + link step device time): independent of how many "
                       "of the window's coin flips came up non-link"}
        # per-kernel device time of the NON-LINK steps (stamp k+1 - stamp k: the kernel plus one boundary) against the
        # algorithmic bytes of SURVEY 8(d)
        nl = np.array([log[i][0] == m for i in range(kept)], dtype=bool)
        ok = nl & (st[:, 0] > 0) & (st[:, 5] >= st[:, 4]) & (st[:, 4] >= st[:, 3]) & (st[:, 3] >= st[:, 2]) & \
            (st[:, 2] >= st[:, 1]) & (st[:, 1] >= st[:, 0])
        if ok.any():
            d = lambda a, b: float((st[ok, b] - st[ok, a]).mean()) * 1e-9  # noqa: E731  (seconds)
            # update_pi folded into the gradient kernel: FUSE is the third template argument of both gradient kernels
            mf = re.search(r"<\s*\d+,\s*\d+,\s*(true|false)", names["beta_grads"])
            fused = bool(mf) and mf.group(1) == "true"
            nodes = m + 1
            b_phi = phi_bytes_per_node(K, n) * nodes
            b_pi = (8 * K + 8) * nodes
            b_grads = (4 * K + 72) * m + 4 * K  # the shared end point's row once (node-stratified mini-batch)

            def rec(name, secs, nbytes):
                return {"kernel": name, "avg_ms": round(secs * 1e3, 5),
                        "algorithmic_bytes": int(nbytes) if nbytes else None,
                        "achieved_GBps": round(nbytes / secs / 1e9, 1) if nbytes and secs > 0 else None,
                        "frac": round(nbytes / secs / 1e9 / 8000.0, 4) if nbytes and secs > 0 else None}
            kernels = {"update_phi": rec(names["update_phi"], d(0, 1), b_phi)}
            if fused:
                kernels["update_pi+beta_grads"] = rec(names["beta_grads"], d(1, 3), b_pi + b_grads)
            else:
                kernels["update_pi"] = rec(names["update_pi"], d(1, 2), b_pi)
                kernels["beta_grads"] = rec(names["beta_grads"], d(2, 3), b_grads)
            kernels["sum_grads+update_theta"] = rec("sum_update_theta_kernel", d(3, 4), None)
            kernels["wait_for_sampler"] = {"avg_ms": round(d(4, 5) * 1e3, 5),
                                           "what": "the main chain's last kernel polling for the next mini-batch"}
            kernels["source"] = ("device stamps of the %d non-link steps of the timed window; each duration includes one "
                                 "kernel boundary" % int(ok.sum()))
            okl = (~nl) & (st[:, 0] > 0) & (st[:, 5] >= st[:, 4]) & (st[:, 4] >= st[:, 3]) & (st[:, 3] >= st[:, 2]) & \
                (st[:, 2] >= st[:, 1]) & (st[:, 1] >= st[:, 0])
            if okl.any():  # link steps (a few dozen nodes: latency, not bytes): where their time goes
                dl = lambda a, b: round(float((st[okl, b] - st[okl, a]).mean()) * 1e-6, 5)  # noqa: E731  (ms)
                kernels["link_steps_ms"] = {"update_phi": dl(0, 1), "update_pi": dl(1, 2), "beta_grads": dl(2, 3),
                                            "sum_grads+update_theta": dl(3, 4), "wait_for_sampler": dl(4, 5),
                                            "steps": int(okl.sum()), "update_phi_kernel": names.get("update_phi_small") or names["update_phi"]}
    else:
        launches = [(a.elapsed_time(b) * 1e-3, nn, g) for a, b, nn, g in ev]
    big = [(t, nn, g) for t, nn, g in launches if nn > m // 2]
    roofline = None
    if big:
        per_node = phi_bytes_per_node(K, n)
        nodes_per_launch = float(np.mean([g if nn <= 65535 else nn * (g / 65535.0) for _, nn, g in big]))
        avg_s = float(np.mean([t for t, _, _ in big]))
        achieved = per_node * nodes_per_launch / avg_s / 1e9
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "phi_traffic.json")
        if workload == "C3" and world == 1 and cfg.phi_wg_size == 64 and os.path.exists(tpath):  # the PMC passes were taken on this case
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
                traffic_source = "profiles/phi_traffic.json (stored rocprofv3 --pmc passes of this workload, not measured in this run)"
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": names["update_phi"], "achieved": round(achieved, 1), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                    "traffic_source": traffic_source,
                    "avg_launch_ms": round(avg_s * 1e3, 4), "launches": len(big),
                    "bytes_per_launch": int(per_node * nodes_per_launch),
                    "timer": "device wall-clock stamps written by the kernels (descriptor loop)" if lrn.loop is not None
                             else "HIP events on the launch stream"}
        if step_classes and step_classes["nonlink"]["ms_per_step"]:
            # step-level: algorithmic bytes of ALL kernels of a non-link iteration / its device time
            step_bytes = (m + 1) * (4 * K * (n + 4) + 68 * n + 8) + m * (8 * K + 72) + 24 * K
            st_s = step_classes["nonlink"]["ms_per_step"] * 1e-3
            roofline["step"] = {"bytes_per_nonlink_step": int(step_bytes), "achieved": round(step_bytes / st_s / 1e9, 1),
                                "frac": round(step_bytes / st_s / 1e9 / 8000.0, 4)}
            pi_bytes = 4.0 * cfg.N * K
            if pi_bytes <= 256e6:
                roofline["step"]["note"] = ("pi is %.0f MB: it fits the 256 MB Infinity Cache, so this configuration's "
                                            "ceiling is the cache's bandwidth, not HBM's 8 TB/s (the fraction is still "
                                            "quoted against 8 TB/s)" % (pi_bytes / 1e6))
        if kernels:
            roofline["kernels"] = kernels
            if ppx_ms and cfg.ppx_wg_size:
                H = int(lrn.heldoutPerplexity.edges.numel()) if hasattr(lrn.heldoutPerplexity, "edges") else None
                if H:
                    b_ppx = (8 * K + 88) * H
                    roofline["kernels"]["perplexity"] = {
                        "kernel": names["perplexity"], "avg_ms": round(ppx_ms, 4), "algorithmic_bytes": int(b_ppx),
                        "achieved_GBps": round(b_ppx / (ppx_ms * 1e-3) / 1e9, 1),
                        "frac": round(b_ppx / (ppx_ms * 1e-3) / 1e9 / 8000.0, 4),
                        "note": "whole HeldoutPerplexity() call (kernel + reduction + the 32-byte read-back), host-timed"}
    # ---- the shader clock the chip holds UNDER this workload: a few more (untimed) steps with the probe's idle waves
    # beside them on their own stream (ammsb_clock_probe) -- a roofline fraction is read against the clocks it ran at
    device_state = gpu_state.summarize(state_before, state_after)
    try:
        from mcmc_ammsb_gpu_amd import ops as _ops
        probe = _ops.ClockProbe(lrn.ctx, 64)
        n_probe = 12 if m >= 32768 else 60
        lrn.Run(n_probe // 3)
        probe.launch(600)
        lrn.Run(n_probe - n_probe // 3)
        clk = probe.read()
        sync()
        device_state["shader_clock_under_load_mhz"] = clk["mhz"]
        device_state["shader_clock_per_xcd_mhz"] = clk["mhz_per_xcd"]
        device_state["shader_clock_how"] = ("s_memtime / s_memrealtime of idle probe waves on a second stream beside %d "
                                            "untimed steps right after the timed window" % n_probe)
    except Exception as e:  # a measurement aid: never lose the bench line over it
        device_state["shader_clock_under_load_mhz"] = None
        device_state["shader_clock_how"] = "probe failed: %r" % (e,)
    if roofline is not None:
        roofline["device_state"] = device_state
    return {"settle_steps": settle_steps, "settle_log": settle_log, "device_state": device_state,
            "value": edges_done / dt, "ms_per_step": dt * 1e3 / steps, "dt": dt, "edges_done": int(edges_done),
            "host_enqueue_ms_per_step": t_enq * 1e3 / steps, "step_classes": step_classes,
            "value_per_class": value_per_class, "roofline": roofline, "ppx_ms": ppx_ms, "ppx": ppx,
            "kernel_names": names}


def sub_record(r, cfg, steps, warmup, loop):
    """The part of a measure() result that goes into reference_default_wg / small_configs."""
    rf = r["roofline"] or {}
    return {"value": r["value"], "unit": "edges/s", "ms_per_step": r["ms_per_step"], "steps": steps, "warmup": warmup,
            "value_per_class": r["value_per_class"], "step_classes": r["step_classes"],
            "phi_wg": cfg.phi_wg_size, "beta_wg": cfg.beta_wg_size, "ppx_wg": cfg.ppx_wg_size,
            "kernels_dispatched": r["kernel_names"],
            "roofline": {k: rf.get(k) for k in ("kernel", "achieved", "frac", "avg_launch_ms", "launches", "step", "kernels",
                                                "device_state")},
            "settle_steps": r.get("settle_steps", 0),
            "ppx_eval_ms": r["ppx_ms"], "loop": loop, "host_enqueue_ms_per_step": r["host_enqueue_ms_per_step"]}


def main():
    args = parse()
    spawn_ranks_if_needed(args, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    wd = Watchdog(args.watchdog_s if args.watchdog_s >= 0 else (300.0 if world > 1 else 0.0), rank, world)
    args.phase = wd.phase
    wd.phase("import torch")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the MI355X path has no CPU fallback")
    # AMMSB_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share
    # devices, the exchange is staged through host memory) -- for checking the code path, never for numbers
    backend = os.environ.get("AMMSB_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    wd.phase("rendezvous (init_process_group, %s)" % backend)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__ as ge
    wd.phase("build (rank 0) + first barrier")
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    wd.phase("synthetic graph + edge sets (host)")
    import ammsb_pkg
    ammsb_pkg.load()
    from mcmc_ammsb_gpu_amd import hostlib
    from mcmc_ammsb_gpu_amd.learner import Config, Learner

    N, K, m, n, deg, k_true = WORKLOADS[args.workload]
    t_setup = time.perf_counter()

    def note(msg):  # progress on stderr: the large workloads spend minutes of host time before the first launch
        if rank == 0:
            print("[bench %6.1fs] %s" % (time.perf_counter() - t_setup, msg), file=sys.stderr, flush=True)
    edges = hostlib.generate_graph(N, k_true, deg, seed=20260101)
    note("graph: %d edges" % edges.size)
    ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
    note("training / held-out sets built")
    use_graph = args.loop == "graph" or (args.loop == "auto" and world == 1 and not args.host_sampling)

    def make_cfg(K_, m_, n_, phi_wg, beta_wg, ppx_wg, graph):
        return Config.from_cli_defaults(K=K_, mini_batch_size=m_, num_node_sample=n_, strategy="Node",
                                        phi_wg_size=phi_wg, beta_wg_size=beta_wg, ppx_wg_size=ppx_wg,
                                        device_sampling=not args.host_sampling,
                                        graph_launch=graph, graph_timestamps=graph, phi_exchange=args.exchange)
    cfg = make_cfg(K, m, n, pick_wg(K, args.phi_wg, 16),   # K=1024 -> 64: the LDS-streamed kernel
                   pick_wg(K, args.beta_wg, 16), pick_wg(K, args.ppx_wg, 16), use_graph)
    wd.phase("Learner() incl. the split calibration's update_phi / all-gather / point-to-point timings")
    lrn = Learner(cfg, ds, rank=rank, world_size=world)
    setup_s = time.perf_counter() - t_setup
    pi_placement = getattr(lrn, "pi_placement", None)
    lrn_E, lrn_H = int(ds.E), int(ds.heldout_edges.size)
    graphs = lrn.loop is not None
    note("learner ready (%s loop)" % ("device-descriptor" if graphs else "eager"))

    r = measure(args, lrn, cfg, m, args.steps, args.warmup, world, dist, torch, args.workload, args.ppx_calls,
                settle_s=args.settle_s)
    dt, edges_done, roofline = r["dt"], r["edges_done"], r["roofline"]

    # ---- the same window again after --sustained-s seconds of load: update_phi holds the package at its power cap, and
    # what the chip delivers once its power controller has settled is what a long run sees (profiles/README.md)
    sustained = None
    if args.sustained_s > 0:
        try:
            r2 = measure(args, lrn, cfg, m, args.steps, 0, world, dist, torch, args.workload, 0, settle_s=args.sustained_s)
            rf2 = r2["roofline"] or {}
            sustained = {"after_s": args.sustained_s, "untimed_steps": r2["settle_steps"], "steps": args.steps,
                         "value": r2["value"], "unit": "edges/s", "ms_per_step": r2["ms_per_step"],
                         "value_per_class": (r2["value_per_class"] or {}).get("value"),
                         "step_classes": r2["step_classes"],
                         "update_phi_ms": rf2.get("avg_launch_ms"), "frac": rf2.get("frac"),
                         "device_state": r2["device_state"],
                         "what": "a second window of the same --steps after that many seconds of untimed iterations: the "
                                 "throughput once the package's power controller has settled under this load (`value` "
                                 "above is the contract's window: --warmup steps after start-up, then --steps)"}
        except Exception as e:  # a reported extra: never lose the line over it
            sustained = {"error": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(args, lrn, cfg, ds, m + 1)
        except Exception as e:  # the baseline is a reported extra; never lose the GPU number over it
            cpu = {"value": None, "unit": "mini-batch edges/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    wd.phase("closing HeldoutPerplexity()")
    final_ppx = lrn.HeldoutPerplexity()
    loop_note = loop_form(graphs)
    fallback = getattr(lrn, "loop_fallbacks", 0)
    if fallback:
        loop_note += "; %d run(s) finished on the stream-event hand-over after a device-side wait gave up" % fallback
    phi_split = None if world == 1 else dict(getattr(lrn, "calibration", {}), replicated_groups=lrn.g_rep,
                                             groups_per_block=lrn.cc, chunks=lrn.nch,
                                             beta_gradient="%s%s" % (lrn.grads_mode, " (update_pi folded into its launch)"
                                                                     if lrn.grads_mode == "replicated" and lrn.grads_fused else ""))
    if world > 1:
        # what makes a multi-GPU line readable by itself: a traced (untimed) window right after the timed one -- per
        # non-link step where the time went (own blocks, replicated groups, each chunk's exchange and what of it was
        # hidden, update_pi, the gradient's all-gather), next to the calibration's prediction above
        wd.phase("traced steps (phi_split.trace)")
        try:
            lrn.shard_trace = []
            traced = 0
            while traced < 64 and sum(1 for t_ in lrn.shard_trace if t_["n_nodes"] > m // 2) < 6:
                lrn.Run(4)
                traced += 4
            lrn.drain()
            rep = lrn.shard_report()
            lrn.shard_trace = None
            phi_split["trace"] = rep
            if rep.get("steps") and phi_split.get("phi_ms"):
                phi_split["trace"]["predicted_vs_measured"] = {
                    "predicted_phi_ms": phi_split.get("predicted_phi_ms"), "measured_phi_phase_ms": rep.get("phi_phase_ms"),
                    "calibrated_full_exchange_ms": phi_split.get("xchg_ms"), "measured_exchange_ms": rep.get("exchange_ms")}
        except Exception as e:  # diagnostics must not cost the line
            phi_split["trace"] = {"error": repr(e)}
    wd.phase("single-rank extras" if world == 1 else "closing the learner")
    lrn.close()
    del lrn
    torch.cuda.empty_cache()

    # ---- the same workload at the reference's DEFAULT work-group sizes (32 for phi, beta and perplexity, main.cc:61-64)
    ref_wg = None
    if rank == 0 and world == 1 and args.extras and (cfg.phi_wg_size, cfg.beta_wg_size, cfg.ppx_wg_size) != (32, 32, 32):
        try:
            cfg32 = make_cfg(K, m, n, 32, 32, 32, use_graph)
            l32 = Learner(cfg32, ds, rank=0, world_size=1)
            st32 = max(10, min(args.steps, 40))
            r32 = measure(args, l32, cfg32, m, st32, min(args.warmup, 5), 1, dist, torch, args.workload, 2,
                          settle_s=args.settle_s)
            ref_wg = sub_record(r32, cfg32, st32, min(args.warmup, 5), loop_form(l32.loop is not None))
            ref_wg["what"] = ("%s with --phi-wg / --beta-wg / --ppx-wg left at the reference's defaults (32, main.cc:61-64): "
                              "%d columns per work-item" % (args.workload, (K + 31) // 32))
            l32.close()
            del l32
            torch.cuda.empty_cache()
            note("reference-default work-groups: %.3e edges/s" % ref_wg["value"])
        except Exception as e:
            ref_wg = {"error": repr(e)}

    # ---- the other single-GPU configurations of BASELINE.json, each with its own graph and Learner: the small ones
    # (launch-latency territory) and, further down, the largest (C5: N = 10M, K = 4096, degree 64)
    def other_config(name, st_n, wu_n, ppx_n, settle):
        t_cfg = time.perf_counter()
        N2, K2, m2, n2, deg2, kt2 = WORKLOADS[name]
        e2 = hostlib.generate_graph(N2, kt2, deg2, seed=20260101)
        ds2 = hostlib.Dataset.robust(N2, e2, heldout_ratio=0.01, rand_seed=1)
        c2 = make_cfg(K2, m2, n2, pick_wg(K2, 0, 16), pick_wg(K2, 0, 16), pick_wg(K2, 0, 16), use_graph)
        l2 = Learner(c2, ds2, rank=0, world_size=1)
        setup2 = time.perf_counter() - t_cfg
        note("%s: learner ready after %.1f s" % (name, setup2))
        r2 = measure(args, l2, c2, m2, st_n, wu_n, 1, dist, torch, name, ppx_n, settle_s=settle)
        rec = sub_record(r2, c2, st_n, wu_n, loop_form(l2.loop is not None))
        rec["workload"] = "%s: synthetic a-MMSB graph N=%d avg-degree=%d K=%d mini-batch=%d n=%d strategy=Node" \
            % (name, N2, deg2, K2, m2, n2)
        rec["E"], rec["heldout_edges"], rec["setup_s"] = int(ds2.E), int(ds2.heldout_edges.size), round(setup2, 1)
        rec["pi_bytes"] = 4 * N2 * K2
        l2.close()
        del l2, ds2, e2
        torch.cuda.empty_cache()
        note("%s: %.4f ms per step" % (name, rec["ms_per_step"]))
        return rec

    small = None
    if rank == 0 and world == 1 and args.extras and args.workload == "C3":
        small = {}
        for name, st_small in (("C1", 3000), ("C2", 2000)):
            try:
                small[name] = other_config(name, st_small, 200, 2, 0.5)
            except Exception as e:
                small[name] = {"error": repr(e)}

    cpp = None
    if rank == 0 and world == 1 and args.cpp_dropin and args.workload in ("C1", "C2", "C3"):
        try:
            cpp = cpp_dropin(args, hostlib, N, K, m, n, cfg.phi_wg_size, edges, note)
        except Exception as e:
            cpp = {"error": repr(e)}

    # ---- BASELINE.json configs[4], C5: N = 10M, K = 4096, average degree 64 -- pi is 164 GB in ONE allocation (the
    # reference cannot address it: 32-bit row offsets, partitioned-alloc.h:24-28; and its perplexity kernel would
    # allocate 52 GB of scratch, perplexity.cc:245-247).  About two minutes of host set-up (graph, cuckoo sets, CSR).
    large = None
    if rank == 0 and world == 1 and args.extras and args.workload == "C3" and args.large != "none":
        del ds, edges
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        which = args.large
        if which == "auto":
            free_hbm, _ = torch.cuda.mem_get_info()
            try:
                host_free = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
            except (ValueError, OSError):
                host_free = 0
            which = "C5" if free_hbm > 215e9 and host_free > 48e9 else "C5s"
            note("large_configs: %s (free HBM %.0f GB, free host memory %.0f GB)" % (which, free_hbm / 1e9, host_free / 1e9))
        large = {}
        try:
            large[which] = other_config(which, 40, 6, 3, 1.0)
            if which != "C5":
                large[which]["stand_in"] = ("C5s = C5's rows (K = 4096, mini-batch 65536) over 2M vertices at degree 32: "
                                            "taken because this box could not hold C5 (N = 10M: 164 GB of pi)")
        except Exception as e:
            large[which] = {"error": repr(e)}
        ds = edges = None

    if rank == 0:
        E_main, H_main = int(lrn_E), int(lrn_H)
        out = {
            "metric": "mini-batch edges/s (SG-MCMC a-MMSB learner loop)",
            "value": edges_done / dt,
            "unit": "edges/s",
            "n_gpus": dist.get_world_size() if world > 1 else 1,
            "rccl_ranks": (dist.get_world_size() if backend == "nccl" else 0) if world > 1 else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if backend == "nccl" or world == 1 else "synthetic (REHEARSAL: %s backend, ranks share GPUs)" % backend,
            "config": {"workload": "%s: synthetic a-MMSB graph N=%d avg-degree=%d K=%d mini-batch=%d n=%d strategy=Node"
                                   % (args.workload, N, deg, K, m, n),
                       "E": E_main, "heldout_edges": H_main,
                       "phi_wg": cfg.phi_wg_size, "beta_wg": cfg.beta_wg_size, "ppx_wg": cfg.ppx_wg_size,
                       "sampling": "host(rand_r): the reference's stream" if args.host_sampling else
                                   "device: same distribution as sample.cc, NOT its rand_r stream (same seed does not "
                                   "reproduce a reference trajectory; --host-sampling does)",
                       "loop": loop_note,
                       "host": "python (ctypes -> C ABI); see cpp_dropin for the C++ mcmc::Learner",
                       "parallelism": ("one GPU" if world == 1 else
                                       "replicated pi, node-sharded phi x%d, beta gradient %s" % (world, phi_split["beta_gradient"])),
                       "phi_split": phi_split},
            "host_enqueue_ms_per_step": r["host_enqueue_ms_per_step"],
            "step_classes": r["step_classes"],
            "value_per_class": r["value_per_class"],
            "ppx_eval_ms": r["ppx_ms"],
            "perplexity": r["ppx"],
            "perplexity_after": final_ppx,
            "mini_batch_edges": int(edges_done),
            "setup_s": round(setup_s, 1),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "reference_default_wg": ref_wg,
            "small_configs": small,
            "large_configs": large,
            "sustained": sustained,
            "pi_placement": pi_placement,
            "settle": {"seconds": args.settle_s, "steps": r.get("settle_steps", 0), "auto": r.get("settle_log"),
                       "why": "untimed iterations in front of the warm-up steps: the clocks' ramp from idle (a few hundred "
                              "milliseconds) is not part of the measurement; `sustained` is the window after seconds of load"},
            "cpp_dropin": cpp,
        }
        wd.phase("done")
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
