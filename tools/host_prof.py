"""development aid: host vs device time per step of Learner.Run at C1 / C2 for the loop's launch forms
(AMMSB_LOOP_LAUNCH=graph|serial, AMMSB_LOOP_HANDSHAKE=event), each in a fresh learner, interleaved repeats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ammsb_pkg
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import hostlib
from mcmc_ammsb_gpu_amd.learner import Config, Learner

t0 = time.perf_counter(); x = 0
for i in range(2_000_000): x += i
print("host speed: 2M-iteration python loop %.0f ms, cpus %d" % ((time.perf_counter() - t0) * 1e3, len(os.sched_getaffinity(0))), flush=True)
which = sys.argv[1:] or ["C1", "C2"]
for name, (N, K, m) in {"C1": (10_000, 32, 1024), "C2": (100_000, 256, 8192)}.items():
    if name not in which:
        continue
    edges = hostlib.generate_graph(N, min(K, 64), 32, seed=20260101)
    ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
    wg = 64 if K >= 256 else 32
    variants = [("default", wg, False), ("graph", wg, False), ("event", wg, False)]
    if os.environ.get("HOST_PROF_VARIANTS"):
        variants = [("default", wg, False), ("default", 64, False), ("default", wg, True), ("default", 64, True)]
    for rep in range(2):
        for mode, wg, stamps in variants:
            os.environ.pop("AMMSB_LOOP_LAUNCH", None); os.environ.pop("AMMSB_LOOP_HANDSHAKE", None)
            if mode == "graph": os.environ["AMMSB_LOOP_LAUNCH"] = "graph"
            if mode == "event": os.environ["AMMSB_LOOP_HANDSHAKE"] = "event"
            cfg = Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=32, strategy="Node", phi_wg_size=wg,
                                           beta_wg_size=wg, ppx_wg_size=wg, device_sampling=True, graph_launch=True, graph_timestamps=stamps)
            lrn = Learner(cfg, ds)
            lrn.Run(300); lrn.drain()
            steps = 3000
            t0 = time.perf_counter(); lrn.Run(steps); t1 = time.perf_counter(); lrn.drain(); t2 = time.perf_counter()
            print("%s %-7s wg %d stamps %d rep %d: enqueue %.1f us/step, total %.1f us/step" % (name, mode, wg, stamps, rep, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6), flush=True)
            if os.environ.get("HOST_PROF_SHORT"):
                for steps in (100, 1000, 1000, 513, 512):
                    t0 = time.perf_counter(); lrn.Run(steps); t1 = time.perf_counter(); lrn.drain(); t2 = time.perf_counter()
                    print("   Run(%d): enqueue %.2f ms, total %.2f ms" % (steps, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
            lrn.close()
