"""update_phi's launch time against the time since the load began (C3, descriptor loop): chunks of 50 iterations back
to back for --seconds, per chunk the mean update_phi time of its non-link steps (device stamps), the package power
and the shader clock sysfs reports.  Then --idle seconds of nothing and the same again (does the transient repeat?).
Usage: python tools/phi_timeline.py [seconds] [idle]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.build()
import ammsb_pkg  # noqa: E402

ammsb_pkg.load()
import bench  # noqa: E402
from mcmc_ammsb_gpu_amd import gpu_state, hostlib  # noqa: E402
from mcmc_ammsb_gpu_amd.learner import Config, Learner  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 14.0
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
N, K, m, n, deg, k_true = bench.WORKLOADS["C3"]
edges = hostlib.generate_graph(N, k_true, deg, seed=20260101)
ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
cfg = Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, strategy="Node", phi_wg_size=64, beta_wg_size=64,
                               ppx_wg_size=64, device_sampling=True, graph_launch=True, graph_timestamps=True)
lrn = Learner(cfg, ds)
lrn.Run(2)
lrn.drain()
time.sleep(idle)
dev = torch.cuda.current_device()
for phase in ("first", "after_idle"):
    t_begin = time.perf_counter()
    rows = []
    while time.perf_counter() - t_begin < seconds:
        first = lrn.phiUpdater.count_calls + 1
        lrn.step_log = []
        t0 = time.perf_counter()
        lrn.Run(50)
        lrn.drain()
        t1 = time.perf_counter()
        st = lrn.loop.step_stamps(first, 50)
        ne = np.concatenate(lrn.step_log)
        non = ne == m
        phi_ms = (st[non, 1] - st[non, 0]).mean() * 1e-6 if non.any() else float("nan")
        g = gpu_state.read(dev)
        rows.append({"t": round(t0 - t_begin, 2), "phi_ms": round(float(phi_ms), 4), "nonlink": int(non.sum()),
                     "ms_per_step": round((t1 - t0) * 1e3 / 50, 4), "power_w": g.get("power_w"), "sclk": g.get("sclk_mhz"),
                     "temp": g.get("junction_temp_c")})
    lrn.step_log = None
    print(phase, json.dumps(rows), flush=True)
    a = np.array([r["phi_ms"] for r in rows])
    t = np.array([r["t"] for r in rows])
    for lo, hi in ((0, 1), (1, 2), (2, 4), (4, 6), (6, 9), (9, 100)):
        sel = (t >= lo) & (t < hi) & np.isfinite(a)
        if sel.any():
            print("  %s t in [%g, %g) s: update_phi %.3f ms (min %.3f max %.3f, %d chunks)" % (phase, lo, hi, a[sel].mean(), a[sel].min(), a[sel].max(), int(sel.sum())), flush=True)
    time.sleep(idle)
lrn.close()
