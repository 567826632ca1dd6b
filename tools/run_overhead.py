"""What one Learner.Run(n) call costs beyond its steps' device time (descriptor loop, C3): host time of the call to
the end of the drain, against the device span from the first step's update_phi start to the last step's
AMMSB_STAMP_NEXT, for n = 1 .. 100.  Usage: python tools/run_overhead.py [workload]"""
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
from mcmc_ammsb_gpu_amd import hostlib  # noqa: E402
from mcmc_ammsb_gpu_amd.learner import Config, Learner  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
N, K, m, n, deg, k_true = bench.WORKLOADS[wl]
edges = hostlib.generate_graph(N, k_true, deg, seed=20260101)
ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
wg = bench.pick_wg(K, 0, 16)
for strategy in ("Node", "NodeLink"):
    cfg = Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, strategy=strategy, phi_wg_size=wg,
                                   beta_wg_size=wg, ppx_wg_size=wg, device_sampling=True, graph_launch=True,
                                   graph_timestamps=True)
    lrn = Learner(cfg, ds)
    lrn.Run(200)
    lrn.drain()
    torch.cuda.synchronize()
    for steps in (1, 2, 5, 20, 100):
        rows = []
        for rep in range(6):
            first = lrn.phiUpdater.count_calls + 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lrn.Run(steps)
            t_enq = time.perf_counter()
            lrn.drain()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            st = lrn.loop.step_stamps(first, steps)  # ns
            span = (st[-1, 5] - st[0, 0]) * 1e-6  # first update_phi start -> last step's "next mini-batch available"
            last = (st[-1, 5] - st[-1, 0]) * 1e-6
            rows.append(((t1 - t0) * 1e3, (t_enq - t0) * 1e3, span, last))
        a = np.array(rows[1:])
        print("%s %-8s n=%3d  call+drain %.3f ms (enqueue side %.3f) | device span %.3f | call - span = %.3f ms (min %.3f max %.3f) | last step %.3f"
              % (wl, strategy, steps, a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), (a[:, 0] - a[:, 2]).mean(),
                 (a[:, 0] - a[:, 2]).min(), (a[:, 0] - a[:, 2]).max(), a[:, 3].mean()), flush=True)
    lrn.close()
    del lrn
    torch.cuda.empty_cache()
