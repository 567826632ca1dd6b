"""Host (enqueue-side) cost per iteration of the multi-GPU schedule, measured on ONE GPU: a one-rank RCCL group with
force_exchange runs the sharded eager loop exactly as a rank of a multi-GPU job issues it (chunked update_phi blocks,
replicated groups on their stream, in-place all-gathers, rank-ordered gradient sum).  With 8 ranks a non-link step's
device time falls to ~0.45 ms and a link step's stays ~0.06 ms: whatever the host needs per step beyond that bounds the
job.  Usage: python tools/shard_host_cost.py [rho] [chunks]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.build()
import ammsb_pkg  # noqa: E402

ammsb_pkg.load()
import bench  # noqa: E402
from mcmc_ammsb_gpu_amd import hostlib  # noqa: E402
from mcmc_ammsb_gpu_amd.learner import Config, Learner  # noqa: E402

rho = float(sys.argv[1]) if len(sys.argv) > 1 else 0.09
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29631", rank=0, world_size=1, device_id=torch.device("cuda", 0))
N, K, m, n, deg, k_true = bench.WORKLOADS["C3"]
edges = hostlib.generate_graph(N, k_true, deg, seed=20260101)
ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
for strategy in ("NodeLink", "NodeNonLink", "Node"):
    cfg = Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, strategy=strategy, phi_wg_size=64,
                                   beta_wg_size=64, ppx_wg_size=64, device_sampling=True, graph_launch=False,
                                   phi_chunks=chunks, phi_replicate=rho, force_exchange=True)
    lrn = Learner(cfg, ds, rank=0, world_size=1)
    lrn.Run(30)
    lrn.drain()
    res = []
    for rep in range(4):
        steps = 12
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lrn.Run(steps)
        t1 = time.perf_counter()
        lrn.drain()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append(((t1 - t0) * 1e3 / steps, (t2 - t0) * 1e3 / steps))
    print("%-12s rho %.2f chunks %d (g_rep %d, nch %d): host enqueue %.3f ms per step (runs: %s) | with the device %.3f ms per step"
          % (strategy, rho, chunks, lrn.g_rep, lrn.nch, min(r[0] for r in res), " ".join("%.3f" % r[0] for r in res),
             min(r[1] for r in res)), flush=True)
    if strategy == "NodeLink":
        pr = cProfile.Profile()
        pr.enable()
        lrn.Run(200)
        pr.disable()
        lrn.drain()
        st = pstats.Stats(pr, stream=sys.stdout)
        st.sort_stats("cumulative").print_stats(28)
    lrn.close()
    del lrn
    torch.cuda.empty_cache()
dist.destroy_process_group()
