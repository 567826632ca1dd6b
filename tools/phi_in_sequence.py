"""update_phi alone (back to back) against update_phi inside the step's kernel sequence (update_phi, update_pi +
gradient in one launch, theta step), one process, one pi, no sampling chain beside it: is the 7 % between "alone" and
"in the loop" the neighbours on the device or the kernels in front of it?  Usage: python tools/phi_in_sequence.py"""
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

N, K, m, n, deg, k_true = bench.WORKLOADS["C3"]
edges = hostlib.generate_graph(N, k_true, deg, seed=20260101)
ds = hostlib.Dataset.robust(N, edges, heldout_ratio=0.01, rand_seed=1)
cfg = Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=n, strategy="NodeNonLink", phi_wg_size=64,
                               beta_wg_size=64, ppx_wg_size=64, device_sampling=True, graph_launch=False)
lrn = Learner(cfg, ds)
print("placement", lrn.pi_placement, flush=True)
lrn.Run(3)
lrn.drain()
phi, beta = lrn.phiUpdater, lrn.betaUpdater
s = lrn.samples[lrn.phase]
lrn.futures[lrn.phase].result()
torch.cuda.synchronize()
cur = torch.cuda.current_stream()


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record(cur)
    return e


def alone(k):
    marks = []
    for _ in range(k):
        a = ev()
        phi.update_phi(s.dev_nodes, s.neighbor_sampler.GetData(), s.n_nodes)
        marks.append((a, ev()))
    return marks


def sequence(k, fused=True, theta=True):
    marks = []
    for _ in range(k):
        a = ev()
        phi.update_phi(s.dev_nodes, s.neighbor_sampler.GetData(), s.n_nodes)
        b = ev()
        if fused:
            g = beta.update_pi_and_grads(phi, s.dev_nodes, s.dev_edges, s.n_edges)
        else:
            phi.update_pi(s.dev_nodes, s.n_nodes)
            g = beta.calculate_grads(s.dev_edges, s.n_edges)
        if theta:
            beta.update_theta(1.0, g)
        marks.append((a, b))
    return marks


t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.5:
    alone(6)
    sequence(6)
    torch.cuda.synchronize()
import ctypes as C  # noqa: E402
lib = lrn.ctx.lib
lib.ammsb_debug_beta_pi_nt.argtypes = [C.c_int]


def nt(on):
    lib.ammsb_debug_beta_pi_nt(on)


forms = (("alone", lambda: alone(4), None),
         ("in sequence: fused pi+grads (plain stores)", lambda: sequence(4), 0),
         ("in sequence: fused pi+grads (nt stores)", lambda: sequence(4), 1),
         ("in sequence: update_pi, grads", lambda: sequence(4, fused=False), None))
res = {name: [] for name, _, _ in forms}
tot = {name: [] for name, _, _ in forms}
for r in range(8):
    for name, fn, ntv in forms:
        if ntv is not None:
            nt(ntv)
        a0 = ev()
        mk = fn()
        a1 = ev()
        torch.cuda.synchronize()
        res[name] += [a.elapsed_time(b) for a, b in mk[1:]]  # (the first launch of a turn follows another turn's kernels)
        tot[name].append(a0.elapsed_time(a1) / 4)
nt(-1)
for name, v in res.items():
    v = sorted(v)
    print("%-46s update_phi median %.4f ms  min %.4f  max %.4f | whole step %.4f ms" % (
        name, float(np.median(v)), v[0], v[-1], float(np.median(tot[name]))), flush=True)
lrn.close()
