"""debug: where does update_phi differ from the oracle?  usage: python tools/dbg_pair.py N K n nodes L noise"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import ammsb_pkg
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import ops as hip
import oracle_lib as orc
orc.build()
from test_gpu_parity import Problem
N, K, n, n_nodes, L, noise = [int(x) for x in sys.argv[1:7]]
pr = Problem(orc, hip, N, K, n, n_nodes)
upd = hip.PhiUpdater(pr.ctx, pr.beta, pr.pi, pr.phi_sum, pr.dset, n_nodes, (42, 43), L, phi_disable_noise=not noise)
seeds = orc.rng_init(n_nodes * L, 42, 43)
upd(pr.nodes, pr.nb, n_nodes)
pr.sync()
print(pr.ctx.kernel_names())
want = orc.update_phi(pr.p_orc, pr.beta_h, pr.pi_h.reshape(-1), pr.phi_sum_h, pr.oset, pr.nodes_h, pr.nb_h.reshape(-1), 1, seeds, L, 1, bool(noise))
got = upd.phi_vec.cpu().numpy()[:n_nodes]
bad = got.view(np.uint32) != want.view(np.uint32)
print("mismatching elements:", int(bad.sum()), "of", bad.size)
rows = np.where(bad.any(1))[0]
print("bad rows:", rows[:40], "count", rows.size)
if rows.size:
    r = rows[0]
    cols = np.where(bad[r])[0]
    print("row", r, "bad cols", cols[:64], "count", cols.size)
    print("got", got[r, cols[:8]], "want", want[r, cols[:8]])
    rel = np.abs(got - want) / np.abs(want)
    print("max rel", rel.max(), "links per bad row:", [(int(x), int(pr.oset.has(orc.make_edge(np.full(n, pr.nodes_h[x]), pr.nb_h[x])).sum())) for x in rows[:10]])
print("streams equal:", np.array_equal(upd.rand.host(), seeds))
