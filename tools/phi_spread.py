"""Why does one C3 update_phi launch take 1.53 ms in one process / on one box and 1.75 ms in another?  

One process, several placements of pi, each timed with HIP events, with the shader clock read beside the launch
(ammsb_clock_probe) and -- under the -DAMMSB_PHI_TRACE build (tools/phi_trace.sh build; AMMSB_HIP_LIB=tools/ab/trace/
libammsb_hip_trace.so) -- per-block stamps: cycles and wall time of every block, which XCD it ran on, when each XCD
finished.  Prints one JSON line per variant.

    python tools/phi_spread.py [variants]      variants: comma list of A (one allocation, the default), B32 (32 row
                                               blocks), Cpad (pi allocated behind 6 GB of other buffers), Dmemset
                                               (first touched by a memset), default A,B32,Cpad,Dmemset,A
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import ammsb_pkg
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import gpu_state, hostlib, ops

N, K, n, m = 1_000_000, 1024, 32, 65536
WARM = int(os.environ.get("SPREAD_WARM", "300"))
TIMED = int(os.environ.get("SPREAD_TIMED", "30"))
variants = (sys.argv[1] if len(sys.argv) > 1 else "A,B32,Cpad,Dmemset,A").split(",")

p = ops.make_params(N, K, E=16 * N, num_node_sample=n)
ctx = ops.Context(p)
lib = ctx.lib
has_blocks = hasattr(lib, "ammsb_debug_blocks")
if has_blocks:
    lib.ammsb_debug_blocks.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
rng = np.random.default_rng(0)
theta = ctx.from_numpy(rng.gamma(1.0, 1.0, 2 * K).astype(np.float32))
beta = ctx.zeros((2 * K,), torch.float32)
ops.beta_from_theta(ctx, theta, beta)
u = rng.integers(0, N, 200000, dtype=np.uint64)
v = rng.integers(0, N, 200000, dtype=np.uint64)
e = np.unique((np.minimum(u, v) << np.uint64(32)) | np.maximum(u, v))
hs = hostlib.HostSet(e)
dset = ops.DeviceSet(ctx, hs.Serialize(), hs.BinsPerBucket(), hs.PrimeIdx())
nn = m + 1
nodes = ctx.from_numpy(rng.permutation(N)[:nn].astype(np.uint32))
nbrs = ctx.from_numpy(rng.integers(0, N, size=(nn, n), dtype=np.uint32))
probe = ops.ClockProbe(ctx, 64)
print(json.dumps({"pid": os.getpid(), "lib": os.environ.get("AMMSB_HIP_LIB", "default"), "state_at_start": gpu_state.read(0),
                  "device": torch.cuda.get_device_name(0)}), flush=True)


def run(variant):
    junk = None
    if variant == "Cpad":
        junk = torch.empty(6 << 30, dtype=torch.uint8, device="cuda")
    nblk = int(variant[1:]) if variant[0] in "BV" and variant[1:].isdigit() else 1
    pi = ops.RowPartitionedMatrix(ctx, N, K, rows_in_block=(N + nblk - 1) // nblk if variant[0] == "B" else 0)
    if variant[0] == "V" and nblk > 1:  # ONE allocation presented to the kernels as nblk row blocks (the multi-block code path)
        whole = pi.blocks[0]
        rib = (N + nblk - 1) // nblk
        pi.blocks = [whole[r:r + rib] for r in range(0, N, rib)]
        pi.rows_in_block = rib
        for i, b in enumerate(pi.blocks):
            pi.desc.blocks[i] = b.data_ptr()
        pi.desc.rows_in_block = rib
        pi.desc.num_blocks = len(pi.blocks)
        pi._whole = whole
    if variant == "Dmemset":
        for b in pi.blocks:
            b.zero_()
    phi_sum = ctx.zeros((N,), torch.float32)
    ops.RandomGammaAndNormalize(ctx, 1.0, 1.0, pi, phi_sum)
    upd = ops.PhiUpdater(ctx, beta, pi, phi_sum, dset, nn, (42, 43), 64)
    upd.count_calls = 1
    s0 = gpu_state.read(0)
    for _ in range(WARM):
        upd.update_phi(nodes, nbrs, nn)
    torch.cuda.synchronize()
    ts, clocks = [], []
    for i in range(TIMED):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        upd.update_phi(nodes, nbrs, nn)
        b.record()
        if i % 10 == 5:
            probe.launch(400)  # beside this launch (its own stream): the clock the chip holds under update_phi
        torch.cuda.synchronize()
        if i % 10 == 5:
            clocks.append(probe.read())
        else:
            ts.append(a.elapsed_time(b))
    s1 = gpu_state.read(0)
    rec = {"variant": variant, "pi_blocks": len(pi.blocks), "pi_ptr": hex(pi.blocks[0].data_ptr()),
           "phi_vec_ptr": hex(upd.phi_vec.data_ptr()), "kernel": ctx.kernel_names()["update_phi"],
           "ms_median": round(float(np.median(ts)), 4), "ms_min": round(min(ts), 4), "ms_max": round(max(ts), 4),
           "clock_under_load_mhz": [c["mhz"] for c in clocks], "clock_per_xcd": clocks[-1]["mhz_per_xcd"] if clocks else None,
           "state": gpu_state.summarize(s0, s1)}
    if has_blocks:
        nb = 65535
        buf = (C.c_ulonglong * (5 * nb))()
        assert lib.ammsb_debug_blocks(buf, nb) == 0
        t = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 5).astype(np.int64)
        c0, c1, w0, w1, hw = (t[:, i] for i in range(5))
        xcc = (hw >> 32) & 0xf
        ok = (w1 - w0) > 200
        T0 = w0.min()
        per = {}
        for x in range(8):
            sel = ok & (xcc == x)
            if sel.any():
                per[x] = {"blocks": int(sel.sum()), "mhz": round(float(np.median((c1 - c0)[sel] / ((w1 - w0)[sel] / 100.0))), 1),
                          "block_cycles_median": int(np.median((c1 - c0)[sel])),
                          "block_us_median": round(float(np.median((w1 - w0)[sel])) / 100.0, 2),
                          "first_start_us": round(float((w0[sel] - T0).min()) / 100.0, 2),
                          "last_end_us": round(float((w1[sel] - T0).max()) / 100.0, 1)}
        rec["blocks"] = {"span_us": round(float(w1.max() - T0) / 100.0, 1),
                         "mhz": round(float(np.median((c1 - c0)[ok] / ((w1 - w0)[ok] / 100.0))), 1),
                         "block_cycles_median": int(np.median((c1 - c0)[ok])),
                         "block_us_median": round(float(np.median((w1 - w0)[ok])) / 100.0, 2), "per_xcd": per}
    print(json.dumps(rec), flush=True)
    del upd, pi, phi_sum, junk
    torch.cuda.empty_cache()


for vnt in variants:
    run(vnt)
