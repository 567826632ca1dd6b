"""Does WHERE a 4 GB table lands in HBM change what a random 4 KiB-row gather over it gets?  One process, several
candidate allocations of the same size, the same gather (torch.index_select of 131 072 random rows into one output
buffer) timed round-robin over the candidates so that time-dependent effects hit all alike.
Usage: python tools/placement_probe.py [candidates] [rounds]"""
import sys
import time

import torch

cands = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N, K, R = 1_000_000, 1024, 131072
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(1)
idx = torch.randint(0, N, (R,), generator=g).to(dev)
out = torch.empty((R, K), dtype=torch.float32, device=dev)
bufs, holes = [], []
for c in range(cands):
    b = torch.empty((N, K), dtype=torch.float32, device=dev)
    b.normal_()
    bufs.append(b)
    if c % 2 == 0:  # odd-sized spacers between some candidates: different alignment / buddy blocks for the next one
        holes.append(torch.empty(((c + 1) * 37_000_001,), dtype=torch.uint8, device=dev))
print("addresses", [hex(b.data_ptr()) for b in bufs], flush=True)
# warm clocks
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.0:
    for b in bufs:
        torch.index_select(b, 0, idx, out=out)
    torch.cuda.synchronize()
ms = [[] for _ in bufs]
for r in range(rounds):
    for c, b in enumerate(bufs):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(8):
            torch.index_select(b, 0, idx, out=out)
        e.record()
        e.synchronize()
        ms[c].append(a.elapsed_time(e) / 8)
import statistics
for c in range(cands):
    v = sorted(ms[c])
    print("candidate %d: median %.4f ms  p10 %.4f  p90 %.4f  -> %.0f GB/s read" % (c, statistics.median(v), v[len(v) // 10], v[-len(v) // 10 - 1], R * K * 4 / (statistics.median(v) * 1e-3) / 1e9), flush=True)
