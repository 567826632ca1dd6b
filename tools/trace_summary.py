#!/usr/bin/env python3
"""Per-kernel durations from a rocprofv3 kernel trace, split by launch size.

    python tools/trace_summary.py gpurun_out/r01c/trace/bench_kernel_trace.csv profiles/r01_bench_launch_classes.json

rocprofv3's --stats averages every launch of a kernel; update_phi (and the other per-mini-batch kernels) alternate
between chip-filling non-link launches (65 537 nodes) and link launches of a few dozen nodes, so the plain average
says little.  This groups a kernel's launches by grid size and reports count / mean / min / max per class, which is
what bench.py's `roofline.avg_launch_ms` (HIP events around the non-link launches) has to agree with."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    groups = defaultdict(list)
    for r in csv.DictReader(open(src)):
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        groups[short(r["Kernel_Name"])].append((grid, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    out = {}
    for k, v in sorted(groups.items()):
        gmax = max(g for g, _ in v)
        big = [d for g, d in v if g * 2 > gmax]
        small = [d for g, d in v if g * 2 <= gmax]
        ent = {"max_grid_threads": gmax,
               "large_launches": {"count": len(big), "mean_us": round(sum(big) / len(big), 2), "min_us": round(min(big), 2),
                                  "max_us": round(max(big), 2)}}
        if small:
            ent["small_launches"] = {"count": len(small), "mean_us": round(sum(small) / len(small), 2),
                                     "min_us": round(min(small), 2), "max_us": round(max(small), 2)}
        out[k] = ent
    json.dump(out, open(dst, "w"), indent=1)
    for k in out:
        if k.startswith(("update_phi", "update_pi", "beta_grads", "sum_partials", "ppx_")):
            print(k, out[k])


if __name__ == "__main__":
    main()
