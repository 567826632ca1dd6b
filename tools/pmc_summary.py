#!/usr/bin/env python3
"""Summarise the two rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE are collected in separate runs:
the TCC counters do not fit one pass, MI355X_MICROARCH.md) into profiles/<tag>_bench_pmc_summary.json and
profiles/phi_traffic.json (read by bench.py for roofline.traffic).

    python tools/pmc_summary.py gpurun_out/r01b/fetch/f_counter_collection.csv \
                                gpurun_out/r01b/write/w_counter_collection.csv r01

Per kernel the mean is taken over its largest-grid launches (the non-link mini-batches).  Units: rocprofv3
reports both counters in KB; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes, so read bytes are
2 x FETCH_SIZE (same guide, HBM section)."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def per_kernel(path, counter):
    rows = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            rows[short(r["Kernel_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    out = {}
    for k, v in rows.items():
        g = max(x for x, _ in v)
        big = [c for x, c in v if x == g]
        out[k] = (sum(big) / len(big), len(big), g)
    return out


def main():
    fetch_csv, write_csv, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    summary = {}
    for k in sorted(set(f) | set(w)):
        summary[k] = {"FETCH_SIZE_KB_big_launch": f.get(k, (None,))[0], "WRITE_SIZE_KB_big_launch": w.get(k, (None,))[0],
                      "big_launches": f.get(k, (0, 0))[1], "grid_size": f.get(k, (0, 0, 0))[2]}
        if f.get(k) and w.get(k):
            summary[k]["hbm_bytes_big_launch"] = 2 * f[k][0] * 1024 + w[k][0] * 1024
    json.dump(summary, open(os.path.join(ROOT, "profiles", "%s_bench_pmc_summary.json" % tag), "w"), indent=1)
    phi = [k for k in summary if k.startswith("update_phi")]
    if phi:
        k = max(phi, key=lambda x: summary[x].get("hbm_bytes_big_launch", 0))
        K, n, nodes = 1024, 32, 65537
        doc = {"kernel": k,
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 20 "
                         "--warmup 4 --no-cpu-baseline`, mean over the non-link (65537-node) launches",
               "FETCH_SIZE_bytes_raw": summary[k]["FETCH_SIZE_KB_big_launch"] * 1024,
               "WRITE_SIZE_bytes": summary[k]["WRITE_SIZE_KB_big_launch"] * 1024,
               "correction": "gfx950 FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM): read bytes = 2 x FETCH_SIZE",
               "hbm_bytes_per_launch": summary[k]["hbm_bytes_big_launch"],
               "algorithmic_bytes_per_launch": (4 * K * (n + 2) + 68 * n + 8) * nodes}
        json.dump(doc, open(os.path.join(ROOT, "profiles", "phi_traffic.json"), "w"), indent=1)
        print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
