#!/bin/bash
# SQ counters of the C3 beta gradient kernel (0.106 ms for 65536 edges = 0.32 of the HBM roofline): what a wave spends
# its time on.  usage (GPU box): tools/pmc_beta.sh [workload]
W=${1:-C3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_beta_$W; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O -o p -- python3 bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --cpp-dropin 0 --extras 0 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 - "$O" <<'PY'
import csv, glob, collections, re, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if any(x in k for x in ("beta_grads", "update_pi", "update_phi", "ppx_")) and int(r["Grid_Size"]) >= 64 * 1024:
        mm = re.search(r"(\w+_kernel(<[^>]*>)?)", k)
        acc[(mm.group(1) if mm else k[:60]) + " grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    waves = m.get("SQ_WAVES", 0)
    print(k, "launches", len(c["SQ_WAVES"]), {n: round(v) for n, v in m.items()})
    if waves:
        print("  per wave: VALU %.0f, LDS %.0f; share of wave cycles: VALU active %.2f, wait_inst_any %.2f, wait_any %.2f" % (
            m["SQ_INSTS_VALU"] / waves, m["SQ_INSTS_LDS"] / waves, m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"],
            m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]))
PY
