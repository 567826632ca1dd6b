#!/bin/bash
# SQ counters of the K = 256 update_phi kernel (C2): how many vector instructions a row costs and how busy the
# vector pipes are -- the kernel sits at 0.52 of the HBM roofline; is it waiting for memory or issuing?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_c2; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O -o p -- python3 bench.py --workload C2 --steps 60 --warmup 10 --no-cpu-baseline --cpp-dropin 0 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_c2/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "update_phi" in r["Kernel_Name"] and int(r["Grid_Size"]) > 100000:
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k, {n: round(v) for n, v in m.items()})
    waves = m.get("SQ_WAVES", 0)
    if waves:
        print("  VALU instructions per wave %.0f = %.1f per neighbour row (33 rows)" % (m["SQ_INSTS_VALU"] / waves, m["SQ_INSTS_VALU"] / waves / 33))
        print("  LDS instructions per wave %.0f" % (m["SQ_INSTS_LDS"] / waves))
        print("  wave cycles: active-VALU %.2f, wait_inst_any %.2f, wait_any %.2f of SQ_WAVE_CYCLES" % (
            m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]))
        print("  GRBM_GUI_ACTIVE (sum over 8 XCDs) %.0f" % m["GRBM_GUI_ACTIVE"])
PY
