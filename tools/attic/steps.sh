#!/bin/bash
# per-kernel device times of the small configurations (stamps of the descriptor loop): tools/steps.sh [-w WG] [C2 C1 ...]
mkdir -p gpurun_out/r03
wgflags=""
if [ "$1" = "-w" ]; then wgflags="--phi-wg $2 --beta-wg $2 --ppx-wg $2"; shift 2; fi
for w in ${@:-C2 C1}; do
  python bench.py --workload $w --steps 3000 --warmup 200 --no-cpu-baseline --cpp-dropin 0 --extras 0 $wgflags > gpurun_out/r03/steps_$w.json 2> gpurun_out/r03/steps_$w.err || { tail -n 5 gpurun_out/r03/steps_$w.err; exit 1; }
  python - gpurun_out/r03/steps_$w.json "$w $wgflags" <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); k = b["roofline"]["kernels"]
print(sys.argv[2], "%.4g edges/s" % b["value"], "ms/step %.4f" % b["ms_per_step"], {n: v.get("avg_ms") for n, v in k.items() if isinstance(v, dict) and "avg_ms" in v})
print("   link steps:", k.get("link_steps_ms"))
PY
done
