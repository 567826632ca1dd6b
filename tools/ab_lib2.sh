#!/bin/bash
# Same-box A/B of two builds of libammsb_hip.so on C2 (wg 64 and 32) and C3: tools/ab_lib2.sh <other_lib.so> [out_dir]
other=$1; out=${2:-gpurun_out/r03/ab_lib2}
mkdir -p "$out"
for rep in 1 2; do for which in head other; do
  lib=""; [ $which = other ] && lib="$other"
  for cfg in "C2 64 2000 200" "C2 32 2000 200" "C3 64 20 5"; do
    set -- $cfg
    AMMSB_HIP_LIB=$lib python bench.py --workload $1 --steps $3 --warmup $4 --no-cpu-baseline --cpp-dropin 0 --extras 0 --phi-wg $2 --beta-wg $2 --ppx-wg $2 > "$out/${which}_$1_$2_$rep.json" 2>/dev/null || echo failed
    python - "$out/${which}_$1_$2_$rep.json" "$which $1 wg=$2 rep=$rep" <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
k = b["roofline"]["kernels"]
print(sys.argv[2], "ms/step %.4f" % b["ms_per_step"], "nonlink %.4f link %.4f" % (b["step_classes"]["nonlink"]["ms_per_step"], b["step_classes"]["link"]["ms_per_step"]),
      "phi %.4f ms (%s) frac %.3f" % (k["update_phi"]["avg_ms"], k["update_phi"]["kernel"][:40], k["update_phi"]["frac"]), flush=True)
PY
  done
done; done
