#!/bin/bash
# Same-box A/B of the short-row update_phi forms on C2: AMMSB_PHI_PAIR = 0 (one node per wave) / 1 / 2 / 3
# (two nodes per wave: ring 4 x 2 rows, ring 8 x 2, ring 8 x 4) at the work-group sizes 64 and 32.
# usage: tools/ab_c2.sh [out_dir]   (run on the GPU box)
out=${1:-gpurun_out/r03/ab_c2}
mkdir -p "$out"
for rep in 1 2; do
  for wg in 64 32; do
    for pair in 0 1 2 3; do
      AMMSB_PHI_PAIR=$pair python bench.py --workload C2 --steps 2000 --warmup 200 --no-cpu-baseline --cpp-dropin 0 --extras 0 \
        --phi-wg $wg --beta-wg $wg --ppx-wg $wg > "$out/c2_wg${wg}_pair${pair}_$rep.json" 2> "$out/c2_wg${wg}_pair${pair}_$rep.err" || echo "failed wg=$wg pair=$pair"
      python - "$out/c2_wg${wg}_pair${pair}_$rep.json" "wg=$wg pair=$pair rep=$rep" <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
k = b["roofline"]["kernels"]
print(sys.argv[2], "ms/step %.4f" % b["ms_per_step"], "nonlink %.4f link %.4f" % (b["step_classes"]["nonlink"]["ms_per_step"], b["step_classes"]["link"]["ms_per_step"]),
      "phi %.4f ms (%s) frac %.3f" % (k["update_phi"]["avg_ms"], k["update_phi"]["kernel"], k["update_phi"]["frac"]), flush=True)
PY
    done
  done
done
