#!/bin/bash
# Same-box A/B of the persistent streaming update_phi (K = 256) on C2: AMMSB_PHI_STREAM = 0 / 1, work-group sizes 64 and 32.
# usage: tools/ab_stream.sh [out_dir]   (run on the GPU box)
out=${1:-gpurun_out/r03/ab_stream}
mkdir -p "$out"
for rep in 1 2; do
  for wg in 64 32; do
    for st in 0 1; do
      AMMSB_PHI_STREAM=$st python bench.py --workload C2 --steps 2000 --warmup 200 --no-cpu-baseline --cpp-dropin 0 --extras 0 \
        --phi-wg $wg --beta-wg $wg --ppx-wg $wg > "$out/c2_wg${wg}_stream${st}_$rep.json" 2> "$out/c2_wg${wg}_stream${st}_$rep.err" || echo "failed wg=$wg stream=$st"
      python - "$out/c2_wg${wg}_stream${st}_$rep.json" "wg=$wg stream=$st rep=$rep" <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
k = b["roofline"]["kernels"]
print(sys.argv[2], "ms/step %.4f" % b["ms_per_step"], "nonlink %.4f link %.4f" % (b["step_classes"]["nonlink"]["ms_per_step"], b["step_classes"]["link"]["ms_per_step"]),
      "phi %.4f ms (%s) frac %.3f" % (k["update_phi"]["avg_ms"], k["update_phi"]["kernel"][:40], k["update_phi"]["frac"]), flush=True)
PY
    done
  done
done
