#!/bin/bash
# Same-box A/B of the K = 1024 update_phi ring depth on C3: AMMSB_PHI_RING unset (two slots) / 4, AMMSB_PHI_NB=2.
out=${1:-gpurun_out/r03/ab_c3_ring}
mkdir -p "$out"
for rep in 1 2; do
  for v in "base:" "ring4:AMMSB_PHI_RING=4" "nb2:AMMSB_PHI_NB=2"; do
    tag=${v%%:*}; kv=${v#*:}
    env $kv python bench.py --steps 20 --warmup 5 --no-cpu-baseline --cpp-dropin 0 --extras 0 > "$out/c3_${tag}_$rep.json" 2> "$out/c3_${tag}_$rep.err" || echo "failed $tag"
    python - "$out/c3_${tag}_$rep.json" "$tag rep=$rep" <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
k = b["roofline"]["kernels"]
print(sys.argv[2], "value %.4g ms/step %.4f" % (b["value"], b["ms_per_step"]), "nonlink %.4f link %.4f" % (b["step_classes"]["nonlink"]["ms_per_step"], b["step_classes"]["link"]["ms_per_step"]),
      "phi %.4f ms frac %.3f" % (k["update_phi"]["avg_ms"], k["update_phi"]["frac"]), flush=True)
PY
  done
done
