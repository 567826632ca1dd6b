#!/bin/bash
# Same-box A/B of the non-temporal hint on update_phi's neighbour-row requests: AMMSB_PHI_NT = 1 / 0, on C2 (pi = 100 MB,
# fits the 256 MB Infinity Cache) and C3 (pi = 4 GB).  usage: tools/ab_nt.sh [out_dir]   (run on the GPU box)
out=${1:-gpurun_out/r03/ab_nt}
mkdir -p "$out"
for w in C2 C3; do
  steps=2000; warm=200; [ $w = C3 ] && steps=20 && warm=5
  for rep in 1 2; do for r in 1 0; do
    AMMSB_PHI_NT=$r python bench.py --workload $w --steps $steps --warmup $warm --no-cpu-baseline --cpp-dropin 0 --extras 0 > "$out/${w}_nt${r}_$rep.json" 2>/dev/null || echo "failed $w nt=$r"
    python - "$out/${w}_nt${r}_$rep.json" "$w nt=$r rep=$rep" <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
k = b["roofline"]["kernels"]
print(sys.argv[2], "ms/step %.4f" % b["ms_per_step"], "nonlink %.4f link %.4f" % (b["step_classes"]["nonlink"]["ms_per_step"], b["step_classes"]["link"]["ms_per_step"]),
      "phi %.4f ms (%s) frac %.3f" % (k["update_phi"]["avg_ms"], k["update_phi"]["kernel"][:40], k["update_phi"]["frac"]), flush=True)
PY
  done; done
done
