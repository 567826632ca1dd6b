#!/bin/bash
# gradient kernel: partial-row slots (AMMSB_BETA_SLOTS) on C3 and C2.  usage: tools/ab_beta_slots.sh   (GPU box)
mkdir -p gpurun_out/r03/ab_slots
for w in C3 C2; do
  st=20; wu=5; [ $w = C2 ] && st=2000 && wu=200
  for rep in 1 2; do for sl in 2048 3072 4096; do
    AMMSB_BETA_SLOTS=$sl python bench.py --workload $w --steps $st --warmup $wu --no-cpu-baseline --cpp-dropin 0 --extras 0 > gpurun_out/r03/ab_slots/${w}_$sl.json 2>/dev/null || echo failed
    python - gpurun_out/r03/ab_slots/${w}_$sl.json "$w slots=$sl rep=$rep" <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); k = b["roofline"]["kernels"]
print(sys.argv[2], "ms/step %.4f nonlink %.4f" % (b["ms_per_step"], b["step_classes"]["nonlink"]["ms_per_step"]), {n: v.get("avg_ms") for n, v in k.items() if isinstance(v, dict) and "avg_ms" in v and n != "perplexity"}, flush=True)
PY
  done; done
done
