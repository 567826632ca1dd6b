#!/bin/bash
# update_phi_lds2_kernel<4, 8, 4, 64> on C2 with its residency capped by extra LDS per block (AMMSB_PHI_LDS_PAD):
# 9856 B -> 16 blocks per CU; +1024 -> 14; +3072 -> 12; +6144 -> 10; +10240 -> 8
mkdir -p gpurun_out/r03/ab_occ
for rep in 1 2; do for pad in 0 1024 3072 6144 10240; do
  AMMSB_PHI_LDS_PAD=$pad python bench.py --workload C2 --steps 2000 --warmup 200 --no-cpu-baseline --cpp-dropin 0 --extras 0 > gpurun_out/r03/ab_occ/x.json 2>/dev/null || echo failed
  python - gpurun_out/r03/ab_occ/x.json "pad=$pad rep=$rep" <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); k = b["roofline"]["kernels"]
print(sys.argv[2], "ms/step %.4f nonlink %.4f" % (b["ms_per_step"], b["step_classes"]["nonlink"]["ms_per_step"]), "phi %.4f ms frac %.3f" % (k["update_phi"]["avg_ms"], k["update_phi"]["frac"]), flush=True)
PY
done; done
