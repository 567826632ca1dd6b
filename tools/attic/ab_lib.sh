#!/bin/bash
# Same-box A/B of two builds of the device library on a bench workload: tools/ab_lib.sh WORKLOAD OLD.so [reps] [extra bench flags]
W=$1; OLD=$2; REPS=${3:-3}; shift 3
for rep in $(seq $REPS); do
  for tag in old new; do
    if [ $tag = old ]; then export AMMSB_HIP_LIB=$OLD; else unset AMMSB_HIP_LIB; fi
    python bench.py --workload $W --steps ${STEPS:-1500} --warmup 100 --no-cpu-baseline --cpp-dropin 0 --extras 0 "$@" > /tmp/ab_$tag.json 2>/dev/null
    python - /tmp/ab_$tag.json "$W $tag rep=$rep" <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
k = b["roofline"]["kernels"]
ls = k.get("link_steps_ms", {})
print(sys.argv[2], "ms/step %.4f" % b["ms_per_step"], "nonlink %.4f link %.4f" % (b["step_classes"]["nonlink"]["ms_per_step"], b["step_classes"]["link"]["ms_per_step"]),
      "| phi %.4f" % k["update_phi"]["avg_ms"], " ".join("%s %.4f" % (n, v["avg_ms"]) for n, v in k.items() if isinstance(v, dict) and "avg_ms" in v and n not in ("update_phi", "perplexity")),
      "| link phi %.4f grads %.4f" % (ls.get("update_phi", 0), ls.get("beta_grads", 0)), "ppx %.4f" % (b["ppx_eval_ms"] or 0), flush=True)
PY
  done
done
