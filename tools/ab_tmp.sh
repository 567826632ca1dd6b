for cfg in "C3 64 20 5" "C3 32 20 5" "C2 64 2000 200" "C1 64 3000 200"; do set -- $cfg
python bench.py --workload $1 --steps $3 --warmup $4 --no-cpu-baseline --cpp-dropin 0 --extras 0 --phi-wg $2 --beta-wg $2 --ppx-wg $2 > gpurun_out/r03/x.json 2>/dev/null
python - gpurun_out/r03/x.json "$1 wg=$2" <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); k = b["roofline"]["kernels"]
print(sys.argv[2], "%.4g" % b["value"], "ms/step %.4f" % b["ms_per_step"], {n: v.get("avg_ms") for n, v in k.items() if isinstance(v, dict) and "avg_ms" in v and n != "perplexity"}, "link beta", k.get("link_steps_ms", {}).get("beta_grads"), flush=True)
PY
done
