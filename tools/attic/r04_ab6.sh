#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
AMMSB_PHI_LDS3=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_golden.py -x -q -m gpu -k "phi or golden or fullsize or c3" > gpurun_out/r04/gputests6.log 2>&1 || { tail -40 gpurun_out/r04/gputests6.log; exit 1; }
tail -3 gpurun_out/r04/gputests6.log
for i in 1 2 3; do
for f in 0 1; do
  AMMSB_PHI_LDS3=$f timeout -k 10 300 python bench.py --steps 100 --warmup 10 --large none --extras 0 --no-cpu-baseline --cpp-dropin 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('lds3=$f', round(d['value']/1e6,3), d['step_classes']['nonlink']['ms_per_step'], r['kernel'], {k:v.get('avg_ms') for k,v in r['kernels'].items() if isinstance(v,dict) and 'avg_ms' in v})" || exit 1
done
done
AMMSB_HIP_LIB=$PWD/tools/ab/trace/libammsb_hip_trace.so AMMSB_PHI_LDS3=1 SPREAD_WARM=100 timeout -k 10 200 python tools/phi_spread.py A > gpurun_out/r04/spread_lds3.jsonl 2>/dev/null
python - <<'PY'
import json
for l in open('gpurun_out/r04/spread_lds3.jsonl'):
    d=json.loads(l)
    if 'variant' in d:
        b=d['blocks']; print(d['kernel'], d['ms_median'], 'block us', b['block_us_median'], 'cycles', b['block_cycles_median'], 'mhz', b['mhz'])
PY
