#!/bin/bash
# same-box A/B of two builds of the device library (AMMSB_HIP_LIB): update_phi's per-launch time and the step time
# usage: tools/ab_phi.sh WORKLOAD STEPS [pairs]
W=${1:-C2}; S=${2:-1000}; P=${3:-3}
O=gpurun_out/ab; mkdir -p $O
for i in $(seq 1 $P); do
  for lib in prev new; do
    if [ $lib = prev ]; then export AMMSB_HIP_LIB=$PWD/tools/ab/libammsb_hip_prev.so; else unset AMMSB_HIP_LIB; fi
    timeout -k 10 400 python bench.py --workload $W --steps $S --warmup 100 --no-cpu-baseline --cpp-dropin 0 > $O/${lib}_${W}_$i.log 2>&1 || exit 1
    tail -1 $O/${lib}_${W}_$i.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib $W', 'ms/step %.4f' % d['ms_per_step'], 'phi launch ms %.4f' % r['avg_launch_ms'], 'frac %.3f' % r['frac'], {k:round(v['ms_per_step'],4) for k,v in d['step_classes'].items() if isinstance(v,dict)})"
  done
done
