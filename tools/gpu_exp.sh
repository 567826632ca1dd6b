# development aid: new GPU tests + wave-slot A/B (ytab in LDS vs in global memory) on one box
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_setbuild.py tests/test_gpu_reference_assertions.py tests/test_gpu_graph_loop.py tests/test_gpu_configs.py -x -q -m gpu -s > gpurun_out/r2_e_tests.log 2>&1 || { tail -40 gpurun_out/r2_e_tests.log; exit 1; }
grep -n "device cuckoo build\|passed\|failed" gpurun_out/r2_e_tests.log
python tools/occ.py 2>/dev/null
AMMSB_HIP_LIB=$PWD/build_tools/libammsb_hip_ytablds.so python tools/occ.py 2>/dev/null
summ() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
sc=d['step_classes'] or {}
print(sys.argv[2], 'edges/s %.3e' % d['value'], 'ms/step', round(d['ms_per_step'],4),
      'nonlink', sc.get('nonlink',{}).get('ms_per_step'), 'phi_ms', d['roofline'] and d['roofline']['avg_launch_ms'], 'frac', d['roofline'] and d['roofline']['frac'])
PY
}
for rep in 1 2 3; do
  for v in new ytablds; do
    if [ $v = new ]; then unset AMMSB_HIP_LIB; else export AMMSB_HIP_LIB=$PWD/build_tools/libammsb_hip_$v.so; fi
    timeout -k 10 300 python bench.py --workload C3 --steps 200 --warmup 20 --no-cpu-baseline --cpp-dropin 0 > gpurun_out/r2_e_C3_$v.log 2>/dev/null || exit 1
    summ gpurun_out/r2_e_C3_$v.log "C3 $v rep$rep"
  done
done
