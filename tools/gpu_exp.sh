# development aid: GPU tests of the loop + benches of the three single-GPU workloads on one box
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-f}
timeout -k 10 900 python -m pytest tests/test_gpu_graph_loop.py tests/test_gpu_learner.py tests/test_gpu_reference_assertions.py -x -q -m gpu > gpurun_out/r2_${T}_tests.log 2>&1 || { tail -40 gpurun_out/r2_${T}_tests.log; exit 1; }
tail -3 gpurun_out/r2_${T}_tests.log
summ() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
sc=d['step_classes'] or {}
print(sys.argv[2], 'edges/s %.3e' % d['value'], 'ms/step', round(d['ms_per_step'],4), 'enq', round(d['host_enqueue_ms_per_step'],4),
      'nonlink', sc.get('nonlink',{}).get('ms_per_step'), 'link', sc.get('link',{}).get('ms_per_step'),
      'phi_ms', d['roofline'] and d['roofline']['avg_launch_ms'], 'frac', d['roofline'] and d['roofline']['frac'])
PY
}
for wl in C1 C2 C3; do
  timeout -k 10 300 python bench.py --workload $wl --steps 600 --warmup 50 --no-cpu-baseline --cpp-dropin 0 > gpurun_out/r2_${T}_${wl}.log 2> gpurun_out/r2_${T}_${wl}.err || { tail -5 gpurun_out/r2_${T}_${wl}.err; exit 1; }
  summ gpurun_out/r2_${T}_${wl}.log "$wl graph"
done
