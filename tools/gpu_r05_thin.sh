cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pointwise_thin.py tests/test_lres_models.py tests/test_trainer_gpu.py -m gpu -q --no-header -x 2>&1 | tail -12 > gpurun_out/r05_thin_tests.log; cat gpurun_out/r05_thin_tests.log
for v in 1 0 1 0; do
  LVG_THIN_POINTWISE=$v LVG_BENCH_LEGS=forward_only,train_lres timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print({k: v for k, v in d['train_lres'].items() if k in ('error',)}, 'LVG_THIN_POINTWISE=$v', 'step', d['ms_per_step'], 'ms', d['value'], 'frames/s; forward_only', d['forward_only']['ms_per_step'], 'ms; train_lres', d['train_lres'].get('ms_per_step'), 'ms')
" | tee -a gpurun_out/r05_thin_ab.log
done
