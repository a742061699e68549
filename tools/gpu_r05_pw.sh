cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/r05_pointwise_ab.log
timeout 600 python -m pytest tests/test_pointwise_thin.py tests/test_lres_models.py tests/test_conv3d_frames.py -m gpu -q --no-header -x 2>&1 | grep -v "^\[W\|Gloo\|amdgpu.ids" | tail -8 | tee gpurun_out/r05_pointwise_tests.log
for cfg in "1 1" "0 0" "1 0" "1 1" "0 0"; do
  set -- $cfg
  LVG_POINTWISE_WGRAD_HAND=$1 LVG_POINTWISE_HAND_ALL=$2 LVG_BENCH_LEGS=train_lres timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('WGRAD_HAND=$1 HAND_ALL=$2', 'step', d['ms_per_step'], 'ms', d['value'], 'frames/s; train_lres', d['train_lres'].get('ms_per_step'), 'ms', d['train_lres'].get('error', ''))
" | tee -a gpurun_out/r05_pointwise_ab.log
done
