cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/r05_split_stack_ab.log
timeout 600 python -m pytest tests/test_conv3d_frames.py -m gpu -q --no-header -x -k "split" 2>&1 | grep -v "^\[W\|Gloo\|amdgpu.ids" | tail -4 | tee gpurun_out/r05_split_tests.log
for v in 1 0 1; do
  LVG_SPLIT_STACK_HIP=$v LVG_BENCH_LEGS=fp32 timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('LVG_SPLIT_STACK_HIP=$v fp32', d['fp32'].get('ms_per_step'), 'ms', d['fp32'].get('value'), 'frames/s', d['fp32'].get('error', ''))
" | tee -a gpurun_out/r05_split_stack_ab.log
done
timeout 600 python -m pytest tests/test_lres_models.py -m gpu -q --no-header -x 2>&1 | grep -v "^\[W\|Gloo\|amdgpu.ids" | tail -3 | tee -a gpurun_out/r05_split_tests.log
