cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/r05_small_frame_ab.log
timeout 600 python -m pytest tests/test_pointwise_thin.py tests/test_lres_models.py tests/test_conv3d_frames.py -m gpu -q --no-header -x 2>&1 | grep -v "^\[W\|Gloo\|amdgpu.ids" | tail -6 | tee gpurun_out/r05_small_frame_tests.log
for cfg in "1 64" "0 64" "1 48" "1 64" "0 64" "1 48"; do
  set -- $cfg
  LVG_SMALL_FRAME_WGRAD=$1 LVG_HAND_CONV_MIN_TILES=$2 timeout 400 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('SMALL_FRAME_WGRAD=$1 MIN_TILES=$2', 'step', d['ms_per_step'], 'ms', d['value'], 'frames/s')
" | tee -a gpurun_out/r05_small_frame_ab.log
done
