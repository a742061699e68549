cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_lres_models.py tests/test_trainer_gpu.py -m gpu -q -x --no-header -rf 2>&1 | tail -4
for f in 0 1; do
  echo "== LVG_SIDE_STREAM_TERMS=$f"
  LVG_SIDE_STREAM_TERMS=$f timeout 300 python bench.py --no-cpu-baseline --no-extra-legs 2>&1 | tail -1 | cut -c1-330
done
echo "== eager, side=1"
LVG_SIDE_STREAM_TERMS=1 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --graph off 2>&1 | tail -1 | cut -c1-200
