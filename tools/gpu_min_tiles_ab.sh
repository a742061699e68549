# The 3 x 4-pixel layers (72 tiles) on the hand-written convolution (LVG_HAND_CONV_MIN_TILES=64) against the library route (default 128), same call;
# and the trainer tests.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_trainer_gpu.py -m gpu -q --no-header -rf > gpurun_out/r04_trainer_tests.log 2>&1; tail -4 gpurun_out/r04_trainer_tests.log
for v in 128 64 128 64; do
  LVG_HAND_CONV_MIN_TILES=$v timeout 300 python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r04_bench_min_tiles_$v.log 2>&1
  echo "min tiles $v: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r04_bench_min_tiles_$v.log | head -1)" | tee -a gpurun_out/r04_min_tiles_ab.log
done
