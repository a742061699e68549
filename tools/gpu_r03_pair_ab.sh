#!/bin/bash
# r03: pixel-pair path of the first discriminator block + 1x1 weight gradients as GEMMs: parity, then same-call A/B of the default step
mkdir -p gpurun_out
export PYTHONPATH=long-video-gan_amd
( timeout 900 python -m pytest tests/test_conv3d_frames.py tests/test_lres_models.py tests/test_trainer_gpu.py -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/r03_pair_tests.log
tail -4 gpurun_out/r03_pair_tests.log
{
for cfg in "1 1" "0 1" "1 0" "1 1" "0 0"; do
  set -- $cfg
  LVG_HAND_PAIR=$1 LVG_POINTWISE_WGRAD_GEMM=$2 timeout 300 python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r03_bench_pair_$1_$2.log 2>gpurun_out/r03_bench_pair.err
  echo "HAND_PAIR=$1 WGRAD_GEMM=$2: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r03_bench_pair_$1_$2.log | head -1) $(grep -o '"step_ms_in_custom_ops": [0-9.]*' gpurun_out/r03_bench_pair_$1_$2.log)"
done
} 2>&1 | tee gpurun_out/r03_pair_ab.log
tail -3 gpurun_out/r03_bench_pair.err
