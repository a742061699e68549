cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_modconv_epilogue.py tests/test_lres_models.py tests/test_trainer_gpu.py tests/test_pixel_path.py -m gpu -q --no-header -rf -x > gpurun_out/r02_boundary_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02_boundary_tests.log
tail -12 gpurun_out/r02_boundary_tests.log
for v in 1 0 1; do
  LVG_FUSE_BOUNDARY=$v timeout 300 python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r02_bench_boundary_$v.log 2>&1
  echo "FUSE_BOUNDARY=$v: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r02_bench_boundary_$v.log | tr '\n' ' ')" | tee -a gpurun_out/r02_boundary_ab.log
done
LVG_FUSE_BOUNDARY=1 timeout 300 python bench.py --no-extra-legs --no-cpu-baseline --forward-only 2>&1 | grep -o '"value": [0-9.]*' | head -1
LVG_FUSE_BOUNDARY=0 timeout 300 python bench.py --no-extra-legs --no-cpu-baseline --forward-only 2>&1 | grep -o '"value": [0-9.]*' | head -1
