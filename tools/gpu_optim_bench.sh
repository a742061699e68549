cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_optim.py tests/test_trainer_gpu.py tests/test_train_sres.py -m gpu -q -x --no-header -rf 2>&1 | tail -4
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs 2>&1 | tail -1 | cut -c1-330
