# Graph mode of the two trainers: tests, then eager against graph replay in the same process.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_trainer_gpu.py tests/test_train_sres.py -q -m gpu --no-header -rf > gpurun_out/r04_graph_trainer_test.log 2>&1; tail -15 gpurun_out/r04_graph_trainer_test.log
timeout 600 python tools/train_sres_step_time.py 3 > gpurun_out/r04_train_sres_step_time.log 2>&1; tail -4 gpurun_out/r04_train_sres_step_time.log
