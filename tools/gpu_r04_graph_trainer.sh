cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_trainer_gpu.py -x -q -m gpu > gpurun_out/r04_graph_trainer_test.log 2>&1; tail -15 gpurun_out/r04_graph_trainer_test.log
timeout 600 python tools/train_step_time.py 32 4 4 > gpurun_out/r04_train_step_time.log 2>&1; tail -8 gpurun_out/r04_train_step_time.log
timeout 300 python tools/train_step_time.py 4 1 6 >> gpurun_out/r04_train_step_time.log 2>&1; tail -3 gpurun_out/r04_train_step_time.log
