# weight-gradient split-K: the shipped rule (one full round of workgroups, rounded down) against fixed workgroup targets
# (LVG_WGRAD_TARGET; splits = ceil(target / tiles)); timing incl. the range sum
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== shipped rule"; timeout 200 python tools/wgrad_bench.py 5 2>&1 | grep "hand" | cut -c1-82
echo "== LVG_WGRAD_TARGET=1024"; LVG_WGRAD_TARGET=1024 timeout 200 python tools/wgrad_bench.py 5 2>&1 | grep "hand" | cut -c1-82
} 2>&1 | tee gpurun_out/r02_wgrad_splits2.log
timeout 300 python -m pytest tests/test_conv3d_frames.py -m gpu -q --no-header -x -k wgrad 2>&1 | tail -2
for v in rule 1024 rule; do
  if [ $v = 1024 ]; then export LVG_WGRAD_TARGET=1024; else unset LVG_WGRAD_TARGET; fi
  timeout 300 python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r02_bench_wsplit_$v.log 2>&1
  echo "$v: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r02_bench_wsplit_$v.log | tr '\n' ' ')" | tee -a gpurun_out/r02_wgrad_splits2.log
done
