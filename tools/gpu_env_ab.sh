# A/B of environment switches on the default bench line (no extra legs): bash tools/gpu_env_ab.sh "VAR=1" "VAR2=0" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "LVG_NONE=0" "$@" "LVG_NONE=0"; do
  env $v timeout 600 python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/ab.log 2> gpurun_out/ab.err || tail -3 gpurun_out/ab.err
  echo "$v: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab.log)"
done
