# One `rocprofv3 --kernel-trace --stats` per-kernel table per bench leg -> gpurun_out/${ROUND}_window_<leg>.csv (whole-process tables of a process that
# runs just that leg's step: warm-up iterations included, the roofline replays and the CPU leg excluded).
R=${ROUND:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
prof() {   # name, command...
    local name=$1; shift
    rm -rf gpurun_out/prof_leg
    LVG_BENCH_NO_ROOFLINE=1 timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_leg -o leg -- "$@" > gpurun_out/${R}_window_${name}.run.log 2>&1
    echo "$name rc=$?"
    find gpurun_out/prof_leg -name "*kernel_stats.csv" -exec cp {} gpurun_out/${R}_window_${name}.csv \;
    rm -rf gpurun_out/prof_leg
    head -8 gpurun_out/${R}_window_${name}.csv | cut -c1-160
}
prof main python bench.py --no-extra-legs --no-cpu-baseline
prof fp32 python bench.py --no-extra-legs --no-cpu-baseline --dtype fp32 --steps 3 --warmup 1
prof sres python tools/sres_step.py 3
prof train_lres python tools/train_step_time.py 32 4 2
prof train_sres python tools/train_sres_step_time.py 2
