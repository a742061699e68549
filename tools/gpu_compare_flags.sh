# One gpurun call comparing the opt-in step variants of lvg.models.lres on the default bench workload.
# Usage: gpurun --timeout 900 -- 'bash tools/gpu_compare_flags.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {   # name, env...
    name=$1; shift
    env "$@" timeout -s INT 240 python bench.py --no-cpu-baseline > gpurun_out/flags_$name.log 2>&1
    echo "$name: $(grep -o '"value": [0-9.]*' gpurun_out/flags_$name.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/flags_$name.log | head -1)"
}
run base        LVG_TAP_STACK=0 LVG_POINTWISE_GEMM=0
run tap         LVG_TAP_STACK=1 LVG_POINTWISE_GEMM=0
run tap_gemm    LVG_TAP_STACK=1 LVG_POINTWISE_GEMM=1
run gemm        LVG_TAP_STACK=0 LVG_POINTWISE_GEMM=1
# keep whatever MIOpen recorded for the new shapes
mkdir -p gpurun_out/miopen_db && cp -r long-video-gan_amd/miopen_db/* gpurun_out/miopen_db/
