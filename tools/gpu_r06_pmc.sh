#!/bin/bash
# r06: issue / wait counters of one filtered_lrelu launch (one rocprofv3 --pmc pass each; LVG_LIB selects an ablation build).
#   bash tools/gpu_r06_pmc.sh <tag> <case> <dtype> <mode> <impl> [variant]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-a}; CASE=${2:-L8}; DT=${3:-1}; MODE=${4:-0}; IMPL=${5:-5}; VAR=${6:-}
[ -n "$VAR" ] && export LVG_LIB=$PWD/long-video-gan_amd/lib/variant_$VAR.so
OUT=gpurun_out/r06_pmc_${tag}_${CASE}_${MODE}_${IMPL}${VAR:+_$VAR}; mkdir -p $OUT
run() { timeout 120 rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -o p -- tools/bin/flrelu_check one $CASE $DT $MODE $IMPL 2 > $OUT/$1.log 2>&1 || echo "pass $1 failed: $(tail -2 $OUT/$1.log)"; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VALU"
run c "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
python tools/pmc_summary.py filtered_lrelu $OUT > $OUT/summary.csv
find $OUT -name "*.csv" ! -name summary.csv -delete; find $OUT -name "*.db" -delete
echo "== $CASE mode $MODE impl $IMPL $VAR"; cat $OUT/summary.csv
