#!/bin/bash
# r05: where the waves of the row-band filtered_lrelu kernel wait: LDS conflicts, pipe-busy and wait counters (one rocprofv3 --pmc pass each).
#   bash tools/gpu_r05_band_pmc2.sh <tag> <case> <dtype> <mode> <impl>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-a}; CASE=${2:-L8}; DT=${3:-1}; MODE=${4:-0}; IMPL=${5:-4}
OUT=gpurun_out/pmc2_${tag}_${CASE}_${MODE}_${IMPL}; mkdir -p $OUT
run() { timeout 120 rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -o p -- tools/bin/flrelu_check one $CASE $DT $MODE $IMPL 2 > $OUT/$1.log 2>&1 || echo "pass $1 failed: $(tail -2 $OUT/$1.log)"; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VALU"
run c "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_WAIT_IFETCH GRBM_GUI_ACTIVE SQ_INSTS_SALU"
python tools/pmc_summary.py filtered_lrelu $OUT > $OUT/summary.csv
find $OUT -name "*.csv" ! -name summary.csv -delete; find $OUT -name "*.db" -delete
echo "== $CASE mode $MODE impl $IMPL"; cat $OUT/summary.csv
