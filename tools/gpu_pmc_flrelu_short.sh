# Instruction-mix PMC passes over one filtered_lrelu layer: bash tools/gpu_pmc_flrelu_short.sh <L8|L10|L13> <dtype> <mode> <impl> <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CASE=${1:-L8}; DT=${2:-1}; MODE=${3:-0}; IMPL=${4:-2}; TAG=${5:-pmc}
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { timeout 120 rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -o p -- tools/bin/flrelu_check one $CASE $DT $MODE $IMPL 2 > $OUT/$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
run c "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"
python tools/pmc_summary.py filtered_lrelu $OUT > $OUT/summary.csv
find $OUT -name "*.csv" ! -name summary.csv -delete; find $OUT -name "*.db" -delete
echo "== $CASE mode $MODE impl $IMPL"; cat $OUT/summary.csv
