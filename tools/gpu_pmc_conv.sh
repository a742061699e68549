# PMC passes over the hand-written convolution on one shape: bash tools/gpu_pmc_conv.sh <shape filter> <tag>   (env LVG_CONV_* select the variant)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FILT=${1:-80x512->512}; TAG=${2:-r02_conv_pmc}
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { timeout 150 rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -o p -- python tools/conv_bench.py 3 "$FILT" > $OUT/$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"
run c "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"
run d "FETCH_SIZE"
run e "WRITE_SIZE"
python tools/pmc_summary.py conv3d_igemm $OUT > $OUT/summary.csv
find $OUT -name "*.csv" ! -name summary.csv -delete; find $OUT -name "*.db" -delete
cat $OUT/summary.csv; tail -2 $OUT/a.log
