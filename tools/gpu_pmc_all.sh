cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc_flrelu.sh L8 1 1 2 r02_flrelu_mfma_L8_u2d2_fwd_write > /dev/null 2>&1
bash tools/gpu_pmc_flrelu.sh L8 1 2 2 r02_flrelu_mfma_L8_u2d2_bwd_read > /dev/null 2>&1
bash tools/gpu_pmc_flrelu.sh L8 1 0 2 r02_flrelu_mfma_L8_u2d2_fwd_nomask > /dev/null 2>&1
bash tools/gpu_pmc_flrelu.sh L10 1 1 2 r02_flrelu_mfma_L10_u4d2_fwd_write > /dev/null 2>&1
bash tools/gpu_pmc_flrelu.sh L10 1 2 2 r02_flrelu_mfma_L10_u2d4_bwd_read > /dev/null 2>&1
bash tools/gpu_pmc_flrelu.sh L13 1 1 2 r02_flrelu_mfma_L13_u2d2_fwd_write > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_flrelu_stats -o t -- tools/bin/flrelu_check time > gpurun_out/r02_flrelu_time_under_rocprof.log 2>&1
cp $(find gpurun_out/r02_flrelu_stats -name "*kernel_stats.csv" | head -1) gpurun_out/r02_flrelu_check_time_kernel_stats.csv
rm -rf gpurun_out/r02_flrelu_stats
timeout 100 tools/bin/flrelu_check time > gpurun_out/r02_flrelu_time.log 2>&1
for d in gpurun_out/r02_flrelu_mfma_*; do echo "== $d"; grep -E "SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_MFMA|SQ_VALU_MFMA_BUSY|SQ_WAVE_CYCLES|SQ_WAIT_ANY|FETCH_SIZE|WRITE_SIZE|SQ_BUSY_CYCLES|GRBM_GUI" $d/summary.csv | tr '\n' ' '; echo; done
cat gpurun_out/r02_flrelu_time.log
