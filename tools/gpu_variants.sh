cd $GRAFT_REPO_ROOT
timeout 100 tools/bin/flrelu_check check > gpurun_out/variants_check.log 2>&1; grep -c OK gpurun_out/variants_check.log; grep -E "FAIL|check:" gpurun_out/variants_check.log | head -5
for v in long-video-gan_amd/lib/variant_*.so; do echo "== $v"; LVG_LIB=$PWD/$v timeout 60 tools/bin/flrelu_check time 2>&1 | grep "impl=MFMA" | grep -E "^L8|^L10 .*bwd|^L13 .*fwd "; done
