cd $GRAFT_REPO_ROOT
for v in long-video-gan_amd/lib/variant_abl*.so; do echo "== $v"; for m in 0 1 2; do LVG_LIB=$PWD/$v timeout 60 tools/bin/flrelu_check one L8 1 $m 2 10 2>&1 | grep "impl=MFMA" | cut -c1-60; done; done
