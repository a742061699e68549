cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for filt in "64->64@36x64k1" "128->128@18x32k1" "64->64@32x32k5"; do
echo "== $filt abl=0: $(timeout 100 python tools/conv_bench.py 5 "$filt" 2>&1 | grep hand | cut -c30-60)"
for abl in $ABLS; do
  echo "== $filt abl=$abl: $(LVG_HIP_LIB=$PWD/long-video-gan_amd/lib/variant_conv_abl$abl.so timeout 100 python tools/conv_bench.py 5 "$filt" 2>&1 | grep hand | cut -c30-60)"
done; done 2>&1 | tee gpurun_out/r02_conv_abl_smallk.log
