#!/bin/bash
# Measurement builds of liblvg_hip.so with filtered_lrelu_wave.hip compiled under extra -D flags.
#   tools/build_flrelu_variants.sh name1:"-DLVG_WAVE_LATE_XWRITE=1" name2:"-DLVG_WABL=2" ...  -> long-video-gan_amd/lib/variant_<name>.so
set -e
cd "$(dirname "$0")/../long-video-gan_amd/csrc"
make -s
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-honor-nans $flags -c filtered_lrelu_wave.hip -o ../build/variant_flw_$name.o_
  objs=$(ls ../build/*.o | grep -v filtered_lrelu_wave)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../build/variant_flw_$name.o_ -o ../lib/variant_$name.so
  echo "built variant_$name.so ($flags)"
done
