#!/bin/bash
# Measurement builds of liblvg_hip.so with one filtered_lrelu source compiled under extra -D flags.
#   [SRC=filtered_lrelu_band] tools/build_flrelu_variants.sh name1:"-DLVG_BABL=32" name2:"-DLVG_WABL=2" ...  -> long-video-gan_amd/lib/variant_<name>.so
# SRC defaults to filtered_lrelu_wave (the round-4 kernel).
set -e
SRC=${SRC:-filtered_lrelu_wave}
cd "$(dirname "$0")/../long-video-gan_amd/csrc"
make -s
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-honor-nans $flags -c $SRC.hip -o ../build/variant_$name.o_
  objs=$(ls ../build/*.o | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../build/variant_$name.o_ -o ../lib/variant_$name.so
  echo "built variant_$name.so ($SRC: $flags)"
done
