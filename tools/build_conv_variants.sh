#!/bin/bash
# Measurement builds of liblvg_hip.so with parts of the conv3d_igemm K loop compiled out (LVG_CONV_ABL bits).
set -e
cd "$(dirname "$0")/../long-video-gan_amd/csrc"
for abl in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DLVG_CONV_ABL=$abl -c conv3d_igemm.hip -o ../build/conv3d_igemm_abl$abl.o
  objs=$(ls ../build/*.o | grep -v conv3d_igemm)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../build/conv3d_igemm_abl$abl.o -o ../lib/variant_conv_abl$abl.so
done
