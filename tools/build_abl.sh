#!/bin/bash
# Dev aid: variant builds of one kernel source into tools/abl/ for tools/fused_bench.py (LSAM_LIB=...).
# usage: tools/build_abl.sh SRC NAME "-DFLAG=.." [NAME2 "flags2" ...]     (SRC: tailblock, mlpblock, winblock, ...)
set -e
cd "$(dirname "$0")/../image_segmentation_amd/csrc"
mkdir -p build_abl ../../tools/abl
src=$1; shift
objs=""
for o in gemm norm_elem attention winblock mlpblock tailblock patchembed lateral neck heads heads_bwd kvattn qkvattn hiera capi; do
  [ $o = $src ] || objs="$objs build/$o.o"
done
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags -c $src.hip -o build_abl/${src}_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/abl/liblsam_$name.so $objs build_abl/${src}_$name.o
done
