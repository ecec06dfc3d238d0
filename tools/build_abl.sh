#!/bin/bash
# Dev aid: variant builds of the tail kernel into tools/abl/ for tools/fused_bench.py (LSAM_LIB=...).
# usage: tools/build_abl.sh NAME "-DFLAG=.. -DFLAG2=.." [NAME2 "flags2" ...]
set -e
cd "$(dirname "$0")/../image_segmentation_amd/csrc"
mkdir -p build_abl ../../tools/abl
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags -c tailblock.hip -o build_abl/tb_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/abl/liblsam_$name.so build/gemm.o build/norm_elem.o build/attention.o build/winblock.o build/mlpblock.o build_abl/tb_$name.o build/hiera.o build/capi.o
done
