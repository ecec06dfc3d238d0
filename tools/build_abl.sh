#!/bin/bash
# Dev aid: ablation builds of the tail kernel (TAIL_ABL bit mask) into tools/abl/ for tools/fused_bench.py.
set -e
cd "$(dirname "$0")/../image_segmentation_amd/csrc"
mkdir -p build_abl ../../tools/abl
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DTAIL_ABL=$v -c tailblock.hip -o build_abl/tb_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/abl/liblsam_abl$v.so build/gemm.o build/norm_elem.o build/attention.o build/winblock.o build/mlpblock.o build_abl/tb_$v.o build/hiera.o build/capi.o
done
