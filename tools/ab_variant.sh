#!/bin/bash
# Timing-only variants of one source file, each compiled with extra flags into its own copy of the tools library and run through a
# tools script on the GPU box (from the repo root, after csrc/build.sh tools):
#   bash tools/ab_variant.sh conv_mfma "python3 tools/bench_s2.py" "-DS2_ABL_CT=0" "-DS2_ABL_CT=1" ...
set -e
SRC=$1; CMD=$2; shift 2
OBJ=build/obj_tools
mkdir -p gpurun_out/ab_variant
i=0
for flags in "$@"; do
  i=$((i + 1))
  PERFILE=""; [ "$SRC" = "warp_corr" ] && PERFILE="-fno-slp-vectorize"
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -DPIVLFN_TOOLS -DPIVLFN_STAMPS $PERFILE $flags -c piv_liteflownet-pytorch_amd/csrc/$SRC.hip -o gpurun_out/ab_variant/${SRC}_$i.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/ab_variant/lib_$i.so $(ls $OBJ/*.o | grep -v "/$SRC.o") gpurun_out/ab_variant/${SRC}_$i.o
  echo "== variant $i: $flags"
  PIVLFN_TOOLS_LIB=$PWD/gpurun_out/ab_variant/lib_$i.so $CMD 2>&1 | grep -v amdgpu.ids | cut -c1-240
done
