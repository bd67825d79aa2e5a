#!/bin/bash
# A/B of the Winograd kernel against variants compiled with extra flags (one library each), interleaved rounds on one box:
#   bash tools/wino_ab.sh "-DWINO_PRIO=1" "-DWINO_PRIO=2" ...     (run from the repo root after csrc/build.sh)
set -e
OBJ=build/obj
mkdir -p gpurun_out/wino_ab
i=0
for flags in "$@"; do
  i=$((i + 1))
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $flags -c piv_liteflownet-pytorch_amd/csrc/conv_wino.hip -o gpurun_out/wino_ab/wino_$i.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/wino_ab/lib_$i.so $(ls $OBJ/*.o | grep -v conv_wino.o) gpurun_out/wino_ab/wino_$i.o
  echo "== variant $i: $flags"
  python3 tools/bench_wino.py --levels ${LEVELS:-1} --layers ${LAYERS:-128x128,128x64,64x64,32x32} --rounds ${ROUNDS:-4} --ab gpurun_out/wino_ab/lib_$i.so 2>&1 | grep -v amdgpu.ids | cut -c1-260
done
