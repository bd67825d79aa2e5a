#!/bin/bash
# Builds libpivlfn.so (gfx950) in-tree: piv_liteflownet-pytorch_amd/pivlfn/libpivlfn.so
set -e
cd "$(dirname "$0")"
OUT=../pivlfn/libpivlfn.so
OBJ=../../build/obj
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $PIVLFN_EXTRA_FLAGS"   # PIVLFN_EXTRA_FLAGS=-DPIVLFN_STAMPS: instrumented build for tools/bench_ops.py conv_stamps
pids=()
for f in conv_mfma conv_f16 conv_head warp_corr corr_bwd ops net api; do
  if [ ! -f "$OBJ/$f.o" ] || [ "$f.hip" -nt "$OBJ/$f.o" ] || [ common.h -nt "$OBJ/$f.o" ] || [ ../../include/pivlfn.h -nt "$OBJ/$f.o" ]; then
    hipcc $FLAGS -c "$f.hip" -o "$OBJ/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/conv_mfma.o "$OBJ"/conv_f16.o "$OBJ"/conv_head.o "$OBJ"/warp_corr.o "$OBJ"/corr_bwd.o "$OBJ"/ops.o "$OBJ"/net.o "$OBJ"/api.o
echo "built $OUT"
