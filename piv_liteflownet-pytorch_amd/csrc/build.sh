#!/bin/bash
# Builds libpivlfn.so (gfx950) in-tree: piv_liteflownet-pytorch_amd/pivlfn/libpivlfn.so
#   build.sh          the production library (every object rebuilt when ANY source/header of csrc/ or include/ is newer)
#   build.sh tools    additionally tools/libpivlfn_tools.so: -DPIVLFN_TOOLS -DPIVLFN_STAMPS (A/B knobs, ablation masks, in-kernel
#                     stamps); only tools/*.py load it
set -e
cd "$(dirname "$0")"
SRCS="conv_mfma conv_wino conv_wino_b3 conv_f16 conv_split conv_head warp_corr corr_bwd ops net api"
build_one() {   # $1 = object dir, $2 = output .so, $3 = extra flags, $4 = extra sources (research kernels of tools/kernels/)
  local OBJ="$1" OUT="$2" EXTRA="$4"
  mkdir -p "$OBJ"
  local FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $3 $PIVLFN_EXTRA_FLAGS"
  # a change to any header (or to the flags) rebuilds everything: no stale objects
  local STAMP="$OBJ/.flags"
  if [ ! -f "$STAMP" ] || [ "$(cat "$STAMP")" != "$FLAGS" ]; then rm -f "$OBJ"/*.o; echo "$FLAGS" > "$STAMP"; fi
  local hdr_new=0
  for h in *.h ../../include/*.h build.sh; do
    for o in "$OBJ"/*.o; do [ -f "$o" ] && [ "$h" -nt "$o" ] && hdr_new=1; done
  done
  [ "$hdr_new" = 1 ] && rm -f "$OBJ"/*.o
  local pids=() objs=()
  for f in $SRCS; do
    objs+=("$OBJ/$f.o")
    if [ ! -f "$OBJ/$f.o" ] || [ "$f.hip" -nt "$OBJ/$f.o" ]; then
      # warp_corr: the SLP vectoriser packs the consumers' fma chains into v_pk_fma_f32 behind 2-4 v_mov each (measured in the ISA)
      # conv_wino_b3: v_pk_add_f32 / v_pk_fma_f32 cost 11 cycles of the wave's issue time against 2 x 4-5 for the scalar pair
      # (tools/micro/mfma_shadow.hip), and that kernel's one wave per SIMD is issue-bound: 1 % (profiles/r06_b3_lds_ring_ab.log)
      local PERFILE=""; { [ "$f" = "warp_corr" ] || [ "$f" = "conv_wino_b3" ]; } && PERFILE="-fno-slp-vectorize"
      hipcc $FLAGS $PERFILE -c "$f.hip" -o "$OBJ/$f.o" &
      pids+=($!)
    fi
  done
  for f in $EXTRA; do
    objs+=("$OBJ/$f.o")
    if [ ! -f "$OBJ/$f.o" ] || [ "../../tools/kernels/$f.hip" -nt "$OBJ/$f.o" ]; then
      hipcc $FLAGS -I. -c "../../tools/kernels/$f.hip" -o "$OBJ/$f.o" &
      pids+=($!)
    fi
  done
  for p in "${pids[@]}"; do wait "$p"; done
  hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${objs[@]}"
  echo "built $OUT"
}
build_one ../../build/obj ../pivlfn/libpivlfn.so ""
if [ "$1" = "tools" ]; then
  # research kernels that no user path launches (tools/kernels/): round 5's persistent Winograd kernel with specialised waves and round
  # 4's F(4x4, 3x3) kernel -- both measured slower than conv_wino.hip, kept with their tests for A/B runs
  build_one ../../build/obj_tools ../../tools/libpivlfn_tools.so "-DPIVLFN_TOOLS -DPIVLFN_STAMPS" "conv_wino_ws conv_wino4"
fi
