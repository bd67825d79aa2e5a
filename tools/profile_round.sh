#!/bin/bash
# Everything profiles/rNN_* is made from, in one go on the GPU box (from the repo root):  bash tools/profile_round.sh
# PARTS=short: only the kernel table, the warp+correlation counters and the split-operand Winograd counters (what bench.py reads)
# (rocprofv3 runs with the program itself after `--`; counters in their own --pmc passes, never combined with other trace domains)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=gpurun_out/${PROFILE_ROUND:-r06}
RN=${PROFILE_ROUND:-r06}
mkdir -p $R
# 1. kernel trace + stats of the default bench (fp32: the instruction + Winograd), per-level timeline
rocprofv3 --kernel-trace --output-format csv -d $R/prof_fp32 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-arithmetic --lean > $R/prof_fp32.log 2>&1
python3 tools/stats_md.py $R/prof_fp32 3 > $R/kernel_stats_fp32.md
python3 tools/level_timeline.py $R/prof_fp32 > $R/timeline_fp32.txt
echo "stats done"
# 2. warp+correlation counter passes (level 3, level 1, batch-8 level 3)
PMC_ROUND=$RN bash tools/pmc_l3.sh > $R/pmc_l3.log 2>&1
cp gpurun_out/pmc_l3/${RN}_pmc_*.json $R/
echo "pmc_l3 done"
[ "$PARTS" = "short" ] && { bash tools/pmc_b3.sh > $R/pmc_b3.log 2>&1; cp gpurun_out/pmc_b3/${RN}_pmc_b3.json $R/; echo "short: done"; exit 0; }
# 3. config #3 (32 x 512^2) counter passes on the level-3 warp+correlation
bash tools/pmc_config3.sh > $R/${RN}_pmc_config3.json 2> $R/pmc_config3.err || echo "pmc_config3 failed"
echo "config3 done"
# 4. Winograd kernel counters: the split-operand kernel beside the fp32 one (traffic, matrix-pipe busy, clock); busy / waits and
#    instruction mix of the fp32 one
bash tools/pmc_b3.sh > $R/pmc_b3.log 2>&1 || echo "pmc_b3 failed"
cp gpurun_out/pmc_b3/${RN}_pmc_b3.json $R/ || true
bash tools/pmc_wino.sh > $R/${RN}_pmc_wino_busy.json 2> $R/pmc_wino_busy.err
PMC="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM" bash tools/pmc_wino.sh > $R/${RN}_pmc_wino_insts.json 2> $R/pmc_wino_insts.err
echo "pmc_wino done"
# 5. counters of the separable distance convolutions (conv_col7 / conv_row7); micro-benchmarks: where the waves of a workgroup land,
#    what a dependent chain of fp32 MFMAs sustains
bash tools/pmc_dist.sh > $R/${RN}_pmc_dist.json 2> $R/pmc_dist.err || echo "pmc_dist failed"
bash tools/pmc_netc.sh > $R/${RN}_pmc_netc.json 2> $R/pmc_netc.err || echo "pmc_netc failed"
# micro-benchmarks: a build failure is shown and the benchmark skipped (no stale binary is run); one killed at its limit (a hung
# kernel) ends the script -- nothing further may touch the GPU after that
for m in wave_simd mfma_chain mfma_neighbour ws_step ws_stall ws_gap ws_flag wino_bf16_loop mfma_shadow; do
  if [ ! -x tools/micro/$m ] || [ tools/micro/$m.hip -nt tools/micro/$m ]; then
    hipcc --offload-arch=gfx950 -O3 -w tools/micro/$m.hip -o tools/micro/$m || { echo "build of tools/micro/$m failed: skipped"; rm -f tools/micro/$m; continue; }
  fi
  set +e
  timeout -k 10 120 tools/micro/$m > $R/${RN}_$m.log 2>&1
  rc=$?
  set -e
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tools/micro/$m was killed at its limit: stopping"; exit $rc; fi
done
python3 tools/bench_dist.py > $R/${RN}_bench_dist.log 2>&1 || true
python3 tools/bench_s2.py > $R/${RN}_bench_s2.log 2>&1 || true
python3 tools/bench_wino.py --levels 1 --layers 128x128,128x64,64x64,32x32 --rounds 3 > $R/${RN}_bench_wino_f4.log 2>&1 || true
python3 tools/bench_wino.py --masks 7680,7936 --levels 1,2 --rounds 3 > $R/${RN}_bench_wino_ws.log 2>&1 || true
# 6. the bench line itself (un-profiled), with the CPU baseline and the arithmetic table
python3 bench.py > $R/bench_default.json 2> $R/bench_default.err
echo "all done"
