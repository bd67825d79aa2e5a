#!/bin/bash
# Which part of conv_c3k7_kernel<true> (NetC.conv1 + the two 1 x 1 layers of level 1, 1.34 GB of stores) takes the time: the forward of
# the tools build with the kernel's compile-time ablation instances (selected by pivlfn_tune knob 7) stepping through MASKS, one rocprofv3 --kernel-trace pass;
# the kernel's launches come in the order of the masks, 8 per mask (3 warm-up + 5 timed forwards).
#   bash tools/c3k7_ablate.sh  > profiles/rNN_c3k7_ablation.log      (from the repo root on the GPU box)
set -e
OUT=$PWD/gpurun_out/c3k7
mkdir -p "$OUT"
export TMPDIR=/tmp
MASKS=${MASKS:-0,3,1,2,4,5,12,15,28,31,0}
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 tools/net_ab.py --knob ${KNOB:-7} --masks $MASKS --rounds 1 --steps 5 > "$OUT/run.log" 2>&1
python3 - "$OUT" "$MASKS" <<'PY'
import csv, glob, os, sys
out, masks = sys.argv[1], [int(m) for m in sys.argv[2].split(",")]
rows = []
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows += [r for r in csv.DictReader(open(p)) if "conv_c3k7_kernel<true" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
assert len(du) == 8 * len(masks), (len(du), len(masks))
what = {1: "no stores of the 1x1 layers", 2: "no conv1 stores", 4: "no MFMAs of the 1x1 layers", 8: "no conv1 MFMAs", 16: "no patch loads"}
print("conv_c3k7_kernel<true> at 1024x1024, both frames; us per launch (mean of the 5 timed forwards; min)")
for i, m in enumerate(masks):
    d = du[8 * i + 3: 8 * i + 8]
    print(f"mask {m:2d}: {sum(d) / len(d):7.1f}  {min(d):7.1f}   " + ("as shipped" if m == 0 else ", ".join(v for k, v in what.items() if m & k)))
PY
