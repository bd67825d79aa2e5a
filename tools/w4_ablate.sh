#!/bin/bash
# Timing-only ablations of the F(4x4, 3x3) kernel: one library per compile-time mask (1 no MFMAs, 2 no transform, 4 no weight loads,
# 8 no patch loads, 16 no epilogue), each timed on the 128 -> 128 layer at 1024 x 1024 (results are wrong by construction).
# Run on the GPU box from the repo root after csrc/build.sh:  bash tools/w4_ablate.sh "0 1 2 4 8 12 16"
set -e
MASKS=${1:-"0 1 2 4 8 12 16 31"}
OBJ=build/obj
mkdir -p gpurun_out/w4abl
for m in $MASKS; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -DW4_ABL_CT=$m -c -Ipiv_liteflownet-pytorch_amd/csrc tools/kernels/conv_wino4.hip -o gpurun_out/w4abl/w4_$m.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/w4abl/lib_$m.so $(ls $OBJ/*.o | grep -v conv_wino4.o) gpurun_out/w4abl/w4_$m.o
done
python3 - "$MASKS" <<'PY'
import ctypes, sys, os, torch
sys.path.insert(0, "piv_liteflownet-pytorch_amd")
from pivlfn import _lib
masks = sys.argv[1].split()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
n, ci, co = 1024, 128, 128
g = torch.Generator().manual_seed(1)
w = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).contiguous()
b = torch.randn(co, generator=g).contiguous()
x = torch.randn(1, n, n, ci, device=dev)
y = torch.empty(1, n, n, co, device=dev)
for m in masks:
    lib = ctypes.CDLL(os.path.abspath(f"gpurun_out/w4abl/lib_{m}.so"))
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
    h = ctypes.c_void_p()
    assert lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), co, ci, 3, 3, ctypes.byref(h)) == 0
    def run():
        assert lib.pivlfn_conv2d_nhwc_wino4(h, x.data_ptr(), ci, y.data_ptr(), co, 1, n, n, 1, st) == 0
    for _ in range(3): run()
    ts = []
    for _ in range(3):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(10): run()
        e.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) * 100)
    print(f"mask {m:>3}: min {min(ts):8.1f} us", flush=True)
PY
