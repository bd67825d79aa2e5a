#!/usr/bin/env python3
"""bench.py -- PIV image-pairs/s of the MI355X-native PIV-LiteFlowNet-en forward (BASELINE.json metric).

One "step" = one pass of the hot path (pivlfn_forward: the whole coarse-to-fine pipeline, device-resident
[B,3,H,W] inputs -> device-resident [B,2,H,W] flow) over one batch of synthetic particle-image pairs.
Default workload = BASELINE.json configs[1]: PIV-LiteFlowNet-en, batch 1, 1024x1024, fp32, 1 GPU.
For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank runs its own pairs (weak scaling;
pairs are independent) and the flows are reassembled with one RCCL all-gather per step,
issued asynchronously so it overlaps the next step.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     the level-3 warp+correlation kernel, algorithmic bytes (24 707 072 B/pair at
                  1024^2) / its launch duration measured live with hipEvents around that launch in every timed step;
  "cpu_baseline": the oracle (torch CPU convs + CPU correlation) timed on this box's host cores on ONE pair
                  of the same workload (reported baseline only).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "piv_liteflownet-pytorch_amd"))

import numpy as np
import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1, help="pairs per step per GPU (default 1 = BASELINE configs[1])")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--model", default="piv", choices=["piv", "hui"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-arithmetic", action="store_true", help="skip the side-by-side timing of the three fp32 conv arithmetics")
    ap.add_argument("--lean", action="store_true", help="only the timed forwards (no level-1 / batch-8 / conv roofline extras): for profiler runs")
    ap.add_argument("--cpu-runs", type=int, default=3, help="timed warm runs of the CPU baseline (median reported)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp32_wino_mfma32", "fp32_direct", "fp32_split", "fp32_split3", "fp16"],
                    help="how the large convolutions multiply: fp32 (the library's default and the headline: the fp32 matrix instruction, "
                         "3x3 stride-1 layers by Winograd F(2x2,3x3) in fp32), fp32_direct (the same instruction, direct convolution "
                         "everywhere), fp32_split (fp32 operands as three fp16 pieces = all 24 bits, six partial products on the fp16 "
                         "matrix cores, exact to 2^-32); fp32_split3 (two pieces, 22-23 operand bits) and fp16 (BASELINE config #5) are "
                         "narrower than fp32 and reported under their own metric names, never as the headline")
    ap.add_argument("--profile-level", type=int, default=3, help="level whose warp+correlation launch is event-timed")
    return ap.parse_args()


def l3_algorithmic_bytes(B, H, W, L, C, stride):
    """4*[C*Ho*Wo (f1 touched) + C*h*w (f2) + 2*h*w (flow) + 49*Ho*Wo (out)] per pair."""
    h, w = H >> (L - 1), W >> (L - 1)
    Ho, Wo = -(-h // stride), -(-w // stride)
    flow = 2 * h * w if L < 6 else 0
    return 4 * B * (C * Ho * Wo + C * h * w + flow + 49 * Ho * Wo)


def host_cores():
    """Cores this process may really use: cgroup quota if there is one, else the affinity mask, capped at 16
    (the GPU box gives a 1-GPU job a 16-core share; 256 threads on it ran 20x slower)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("PIVLFN_CPU_THREADS", "16"))))


def cpu_baseline(model, size, wts, i1, i2, runs=3):
    """The oracle (CPU restatement of the reference's forward) on ONE pair of the workload: median of `runs` warm runs after
    one untimed run, threads = the cores this process may use."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pivlfn_oracle as orc
    nthreads = host_cores()
    torch.set_num_threads(nthreads)
    net = orc.make_net(model, wts, corr="torch")
    ts = []
    with torch.no_grad():
        out = net.forward(i1[:1], i2[:1])                                                   # warm-up: pages, thread pool, caches
        for _ in range(runs):
            t0 = time.perf_counter()
            out = net.forward(i1[:1], i2[:1])
            ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    return out, {"value": 1.0 / dt, "unit": "image-pairs/s", "cores": nthreads, "kind": "port",
                 "sample": f"1 pair {size}x{size} fp32, oracle/pivlfn_oracle.py (torch {torch.__version__} CPU convs + "
                           f"slicing correlation); median of {runs} warm runs: {dt:.2f} s (all: {', '.join(f'{t:.2f}' for t in ts)} s)"}


def kernel_source_hash():
    """sha256 over the sources of the roofline kernel: counter traffic from profiles/ is only quoted for the build it was taken on."""
    import hashlib
    h = hashlib.sha256()
    for rel in ("piv_liteflownet-pytorch_amd/csrc/warp_corr.hip", "piv_liteflownet-pytorch_amd/csrc/common.h"):
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]


def counter_traffic(name):
    """HBM bytes per launch from the committed --pmc passes (profiles/<name>), or (None, why)."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, f"profiles/{name} not present"
    try:
        j = json.load(open(path))
    except Exception as e:          # noqa: BLE001
        return None, f"profiles/{name}: {e}"
    if j.get("kernel_source_sha256_16") != kernel_source_hash():
        return None, f"profiles/{name} was taken on other kernel sources ({j.get('kernel_source_sha256_16')} != {kernel_source_hash()}): re-run tools/pmc_l3.sh"
    return j.get("hbm_bytes_per_launch"), f"profiles/{name}"


def counter_file_value(name, key):
    """Another field of the same committed counter file (the rocprofv3 --kernel-trace duration of the launch), or None."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name))).get(key)
    except Exception:          # noqa: BLE001
        return None


DTYPE = {       # the arithmetic type the path computes in (short), and what that means (dtype_note)
    "fp32": ("f32",
             "fp32 data, operands (all 24 significand bits of every multiplicand), products and accumulators.  The 3x3 stride-1 layers (95 % of "
             "the multiplies) by Winograd F(2x2,3x3) -- the minimal-filtering algorithm cuDNN / MIOpen use for fp32 3x3 layers: 2.25x fewer "
             "multiplies, input / output transforms are fp32 additions, filter transform in float64 rounded once to fp32.  WHAT MULTIPLIES WHAT: in "
             "the Winograd layers with whole 64-channel output groups, >= 48 input channels and >= 256x256 outputs per image (csrc/conv_wino_b3.hip; "
             "two thirds of the forward's multiplies) each fp32 operand U or V is split EXACTLY into three bf16 values h + m + l (8 + 8 + 8 "
             "significand bits, fp32's exponent range: no narrower input domain, no scaling) and U*V is accumulated in fp32 on "
             "v_mfma_f32_32x32x16_bf16 from the six piece products hh, hm, mh, hl, mm, lh, each exact; dropped: ml + lm + ll <= 2^-23 |U V| (2^-26 "
             "typically), below the rounding of an fp32 fma -- measured against float64 the layer error is at or below that of both "
             "fp32-instruction kernels, and 6, 8 and 9 piece products agree to three digits (tests/test_gpu_wino_b3.py); bit-identical across "
             "batch.  Every other convolution (the remaining Winograd layers, all direct ones) multiplies fp32 x fp32 on v_mfma_f32_32x32x2_f32 "
             "(exact fp32 fma chains)"),
    "fp32_wino_mfma32": ("f32", "the default of rounds 3-5: as fp32 with every Winograd layer on v_mfma_f32_32x32x2_f32 (csrc/conv_wino.hip)"),
    "fp32_direct": ("f32", "v_mfma_f32_32x32x2_f32, direct convolution in every layer"),
    "fp32_split": ("f32 (products from fp16x6 split operands on the f16 MFMA, f32 accumulate)",
                   "fp32 data, accumulators and results; each operand of the large convolutions enters the fp16 matrix cores as three fp16 "
                   "pieces (all 24 bits): 6 partial products per fp32 product, exact to 2^-32"),
    "fp32_split3": ("f32 data, 22-23-bit multiplicands (fp16x3 split operands on the f16 MFMA, f32 accumulate)",
                    "NOT an fp32-wide arithmetic: each operand of the large convolutions enters the fp16 matrix cores as two fp16 pieces "
                    "(22-23 significand bits), 3 partial products per product, relative product error <= 2^-21"),
    "fp16": ("f16 multiplicands, f32 accumulate", "BASELINE config #5: operands rounded to fp16"),
}
FP32_WIDE = ("fp32", "fp32_wino_mfma32", "fp32_direct", "fp32_split")      # arithmetics that keep all 24 operand bits: only these may carry the fp32 metric name


def arithmetic_modes(net, i1, i2, steps, dev):
    """The same forward under each conv arithmetic, timed back to back in this process (single GPU): pairs/s and the largest flow
    difference from the direct fp32-instruction result."""
    keep = net.precision
    res, ref = {}, None
    for mode in ("fp32_direct", "fp32_wino_mfma32", "fp32", "fp32_split", "fp32_split3"):
        net.precision = mode
        for _ in range(3):
            flow = net(i1, i2)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            flow = net(i1, i2)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        if ref is None:
            ref = flow.clone()
        res[mode] = {"pairs_per_s": round(steps * i1.shape[0] / dt, 3), "ms_per_step": round(dt / steps * 1e3, 3),
                     "max_abs_px_vs_fp32_direct": round(float((flow - ref).abs().max()), 7),
                     "operand_bits": 24 if mode in FP32_WIDE else 23}
    net.precision = keep
    res["what"] = ("fp32_direct = v_mfma_f32_32x32x2_f32, direct convolution everywhere; fp32_wino_mfma32 (the headline of rounds 3-5) = the same "
                   "instruction with the 3x3 stride-1 layers by Winograd F(2x2,3x3); fp32 (the headline) = that with the large Winograd layers' "
                   "operands split exactly into three bf16 pieces on v_mfma_f32_32x32x16_bf16 (six exact piece products per product, all 24 bits: "
                   "dtype_note); fp32_split / fp32_split3 = the residual-free convs with >= 256x256 / >= 64x64 "
                   "outputs per image on v_mfma_f32_32x32x16_f16 with fp32 operands split into 3 / 2 fp16 pieces (6 / 3 partial products, fp32 "
                   "accumulate); fp32_split3 keeps 22-23 operand bits and is not an fp32-wide arithmetic")
    return res


def conv_roofline(dev, precision, launches=40):
    """The dominant kernel of the forward -- the 128 -> 128 3x3 convolution of level 1 (1024 x 1024, batch 1) -- standalone through
    the C ABI: matrix-core work actually executed per launch / its duration, against the dense MFMA peak of the instruction used.
    Every launch has its own event pair: min / median / max over `launches` launches."""
    import ctypes
    from pivlfn import _lib
    lib = _lib.load()
    co = ci = 128
    n = 1024
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).contiguous()
    b = torch.randn(co, generator=g).contiguous()
    h = ctypes.c_void_p()
    _lib.check(lib.pivlfn_conv_create(w.data_ptr(), b.data_ptr(), co, ci, 3, 3, ctypes.byref(h)), "conv_create")
    x = torch.randn(1, n, n, ci, device=dev)
    y = torch.empty(1, n, n, co, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    terms = {"fp32_split3": 3, "fp32_split": 6}.get(precision, 0)

    def launch():
        if terms:
            _lib.check(lib.pivlfn_conv2d_nhwc_split(h, x.data_ptr(), ci, y.data_ptr(), co, 1, n, n, 1, 1, 1, 1, terms, st), "conv")
        elif precision == "fp32":
            _lib.check(lib.pivlfn_conv2d_nhwc_wino_b3(h, x.data_ptr(), ci, y.data_ptr(), co, 1, n, n, 1, 6, st), "conv")
        elif precision == "fp32_wino_mfma32":
            _lib.check(lib.pivlfn_conv2d_nhwc_wino(h, x.data_ptr(), ci, y.data_ptr(), co, 1, n, n, 1, st), "conv")
        else:
            _lib.check(lib.pivlfn_conv2d_nhwc(h, x.data_ptr(), ci, y.data_ptr(), co, None, 0, 1, n, n, 1, 1, 1, 1, st), "conv")
    for _ in range(10):
        launch()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
    for a, e in evs:
        a.record()
        launch()
        e.record()
    torch.cuda.synchronize(dev)
    lib.pivlfn_conv_destroy(h)
    ts = sorted(a.elapsed_time(e) * 1e-3 for a, e in evs)
    t = ts[len(ts) // 2]
    spread = {"min_launch_us": round(ts[0] * 1e6, 1), "median_launch_us": round(t * 1e6, 1), "max_launch_us": round(ts[-1] * 1e6, 1),
              "launches_timed": launches, "timer": "one hipEvent pair per launch on the launch stream (the pair's own ~10 us marker cost is included)"}
    flop = 2.0 * n * n * co * ci * 9
    if terms:
        return {"bound": "mfma", "achieved": round(terms * flop / t / 1e12, 1), "peak": 2516.0, "unit": "TFLOP/s", "frac": round(terms * flop / t / 2.516e15, 4),
                "traffic": None, "kernel": f"conv_split kernels, 128->128 3x3 at 1024x1024 B=1, {terms} fp16 partial products per fp32 product (v_mfma_f32_32x32x16_f16)",
                "fp32_equivalent_tflops": round(flop / t / 1e12, 1), **spread,
                "sustainable_on_random_operands_tflops": 1600.0,
                "note": "fp16 matrix work actually executed (terms x 2 x pixels x Cin x Cout x 9) / median duration; the chip sustains ~1600 TFLOP/s of this "
                        "instruction on random operands (1.70 GHz under power, tools/micro/mfma_f16_power.hip, profiles/r02_mfma_f16_power.log)"}
    if precision == "fp32":
        ex = 6 * flop / 2.25      # F(2x2,3x3): 16 multiplies per 2x2 output tile and channel pair instead of 36, six bf16 piece products each
        cnt = counter_file_value("r06_pmc_b3.json", "pivlfn::conv_wino_b3_kernel<6>") or {}
        return {"bound": "mfma", "achieved": round(ex / t / 1e12, 1), "peak": 2516.0, "unit": "TFLOP/s", "frac": round(ex / t / 2.516e15, 4), "traffic": None,
                "kernel": "conv_wino_b3_kernel<6>, 128->128 3x3 at 1024x1024 B=1 (Winograd F(2x2,3x3), operands split exactly into 3 bf16 pieces, 6 piece "
                          "products per product on v_mfma_f32_32x32x16_bf16)",
                "fp32_winograd_equivalent_tflops": round(flop / 2.25 / t / 1e12, 1), "direct_equivalent_tflops": round(flop / t / 1e12, 1), **spread,
                "sustainable_on_random_operands_tflops": 1800.0,
                "counters": {k: cnt.get(k) for k in ("mfma_busy_frac", "clock_GHz", "valu_per_mfma", "hbm_read_MB", "hbm_write_MB", "median_us")} if cnt else None,
                "note": "achieved = bf16 matrix work actually executed (6 x 2 x 16/4 x pixels x Cin x Cout) / median duration against the dense bf16 "
                        "peak; a bare loop of this instruction on random data sustains ~1.8 PFLOP/s at 1.74 GHz (power), the kernel's own K loop with its "
                        "transform / split vector work and weight-fragment loads 1.05-1.2 PFLOP/s (tools/micro/wino_bf16_loop.hip, "
                        "profiles/r06_wino_bf16_loop.log); counters (profiles/r06_pmc_b3.json, quoted in `counters`): the matrix pipe is busy ~38 % of the launch "
                        "at ~2.0 GHz with 7-8 vector instructions per MFMA -- one wave per SIMD (256 accumulator registers) can issue the step's ~770 "
                        "instructions no faster (a plain vector instruction takes 4 cycles in an MFMA's shadow, an MFMA ~10 of issue: "
                        "tools/micro/mfma_shadow.hip), and the workgroups pull 8 TB/s out of the L2s, 80 % of it weight fragments; a variant with every "
                        "memory request a whole step ahead (LDS ring + LDS-DMA) runs at parity (profiles/r06_b3_lds_ring_ab.log): DESIGN.md 4.2f"}
    if precision == "fp32_wino_mfma32":
        ex = flop / 2.25          # F(2x2,3x3): 16 multiplies per 2x2 output tile and channel pair instead of 36
        return {"bound": "mfma", "achieved": round(ex / t / 1e12, 1), "peak": 157.3, "unit": "TFLOP/s", "frac": round(ex / t / 157.3e12, 4), "traffic": None,
                "kernel": "conv_wino_kernel<1, 2>, 128->128 3x3 at 1024x1024 B=1 (Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32)",
                "direct_equivalent_tflops": round(flop / t / 1e12, 1), **spread,
                "note": "achieved = fp32 matrix work actually executed (2 x 16/4 x pixels x Cin x Cout) / median duration; a wave's own vector "
                        "instructions (input / output transforms: 1.66 per MFMA, profiles/r04_pmc_wino_insts.json) delay its next MFMA -- in-order issue; "
                        "a kernel whose vector work sits in other waves was built in round 5 and is slower (DESIGN.md 4.2e, "
                        "profiles/r05_wino_ws_variants.log) -- and the matrix pipe is busy ~75 % of the "
                        "launch (profiles/r04_pmc_wino_busy.json, clock from GRBM_GUI_ACTIVE / duration in the same file)"}
    return {"bound": "mfma", "achieved": round(flop / t / 1e12, 1), "peak": 157.3, "unit": "TFLOP/s", "frac": round(flop / t / 157.3e12, 4), "traffic": None,
            "kernel": "conv_mfma2_kernel, 128->128 3x3 at 1024x1024 B=1 (v_mfma_f32_32x32x2_f32)", **spread}


def l3_throughput_regime(dev, batch=8, launches=40):
    """The same level-3 warp+correlation kernel family outside the latency-bound batch-1 launch: batch 8 of the 1024x1024 level-3
    shapes (2048 tiles), standalone through the C ABI on torch's current stream, timed with events around `launches` launches."""
    from pivlfn import _lib
    lib = _lib.load()
    C, n, s = 64, 256, 2
    g = torch.Generator(device=dev).manual_seed(7)
    f1 = torch.randn(batch, n, n, C, device=dev, generator=g)
    f2 = torch.randn(batch, n, n, C, device=dev, generator=g)
    # a smooth sub-pixel flow, as the network's own are: every f2 pixel is a bilinear tap of some sample, so the bytes the launch has
    # to move are the algorithmic ones (independent per-pixel noise would leave a third of f2 untouched and flatter the rate)
    yy, xx = torch.meshgrid(torch.arange(n, device=dev, dtype=torch.float32), torch.arange(n, device=dev, dtype=torch.float32), indexing="ij")
    ph = torch.arange(batch, device=dev, dtype=torch.float32).view(batch, 1, 1)
    fl = torch.zeros(batch, n, n, 4, device=dev)
    fl[..., 0] = 0.8 * torch.sin(yy * (6.2832 * 3 / n) + ph)
    fl[..., 1] = 0.8 * torch.cos(xx * (6.2832 * 2 / n) + 0.5 * ph)
    out = torch.empty(batch, n // s, n // s, 56, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream

    def launch():
        _lib.check(lib.pivlfn_warp_corr_nhwc(f1.data_ptr(), f2.data_ptr(), fl.data_ptr(), 1.25, out.data_ptr(), batch, C, n, n, s, 1, st), "wc")
    for _ in range(20):          # clocks and caches in their steady state for this launch pattern
        launch()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(launches):
        launch()
    b.record()
    torch.cuda.synchronize(dev)
    t_b2b = a.elapsed_time(b) / launches * 1e-3
    # the kernel's own duration: the same launches back to back, every dispatch with its own start / stop events (what rocprofv3
    # reports per kernel; the ~3 us between two dependent kernels of a stream belong to no kernel)
    us = ctypes.c_double(0.0)
    _lib.check(lib.pivlfn_warp_corr_nhwc_timed(f1.data_ptr(), f2.data_ptr(), fl.data_ptr(), 1.25, out.data_ptr(), batch, C, n, n, s, 1,
                                               launches, ctypes.byref(us), st), "wc timed")
    t_disp = us.value * 1e-6
    # ... and one dispatch at a time on an idle stream (host synchronisation in between): the start stamp is then taken when the kernel
    # starts, so stop - start is the kernel alone -- the quantity rocprofv3's kernel trace reports
    iso = []
    for _ in range(launches):
        torch.cuda.synchronize(dev)
        _lib.check(lib.pivlfn_warp_corr_nhwc_timed(f1.data_ptr(), f2.data_ptr(), fl.data_ptr(), 1.25, out.data_ptr(), batch, C, n, n, s, 1,
                                                   1, ctypes.byref(us), st), "wc timed")
        iso.append(us.value * 1e-6)
    torch.cuda.synchronize(dev)
    t_iso = sum(iso) / len(iso)
    # The kernel's duration is what the roofline divides by -- and what the committed rocprofv3 summary must agree with: the isolated
    # dispatches (38.9 us where rocprofv3 reads 37.4-38.0).  The back-to-back figure of rounds 1-5 (42.5 us on the same box) contains
    # the 3-4 us between two dependent launches of a stream, which belong to no kernel; it stays in the line as avg_back_to_back_us.
    t = t_iso
    alg = l3_algorithmic_bytes(batch, 1024, 1024, 3, C, s)
    traffic, traffic_src = counter_traffic("r06_pmc_l3b8_warp_corr.json") if batch == 8 else (None, "no counter pass for this batch")
    return {"bound": "hbm", "achieved": round(alg / t / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(alg / t / 8e12, 4),
            "traffic": traffic, "traffic_source": traffic_src,
            "avg_launch_us": round(t * 1e6, 2), "avg_back_to_back_us": round(t_b2b * 1e6, 2), "frac_back_to_back": round(alg / t_b2b / 8e12, 4),
            "avg_dispatch_event_us": round(t_disp * 1e6, 2),
            "rocprofv3_kernel_trace_avg_us": counter_file_value("r06_pmc_l3b8_warp_corr.json", "rocprofv3_kernel_trace_avg_us") if batch == 8 else None,
            "algorithmic_bytes_per_launch": alg, "launches_timed": launches,
            "kernel": "warp_corr_v6_kernel<true, 2> (persistent workgroups, sliding window over runs of 8 tiles)",
            "timer": "avg_launch_us (what frac divides by): start/stop events attached to the dispatch (pivlfn_warp_corr_nhwc_timed), one dispatch "
                     "at a time on an idle stream, mean of `launches_timed` -- the kernel alone, the quantity rocprofv3 reports "
                     "(rocprofv3_kernel_trace_avg_us: a plain --kernel-trace pass over the same launches on the profile box, "
                     "profiles/r06_pmc_l3b8_warp_corr.json); avg_back_to_back_us / frac_back_to_back: one event pair around all launches queued "
                     "back to back / launches -- the figure rounds 1-5 quoted as frac; it contains the 3-4 us between two dependent kernels of a "
                     "stream; avg_dispatch_event_us: the per-dispatch events with the launches queued back to back -- a dispatch's start stamp is "
                     "then taken while its predecessor still runs, so it reads LONGER than the kernel",
            "workload": f"level-3 warp+correlation of batch {batch} x 1024x1024 (C=64, stride 2, 2048 tiles), back-to-back launches"}


def self_launch(args) -> int:
    """`python bench.py --gpus N` without an outer launcher: start N fresh ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, one device each), relay rank 0's output and SUPERVISE them: every rank is polled; when one exits non-zero the others
    are terminated (then killed), the dead rank is named with the tail of its stderr and the parent returns non-zero within seconds;
    an overall deadline (PIVLFN_BENCH_DEADLINE_S, default 1500 s) bounds the whole run.  The parent makes no GPU call and never
    re-executes itself or a rank: a failed run is reported, not retried."""
    import socket
    import subprocess
    import tempfile
    import threading
    backend = os.environ.get("PIVLFN_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()      # (counting devices does not initialise one on this image; nothing below depends on that)
    if backend == "nccl" and ndev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but {ndev} device(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    deadline = time.monotonic() + float(os.environ.get("PIVLFN_BENCH_DEADLINE_S", "1500"))
    procs, errs = [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        errs.append(tempfile.TemporaryFile(mode="w+"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=errs[r], text=True))
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout), daemon=True)      # rank 0's stdout never blocks on a full pipe
    reader.start()

    def tail(r, n=12):
        errs[r].seek(0)
        return "".join(errs[r].readlines()[-n:])

    def stop_all():
        for q in procs:
            if q.poll() is None:
                q.terminate()
        t_kill = time.monotonic() + 5.0
        for q in procs:
            try:
                q.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                q.kill()
                q.wait()

    failed = None
    while True:
        rcs = [q.poll() for q in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = f"rank {bad[0][0]} exited with code {bad[0][1]}" + (f" (also: {bad[1:]})" if len(bad) > 1 else "")
            dead = bad[0][0]
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > deadline:
            failed, dead = "deadline passed with ranks still running: " + str([r for r, rc in enumerate(rcs) if rc is None]), None
            break
        time.sleep(0.2)
    if failed:
        stop_all()
    reader.join(timeout=5.0)
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    if failed:
        print(f"bench.py: {failed}; the other ranks were stopped", file=sys.stderr)
        for r in ([dead] if dead is not None else range(args.gpus)):
            print(f"--- stderr tail of rank {r} ---\n{tail(r)}", file=sys.stderr)
        return 1
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if os.environ.get("PIVLFN_BENCH_FAIL_RANK") == str(rank):      # test knob: this rank dies before the rendezvous
        print(f"bench.py: rank {rank} exits 3 on request (PIVLFN_BENCH_FAIL_RANK)", file=sys.stderr)
        sys.exit(3)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    backend = os.environ.get("PIVLFN_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of the N>1 control flow on a 1-GPU box
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    # PIVLFN_BENCH_FORCE_DIST=1: run the collective path with a process group of ONE rank (tests/test_gpu_configs.py executes the
    # RCCL branch this way on the one-GPU box); MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE must then be set by the caller
    use_dist = world > 1 or os.environ.get("PIVLFN_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        tmo = datetime.timedelta(seconds=float(os.environ.get("PIVLFN_BENCH_INIT_TIMEOUT_S", "600")))      # ranks of a fresh box page the image in at different speeds; a DEAD rank is reported by the parent within seconds
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend, timeout=tmo)

    import pivlfn
    from pivlfn import synth

    B, S = args.batch, args.size
    wts = synth.generate_weights(args.model, 0)
    a, b = synth.particle_batch(B, S, S, seed=1234 + 1000 * rank)
    i1c, i2c = torch.from_numpy(a), torch.from_numpy(b)
    i1, i2 = i1c.to(dev), i2c.to(dev)
    net = pivlfn.Network(model=args.model, params=wts).to(dev).eval()
    net.precision = args.precision

    div = 2 ** (net.lowest_level - 1)
    gdev = dev if backend == "nccl" else torch.device("cpu")
    gathered = [torch.empty(world * B, 2, S // div, S // div, device=gdev) for _ in range(2)] if use_dist else None
    pending = [None, None]

    wait_s = [0.0]

    def step(i):
        flow = net(i1, i2)
        if use_dist:
            k = i & 1
            if pending[k] is not None:
                tw = time.perf_counter()
                pending[k].wait()
                wait_s[0] += time.perf_counter() - tw
            if backend == "nccl":
                pending[k] = dist.all_gather_into_tensor(gathered[k], flow, async_op=True)
            else:
                pending[k] = dist.all_gather(list(gathered[k].view(world, B, 2, S // div, S // div).unbind(0)), flow.cpu(), async_op=True)
        return flow

    def fence():
        if use_dist:
            for k in (0, 1):
                if pending[k] is not None:
                    pending[k].wait()
                    pending[k] = None
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        flow = step(i)
    fence()
    L = args.profile_level
    if L:
        net.profile_enable(L)
    wait_s[0] = 0.0
    t0 = time.perf_counter()
    for i in range(args.steps):
        flow = step(i)
    torch.cuda.current_stream(dev).synchronize()      # this rank's own forwards are done (not the pending gathers, not the other ranks)
    t_own = time.perf_counter() - t0            # this rank's own steps, before it waits for the others
    fence()
    dt = time.perf_counter() - t0
    k_ms, k_empty_ms, k_n = net.profile_read() if L else (0.0, 0.0, 0)
    if L:
        net.profile_enable(0)
    devname = torch.cuda.get_device_name(dev)
    per_rank = [{"rank": 0, "pairs_per_s": round(args.steps * B / t_own, 3), "gather_wait_ms_per_step": 0.0, "device_index": local, "device_name": devname}]
    backend_world = 1
    if use_dist:
        backend_world = dist.get_world_size()       # what the backend saw, not what the launcher asked for
        t = torch.tensor([dt], device=gdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor([args.steps * B / t_own, wait_s[0] / args.steps * 1e3], device=gdev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        who = [None] * world
        dist.all_gather_object(who, (local, devname))
        per_rank = [{"rank": r, "pairs_per_s": round(float(v[0]), 3), "gather_wait_ms_per_step": round(float(v[1]), 4),
                     "device_index": who[r][0], "device_name": who[r][1]} for r, v in enumerate(allr)]
    # level-1 launch of the same kernel family (395 MB: beyond the 256 MiB Infinity Cache), timed the same way in a few extra steps
    l1 = None
    if L == 3 and world == 1 and args.model == "piv" and args.size == 1024 and net.lowest_level == 1 and not args.lean:
        net.profile_enable(1)
        for i in range(5):
            step(i)
        fence()
        l1 = net.profile_read()
        net.profile_enable(0)

    if rank == 0:
        pairs = args.steps * B * world
        value = pairs / dt
        C = [0, 64, 64, 64, 96, 128, 192][L] if L else 0
        stride = 1 if L >= 4 else 2
        roof = None
        if L and k_n:
            alg = l3_algorithmic_bytes(B, S, S, L, C, stride)
            t_k = k_ms / k_n * 1e-3                       # start/stop events attached to the dispatch itself
            t_pair = k_empty_ms / k_n * 1e-3              # plain hipEventRecord pair around the same launch (incl. marker cost)
            ach = alg / t_k / 1e9
            traffic, traffic_src = (counter_traffic("r06_pmc_l3_warp_corr.json") if (B == 1 and S == 1024 and L == 3 and args.model == "piv")
                                    else (None, "no counter pass for this workload"))
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
                    "traffic": traffic, "traffic_source": traffic_src, "kernel": f"{'warp_corr_v7_kernel (one tile per CU)' if (C % 64 == 0 and B * (S // (2 ** (L - 1)) // stride // 8) ** 2 <= torch.cuda.get_device_properties(dev).multi_processor_count) else 'warp_corr_v6_kernel (persistent)'} (level {L}: C={C}, stride {stride})",
                    "algorithmic_bytes_per_launch": alg, "avg_launch_us": round(t_k * 1e6, 2), "launches_timed": k_n,
                    "event_record_pair_us": round(t_pair * 1e6, 2),
                    "rocprofv3_kernel_trace_avg_us": counter_file_value("r06_pmc_l3_warp_corr.json", "rocprofv3_kernel_trace_avg_us")
                    if (B == 1 and S == 1024 and L == 3 and args.model == "piv") else None,
                    # what one tile per CU can reach: a CU gathers 200 KB (the 14 x 14 stride-2 positions x 4 taps touch a 28 x 28
                    # region of 64 channels: 3.06 x the tile's own 8.0 KB x 8 of f2) through its own load path, served by the Infinity
                    # Cache at 33.5 GB/s per CU (MI355X_MICROARCH.md, Indexed rows: a lower bound of the rate), behind a
                    # flow -> taps round trip (~1 us) that cannot overlap it; plus the output's way out.  DESIGN.md 4.1b.
                    # (renamed in round 6: this was called floor_us, but the guide lists 33.5 GB/s under per-CU rates that are LOWER bounds --
                    #  it is an estimate of one share of the launch, not a floor; the same launch on L2-resident inputs takes 7.8-7.9 us
                    #  standalone: profiles/r06_wc_descriptor_ab.log)
                    "model_us": round(200e3 / 33.5e9 * 1e6 + 1.0, 2) if (B == 1 and S == 1024 and L == 3 and args.model == "piv") else None,
                    "model_note": "an ESTIMATE of the gather-arrival share, not a lower bound: batch 1 puts ONE tile on each CU, 200 KB per CU "
                                  "at the Infinity-Cache gather rate the guide lists for one CU (33.5 GB/s, itself a lower bound of that rate) = 6.0 us, "
                                  "+ ~1 us flow -> taps dependency; measured: 7.8-7.9 us back to back on cache-resident inputs, 9.7-10.3 us inside "
                                  "the forward; frac is quoted against 8 TB/s all the same",
                    "timer": "HIP start/stop events attached to the dispatch (hipExtLaunchKernelGGL) in every timed step, on the "
                             "stream the kernel runs on; event_record_pair_us = plain hipEventRecord pair around the same launch "
                             "(adds the marker packets' own cost); rocprofv3_kernel_trace_avg_us: a plain --kernel-trace pass "
                             "over the same launch of the same forward, stored in profiles/r06_pmc_l3_warp_corr.json"}
        fp32_grade = args.precision in FP32_WIDE
        out = {
            "metric": (("PIV" if args.model == "piv" else "LiteFlowNet (Hui weights layout)") +
                       (f" image-pairs/s at {S}x{S} fp32" if fp32_grade else
                        (f" image-pairs/s at {S}x{S}, fp16-multiplicand conv mode (BASELINE config #5 variant; not the fp32 headline)" if args.precision == "fp16" else
                         f" image-pairs/s at {S}x{S}, fp32 data with 22-23-bit multiplicands (three-term fp16 split; narrower than fp32, not the fp32 headline)"))),
            "value": round(value, 3), "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE[args.precision][0], "dtype_note": DTYPE[args.precision][1], "data": "synthetic",
            "config": {"workload": f"{'PIV-LiteFlowNet-en' if args.model == 'piv' else 'LiteFlowNet'} forward, batch {B}/GPU, "
                                   f"{S}x{S} synthetic PIV pair, " + ("fp32 (BASELINE configs[1])" if fp32_grade else ("fp16-multiplicand convs" if args.precision == "fp16" else "three-term split convs")),
                       "conv_arithmetic": args.precision,
                       "pairs_per_step_per_gpu": B, "weights": "generated (pivlfn.synth seed 0)",
                       "multi_gpu": "pairs sharded over ranks, async RCCL all-gather of flows per step" if world > 1 else "single GPU"},
            "roofline": roof,
            "whole_net": {"conv_tflops_direct_equivalent": round(value * (2.506 if (args.model == 'piv' and S == 1024) else float('nan')) / world, 2),
                          "fp32_mfma_peak_tflops": 157.3,
                          # multiplies of the DIRECT algorithm (2.506 TFLOP per pair) per second against the fp32 instruction's peak: Winograd
                          # executes 2.25x fewer in the 3x3 stride-1 layers, so this may exceed what direct convolution could reach at all
                          "direct_equivalent_ratio_to_fp32_mfma_peak": round(value / world * 2.506 / 157.3, 4) if (args.model == 'piv' and S == 1024) else None,
                          # layer-boundary bytes of the reference's graph (in + out + weights of every conv, fp32)
                          "layer_boundary_gb_per_pair": 16.57 if (args.model == 'piv' and S == 1024) else None,
                          "hbm_frac_of_8tbs": round(value / world * 16.57 / 8000.0, 4) if (args.model == 'piv' and S == 1024) else None},
        }
        out["per_rank"] = per_rank
        out["backend_world_size"] = backend_world
        if l1 is not None and l1[2]:
            alg1 = l3_algorithmic_bytes(B, S, S, 1, 64, 2)
            t1 = l1[0] / l1[2] * 1e-3
            tr1, src1 = counter_traffic("r06_pmc_l1_warp_corr.json") if B == 1 else (None, "no counter pass for this workload")
            out["roofline_level1"] = {"bound": "hbm", "achieved": round(alg1 / t1 / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                      "frac": round(alg1 / t1 / 8e12, 4), "traffic": tr1, "traffic_source": src1,
                                      "kernel": "warp_corr_v6_kernel<true, 2> (level 1: C=64, stride 2; 395 MB per launch, beyond the Infinity Cache; runs of 16 tiles)",
                                      "algorithmic_bytes_per_launch": alg1, "avg_launch_us": round(t1 * 1e6, 2), "launches_timed": l1[2],
                                      "rocprofv3_kernel_trace_avg_us": counter_file_value("r06_pmc_l1_warp_corr.json", "rocprofv3_kernel_trace_avg_us") if B == 1 else None}
        if world == 1 and args.model == "piv" and S == 1024 and not args.lean:
            out["roofline_batch8"] = l3_throughput_regime(dev)
        if world == 1 and args.precision != "fp16" and args.model == "piv" and S == 1024 and not args.lean:
            out["roofline_conv"] = conv_roofline(dev, args.precision)
        if world == 1 and args.precision != "fp16" and not args.no_arithmetic:
            out["arithmetic"] = arithmetic_modes(net, i1, i2, min(args.steps, 10), dev)
        if not args.no_cpu_baseline and world == 1:
            ref, cb = cpu_baseline(args.model, S, wts, i1c, i2c, runs=args.cpu_runs)
            out["cpu_baseline"] = cb
            err = float((flow[:1].cpu() - ref).abs().max())
            out["parity_vs_oracle_max_abs_px"] = round(err, 7)
            out["max_abs_flow_px"] = round(float(ref.abs().max()), 3)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if use_dist:
        if os.environ.get("PIVLFN_BENCH_FORCE_DIST") == "1" and rank == 0:
            # what the one-rank RCCL run proves: the gathered buffer holds this rank's last flows
            k = (args.steps - 1) & 1
            same = bool(torch.equal(gathered[k].to(flow.device), flow)) if backend == "nccl" else None
            print(json.dumps({"forced_dist": True, "backend": backend, "gathered_equals_flow": same}), flush=True)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
