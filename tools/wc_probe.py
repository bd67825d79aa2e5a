#!/usr/bin/env python3
"""Where does the level-3 warp+correlation launch spend its time?  In-kernel stamps (tools build) in three settings:
  net    the launch inside a 1024x1024 PIV forward (what bench.py's `roofline` object times)
  warm   the same shapes standalone, launched back to back (inputs resident in L2 / Infinity Cache)
  cold   standalone after 1.5 GB of unrelated writes (inputs in HBM only)
Prints per-phase s_memtime deltas (cycles; mean / median / max over workgroups, wave 0 and wave 8) and the launch's span on
the chip-wide 100 MHz clock (first workgroup entry -> last workgroup exit)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _toolslib  # noqa: E402
from pivlfn import _lib  # noqa: E402

PH = ["entry->taps", "(v4: barrier1)", "issue", "arrive+commit", "barrier2", "dots", "reduce+transpose (v4: barrier3)", "barrier (v4: transpose+bar4)", "store issue"]


def report(tag, stamps, nwg, brief=False):
    t = stamps[:nwg * 32].reshape(nwg, 2, 16).astype(np.int64)
    ret = None
    for w, name in ((0, "wave0"), (1, "wave8")):
        r = t[:, w, :]
        ok = r[:, 15] > 0
        if not ok.any():
            continue
        r = r[ok]
        nst = int(r[0, 15])
        if nst < 10:                       # kernels with entry/exit stamps only
            span = (r[:, 13].max() - r[:, 12].min()) * 10.0
            life = (r[:, 13] - r[:, 12]) * 10.0
            print(f"{tag} {name}: {len(r)} workgroups; span {span:.0f} ns, entries spread over {(r[:, 12].max() - r[:, 12].min()) * 10.0:.0f} ns; lifetime mean {life.mean():.0f} max {life.max():.0f} ns")
            ret = span
            continue
        d = np.diff(r[:, :10], axis=1)
        whole = r[:, 9] - r[:, 0]
        span = (r[:, 13].max() - r[:, 12].min()) * 10.0
        start_spread = (r[:, 12].max() - r[:, 12].min()) * 10.0
        if w == 0:
            ret = span
        print(f"{tag} {name}: {len(r)} workgroups; span {span:.0f} ns (first entry -> last exit), entries spread over {start_spread:.0f} ns; "
              f"whole workgroup cycles mean {whole.mean():.0f} max {whole.max()}")
        if brief and w == 1:
            continue
        print("    " + "  ".join(f"{n} {d[:, i].mean():.0f}/{np.median(d[:, i]):.0f}/{d[:, i].max()}" for i, n in enumerate(PH)))
        life = (r[:, 13] - r[:, 12]) * 10.0
        print(f"    workgroup lifetime ns: mean {life.mean():.0f} median {np.median(life):.0f} max {life.max():.0f};  cycles/ns = {whole.mean() / max(1.0, life.mean()):.2f} GHz")
        if w == 0 and not brief:
            for x in range(8):
                m = (r[:, 14] & 15) == x
                if m.any():
                    print(f"      xcc {x}: {m.sum()} wgs, entry {((r[m, 12].min() - r[:, 12].min()) * 10):.0f}..{((r[m, 12].max() - r[:, 12].min()) * 10):.0f} ns, "
                          f"exit max {((r[m, 13].max() - r[:, 12].min()) * 10):.0f} ns, arrive+commit mean {d[m, 3].mean():.0f}")
    return ret


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="net,warm,cold")
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--combos", default="0:0,0:16", help="variant:dbgmask list for the standalone runs (dbg 16 = no L2 prefetch)")
    ap.add_argument("--brief", action="store_true")
    ap.add_argument("--net-masks", default="0", help="pivlfn_tune(1, .) masks to run the in-network measurement under (2048 = single stream)")
    a = ap.parse_args()
    lib = _toolslib.load()
    _lib._lib = lib                    # the Python surface below talks to the tools build
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    stamps = torch.zeros(4096 * 32, dtype=torch.int64, device=dev)
    L = a.level
    C, n, s = {1: (64, 1024, 2), 2: (64, 512, 2), 3: (64, 256, 2), 4: (96, 128, 1), 5: (128, 64, 1)}[L]
    no = n // s
    nwg = (no // 8) ** 2
    what = a.what.split(",")

    if "net" in what:
        import pivlfn
        from pivlfn import synth
        wts = synth.generate_weights("piv", 0)
        x, y = synth.particle_batch(1, 1024, 1024, seed=1234)
        i1, i2 = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
        net = pivlfn.Network(model="piv", params=wts).to(dev).eval()
        for _ in range(3):
            net(i1, i2)
        torch.cuda.synchronize()
        net.profile_enable(L)
        for mask in [int(m) for m in a.net_masks.split(",")] * 2:
            dbg = 0
            lib.pivlfn_tune(1, mask)
            for _ in range(2):
                net(i1, i2)
            torch.cuda.synchronize()
            net.profile_read()
            tot = []
            for rep in range(a.reps):
                stamps.zero_()
                torch.cuda.synchronize()
                _toolslib.set_stamp_buffer(lib, stamps.data_ptr(), "wc")
                net(i1, i2)
                torch.cuda.synchronize()
                _toolslib.set_stamp_buffer(lib, 0, "wc")
                ms, ems, k = net.profile_read()
                tot.append(round(ms / max(1, k) * 1e3, 2))
                report(f"[net mask {mask} rep {rep}]", stamps.cpu().numpy(), nwg, brief=rep > 0 or a.brief)
            print(f"[net mask {mask}] dispatch-event time of the level-{L} launch per rep: {tot} us")
        lib.pivlfn_tune(1, 0)
        net.profile_enable(0)

    g = torch.Generator(device=dev).manual_seed(5)
    f1 = torch.randn(1, n, n, C, device=dev, generator=g)
    f2 = torch.randn(1, n, n, C, device=dev, generator=g)
    fl = torch.zeros(1, n, n, 4, device=dev)
    fl[..., :2] = torch.randn(1, n, n, 2, device=dev, generator=g) * 0.8
    out = torch.empty(1, no, no, 56, device=dev)

    def launch():
        _toolslib.check(lib, lib.pivlfn_warp_corr_nhwc(f1.data_ptr(), f2.data_ptr(), fl.data_ptr(), 1.25, out.data_ptr(), 1, C, n, n, s, 1, st), "wc")

    combos = [tuple(int(v) for v in c.split(":")) for c in a.combos.split(",")]     # variant:dbgmask
    for variant, dbg in combos:
        lib.pivlfn_tune(0, variant)
        lib.pivlfn_tune(2, dbg)
        tag = f"variant {variant} dbg {dbg}"
        if "warm" in what:
            for rep in range(a.reps):
                for _ in range(5):
                    launch()
                stamps.zero_()
                torch.cuda.synchronize()
                _toolslib.set_stamp_buffer(lib, stamps.data_ptr(), "wc")
                launch()
                torch.cuda.synchronize()
                _toolslib.set_stamp_buffer(lib, 0, "wc")
                report(f"[warm {tag} rep {rep}]", stamps.cpu().numpy(), nwg, brief=rep > 0 or a.brief)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                launch()
            e1.record()
            torch.cuda.synchronize()
            print(f"[warm {tag}] 100 back-to-back launches: {e0.elapsed_time(e1) * 10:.2f} us each (includes the kernel boundary)")
        if "cold" in what:
            junk = torch.empty(3 * 128 * 1024 * 1024, dtype=torch.float32, device=dev)     # 1.5 GB
            small = torch.zeros(1024, device=dev)
            spans = []
            for rep in range(a.reps + 2):
                stamps.zero_()
                junk.fill_(float(rep))
                small.add_(1.0)
                torch.cuda.synchronize()
                _toolslib.set_stamp_buffer(lib, stamps.data_ptr(), "wc")
                junk.fill_(float(rep) + 0.5)       # evicts L2 and the Infinity Cache right before the launch, on the same stream
                small.add_(1.0)
                launch()
                torch.cuda.synchronize()
                _toolslib.set_stamp_buffer(lib, 0, "wc")
                spans.append(report(f"[cold {tag} rep {rep}]", stamps.cpu().numpy(), nwg, brief=rep > 0 or a.brief))
            print(f"[cold {tag}] span ns over reps: {spans}")
            del junk
    lib.pivlfn_tune(0, 0)
    lib.pivlfn_tune(2, 0)


if __name__ == "__main__":
    main()
