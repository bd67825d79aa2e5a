#!/usr/bin/env python3
"""ORACLE / TEST INFRASTRUCTURE -- golden vectors for the correlation backward.

The reference's backward exists only as CuPy CUDA text (src/correlation.py:106-234) and cannot execute in this container
(no CUDA, no cupy), so unlike the forward it cannot be pinned by running the reference.  Its pin is the agreement of two
independent statements of the same gradients:
  (a) oracle/corr_oracle.c corr_backward: the two kernels restated loop for loop (ROUND_OFF ceil/floor trick, rbot scratch,
      `sum / (float)C`), and
  (b) torch autograd (float64) through `correlation_torch`, the forward restatement that IS pinned against the reference.
This script records (a) as the fixture and the (a)-vs-(b) error in tests/golden/pin_report_bwd.json.

  python oracle/gen_golden_bwd.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pivlfn_oracle as orc  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")

CASES = [(1, 16, 12, 20, 2), (2, 24, 9, 7, 1), (1, 32, 8, 8, 1), (1, 16, 13, 11, 2), (1, 8, 33, 18, 1), (2, 5, 17, 36, 2),
         (1, 3, 1, 1, 1), (1, 4, 2, 5, 2)]


def main():
    rng = np.random.default_rng(20240607)
    out, report = {}, {}
    for n, (B, C, H, W, s) in enumerate(CASES):
        f1 = rng.standard_normal((B, C, H, W)).astype(np.float32)
        f2 = rng.standard_normal((B, C, H, W)).astype(np.float32)
        go = rng.standard_normal((B, 49, -(-H // s), -(-W // s))).astype(np.float32)
        g1, g2 = orc.correlation_backward_c(f1, f2, go, s)
        a1, a2 = orc.correlation_backward_autograd(torch.from_numpy(f1).double(), torch.from_numpy(f2).double(),
                                                   torch.from_numpy(go).double(), s)
        e1 = float(np.abs(g1 - a1.numpy()).max() / np.abs(a1.numpy()).max())
        e2 = float(np.abs(g2 - a2.numpy()).max() / np.abs(a2.numpy()).max())
        assert e1 < 1e-6 and e2 < 1e-6, (n, e1, e2)
        if s > 1:       # off-grid positions carry exact zeros
            m = np.ones((H, W), bool)
            m[::s, ::s] = False
            assert not g1[:, :, m].any() and not g2[:, :, m].any()
        out.update({f"f1_{n}": f1, f"f2_{n}": f2, f"go_{n}": go, f"stride_{n}": np.int32(s), f"g1_{n}": g1, f"g2_{n}": g2})
        report[f"case_{n}"] = {"shape": [B, C, H, W, s], "c_vs_autograd64_rel_first": e1, "c_vs_autograd64_rel_second": e2}
    np.savez_compressed(os.path.join(GOLD, "corr_bwd_cases.npz"), **out)
    with open(os.path.join(GOLD, "pin_report_bwd.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
