"""Brightness / contrast modification of input frames for `run.py -b/-c` (reference: run.py:88-97 `image_mod`, which applies
torchvision's `adjust_brightness` then `adjust_contrast` to the PIL image, i.e. PIL's ImageEnhance.Brightness / .Contrast).

Restated on uint8 tensors so that it runs on the device the frames already live on (any torch device; [..., H, W, 3] RGB):
  blend(d, x, f)  = d + f * (x - d) in float32, truncated to uint8; clipped to [0, 255] when f is outside [0, 1]
                    (PIL's ImagingBlend: its interpolation branch cannot leave the range, its extrapolation branch clips)
  brightness(x,f) = blend(0, x, f)
  contrast(x, f)  = blend(m, x, f), m = round-half-up mean of the luma L = (19595 R + 38470 G + 7471 B + 32768) >> 16 of
                    the whole frame (PIL's "L" conversion and ImageStat mean), one value per frame
Checked bit for bit against PIL in tests/test_host.py.
"""
from __future__ import annotations

import torch


def _blend(base: torch.Tensor, img: torch.Tensor, factor: float) -> torch.Tensor:
    f = torch.tensor(float(factor), dtype=torch.float32, device=img.device)
    diff = (img.to(torch.int32) - base.to(torch.int32)).to(torch.float32)
    t = base.to(torch.float32) + f * diff                    # two roundings, as the C expression (no fused multiply-add)
    if not 0.0 <= float(factor) <= 1.0:
        t = t.clamp(0.0, 255.0)
    return t.to(torch.uint8)                                 # truncation toward zero, the C cast


def adjust_brightness(img: torch.Tensor, factor: float) -> torch.Tensor:
    return _blend(torch.zeros((), dtype=torch.uint8, device=img.device), img, factor)


def luma(img: torch.Tensor) -> torch.Tensor:
    x = img.to(torch.int32)
    return (x[..., 0] * 19595 + x[..., 1] * 38470 + x[..., 2] * 7471 + 0x8000) >> 16


def adjust_contrast(img: torch.Tensor, factor: float) -> torch.Tensor:
    lum = luma(img)
    count = lum.shape[-1] * lum.shape[-2]
    total = lum.to(torch.int64).sum(dim=(-1, -2), keepdim=True)
    mean = ((2 * total + count) // (2 * count)).to(torch.uint8)            # int(mean + 0.5)
    return _blend(mean.unsqueeze(-1), img, factor)


def image_mod(img: torch.Tensor, brightness_factor: float = 1.0, contrast_factor: float = 1.0) -> torch.Tensor:
    """uint8 [..., H, W, 3] -> uint8, brightness first, then contrast (run.py:93-94)."""
    if img.dtype != torch.uint8 or img.shape[-1] != 3:
        raise ValueError(f"image_mod: expected uint8 [..., H, W, 3], got {img.dtype} {tuple(img.shape)}")
    return adjust_contrast(adjust_brightness(img, brightness_factor), contrast_factor)


def mod_name(brightness: float, contrast: float) -> str:
    """`NNN_NNN` tag of a (brightness, contrast) pair in output names (run.py:125)."""
    return f"{int(brightness * 100):03d}_{int(contrast * 100):03d}"
