"""Drop-in for `estimate()` (inference.py:30-67) and the `Inference` helper class (inference.py:70-213) of the reference.

Same signature and return convention: `estimate(net, img1, img2, tensor=False)` takes two [B,3,H,W] tensors
in [0,1] on the network's device, adapts H and W up to multiples of 32 with a bilinear
(align_corners=False) resize, runs the network in eval mode, resizes the raw flow back to H x W and rescales
u by W/W' and v by H/H'.  tensor=True returns [B,2,H,W]; tensor=False returns an H x W x 2 numpy array
(batch 1, as in the reference).  Both resizes run as HIP kernels (`pivlfn_resize_bilinear`); when a size is
already a multiple of 32 the input resize is the identity and is skipped.

`Inference.parser(net, im1, im2, device)` (inference.py:202-213) takes two PIL images (or uint8 HxWx3 arrays), builds the
[1,3,H,W] tensors ToTensor would and returns `estimate(...)`'s H x W x 2 array; `Inference.images_parsing` walks a folder
(paired or sequential frames) as inference.py:120-171.  The video and DataLoader front ends of the reference need cv2 /
imutils / the training datasets and are outside the hot path: they raise NotImplementedError.
"""
from __future__ import annotations

import ctypes
import math
import os

import numpy as np
import torch

from . import _lib


def _resize(x: torch.Tensor, Ho: int, Wo: int, mul=None) -> torch.Tensor:
    B, C, H, W = x.shape
    out = torch.empty([B, C, Ho, Wo], dtype=torch.float32, device=x.device)
    m = (ctypes.c_float * 2)(*mul) if mul is not None else None
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().pivlfn_resize_bilinear(x.data_ptr(), out.data_ptr(), B, C, H, W, Ho, Wo, m,
                                                      _lib.stream_ptr(x.device)), "estimate: resize")
    return out


def estimate(net: torch.nn.Module, img1: torch.Tensor, img2: torch.Tensor, tensor: bool = False):
    # Ensure that both the first and second images have the same dimension (inference.py:32-33)
    assert (img1.size(2) == img2.size(2))
    assert (img1.size(3) == img2.size(3))
    if not img1.is_cuda:
        raise NotImplementedError("estimate: GPU tensors only")
    W, H = img1.size(3), img1.size(2)
    aw = int(math.floor(math.ceil(W / 32.0) * 32.0))
    ah = int(math.floor(math.ceil(H / 32.0) * 32.0))
    sw, sh = float(W) / float(aw), float(H) / float(ah)
    a = img1.detach().contiguous().float()
    b = img2.detach().contiguous().float()
    if (ah, aw) != (H, W):
        a = _resize(a, ah, aw)
        b = _resize(b, ah, aw)
    with torch.set_grad_enabled(False):
        if net.training:                # the reference calls net.eval() unconditionally (inference.py:52); it walks every submodule
            net.eval()
        raw = net(a, b)
    if raw.shape[2:] == (H, W) and sw == 1.0 and sh == 1.0:
        flow = raw                      # same-size bilinear resize is the identity; scale factors are 1
    else:
        flow = _resize(raw.contiguous(), H, W, mul=(sw, sh))
    if tensor:
        return flow.detach()
    return torch.squeeze(flow).permute(1, 2, 0).detach().cpu().numpy()


def _to_input(image, device) -> torch.Tensor:
    """PIL image / uint8 [H,W,3] array -> float32 [1,3,H,W] in [0,1] on `device` (ToTensor's arithmetic: byte / 255)."""
    arr = np.asarray(image.convert("RGB") if hasattr(image, "convert") else image)
    if arr.dtype != np.uint8 or arr.ndim != 3 or arr.shape[2] != 3:
        raise ValueError(f"Inference.parser: expected an RGB image, got {arr.dtype} {arr.shape}")
    t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1))).to(torch.float32).div_(255.0)
    return t.unsqueeze(0).to(device)


class Inference:
    """`Inference(net, netname, output_dir, device)`; the piece of it run.py uses is the static `parser`."""

    def __init__(self, net, netname=None, output_dir="./outputs", device="cuda"):
        self.netname = "test" if netname is None else os.path.splitext(os.path.basename(netname))[0]
        self.default = os.path.join(output_dir, self.netname)
        self.device = device
        self.net = net

    @staticmethod
    def parser(net, im1, im2, device="cuda"):
        size1 = im1.size if hasattr(im1, "convert") else np.asarray(im1).shape[:2]
        size2 = im2.size if hasattr(im2, "convert") else np.asarray(im2).shape[:2]
        assert size1 == size2
        return estimate(net, _to_input(im1, device), _to_input(im2, device))

    def images_parsing(self, imgdir: str, pair: bool = True, write: bool = True):
        """Every pair of the folder -> <output_dir>/<netname>/<folder>_parse/<name>_out.flo; returns the flows."""
        import PIL.Image
        from .datasets import image_files_from_folder, pair_files
        from .flo import flowname_modifier, write_flow
        if not isinstance(imgdir, str):
            raise ValueError("Unknown input! Input must be a directory path")
        if not os.path.isdir(imgdir):
            raise ValueError(f"Input directory is NOT found! At {imgdir}")
        outdir = os.path.join(self.default, os.path.basename(imgdir) + "_parse")
        os.makedirs(outdir, exist_ok=True)
        flows = []
        for first, second, _ in pair_files(image_files_from_folder(imgdir, pair=pair), is_pair=pair):
            with PIL.Image.open(first) as a, PIL.Image.open(second) as b:
                flow = self.parser(self.net, a.convert("RGB"), b.convert("RGB"), device=self.device)
            if write:
                write_flow(flow, flowname_modifier(first, outdir, pair=pair))
            flows.append(flow)
        return flows

    def video_parsing(self, *args, **kwargs):
        raise NotImplementedError("Inference.video_parsing needs cv2 / imutils (camera and video front end): outside the hot path")

    def dataloader_parsing(self, *args, **kwargs):
        raise NotImplementedError("Inference.dataloader_parsing needs the reference's training datasets: use run.py or images_parsing")
