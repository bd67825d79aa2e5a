"""Input side of run.py: which files form the image pairs of a folder, and how a frame becomes a tensor.

Behaviour follows the reference's `Run` dataset (src/datasets.py:438-487) and its folder listing (src/utils_data.py:13-33):
  paired mode    every `<base>_img1.<ext>` is matched with `<base>_img2.<ext>`; the pair is called <base>;
  sequence mode  the sorted frames f0, f1, f2, ... give the pairs (f0,f1), (f1,f2), ...; a pair is called after its first frame;
  a pair with a missing file is dropped; start_at / n_images slice the FILE list before pairing.
Frames are decoded with PIL to RGB and scaled to float32 [3,H,W] in [0,1] (what torchvision's ToTensor does; torchvision
itself is not needed).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Sequence, Tuple

import numpy as np
import torch

EXTENSIONS = ("jpg", "jpeg", "png", "bmp", "tif", "ppm")
_FIRST, _SECOND = "_img1", "_img2"


def image_files_from_folder(folder: str, pair: bool = True, upper: bool = True, n_images: int = -1, start_at: int = 0,
                            extensions: Sequence[str] = EXTENSIONS) -> List[str]:
    """Image files of `folder` in the reference's order: extension by extension (lower-case spelling first, then the
    upper-case one when `upper`), names sorted inside each group.  `pair` keeps only first frames (`*_img1.<ext>`)."""
    names = sorted(n for n in os.listdir(folder) if not n.startswith("."))
    by_suffix = {}
    for n in names:
        stem, dot, suffix = n.rpartition(".")
        if dot and (not pair or stem.endswith(_FIRST)):
            by_suffix.setdefault(suffix, []).append(os.path.join(folder, n))
    spellings: Iterable[str] = (s for e in extensions for s in ((e, e.upper()) if upper else (e,)))
    files = [f for s in spellings for f in by_suffix.get(s, ())]
    stop = None if n_images < 0 else start_at + n_images
    return files[start_at:stop]


def pair_files(files: Sequence[str], is_pair: bool) -> List[Tuple[str, str, str]]:
    """(first, second, name) for every pair whose two files exist."""
    if is_pair:
        cand = []
        for first in files:
            stem, ext = os.path.splitext(first)          # stem ends in _img1 (that is how the list was made)
            base = stem[:-len(_FIRST)] if stem.endswith(_FIRST) else stem.rpartition("_")[0]
            cand.append((first, base + _SECOND + ext, os.path.basename(base)))
    else:
        cand = [(a, b, os.path.splitext(os.path.basename(a))[0]) for a, b in zip(files, files[1:])]
    return [c for c in cand if os.path.isfile(c[0]) and os.path.isfile(c[1])]


def read_image(path: str) -> torch.Tensor:
    """Decode to RGB, float32 [3,H,W] in [0,1]."""
    import PIL.Image
    with PIL.Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(np.ascontiguousarray(rgb.transpose(2, 0, 1))).to(torch.float32).div_(255.0)


class Run(torch.utils.data.Dataset):
    """`Run(root, is_pair, n_images, start_at)`: item i = ([img1, img2], name), index taken modulo the length."""

    def __init__(self, root: str, is_pair: bool = True, n_images: int = -1, start_at: int = 0) -> None:
        if not os.path.isdir(root):
            raise ValueError(f"Input image directory is NOT found! '{root}'")
        files = image_files_from_folder(root, pair=is_pair, upper=False, n_images=n_images, start_at=start_at)
        pairs = pair_files(files, is_pair)
        self.image_list = [[a, b] for a, b, _ in pairs]
        self.name_list = [name for _, _, name in pairs]
        self.size = len(pairs)

    def __len__(self) -> int:
        return self.size

    def __getitem__(self, index: int):
        first, second = self.image_list[index % self.size]
        return [read_image(first), read_image(second)], self.name_list[index % self.size]
