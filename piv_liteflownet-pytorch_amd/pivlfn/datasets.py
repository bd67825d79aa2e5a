"""Input side of run.py: the `Run` dataset semantics of the reference (src/datasets.py:438-487) and its file listing
(src/utils_data.py:13-33, 46-56), without torchvision: `ToTensor` is restated as uint8 HWC -> float32 CHW / 255.

is_pair=True : every `*_img1.<ext>` in the folder is paired with `<same base>_img2.<ext>`; name = the common base.
is_pair=False: the sorted frame sequence, pair i = (frame i, frame i+1); name = basename of frame i without extension.
"""
from __future__ import annotations

import os
from glob import glob
from typing import List, Tuple

import numpy as np
import torch

EXTENSIONS = ("jpg", "jpeg", "png", "bmp", "tif", "ppm")


def image_files_from_folder(folder: str, pair: bool = True, upper: bool = True, n_images: int = -1, start_at: int = 0,
                            extensions: Tuple[str, ...] = EXTENSIONS) -> List[str]:
    files: List[str] = []
    for ext in extensions:
        pat = f"*_img1.{ext}" if pair else f"*.{ext}"
        files += sorted(glob(os.path.join(folder, pat)))
        if upper:
            pat_u = f"*_img1.{ext.upper()}" if pair else f"*.{ext.upper()}"
            files += sorted(glob(os.path.join(folder, pat_u)))
    return files[start_at:] if n_images < 0 else files[start_at:start_at + n_images]


def read_image(path: str) -> torch.Tensor:
    """PIL -> RGB -> float32 [3,H,W] in [0,1] (read_gen + ToTensor of the reference)."""
    import PIL.Image
    im = PIL.Image.open(path).convert("RGB")
    a = np.asarray(im, dtype=np.uint8)
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(torch.float32).div_(255.0)


class Run(torch.utils.data.Dataset):
    def __init__(self, root: str, is_pair: bool = True, n_images: int = -1, start_at: int = 0) -> None:
        if not os.path.isdir(root):
            raise ValueError(f"Input image directory is NOT found! '{root}'")
        file_list = image_files_from_folder(root, pair=is_pair, n_images=n_images, start_at=start_at, upper=False)
        prev_file = None
        self.image_list, self.name_list = [], []
        for file in file_list:
            if is_pair:
                imbase, imext = os.path.splitext(os.path.basename(str(file)))
                fbase = imbase.rsplit("_", 1)[0]
                img1, img2 = file, os.path.join(root, str(fbase) + "_img2" + imext)
            else:
                if prev_file is None:
                    prev_file = file
                    continue
                img1, img2 = prev_file, file
                fbase = os.path.splitext(os.path.basename(str(img1)))[0]
                prev_file = file
            if not os.path.isfile(img1) or not os.path.isfile(img2):
                continue
            self.image_list.append([img1, img2])
            self.name_list.append(fbase)
        self.size = len(self.name_list)

    def __len__(self) -> int:
        return self.size

    def __getitem__(self, index: int):
        index = index % self.size
        return [read_image(self.image_list[index][0]), read_image(self.image_list[index][1])], self.name_list[index]
