#!/usr/bin/env python3
"""Counterpart of the reference's run.py on the HIP path: same flags, same output tree.

  python run.py --model piv -i DIR [-i DIR2 ...] -o OUT [-p] [-s N] [-n N] [-b F ...] [-c F ...] [-v 1|2]
                [--weights FILE] [--batch B]

Flags of the reference (run.py:24-42): --start/-s, --num_images/-n, --is_pair/-p, --brightness/-b, --contrast/-c,
--model/-m, --version/-v, --input/-i, --output/-o, --no_cuda.  Output tree (run.py:232-266):
OUT/<netname>/<input-basename>[-<start>_<count|end>]/flow[/left|right]/<name>_out.flo plus args[_left|_right].txt.
Without -b/-c every pair of the folder goes through `estimate` (reference main_dl, run.py:137-168); with -b and/or -c the
folder is read as a frame sequence and every consecutive pair is estimated once per (brightness, contrast) combination on
frames modified like the reference's `image_mod` (run.py:88-134), each flow named <prefix>_<BBB>_<CCC>_<suffix>_out.flo.

Differences, all deliberate:
  * `--weights FILE` names the state dict (the reference hard-codes models/pretrain_torch/*.paramOnly, which are not
    shipped, .MISSING_LARGE_BLOBS); without it the seeded generator of pivlfn.synth stands in and the net is called
    <model>-synthetic;  `--model` defaults to piv (the reference has no default and then fails on a None model);
  * `--no_cuda` (or no GPU) is an error: there is no CPU path, as in src/correlation.py:339-340;
  * pairs of equal size are batched (`--batch`), frames are decoded once on a prefetch thread, staged in pinned memory and
    converted on the device; flows return on a copy stream and a background writer closes the .flo files while the next
    batch computes (pivlfn.pipeline); under torch.distributed.run the pairs are sharded over the ranks
    (pivlfn.dist.shard_bounds);
  * a trailing slash on an input directory is ignored (the reference would name the output directory '');
  * with -b/-c a frame whose file name has no '_' gets the tag appended (<stem>_<BBB>_<CCC>_out.flo) -- the reference splits
    the whole path at its last '_' and then either fails or lets the combinations overwrite each other.
"""
import argparse
import os
import sys
from dataclasses import dataclass
from itertools import product
from typing import List, Optional, Sequence, Tuple

import torch

HERE = os.path.dirname(os.path.realpath(__file__))
sys.path.insert(0, HERE)

from pivlfn import Network                               # noqa: E402
from pivlfn import synth                                 # noqa: E402
from pivlfn.datasets import Run, image_files_from_folder, pair_files     # noqa: E402
from pivlfn.dist import shard_bounds                     # noqa: E402
from pivlfn.flo import FloWriter, flowname_modifier      # noqa: E402
from pivlfn.imagemod import mod_name                     # noqa: E402
from pivlfn.pipeline import PairLoader, stream_pairs     # noqa: E402

parser = argparse.ArgumentParser(description="Inferencing script for LiteFlowNet (MI355X-native path)")
parser.add_argument("--start", "-s", type=int, default=0, help="Input image starting index.")
parser.add_argument("--num_images", "-n", type=int, default=-1, help="Number of image(s) to process from the directory.")
parser.add_argument("--is_pair", "-p", action="store_true", help="To check if the input image format is in pair.")
parser.add_argument("--brightness", "-b", default=None, type=float, nargs="+",
                    help="Add brightness factor to modify all the input images (optional).")
parser.add_argument("--contrast", "-c", default=None, type=float, nargs="+",
                    help="Add contrast factor to modify all the input images (optional).")
parser.add_argument("--model", "-m", type=str, choices=["hui", "piv"], default="piv", help="Select which model to solve the problem!")
parser.add_argument("--version", "-v", type=int, choices=[1, 2], default=1,
                    help="Select the LiteFlowNet model backbone version (i.e., LiteFlowNet or LiteFlowNet2)!")
parser.add_argument("--input", "-i", default=["./images/demo"], type=str, nargs="+", help="Input images directory(ies).")
parser.add_argument("--output", "-o", default="./results", type=str, help="Main output directory.")
parser.add_argument("--no_cuda", action="store_true")
parser.add_argument("--weights", type=str, default=None, help="state dict file (torch.load); default: seeded synthetic weights")
parser.add_argument("--synthetic_weights", action="store_true")
parser.add_argument("--batch", type=int, default=4, help="pairs per forward")
parser.add_argument("--precision", type=str, default=None, choices=["fp32", "fp32_wino_mfma32", "fp32_direct", "fp32_split", "fp32_split3", "fp16"],
                    help="how the large convolutions multiply (not a reference flag; default: the library's, fp32 -- "
                         "see Network.precision)")


@dataclass(frozen=True)
class OutputLayout:
    """Where the results of one input directory go."""
    save: str          # OUT/<netname>/<label>
    flow: str          # <save>/flow[/left|right]
    args_file: str     # <save>/args[_left|_right].txt

    @staticmethod
    def of(out_root: str, netname: str, input_dir: str, start: int, num_images: int) -> "OutputLayout":
        parts = os.path.normpath(input_dir).split(os.sep)
        side = parts[-1].lower() if parts[-1].lower() in ("left", "right") else None      # stereo halves share one parent label
        label = parts[-2] if side and len(parts) > 1 else parts[-1]
        if start != 0 or num_images >= 0:                                                  # a slice of the folder says so in its name
            label += f"-{start}_{'end' if num_images < 0 else num_images}"
        save = os.path.join(out_root, netname, label)
        return OutputLayout(save=save,
                            flow=os.path.join(save, "flow", side) if side else os.path.join(save, "flow"),
                            args_file=os.path.join(save, f"args_{side}.txt" if side else "args.txt"))


def output_dirs(args, imdir, netname):
    """(save, flow dir, args file name) for one input directory."""
    lay = OutputLayout.of(args.output, netname, imdir, args.start, args.num_images)
    return lay.save, lay.flow, os.path.basename(lay.args_file)


class _FrameSequence:
    """Consecutive frames of a folder as pairs, each called by the PATH of its first frame (the -b/-c path names its
    outputs from that path, run.py:126-129).  Listing as the reference's getpair (run.py:48-70): lower-case extensions."""

    def __init__(self, folder: str, n_images: int, start_at: int):
        if not os.path.isdir(folder):
            raise ValueError(f"Input image directory is NOT found! '{folder}'")
        if n_images == 1:
            raise ValueError("--num_images 1 leaves no pair to process")
        files = image_files_from_folder(folder, pair=False, upper=False, n_images=n_images, start_at=start_at)
        pairs = pair_files(files, is_pair=False)
        self.image_list = [[a, b] for a, b, _ in pairs]
        self.name_list = [a for a, _, _ in pairs]

    def __len__(self):
        return len(self.name_list)


def mod_flow_name(first_frame: str, savedir: str, mod: Tuple[float, float]) -> str:
    """<savedir>/<prefix>_<BBB>_<CCC>_<suffix>_out.flo, prefix/suffix = the frame's file name split at its last '_'."""
    stem = os.path.splitext(os.path.basename(first_frame))[0]
    prefix, sep, suffix = stem.rpartition("_")
    tagged = f"{prefix}_{mod_name(*mod)}_{suffix}" if sep else f"{stem}_{mod_name(*mod)}"
    return flowname_modifier(tagged, savedir, pair=False)


def main_dl(net, inputdir, savedir, is_pair, start_id, num_images, device, batch, rank=0, world=1):
    """Every pair of the folder through `estimate` (reference main_dl, run.py:137-168)."""
    os.makedirs(savedir, exist_ok=True)
    ds = Run(root=inputdir, is_pair=is_pair, n_images=num_images, start_at=start_id)
    lo, hi = shard_bounds(len(ds), rank, world)
    print(f"Processing {hi - lo} of {len(ds)} pairs of images (rank {rank}/{world})...")
    loader = PairLoader(ds, lo, hi, batch, depth=2, pin=device.type == "cuda")
    try:
        with FloWriter() as writer:
            n = stream_pairs(net, loader, device,
                             lambda flow, name: writer.submit(flow, flowname_modifier(name, savedir, pair=False)))
    finally:
        loader.close()
    assert n == hi - lo
    return hi - lo


def main_mod(net, inputdir, savedir, start_id, num_images, device, mod_factors: Sequence[Tuple[float, float]], batch,
             rank=0, world=1):
    """Consecutive frames x (brightness, contrast) combinations (reference main, run.py:100-134)."""
    os.makedirs(savedir, exist_ok=True)
    ds = _FrameSequence(inputdir, num_images, start_id)
    lo, hi = shard_bounds(len(ds), rank, world)
    print(f"Processing {hi - lo} of {len(ds)} pairs of images x {len(mod_factors)} modifications (rank {rank}/{world})...")
    loader = PairLoader(ds, lo, hi, batch, depth=2, pin=device.type == "cuda")
    try:
        with FloWriter() as writer:
            n = stream_pairs(net, loader, device,
                             lambda flow, first, mod: writer.submit(flow, mod_flow_name(first, savedir, mod)),
                             mods=list(mod_factors))
    finally:
        loader.close()
    assert n == (hi - lo) * len(mod_factors)
    return n


def load_weights(args) -> Tuple[dict, str]:
    if args.weights and not args.synthetic_weights:
        if not os.path.isfile(args.weights):
            raise ValueError("Unknown params input!")
        return torch.load(args.weights, map_location="cpu"), os.path.splitext(os.path.basename(args.weights))[0]
    tag = args.model + ("2" if args.version == 2 else "")
    return synth.generate_weights(tag, 0), f"{tag}-synthetic"


def main(argv: Optional[List[str]] = None) -> int:
    args = parser.parse_args(argv)
    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("run.py: this build has no CPU path (the reference's correlation has none either, "
                         "src/correlation.py:339-340); a GPU is required")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())   # ranks may share a card
    torch.cuda.set_device(device)
    weights, netname = load_weights(args)
    net = Network(model=args.model, params=weights, version=args.version).to(device).eval()
    if args.precision is not None:
        net.precision = args.precision
    mods = None
    if args.brightness is not None or args.contrast is not None:
        mods = list(product(tuple(args.brightness or (1.0,)), tuple(args.contrast or (1.0,))))
    total = 0
    for i, imdir in enumerate(args.input):
        print(f"---------- Processing images from directory #{str(i).zfill(2)}: '{imdir}'")
        lay = OutputLayout.of(args.output, netname, imdir, args.start, args.num_images)
        os.makedirs(lay.save, exist_ok=True)
        if rank == 0:
            with open(lay.args_file, "w") as f:
                for k, v in sorted(vars(args).items()):
                    f.write(f"{k}: {v}\n")
        if mods is None:
            total += main_dl(net, imdir, lay.flow, args.is_pair, args.start, args.num_images, device, args.batch, rank, world)
        else:
            total += main_mod(net, imdir, lay.flow, args.start, args.num_images, device, mods, args.batch, rank, world)
    print(f"Finish processing {total} flow fields")
    return total


if __name__ == "__main__":
    main()
