#!/usr/bin/env python3
"""Counterpart of the reference's run.py (run.py:24-42 flags, :137-168 main_dl, :203-271 driver) on the HIP path.

  python run.py --model piv -i DIR [-i DIR2 ...] -o OUT [-p] [-s N] [-n N] [--weights FILE] [--batch B]

Same flags and output layout (OUT/<netname>/<input-basename>[-start_num]/flow[/left|right]/<name>_out.flo, args.txt).
Differences: `--weights` names the state dict (the reference hard-codes models/pretrain_torch/*.paramOnly, which are not
shipped); without it, or with `--synthetic_weights`, the seeded generator of pivlfn.synth is used.  `--no_cuda` is an
error (there is no CPU path).  Pairs of equal size are batched (`--batch`); frames are decoded once on a prefetch thread,
staged in pinned memory and converted uint8 -> fp32 on the device; flows come back on a copy stream and `.flo` files are
written by a background writer while the next batch computes (pivlfn.pipeline), and under torch.distributed.run the pairs are sharded over the ranks (pivlfn.dist.shard_bounds).
"""
import argparse
import os
import sys

import torch

HERE = os.path.dirname(os.path.realpath(__file__))
sys.path.insert(0, HERE)

from pivlfn import Network, estimate                     # noqa: E402
from pivlfn.datasets import Run                          # noqa: E402
from pivlfn.dist import shard_bounds                     # noqa: E402
from pivlfn.flo import FloWriter, flowname_modifier      # noqa: E402
from pivlfn.pipeline import PairLoader, stream_pairs     # noqa: E402
from pivlfn import synth                                 # noqa: E402

parser = argparse.ArgumentParser(description="Inferencing script for LiteFlowNet (MI355X-native path)")
parser.add_argument("--start", "-s", type=int, default=0, help="Input image starting index.")
parser.add_argument("--num_images", "-n", type=int, default=-1, help="Number of image(s) to process from the directory.")
parser.add_argument("--is_pair", "-p", action="store_true", help="To check if the input image format is in pair.")
parser.add_argument("--model", "-m", type=str, choices=["hui", "piv"], required=True)
parser.add_argument("--version", "-v", type=int, choices=[1, 2], default=1)
parser.add_argument("--input", "-i", default=["./images/demo"], type=str, nargs="+", help="Input images directory(ies).")
parser.add_argument("--output", "-o", default="./results", type=str, help="Main output directory.")
parser.add_argument("--no_cuda", action="store_true")
parser.add_argument("--weights", type=str, default=None, help="state dict file (torch.load); default: seeded synthetic weights")
parser.add_argument("--synthetic_weights", action="store_true")
parser.add_argument("--batch", type=int, default=4, help="pairs per forward")


def output_dirs(args, imdir, netname):
    """run.py:232-266 of the reference."""
    is_all_flow = (args.start == 0) and (args.num_images < 0)
    num_images = "end" if args.num_images < 0 else args.num_images
    checkname = os.path.basename(os.path.normpath(imdir))
    if checkname.lower() in ["left", "right"]:
        extradir, bname = checkname.lower(), os.path.basename(os.path.dirname(os.path.normpath(imdir)))
    else:
        extradir, bname = None, checkname
    outsubdir = f"{bname}-{args.start}_{num_images}" if not is_all_flow else bname
    save = os.path.join(args.output, netname, outsubdir)
    flodir = os.path.join(save, "flow") if extradir is None else os.path.join(save, "flow", extradir)
    return save, flodir, ("args.txt" if extradir is None else f"args_{extradir}.txt")


def main_dl(net, inputdir, savedir, is_pair, start_id, num_images, device, batch, rank=0, world=1):
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


def main(argv=None):
    args = parser.parse_args(argv)
    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("run.py: this build has no CPU path (the reference's correlation has none either, "
                         "src/correlation.py:339-340); a GPU is required")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    if args.weights and not args.synthetic_weights:
        weights = torch.load(args.weights, map_location="cpu")
        netname = os.path.splitext(os.path.basename(args.weights))[0]
    else:
        weights = synth.generate_weights(args.model + ("2" if args.version == 2 else ""), 0)
        netname = f"{args.model}{'2' if args.version == 2 else ''}-synthetic"
    net = Network(model=args.model, params=weights, version=args.version).to(device).eval()
    total = 0
    for i, imdir in enumerate(args.input):
        print(f"---------- Processing images from directory #{str(i).zfill(2)}: '{imdir}'")
        save, flodir, argsname = output_dirs(args, imdir, netname)
        os.makedirs(save, exist_ok=True)
        if rank == 0:
            with open(os.path.join(save, argsname), "w") as f:
                for k, v in sorted(vars(args).items()):
                    f.write(f"{k}: {v}\n")
        total += main_dl(net, imdir, flodir, args.is_pair, args.start, args.num_images, device, args.batch, rank, world)
    print(f"Finish processing {total} pairs")
    return total


if __name__ == "__main__":
    main()
