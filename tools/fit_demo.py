"""BASELINE config 4 as a whole: a 50 000-iteration fit (the reference's default,
/root/reference/train.py:287) of a synthetic ~1020x768 target with 50 000 Gaussians through
the fused fitting step; prints one JSON line with wall time and PSNR checkpoints."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gaussianimage_amd._testing import synth  # noqa: E402
from gaussianimage_amd.fit import FusedFitter  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50000)
ap.add_argument("--points", type=int, default=50000)
ap.add_argument("--height", type=int, default=768)
ap.add_argument("--width", type=int, default=1020)
ap.add_argument("--kind", default="cholesky")
ap.add_argument("--footprint", type=float, default=1.0,
                help="scale of the initial Cholesky factors (3 = supports like a fitted model's)")
ap.add_argument("--raster-ppl", type=int, default=0, choices=[0, 1, 2, 4],
                help="workgroup shape of the rasterisers (0 = the library's choice)")
ap.add_argument("--spatial-order", action="store_true")
args = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(1)
fit = FusedFitter(args.points, args.height, args.width, kind=args.kind, device=dev, lr=1e-3,
                  keep_image=False, spatial_order=args.spatial_order)
if args.raster_ppl:
    fit.ws.set_raster_ppl(args.raster_ppl)
if args.footprint != 1.0 and args.kind == "cholesky":
    bound = torch.tensor([0.5, 0.0, 0.5], device=dev)
    fit.params["_cholesky"].copy_((fit.params["_cholesky"] + bound) * args.footprint - bound)
gt = synth.target_image(args.height, args.width).to(dev)
marks = {}
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(1, args.iters + 1):
    fit.step(gt)
    if it in (1, 100, 1000, 10000, 25000, args.iters):
        marks[it] = round(fit.psnr(), 3)  # synchronises: 6 times in the whole fit
        print(f"iter {it}: psnr {marks[it]} dB, {time.perf_counter() - t0:.2f} s", flush=True)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(json.dumps({"workload": f"{args.width}x{args.height}, {args.points} Gaussians, {args.kind}",
                  "iterations": args.iters, "wall_s": wall, "ms_per_iter": wall / args.iters * 1e3,
                  "psnr_db": marks, "capacity": fit.ws.info()["capacity"],
                  "max_tile_count": fit.ws.info()["max_tile_count"]}))
