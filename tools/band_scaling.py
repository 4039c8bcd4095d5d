"""Predicts tile-row scaling on one GPU: times the band a rank would own when the 3840x2160 /
200k frame is split over R ranks (rank 0's and a middle rank's band), forward only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import WORKLOADS, make_inputs  # noqa: E402
from gaussianimage_amd.pipeline import FramePipeline  # noqa: E402
from gaussianimage_amd.sharding import band_plan  # noqa: E402

dev = torch.device("cuda:0")
H, W, N, kind, _ = WORKLOADS["uhd200k"]
inputs = make_inputs("uhd200k", dev, seed=1)
tiles_y = (H + 15) // 16
for R in (1, 2, 4, 8):
    plan = band_plan(tiles_y, R)
    for r in sorted({0, R // 2}):
        pipe = FramePipeline(N, H, W, kind=kind, device=dev, rows=plan[r])
        for _ in range(10):
            pipe.forward(*inputs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            pipe.forward(*inputs)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 200 * 1e3
        tp = time.perf_counter()
        for _ in range(200):
            pipe.project(*inputs[:3])
        torch.cuda.synchronize()
        mp = (time.perf_counter() - tp) / 200 * 1e3
        print(f"R={R} rank={r} rows={plan[r]} forward {ms * 1e3:.1f} us  (projection alone "
              f"{mp * 1e3:.1f} us)", flush=True)
