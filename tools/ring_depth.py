"""FrameRing throughput against its depth (fresh ring per measurement, twice each)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from gaussianimage_amd.pipeline import FrameRing  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "kodak30k"
dev = torch.device("cuda:0")
H, W, N, kind, fp = bench.WORKLOADS[name]
inputs = bench.make_inputs(name, dev, seed=1)
for depth in (2, 3, 4, 2, 3, 4):
    ring = FrameRing(N, H, W, kind=kind, device=dev, depth=depth)
    for _ in range(300):
        ring.render(*inputs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3000):
        ring.render(*inputs, wait_inputs=False)
    torch.cuda.synchronize()
    print(f"depth {depth}: {3000 / (time.perf_counter() - t0):.0f} frames/s", flush=True)
