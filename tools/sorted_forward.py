"""Forward frames of a bench workload with the Gaussians stored in tile order (run under
rocprofv3 --kernel-trace to see what the shared list atomics do to the projection)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from gaussianimage_amd.pipeline import FramePipeline  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "uhd200k"
dev = torch.device("cuda:0")
H, W, N, kind, fp = bench.WORKLOADS[name]
inputs = bench.make_inputs(name, dev, seed=1)
means = inputs[0]
tcol = ((0.5 * W * means[:, 0] + 0.5 * W) / 16).floor().clamp_(0, (W - 1) // 16)
trow = ((0.5 * H * means[:, 1] + 0.5 * H) / 16).floor().clamp_(0, (H - 1) // 16)
order = torch.argsort(trow * ((W + 15) // 16) + tcol, stable=True)
if len(sys.argv) > 2 and sys.argv[2] == "random":
    order = torch.arange(N, device=dev)
inputs = tuple(None if t is None else t[order].contiguous() for t in inputs)
run = FramePipeline(N, H, W, kind=kind, device=dev).bind(*inputs)
for _ in range(300):
    run()
torch.cuda.synchronize()
