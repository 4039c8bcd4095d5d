"""The batched fused fit at 24 images per step, one wave per tile (what bench.py's fused_batch
times): a fixed number of steps for kernel traces and counter passes."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
from gaussianimage_amd._testing import synth  # noqa: E402
from gaussianimage_amd.fit import BatchFitter  # noqa: E402

DEV = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "kodak30k"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ppl = int(sys.argv[3]) if len(sys.argv) > 3 else 4
H, W, N, kind, fp = bench.WORKLOADS[wl]
F = 24
gts = torch.stack([synth.target_image(H, W, seed=f)[0] for f in range(F)]).to(DEV).contiguous()
fit = BatchFitter(F, N, H, W, kind=kind, device=DEV)
fit.ws.set_raster_ppl(ppl)
for _ in range(steps):
    fit.step(gts)
torch.cuda.synchronize()
print("overflowed", fit.overflowed_frames, "max list", fit.ws.info()["max_tile_count"])
