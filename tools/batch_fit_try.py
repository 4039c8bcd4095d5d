"""Batched fused fit: ms per step for several batch sizes and both workgroup shapes (config 2)."""
import gc
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
from gaussianimage_amd._testing import synth  # noqa: E402
from gaussianimage_amd.fit import BatchFitter, FusedFitter  # noqa: E402

DEV = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "kodak30k"
H, W, N, kind, fp = bench.WORKLOADS[wl]


def timed(step, n=200, warm=30):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    gc.enable()
    return dt


gt1 = synth.target_image(H, W).to(DEV)
one = FusedFitter(N, H, W, kind=kind, device=DEV, keep_image=False)
t = timed(lambda: one.step(gt1))
print("single: %.1f us per iteration" % (t * 1e6), flush=True)
for F in (4, 8, 16, 24):
    gts = torch.stack([synth.target_image(H, W, seed=f)[0] for f in range(F)]).to(DEV).contiguous()
    for ppl in (0, 1, 4):
        fit = BatchFitter(F, N, H, W, kind=kind, device=DEV)
        fit.ws.set_raster_ppl(ppl)
        t = timed(lambda: fit.step(gts), n=100, warm=20)
        print("F=%d ppl=%d: %.1f us per step = %.2f us per image-iteration -> %.0f image-iterations/s "
              "(overflowed %d, max list %d)" % (F, ppl, t * 1e6, t * 1e6 / F, F / t,
                                                fit.overflowed_frames, fit.ws.info()["max_tile_count"]),
              flush=True)
        del fit
