"""Host time of the reference-style loops through the two operator bindings (config 2)."""
import gc
import sys
import time

import torch

sys.path.insert(0, ".")
from gaussianimage_amd import ops  # noqa: E402
from gaussianimage_amd._testing import harness, synth  # noqa: E402

DEV = "cuda:0"
H, W, N = 512, 768, 30000
gt = synth.target_image(H, W).to(DEV)


def timed(fn, n, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    gc.enable()
    return dt


for compiled in (True, False):
    ops.use_compiled(compiled)
    torch.manual_seed(0)
    for name, model in (("cholesky(3 passes)", harness.CholeskyImageModel(N, H, W, DEV)),
                        ("cholesky(1 pass)", harness.CholeskyImageModel(N, H, W, DEV, passes=1)),
                        ("rs", harness.RSImageModel(N, H, W, DEV))):
        with torch.no_grad():
            fwd = timed(lambda: model(), 2000)
        step = timed(lambda: model.train_iter(gt), 1000)
        print("compiled=%s %-20s model() %.1f us (%.0f FPS)   train_iter %.1f us" % (
            compiled, name, fwd * 1e6, 1 / fwd, step * 1e6), flush=True)
