"""Host time of the reference's FPS loop (model() under no_grad) on the drop-in path."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gaussianimage_amd._testing import harness  # noqa: E402

dev = "cuda:0"
torch.manual_seed(1)
model = harness.CholeskyImageModel(30000, 512, 768, dev).eval()
with torch.no_grad():
    for _ in range(50):
        model()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        model()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("host issue time per model() %.1f us, with drain %.1f us" % (
        (t1 - t0) / 300 * 1e6, (time.perf_counter() - t0) / 300 * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        model()
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
