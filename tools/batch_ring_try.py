"""Two batches in flight on two streams against one stream (config 2, batches of 8)."""
import gc
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
from gaussianimage_amd import _lib  # noqa: E402
from gaussianimage_amd.pipeline import FrameBatch  # noqa: E402

DEV = torch.device("cuda:0")
H, W, N, kind, fp = bench.WORKLOADS["kodak30k"]
B = 8
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
inputs = bench.make_batch_inputs("kodak30k", DEV, B, 1)
depth_list = (1, 2, 3)
for depth in depth_list:
    streams = [torch.cuda.Stream(device=DEV) for _ in range(depth)]
    runs = []
    for st in streams:
        with torch.cuda.stream(st):
            fb = FrameBatch(B, N, H, W, kind=kind, device=DEV)
            _lib.check(_lib.lib.gi2d_workspace_set_option(fb.ws.handle, _lib.OPT_PROJECT_AGG, mode), "opt")
            runs.append((fb, fb.bind(*inputs)))
    for i in range(60):
        runs[i % depth][1]()
    torch.cuda.synchronize()
    gc.disable()
    res = []
    for rep in range(30):
        t0 = time.perf_counter()
        for i in range(20):
            runs[i % depth][1]()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 20)
    gc.enable()
    res.sort()
    print("mode %d depth %d: us per batch min %.1f median %.1f max %.1f -> %.0f frames/s" % (
        mode, depth, res[0] * 1e6, res[len(res) // 2] * 1e6, res[-1] * 1e6, B / res[len(res) // 2]),
        flush=True)
