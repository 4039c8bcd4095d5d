import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from gaussianimage_amd.pipeline import FramePipeline
dev = torch.device("cuda:0")
H, W, N, kind, fp = bench.WORKLOADS["kodak30k"]
inputs = bench.make_inputs("kodak30k", dev, seed=1)
run = FramePipeline(N, H, W, kind=kind, device=dev).bind(*inputs)
for _ in range(50): run()
torch.cuda.synchronize()
for n in (100, 300):
    t0 = time.perf_counter()
    for _ in range(n): run()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(n, "host issue us/frame", (t1 - t0) / n * 1e6, "total us/frame", (time.perf_counter() - t0) / n * 1e6)
