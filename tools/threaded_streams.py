"""Forward throughput with several streams and host threads, each configuration on fresh
pipelines/streams (REPEAT=n repeats a configuration on its streams; DUMMY_STREAMS=n takes n
streams from torch's pool first).  Shows how much the figure depends on which hardware queues
the runtime gives the streams: 97-150 k frames/s for four streams at 768x512 / 30k."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from gaussianimage_amd.pipeline import FramePipeline  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "kodak30k"
dev = torch.device("cuda:0")
H, W, N, kind, fp = bench.WORKLOADS[name]
inputs = bench.make_inputs(name, dev, seed=1)
dummies = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("DUMMY_STREAMS", "0")))]
for threads, per_thread in ((1, 4), (1, 4), (2, 2), (1, 8)):
    runs = []
    for _ in range(threads * per_thread):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            r = FramePipeline(N, H, W, kind=kind, device=dev).bind(*inputs)
            for _ in range(5):
                r()
        print("   stream", hex(st.cuda_stream), flush=True)
        runs.append(r)
    torch.cuda.synchronize()
    frames = 3000

    def work(mine):
        for k in range(frames // threads):
            mine[k % len(mine)]()

    ths = [threading.Thread(target=work, args=(runs[i * per_thread:(i + 1) * per_thread],))
           for i in range(threads)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{threads} threads x {per_thread} streams: {frames / el:.0f} frames/s", flush=True)
    for rep in range(int(os.environ.get("REPEAT", "0"))):
        ths = [threading.Thread(target=work, args=(runs[i * per_thread:(i + 1) * per_thread],))
               for i in range(threads)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        print(f"    again: {frames / (time.perf_counter() - t0):.0f} frames/s", flush=True)
