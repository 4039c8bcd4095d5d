"""Where the host time of the drop-in path goes: cProfile over the reference-style loops."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gaussianimage_amd._testing import harness, synth  # noqa: E402

dev = "cuda:0"
torch.manual_seed(1)
model = harness.CholeskyImageModel(30000, 512, 768, dev)
gt = synth.target_image(512, 768).to(dev)
for _ in range(20):
    model.train_iter(gt)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    model.train_iter(gt)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
