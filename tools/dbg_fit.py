import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianimage_amd import synth
from gaussianimage_amd.fit import FusedFitter
torch.manual_seed(1)
fit = FusedFitter(30000, 512, 768, device="cuda:0", keep_image=False)
gt = synth.target_image(512, 768).to("cuda:0")
for _ in range(60):
    fit.step(gt)
torch.cuda.synchronize()
print(fit.ws.info())
