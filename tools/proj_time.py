"""Projection kernel with and without the tile-list filling (rocprofv3 this for kernel times)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import WORKLOADS, make_inputs  # noqa: E402
from gaussianimage_amd import _lib  # noqa: E402
from gaussianimage_amd.ops import _ptr, _stream  # noqa: E402
from gaussianimage_amd.pipeline import FramePipeline  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "uhd200k"
dev = torch.device("cuda:0")
H, W, N, kind, _ = WORKLOADS[wl]
inputs = make_inputs(wl, dev, seed=1)
pipe = FramePipeline(N, H, W, kind=kind, device=dev)
pipe.forward(*inputs)
torch.cuda.synchronize()
outs = (_ptr(pipe.xys), _ptr(pipe.depths), _ptr(pipe.radii), _ptr(pipe.conics),
        _ptr(pipe.num_tiles_hit))
for fill in (False, True):
    for _ in range(50):
        if fill:
            pipe.forward(*inputs)
        else:
            _lib.check(_lib.lib.gi2d_project_cholesky_fwd(
                _ptr(inputs[0]), _ptr(inputs[1]), N, H, W, pipe.tiles_x, pipe.tiles_y,
                ctypes.byref(pipe.conv), *outs, None, _stream(dev)), "project")
    torch.cuda.synchronize()
print("done")
