"""The C ABI from a C host: examples/c_host/render_batch.c (plain C99, no Python, no torch in its
process) renders a batch through gi2d_render_batch and one frame through the entry point that
allocates nothing; its images must be, bit for bit, what the Python side renders from the same
parameters."""
import os
import subprocess

import numpy as np
import pytest
import torch

from gaussianimage_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lcg_units(count, state=12345):
    out = np.empty(count, np.float32)
    for i in range(count):
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = np.float32(state >> 8) * np.float32(1.0 / 16777216.0)
    return out


def test_c_host_is_built_from_the_current_sources():
    assert os.path.exists(build.C_HOST_BIN), "python -m gaussianimage_amd.build"
    t = os.path.getmtime(build.C_HOST_BIN)
    assert os.path.getmtime(build.C_HOST_SRC) <= t
    out = subprocess.run([build.C_HOST_BIN], capture_output=True, text=True)
    assert out.returncode == 1 and "usage" in out.stderr   # (loads libgi2d.so, touches no GPU)


@pytest.mark.gpu
@pytest.mark.parametrize("F,N,H,W", [(3, 2000, 96, 128), (5, 6000, 200, 176)])
def test_c_host_renders_the_same_bits(tmp_path, F, N, H, W):
    from gaussianimage_amd.pipeline import FrameBatch
    path = str(tmp_path / "images.bin")
    out = subprocess.run([build.C_HOST_BIN, str(F), str(N), str(H), str(W), path],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "noalloc_matches 1" in out.stdout, out.stdout
    got = np.fromfile(path, np.float32).reshape(F, H, W, 3)
    u = _lcg_units(8 * F * N).reshape(F * N, 8)
    f32 = np.float32
    means = (f32(2.0) * u[:, 0:2] - f32(1.0)).reshape(F, N, 2)
    L = np.stack([u[:, 2] + f32(0.5), u[:, 3], u[:, 4] + f32(0.5)], axis=1).reshape(F, N, 3)
    colors = u[:, 5:8].reshape(F, N, 3)
    dev = "cuda:0"
    fb = FrameBatch(F, N, H, W, kind="cholesky", device=dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    img = fb.forward(t(means), t(L), None, t(colors), torch.ones(F, N, device=dev))
    assert np.array_equal(img.cpu().numpy(), got)
