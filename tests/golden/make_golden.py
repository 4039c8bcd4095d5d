"""Generates the golden fixtures of this directory from the CPU oracle.

PARITY UNPINNED: the reference holds no fixtures for this path and its operators cannot be
run here (their source is an un-vendored submodule), so these vectors pin THIS repository's
oracle (regression), not the reference.  They carry the caller's sequence of
/root/reference/gaussianimage_cholesky.py:76-85,101-116 -- raw parameters in, render / loss /
parameter gradients out -- computed by the fp64 differentiable restatement, plus the fp32 C
restatement's operator outputs for the same inputs.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from gaussianimage_amd._testing import synth  # noqa: E402
from oracle import cport, torch_oracle as to  # noqa: E402

CASES = [(48, 64, 64, 0), (48, 64, 512, 1), (80, 96, 64, 2), (80, 96, 512, 0)]


def make(kind, H, W, N, seed):
    gt = synth.target_image(H, W, seed=7)
    if kind == "cholesky":
        raw = synth.cholesky_params(N, seed=seed)
        fwd = to.model_forward_cholesky
        means, L, col, op = synth.cholesky_inputs(N, seed=seed)
        op_in = (means, L)
        proj = cport.project_cholesky_fwd(means.numpy(), L.numpy(), H, W)
    else:
        raw = synth.rs_params(N, seed=seed)
        fwd = to.model_forward_rs
        means, sc, rot, col, op = synth.rs_inputs(N, seed=seed)
        op_in = (means, sc, rot)
        proj = cport.project_rs_fwd(means.numpy(), sc.numpy(), rot.numpy(), H, W)
    leaves = [p.double().requires_grad_(True) for p in raw]
    img = fwd(*leaves, H, W)
    loss = torch.nn.functional.mse_loss(img, gt.double())
    loss.backward()
    xys, depths, radii, conics, nth = proj
    offs, ids = cport.bin_tiles(xys, radii, H, W)
    out, margin = cport.rasterize_sum_fwd(xys, conics, col.numpy(), op.numpy(), H, W, offs, ids,
                                          want_margin=True)
    d = {"H": H, "W": W, "N": N, "seed": seed, "gt": gt.numpy(),
         "render64": img.detach().numpy().astype(np.float32), "loss64": loss.item(),
         "xys": xys, "radii": radii, "conics": conics, "num_tiles_hit": nth,
         "tile_offsets": offs, "ids": ids, "image": out, "margin": margin}
    # the activated operator inputs exactly as the C restatement received them: the GPU test
    # feeds these bits to the projection and asks for bit-equal outputs (activations computed
    # on another device or CPU differ in their last bits, which is not the projection's doing)
    for i, t in enumerate(op_in):
        d[f"in{i}"] = t.numpy()
    for i, (p, leaf) in enumerate(zip(raw, leaves)):
        d[f"param{i}"] = p.numpy()
        d[f"grad{i}"] = leaf.grad.numpy().astype(np.float32)
    return d


if __name__ == "__main__":
    for kind in ("cholesky", "rs"):
        for H, W, N, seed in CASES:
            name = f"{kind}_{W}x{H}_n{N}_s{seed}.npz"
            np.savez_compressed(os.path.join(HERE, name), **make(kind, H, W, N, seed))
            print("wrote", name)
