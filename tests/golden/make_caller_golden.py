"""Generates tests/golden/caller_*.npz by running the REFERENCE'S OWN model files on the CPU.

/root/reference/gaussianimage_cholesky.py and /root/reference/gaussianimage_rs.py are imported
unchanged; `forward()` and `train_iter()` (with the reference's own optimizer.py: Adan, and its
StepLR) run for a few iterations, and what the caller's sequence produces around the operator
boundary is stored: initial parameters, the images of the first forward, and per iteration the
loss, the PSNR, the parameter gradients (captured by an optimizer hook, before the update) and
the updated parameters.

What this pins and what it does not: the three operators behind `gsplat.*` are NOT the
reference's (their source is an un-vendored submodule, /root/reference/.gitmodules:1-3) -- they
resolve here to this repository's differentiable CPU restatement (oracle/torch_oracle.py,
evaluated in float64 and rounded to the caller's float32).  So the operators stay PARITY
UNPINNED; the fixtures pin the SEQUENCE around them -- activations, offsets, clamp / view /
permute, the loss, the optimizer, the scheduler (gaussianimage_cholesky.py:62,74,76-99,101-124;
gaussianimage_rs.py:55-60,74-95) -- as the reference's files execute it.

Modules the model files import at the top and that are not installed in this container
(pytorch_msssim, vector_quantize_pytorch, constriction) are satisfied by empty modules whose
names raise when used: the L2 / quantize=False configuration the fixtures run never touches them.

Only the .npz files travel; nothing of /root/reference is needed to run the tests.

    python tests/golden/make_caller_golden.py        # rewrites tests/golden/caller_*.npz
"""
import importlib
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"

CASES = [("cholesky", 48, 64, 64, 0), ("cholesky", 80, 96, 512, 1),
         ("rs", 48, 64, 64, 0), ("rs", 80, 96, 512, 1)]
ITERS = 3


def _unavailable(name):
    def raiser(*a, **k):
        raise RuntimeError(f"{name} is not installed here and the fixtures must not need it")
    return raiser


def install_modules():
    """`gsplat.*` -> the CPU restatement with the reference's positional signatures; absent
    third-party imports -> empty modules."""
    sys.path.insert(0, ROOT)
    from oracle import torch_oracle as to

    def up(t):
        return t.double()

    def project_gaussians_2d(means2d, L_elements, img_height, img_width, tile_bounds):
        xys, depths, radii, conics, nth = to.project_gaussians_2d(
            up(means2d), up(L_elements), img_height, img_width, tile_bounds)
        return xys.float(), depths.float(), radii, conics.float(), nth

    def project_gaussians_2d_scale_rot(means2d, scales, rotation, img_height, img_width,
                                       tile_bounds):
        xys, depths, radii, conics, nth = to.project_gaussians_2d_scale_rot(
            up(means2d), up(scales), up(rotation), img_height, img_width, tile_bounds)
        return xys.float(), depths.float(), radii, conics.float(), nth

    def rasterize_gaussians_sum(xys, depths, radii, conics, num_tiles_hit, colors, opacity,
                                img_height, img_width, BLOCK_H, BLOCK_W, background=None,
                                return_alpha=False):
        assert BLOCK_H == 16 and BLOCK_W == 16 and not return_alpha
        img = to.rasterize_gaussians_sum(up(xys), radii, up(conics), up(colors), up(opacity),
                                         img_height, img_width, None)
        return img.float()

    pkg = types.ModuleType("gsplat")
    pkg.__path__ = []
    sys.modules["gsplat"] = pkg
    for name, fn in (("project_gaussians_2d", project_gaussians_2d),
                     ("project_gaussians_2d_scale_rot", project_gaussians_2d_scale_rot),
                     ("rasterize_sum", rasterize_gaussians_sum)):
        m = types.ModuleType("gsplat." + name)
        setattr(m, fn.__name__, fn)
        sys.modules["gsplat." + name] = m
        setattr(pkg, name, m)
    for name, attrs in (("pytorch_msssim", ("ms_ssim", "ssim", "SSIM")),
                        ("vector_quantize_pytorch", ("VectorQuantize", "ResidualVQ")),
                        ("constriction", ())):
        if importlib.util.find_spec(name) is None:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, _unavailable(f"{name}.{a}"))
            sys.modules[name] = m
    sys.path.insert(0, REFERENCE)


def target_image(H, W):
    from gaussianimage_amd._testing import synth
    return synth.target_image(H, W, seed=7)


def run_case(kind, H, W, N, seed):
    if kind == "cholesky":
        from gaussianimage_cholesky import GaussianImage_Cholesky as Model
        names = ["_xyz", "_cholesky", "_features_dc"]
    else:
        from gaussianimage_rs import GaussianImage_RS as Model
        names = ["_xyz", "_scaling", "_rotation", "_features_dc"]
    torch.manual_seed(seed)
    model = Model(loss_type="L2", opt_type="adan", num_points=N, H=H, W=W, BLOCK_H=16,
                  BLOCK_W=16, device="cpu", lr=1e-3, quantize=False)
    gt = target_image(H, W)
    d = {"H": H, "W": W, "N": N, "seed": seed, "iters": ITERS, "gt": gt.numpy(),
         "names": np.array(names)}
    for k in names:
        d["init" + k] = getattr(model, k).detach().numpy().copy()
    model.eval()
    with torch.no_grad():
        pkg = model()
    for k in ("render", "gauss_render", "alpha_map"):
        if k in pkg:
            d["forward_" + k] = pkg[k].numpy().copy()
    if kind == "cholesky":
        d["random_colors"] = model.random_colors.numpy().copy()
    d["xys"] = model.xys.numpy().copy()
    d["radii"] = model.radii.numpy().copy()
    # the restatement in float64 must place every radius where the float32 C restatement does
    from oracle import cport
    with torch.no_grad():
        if kind == "cholesky":
            ref = cport.project_cholesky_fwd(model.get_xyz.numpy(),
                                             model.get_cholesky_elements.numpy(), H, W)
        else:
            ref = cport.project_rs_fwd(model.get_xyz.numpy(), model.get_scaling.numpy(),
                                       model.get_rotation.numpy(), H, W)
    assert np.array_equal(ref[2], d["radii"]), "radius on a rounding boundary: pick another seed"

    grads = {}

    def keep_grads(optimizer, args, kwargs):
        for k in names:
            grads[k] = getattr(model, k).grad.detach().numpy().copy()

    model.optimizer.register_step_pre_hook(keep_grads)
    model.train()
    for it in range(1, ITERS + 1):
        if kind == "cholesky":
            loss, psnr = model.train_iter(gt, it)
        else:
            loss, psnr = model.train_iter(gt)
        d[f"loss_{it}"] = np.float64(loss.item())
        d[f"psnr_{it}"] = np.float64(psnr)
        for k in names:
            d[f"grad_{it}{k}"] = grads[k]
            d[f"param_{it}{k}"] = getattr(model, k).detach().numpy().copy()
    assert math.isfinite(d[f"loss_{ITERS}"]) and d[f"loss_{ITERS}"] < d["loss_1"]
    d["lr_after"] = np.float64(model.optimizer.param_groups[0]["lr"])
    return d


if __name__ == "__main__":
    install_modules()
    for kind, H, W, N, seed in CASES:
        name = f"caller_{kind}_{W}x{H}_n{N}_s{seed}.npz"
        np.savez_compressed(os.path.join(HERE, name), **run_case(kind, H, W, N, seed))
        print("wrote", name)
