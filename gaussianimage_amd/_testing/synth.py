"""Synthetic parameter sets with the reference models' own initial distributions.

Cholesky model: /root/reference/gaussianimage_cholesky.py:30-34 (``_xyz`` =
atanh(2(U-0.5)), ``_cholesky`` = U[0,1) + [0.5,0,0.5], ``_features_dc`` = U[0,1),
opacity = 1).  RS model: /root/reference/gaussianimage_rs.py:26-30,55-60.
``footprint`` > 1 scales the covariance factors to imitate fitted (larger)
Gaussians, whose tile lists are longer than at initialisation.
"""
import math

import torch


def _gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return g


def cholesky_params(num_points, seed=1, footprint=1.0):
    """Raw parameters (``_xyz``, ``_cholesky``, ``_features_dc``) on CPU."""
    g = _gen(seed)
    xyz = torch.atanh(2 * (torch.rand(num_points, 2, generator=g) - 0.5))
    chol = torch.rand(num_points, 3, generator=g)
    feat = torch.rand(num_points, 3, generator=g)
    if footprint != 1.0:
        chol = (chol + torch.tensor([0.5, 0.0, 0.5])) * footprint - torch.tensor([0.5, 0.0, 0.5])
    return xyz, chol, feat


def cholesky_inputs(num_points, seed=1, footprint=1.0):
    """Operator inputs (means2d, L_elements, colors, opacity) as the model feeds them."""
    xyz, chol, feat = cholesky_params(num_points, seed, footprint)
    means = torch.tanh(xyz)
    L = chol + torch.tensor([0.5, 0.0, 0.5])
    return means, L, feat, torch.ones(num_points, 1)


def rs_params(num_points, seed=1, footprint=1.0):
    g = _gen(seed)
    xyz = torch.atanh(2 * (torch.rand(num_points, 2, generator=g) - 0.5))
    scaling = torch.rand(num_points, 2, generator=g)
    rotation = torch.rand(num_points, 1, generator=g)
    feat = torch.rand(num_points, 3, generator=g)
    if footprint != 1.0:
        scaling = (scaling + 0.5) * footprint - 0.5
    return xyz, scaling, rotation, feat


def rs_inputs(num_points, seed=1, footprint=1.0):
    xyz, scaling, rotation, feat = rs_params(num_points, seed, footprint)
    return (torch.tanh(xyz), torch.abs(scaling + 0.5), torch.sigmoid(rotation) * 2 * math.pi,
            feat, torch.ones(num_points, 1))


def target_image(H, W, seed=7):
    """Smooth synthetic [1,3,H,W] target in [0,1] (low-pass noise plus ramps)."""
    g = _gen(seed)
    low = torch.rand(1, 3, max(H // 32, 2), max(W // 32, 2), generator=g)
    img = torch.nn.functional.interpolate(low, size=(H, W), mode="bicubic", align_corners=False)
    yy = torch.linspace(0, 1, H).view(1, 1, H, 1)
    xx = torch.linspace(0, 1, W).view(1, 1, 1, W)
    img = 0.7 * img + 0.15 * yy + 0.15 * xx
    return img.clamp(0, 1).contiguous()
