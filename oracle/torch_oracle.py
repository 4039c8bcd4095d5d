"""Differentiable pure-PyTorch restatement of the accumulated-summation path.

TEST INFRASTRUCTURE ONLY -- imported by tests/, never by the product package.

PARITY UNPINNED: the reference's operators live in an un-vendored submodule
(/root/reference/.gitmodules:1-3) and the reference holds no fixtures for them.
This module follows the reference's call sites (gaussianimage_cholesky.py:77,
81-85; gaussianimage_rs.py:75-79) and in-repo conventions (utils.py:96-128) and
exists to check the analytic backward of ``gi2d_oracle.c`` against autograd,
in fp32 and fp64.  It evaluates every pixel against every Gaussian under a
tile-membership mask, so it is for small cases only.
"""
import math

import torch

BLOCK = 16

DEFAULT_CONV = dict(pixel_offset=0.0, center_bias=0.0, alpha_cap=1.0, alpha_min=1.0 / 255.0,
                    bg_weight=0.0)


def _tail(means, sxx, sxy, syy, H, W, tile_bounds, conv):
    det = sxx * syy - sxy * sxy
    ok = (det != 0) & torch.isfinite(det)
    safe = torch.where(ok, det, torch.ones_like(det))
    conics = torch.stack([syy / safe, -sxy / safe, sxx / safe], dim=-1)
    b = 0.5 * (sxx + syy)
    disc = torch.clamp(b * b - det, min=0.1)
    lam = torch.maximum(b + disc.sqrt(), b - disc.sqrt())
    radius = torch.ceil(3.0 * lam.sqrt()).detach()
    ok = ok & torch.isfinite(radius)
    cx = 0.5 * W * means[:, 0] + 0.5 * W + conv["center_bias"]
    cy = 0.5 * H * means[:, 1] + 0.5 * H + conv["center_bias"]
    xys = torch.stack([cx, cy], dim=-1)
    okf = ok.to(means.dtype)
    xys = xys * okf[:, None]
    conics = conics * okf[:, None]
    radii = (radius * okf).to(torch.int32)
    box = tile_boxes(xys.detach(), radii, tile_bounds)
    area = (box[:, 1] - box[:, 0]) * (box[:, 3] - box[:, 2])
    num_tiles_hit = torch.where(ok & (area > 0), area, torch.zeros_like(area)).to(torch.int32)
    depths = torch.zeros_like(cx)
    return xys, depths, radii, conics, num_tiles_hit


def tile_boxes(xys, radii, tile_bounds):
    tx, ty = tile_bounds[0], tile_bounds[1]
    r = radii.to(xys.dtype) / BLOCK
    c = xys / BLOCK
    x0 = torch.clamp((c[:, 0] - r).trunc().long(), 0, tx)
    x1 = torch.clamp((c[:, 0] + r + 1).trunc().long(), 0, tx)
    y0 = torch.clamp((c[:, 1] - r).trunc().long(), 0, ty)
    y1 = torch.clamp((c[:, 1] + r + 1).trunc().long(), 0, ty)
    return torch.stack([x0, x1, y0, y1], dim=-1)


def project_gaussians_2d(means, L, H, W, tile_bounds, conv=DEFAULT_CONV):
    l11, l21, l22 = L[:, 0], L[:, 1], L[:, 2]
    return _tail(means, l11 * l11, l11 * l21, l21 * l21 + l22 * l22, H, W, tile_bounds, conv)


def project_gaussians_2d_scale_rot(means, scales, rot, H, W, tile_bounds, conv=DEFAULT_CONV):
    c, s = torch.cos(rot[:, 0]), torch.sin(rot[:, 0])
    q1, q2 = scales[:, 0] ** 2, scales[:, 1] ** 2
    return _tail(means, c * c * q1 + s * s * q2, c * s * (q1 - q2), s * s * q1 + c * c * q2, H, W,
                 tile_bounds, conv)


def rasterize_gaussians_sum(xys, radii, conics, colors, opacity, H, W, background=None,
                            conv=DEFAULT_CONV):
    """[H,W,3] image; dense evaluation under the tile-membership mask."""
    dt, dev = xys.dtype, xys.device
    tiles_x, tiles_y = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    box = tile_boxes(xys.detach().to(torch.float32), radii, (tiles_x, tiles_y, 1))
    jj = torch.arange(W, device=dev)
    ii = torch.arange(H, device=dev)
    ptx = (jj // BLOCK)[None, :, None]
    pty = (ii // BLOCK)[:, None, None]
    member = ((ptx >= box[None, None, :, 0]) & (ptx < box[None, None, :, 1]) &
              (pty >= box[None, None, :, 2]) & (pty < box[None, None, :, 3]) &
              (radii > 0)[None, None, :])
    px = (jj.to(dt) + conv["pixel_offset"])[None, :, None]
    py = (ii.to(dt) + conv["pixel_offset"])[:, None, None]
    dx = xys[None, None, :, 0] - px
    dy = xys[None, None, :, 1] - py
    a, b, c = conics[:, 0], conics[:, 1], conics[:, 2]
    sigma = 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
    alpha = torch.clamp(opacity[:, 0] * torch.exp(-sigma), max=conv["alpha_cap"])
    on = member & (sigma >= 0) & (alpha >= conv["alpha_min"])
    w = torch.where(on, alpha, torch.zeros_like(alpha))
    out = torch.einsum("hwn,nc->hwc", w, colors)
    if background is not None and conv["bg_weight"] != 0.0:
        out = out + conv["bg_weight"] * background
    return out


def model_forward_cholesky(_xyz, _cholesky, _features_dc, H, W, conv=DEFAULT_CONV):
    """The caller's sequence around the boundary (gaussianimage_cholesky.py:62,74,76-85)."""
    tile_bounds = ((W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK, 1)
    bound = torch.tensor([0.5, 0.0, 0.5], dtype=_xyz.dtype)
    xys, _, radii, conics, _ = project_gaussians_2d(torch.tanh(_xyz), _cholesky + bound, H, W,
                                                    tile_bounds, conv)
    opac = torch.ones(_xyz.shape[0], 1, dtype=_xyz.dtype)
    img = rasterize_gaussians_sum(xys, radii, conics, _features_dc, opac, H, W, None, conv)
    img = torch.clamp(img, 0, 1)
    return img.view(-1, H, W, 3).permute(0, 3, 1, 2).contiguous()


def model_forward_rs(_xyz, _scaling, _rotation, _features_dc, H, W, conv=DEFAULT_CONV):
    """gaussianimage_rs.py:55-60,74-80."""
    tile_bounds = ((W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK, 1)
    scaling = torch.abs(_scaling + 0.5)
    rot = torch.sigmoid(_rotation) * 2 * math.pi
    xys, _, radii, conics, _ = project_gaussians_2d_scale_rot(torch.tanh(_xyz), scaling, rot, H, W,
                                                              tile_bounds, conv)
    opac = torch.ones(_xyz.shape[0], 1, dtype=_xyz.dtype)
    img = rasterize_gaussians_sum(xys, radii, conics, _features_dc, opac, H, W, None, conv)
    img = torch.clamp(img, 0, 1)
    return img.view(-1, H, W, 3).permute(0, 3, 1, 2).contiguous()
