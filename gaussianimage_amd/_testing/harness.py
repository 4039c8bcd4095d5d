"""Stand-ins for the reference's callers, used by the benchmark and the end-to-end tests.

They reproduce the operator sequence of ``GaussianImage_Cholesky`` / ``GaussianImage_RS``
around the boundary -- parameters and their initial distributions
(/root/reference/gaussianimage_cholesky.py:30-42, gaussianimage_rs.py:26-35), activations
(:62,74 / :55-60), ``forward`` (:76-99 / :74-80: the Cholesky model rasterises three colour
sets per call) and ``train_iter`` (:101-124 / :82-95: L2 loss, backward, PSNR with its
``.item()``, Adan step, StepLR) -- through the same ``gsplat.*`` imports the reference uses.
Quantisation, codec and logging are out of scope.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from gsplat.project_gaussians_2d import project_gaussians_2d
from gsplat.project_gaussians_2d_scale_rot import project_gaussians_2d_scale_rot
from gsplat.rasterize_sum import rasterize_gaussians_sum

from ..ops import rasterize_gaussians_sum_multi

from ..adan import Adan


class _Base(nn.Module):
    def __init__(self, num_points, H, W, device, lr=1e-3, BLOCK_H=16, BLOCK_W=16):
        super().__init__()
        self.init_num_points, self.H, self.W = num_points, H, W
        self.BLOCK_H, self.BLOCK_W = BLOCK_H, BLOCK_W
        self.tile_bounds = ((W + BLOCK_W - 1) // BLOCK_W, (H + BLOCK_H - 1) // BLOCK_H, 1)
        self.device = device
        self._xyz = nn.Parameter(torch.atanh(2 * (torch.rand(num_points, 2) - 0.5)))
        self._features_dc = nn.Parameter(torch.rand(num_points, 3))
        self.register_buffer('_opacity', torch.ones((num_points, 1)))
        self.register_buffer('background', torch.ones(3))
        self.lr = lr

    def _finish(self):
        self.to(self.device)
        self.optimizer = Adan(self.parameters(), lr=self.lr)
        self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=20000,
                                                         gamma=0.5)

    def _image(self, out):
        out = torch.clamp(out, 0, 1)
        return out.view(-1, self.H, self.W, 3).permute(0, 3, 1, 2).contiguous()

    def train_iter(self, gt_image):
        image = self.forward()["render"]
        loss = F.mse_loss(image, gt_image)
        loss.backward()
        with torch.no_grad():
            mse_loss = F.mse_loss(image, gt_image)
            psnr = 10 * math.log10(1.0 / mse_loss.item())
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        self.scheduler.step()
        return loss, psnr


class CholeskyImageModel(_Base):
    def __init__(self, num_points, H, W, device, lr=1e-3, passes=3):
        super().__init__(num_points, H, W, device, lr)
        self._cholesky = nn.Parameter(torch.rand(num_points, 3))
        self.register_buffer('cholesky_bound', torch.tensor([0.5, 0, 0.5]).view(1, 3))
        self.random_colors = torch.rand(num_points, 3)
        self.passes = passes  # the reference's forward() makes 3; 1 = the render alone;
        # "fused" = the three images from one multi-set launch
        self._finish()

    def forward(self):
        self.xys, depths, self.radii, conics, num_tiles_hit = project_gaussians_2d(
            torch.tanh(self._xyz), self._cholesky + self.cholesky_bound, self.H, self.W,
            self.tile_bounds)
        colors, opacities = self._features_dc, self._opacity.clone()
        args = (self.xys, depths, self.radii, conics, num_tiles_hit)
        kw = dict(background=self.background, return_alpha=False)
        if self.passes == "fused":
            # the same three images from one launch
            geom = self.random_colors.to(self.xys.device) * 0.5
            img, gauss, alpha = rasterize_gaussians_sum_multi(
                *args, [colors, geom, torch.ones_like(colors)], opacities, self.H, self.W,
                self.BLOCK_H, self.BLOCK_W, background=self.background)
            alpha = alpha.mean(dim=-1, keepdim=True)
            return {"render": self._image(img), "gauss_render": self._image(gauss),
                    "alpha_map": alpha.view(-1, self.H, self.W, 1).permute(0, 3, 1, 2).contiguous(),
                    "final_opacities": opacities}
        out = {"render": self._image(rasterize_gaussians_sum(
            *args, colors, opacities, self.H, self.W, self.BLOCK_H, self.BLOCK_W, **kw))}
        if self.passes >= 3:
            geom = self.random_colors.to(self.xys.device) * 0.5
            out["gauss_render"] = self._image(rasterize_gaussians_sum(
                *args, geom, opacities, self.H, self.W, self.BLOCK_H, self.BLOCK_W, **kw))
            alpha = rasterize_gaussians_sum(*args, torch.ones_like(colors), opacities, self.H,
                                            self.W, self.BLOCK_H, self.BLOCK_W, **kw)
            alpha = alpha.mean(dim=-1, keepdim=True)
            out["alpha_map"] = alpha.view(-1, self.H, self.W, 1).permute(0, 3, 1, 2).contiguous()
            out["final_opacities"] = opacities
        return out


class RSImageModel(_Base):
    def __init__(self, num_points, H, W, device, lr=1e-3):
        super().__init__(num_points, H, W, device, lr)
        self._scaling = nn.Parameter(torch.rand(num_points, 2))
        self._rotation = nn.Parameter(torch.rand(num_points, 1))
        self.register_buffer('bound', torch.tensor([0.5, 0.5]).view(1, 2))
        self._finish()

    def forward(self):
        self.xys, depths, self.radii, conics, num_tiles_hit = project_gaussians_2d_scale_rot(
            torch.tanh(self._xyz), torch.abs(self._scaling + self.bound),
            torch.sigmoid(self._rotation) * 2 * math.pi, self.H, self.W, self.tile_bounds)
        out = rasterize_gaussians_sum(self.xys, depths, self.radii, conics, num_tiles_hit,
                                      self._features_dc, self._opacity, self.H, self.W,
                                      self.BLOCK_H, self.BLOCK_W, background=self.background,
                                      return_alpha=False)
        return {"render": self._image(out)}
