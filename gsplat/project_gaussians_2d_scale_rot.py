"""``from gsplat.project_gaussians_2d_scale_rot import project_gaussians_2d_scale_rot``
(/root/reference/gaussianimage_rs.py:1)."""
from gaussianimage_amd.ops import project_gaussians_2d_scale_rot

__all__ = ["project_gaussians_2d_scale_rot"]
