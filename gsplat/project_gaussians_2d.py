"""``from gsplat.project_gaussians_2d import project_gaussians_2d``
(/root/reference/gaussianimage_cholesky.py:1)."""
from gaussianimage_amd.ops import project_gaussians_2d

__all__ = ["project_gaussians_2d"]
