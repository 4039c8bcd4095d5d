"""``from gsplat.rasterize_sum import rasterize_gaussians_sum``
(/root/reference/gaussianimage_cholesky.py:2)."""
from gaussianimage_amd.ops import rasterize_gaussians_sum

__all__ = ["rasterize_gaussians_sum"]
