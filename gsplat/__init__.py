"""Drop-in for the module paths the reference imports from its ``gsplat`` submodule
(/root/reference/gaussianimage_cholesky.py:1-2, /root/reference/gaussianimage_rs.py:1-2);
the operators run on the HIP kernels of ``gaussianimage_amd``."""
from gaussianimage_amd.ops import (project_gaussians_2d, project_gaussians_2d_scale_rot,
                                   rasterize_gaussians_sum)

__all__ = ["project_gaussians_2d", "project_gaussians_2d_scale_rot", "rasterize_gaussians_sum"]
