"""Drop-in module name for the reference: `from diff_gaussian_rasterization import
GaussianRasterizationSettings, GaussianRasterizer`
(/root/reference/gaussiansplatting/gaussian_renderer/__init__.py:14, gs_renderer.py:10-13)
resolves here when this repository is on sys.path; everything is implemented in
humangaussian_amd (HIP, gfx950)."""
from humangaussian_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
    _RasterizeGaussians,
)
