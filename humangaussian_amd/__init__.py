"""humangaussian_amd - MI355X-native differentiable 3DGS rasterizer for HumanGaussian.

Only what the rasterize hot path needs:
  csrc/            hand-written HIP kernels (gfx950) + the C ABI of include/hgs_rast.h
  _lib.py          build + ctypes binding of libhgs_rast.so
  rasterizer.py    GaussianRasterizationSettings / GaussianRasterizer (reference API mirror)
  renderer.py      render() / Renderer.render() mirrors (the reference's two call sites)
  view_parallel.py view-parallel multi-GPU rendering (one RCCL all-gather per step)
  synth.py         synthetic SMPL-X-like clouds + threestudio-style cameras (bench/tests)
"""
from .rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
    rasterize_gaussians_batch,
)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians",
           "rasterize_gaussians_batch"]
