"""`distCUDA2(points) -> (P,)`: mean squared distance of every point to its 3 nearest neighbours,
the `simple_knn._C.distCUDA2` of the reference's import path
(/root/reference/gaussiansplatting/scene/gaussian_model.py:20,134; gs_renderer.py:14,386-389).
HIP kernels: csrc/knn.hip through `hgs_knn_mean_dist2_grid` of the C ABI (uniform grid + ring search, near-linear;
`brute_force=True`: the exact O(P^2) kernel `hgs_knn_mean_dist2`, the same distances).  No CPU path."""
from __future__ import annotations

import torch

from . import _lib


def distCUDA2(points: torch.Tensor, brute_force: bool = False) -> torch.Tensor:
    return _lib.load_binding().knn_mean_dist2(points, brute_force)
