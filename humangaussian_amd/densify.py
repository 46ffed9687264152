"""Densification bookkeeping of the training step either side of the rasterize path
(SURVEY.md 8(f)-3), each as ONE HIP pass (csrc/bookkeeping.hip) instead of a dozen torch
elementwise kernels launched from Python:

* `add_densification_stats`  <- /root/reference/threestudio/systems/GaussianDreamer.py:253-256 (radii max over the
                                views), :289 (visibility), :385-391 (sum of the per-view `viewspace_points.grad`,
                                `max_radii2D` update) + gaussiansplatting/scene/gaussian_model.py:434-438
* `densify_masks`            <- gaussian_model.py:359-438: the clone / split / prune selections of
                                `densify_and_clone`, `densify_and_split`, `densify_and_prune`, `prune_only`
* `prune_rows`               <- gaussian_model.py:283-337: boolean-mask indexing of every parameter tensor and of
                                both Adam moments (`_prune_optimizer`), one index computation for all tensors

The sampling of new Gaussians (`torch.normal`, gaussian_model.py:371-377) and the optimizer object
surgery stay with the caller: they are not on the path."""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _lib


def add_densification_stats(viewspace_grads: torch.Tensor, radii: torch.Tensor, xyz_gradient_accum: torch.Tensor,
                            denom: torch.Tensor, max_radii2D: torch.Tensor, keep: Optional[torch.Tensor] = None):
    """viewspace_grads (B,P,3) [or (P,3)], radii (B,P) [or (P,)] int32 of the step's views;
    updates xyz_gradient_accum (P,1), denom (P,1), max_radii2D (P,) IN PLACE and returns
    (radii_max (P,) int32, visibility_filter (P,) bool).  `keep` (P,) bool restricts the update (the
    reference excludes near-hand points: GaussianDreamer.py:292-297)."""
    return tuple(_lib.load_binding().densify_stats(viewspace_grads, radii, keep, xyz_gradient_accum, denom, max_radii2D))


def densify_masks(xyz_gradient_accum, denom, scaling, opacity, max_radii2D, grad_threshold: float,
                  percent_dense: float, extent: float, min_opacity: float, max_screen_size: Optional[float] = None,
                  size_thresh: Optional[float] = None, raw_params: bool = False):
    """Returns (clone_mask, split_mask, prune_mask) bool (P,) and counts int32 (3,) on the device.
    `scaling` / `opacity` are `get_scaling` / `get_opacity`, or the raw `_scaling` / `_opacity`
    parameters with raw_params=True (exp / sigmoid are then applied inside the kernel)."""
    return tuple(_lib.load_binding().densify_masks(
        xyz_gradient_accum, denom, scaling, bool(raw_params), opacity, bool(raw_params), max_radii2D,
        float(grad_threshold), float(percent_dense), float(extent), float(min_opacity),
        float(max_screen_size or 0.0), float(size_thresh or 0.0)))


def prune_rows(keep: torch.Tensor, tensors: Sequence[torch.Tensor]):
    """[t[keep] for t in tensors] for (P, ...) fp32 tensors (parameters, exp_avg, exp_avg_sq,
    xyz_gradient_accum, denom, max_radii2D), order preserved."""
    return _lib.load_binding().compact_rows(keep, list(tensors))


def append_rows(tensors: Sequence[torch.Tensor], new_rows: Sequence[Optional[torch.Tensor]]):
    """The clone / split APPEND half of the optimizer surgery (gaussian_model.py:339-357, `cat_tensors_to_optimizer`):
    [cat(t, n)] per tensor; `None` in `new_rows` appends zeros of the other tensors' row count - what the reference
    does for the Adam moments (`exp_avg`, `exp_avg_sq`) and for `xyz_gradient_accum` / `denom` / `max_radii2D`
    (gaussian_model.py:352-357).  Plain `torch.cat` (a copy is the whole operation: nothing to fuse)."""
    n_new = next((n.shape[0] for n in new_rows if n is not None), 0)
    out = []
    for t, n in zip(tensors, new_rows):
        if n is None:
            n = torch.zeros((n_new,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        if n.shape[1:] != t.shape[1:]:
            raise ValueError("append_rows: new rows must have the tensor's trailing dimensions")
        out.append(torch.cat((t, n.to(dtype=t.dtype, device=t.device)), dim=0))
    return out
