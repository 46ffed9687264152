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

* `densify_and_prune`        <- gaussian_model.py:410-423 whole: the above + the sampling of the split children
                                (:369-380) + the surgery on the Adam optimizer object (:283-357), on any object shaped like
                                the reference GaussianModel; pinned to the reference's own method by
                                tests/golden/reference_densify.npz (tests/golden/make_densify_fixture.py)."""
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


# ------------------------------------------------------------------------------ GaussianModel.densify_and_prune, whole

_GROUPS = (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
           ("scaling", "_scaling"), ("rotation", "_rotation"))


def build_rotation(q: torch.Tensor) -> torch.Tensor:
    """(n,4) raw quaternions (w,x,y,z) -> (n,3,3), normalised inside like utils/general_utils.py:78-99."""
    q = q / torch.sqrt((q * q).sum(dim=1, keepdim=True))
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)


def densify_and_prune(pc, max_grad: float, min_opacity: float, extent: float, max_screen_size: Optional[float],
                      N: int = 2, split_samples: Optional[torch.Tensor] = None):
    """`GaussianModel.densify_and_prune(max_grad, min_opacity, extent, max_screen_size)` of
    /root/reference/gaussiansplatting/scene/gaussian_model.py:410-423 - clone (:393-408), split incl. the sampling of the
    children (:359-391), final prune (:414-421) and the surgery on the Adam optimizer object (`cat_tensors_to_optimizer`
    :339-357, `_prune_optimizer` :283-301) - on any object with the reference GaussianModel's attributes (`_xyz`,
    `_features_dc`, `_features_rest`, `_opacity`, `_scaling`, `_rotation` as the single parameters of optimizer groups named
    xyz / f_dc / f_rest / opacity / scaling / rotation; `xyz_gradient_accum`, `denom`, `max_radii2D`, `percent_dense`,
    `optimizer`).  Same result as the reference's sequence (same row order: survivors, clones, the N children of every
    split parent in `repeat(N, 1)` order; masks exact; the children drawn by the same `torch.normal(mean=0, std=scale)`
    call, so the same generator state gives the same samples), in two mask passes (`hgs_densify_masks`), the appends and ONE
    compaction of every parameter, both Adam moments and the three statistics (`hgs_compact_index` + `hgs_gather_rows`)
    instead of the reference's three optimizer rebuilds and ~60 elementwise / indexing kernels.
    `split_samples` ((N * n_split, 3), optional): the children's offsets in their parents' frames instead of a fresh
    draw (tests replay the reference's).  Returns a dict of counts."""
    P = pc._xyz.shape[0]
    dev = pc._xyz.device
    # ---- selections on the current points (gaussian_model.py:393-397, 359-367): one pass on the raw parameters
    clone, split, _, _ = densify_masks(pc.xyz_gradient_accum, pc.denom, pc._scaling, pc._opacity, pc.max_radii2D, max_grad,
                                       pc.percent_dense, extent, min_opacity, max_screen_size=None, raw_params=True)
    ci, si = torch.nonzero(clone).reshape(-1), torch.nonzero(split).reshape(-1)
    nc, ns = int(ci.numel()), int(si.numel())
    # ---- the new rows: clones are copies; children are sampled in the parent's frame (:369-380)
    scal = torch.exp(pc._scaling.detach()[si])                                  # get_scaling of the split parents
    stds = scal.repeat(N, 1)
    if split_samples is None:
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=dev), std=stds)
    else:
        samples = split_samples.to(dev, torch.float32)
        if samples.shape != stds.shape:
            raise ValueError(f"split_samples must be {tuple(stds.shape)}")
    rots = build_rotation(pc._rotation.detach()[si]).repeat(N, 1, 1)
    child = {"_xyz": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + pc._xyz.detach()[si].repeat(N, 1),
             "_scaling": torch.log(scal.repeat(N, 1) / (0.8 * N)),
             "_rotation": pc._rotation.detach()[si].repeat(N, 1),
             "_features_dc": pc._features_dc.detach()[si].repeat(N, 1, 1),
             "_features_rest": pc._features_rest.detach()[si].repeat(N, 1, 1),
             "_opacity": pc._opacity.detach()[si].repeat(N, 1)}
    grown = {attr: torch.cat((getattr(pc, attr).detach(), getattr(pc, attr).detach()[ci], child[attr]), dim=0)
             for _, attr in _GROUPS}
    n_all = P + nc + N * ns
    # ---- the final prune (:414-421) is evaluated on the grown set; `densification_postfix` has zeroed max_radii2D by then
    zeros1 = torch.zeros((n_all, 1), device=dev)
    _, _, prune, _ = densify_masks(zeros1, zeros1, grown["_scaling"], grown["_opacity"], zeros1.reshape(-1), 0.0,
                                   pc.percent_dense, extent, min_opacity, max_screen_size=max_screen_size, raw_params=True)
    keep = ~prune
    keep[:P] &= ~split                                                          # the split parents go (:390-391)
    # ---- one compaction for the parameters and both Adam moments (appended rows start with zero moments: :346-347)
    tensors, slots = [], []
    for name, attr in _GROUPS:
        group = next(g for g in pc.optimizer.param_groups if g["name"] == name)
        assert len(group["params"]) == 1
        st = pc.optimizer.state.get(group["params"][0], None)
        tensors.append(grown[attr])
        slots.append((group, attr, st))
        if st is not None:
            for m in ("exp_avg", "exp_avg_sq"):
                pad = torch.zeros((n_all - P,) + tuple(st[m].shape[1:]), dtype=st[m].dtype, device=dev)
                tensors.append(torch.cat((st[m], pad), dim=0))
    out = iter(prune_rows(keep, tensors))
    for group, attr, st in slots:
        new = torch.nn.Parameter(next(out).requires_grad_(True))
        if st is not None:
            st["exp_avg"], st["exp_avg_sq"] = next(out), next(out)
            del pc.optimizer.state[group["params"][0]]
            pc.optimizer.state[new] = st
        group["params"][0] = new
        setattr(pc, attr, new)
    n_final = pc._xyz.shape[0]
    pc.xyz_gradient_accum = torch.zeros((n_final, 1), device=dev)                # (:354-357 zero them; pruning zeros keeps zeros)
    pc.denom = torch.zeros((n_final, 1), device=dev)
    pc.max_radii2D = torch.zeros((n_final,), device=dev)
    return {"cloned": nc, "split": ns, "children": N * ns, "pruned": int(n_all - ns - n_final), "points": n_final}
