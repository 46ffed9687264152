"""Host-side mirror of the `diff_gaussian_rasterization` Python API (ashawkey fork) on top of
libhgs_rast.so.  Same names, argument meaning, return tuple and error behaviour as the
module the reference imports at
  /root/reference/gaussiansplatting/gaussian_renderer/__init__.py:14  (call :36-51, :86-94)
  /root/reference/gs_renderer.py:10-13                               (call :951-966, :1006-1015)
so those files run unchanged when `diff_gaussian_rasterization` resolves to this package
(the top-level `diff_gaussian_rasterization/` shim re-exports it).

PyTorch is plumbing only: it owns the tensors, the current HIP stream and autograd; all
arithmetic happens in libhgs_rast.so (plain pointers and sizes, include/hgs_rast.h), reached
through the small C++ torch binding `_hgs_torch.so` that plays the role of upstream's `_C`.
There is no CPU path: non-HIP tensors raise.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# ---------------------------------------------------------------------------- plumbing
# The autograd node, the per-device capacity estimates and the one host wait per forward live in
# the C++ torch binding (csrc/torch_binding.cpp -> _hgs_torch.so, on top of the C ABI): at
# 0.3 ms per step a Python autograd.Function cost as much host time as the GPU needs.

import types


def set_stage_events(fwd=None, bwd=None):
    """Measurement hook (bench.py only): sequences of raw hipEvent_t handles (ints) recorded
    after each stage of the following forward / backward calls (HGS_FWD_STAGES /
    HGS_BWD_STAGES in hgs_rast.h), or None to disable."""
    _lib.load_binding().set_stage_events(None if fwd is None else [int(h) for h in fwd],
                                         None if bwd is None else [int(h) for h in bwd])


def _dev_index(device: torch.device) -> int:
    return device.index if device.index is not None else torch.cuda.current_device()


def _state(device: torch.device):
    """Snapshot of the per-device bookkeeping of the binding: capacity / tile_hint used by the
    last call, max_R / max_tile it reported, calls, retries (forwards re-run after a device-side
    overflow), wait_ns, and the decaying per-(views, H, W) estimates."""
    idx = _dev_index(device)
    ns = types.SimpleNamespace(**_lib.load_binding().device_state(idx))
    ns.index = idx
    return ns


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                        cov3Ds_precomp, raster_settings, zero_means2D: bool = False):
    """Replaces upstream's `_RasterizeGaussians.apply` (forward -> `_C.rasterize_gaussians`,
    backward -> `_C.rasterize_gaussians_backward`).
    zero_means2D: `means2D` is UNINITIALISED storage (`torch.empty`); the forward's per-Gaussian kernel writes the zeros
    upstream's `torch.zeros_like(xyz) + 0` holds (ABI v16, `hgs_forward_batch_act_leaf`) - no fill launch."""
    want_grad = torch.is_grad_enabled() and (
        means3D.requires_grad or means2D.requires_grad or opacities.requires_grad
        or (sh is not None and sh.requires_grad)
        or (colors_precomp is not None and colors_precomp.requires_grad)
        or (scales is not None and scales.requires_grad)
        or (rotations is not None and rotations.requires_grad)
        or (cov3Ds_precomp is not None and cov3Ds_precomp.requires_grad))
    rs = raster_settings
    color, radii, depth, alpha = _lib.load_binding().rasterize(
        means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
        rs.bg, rs.viewmatrix, rs.projmatrix, rs.campos, int(rs.image_height), int(rs.image_width),
        float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier), int(rs.sh_degree),
        bool(rs.prefiltered), bool(rs.debug), want_grad, bool(zero_means2D))
    return color, radii, depth, alpha


ACT_OPACITY_SIGMOID, ACT_SCALE_EXP, ACT_ROTATION_NORMALIZE = 1, 2, 4      # HGS_ACT_* of include/hgs_rast.h
# the torch binding's own bit (stripped before the library sees the flags): `means2D` is uninitialised storage, the
# forward zero-fills it inside its per-Gaussian kernel (see rasterize_gaussians)
ZERO_MEANS2D = 1 << 16
# backward only: dL/dscales as the TRUE derivative at scale_modifier != 1 (the default follows the fork, whose backward
# drops the modifier's factor; identical at 1.0, the only value the reference passes) - HGS_GRAD_SCALE_TRUE_DERIVATIVE
GRAD_SCALE_TRUE_DERIVATIVE = 8


def rasterize_gaussians_batch(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                              cov3Ds_precomp, raster_settings_list, activation_flags: int = 0):
    """B views of the same Gaussians in ONE launch set (include/hgs_rast.h: hgs_forward_batch /
    hgs_backward_batch) - what the reference does with a Python loop over
    `render()` at threestudio/systems/GaussianDreamer.py:244-266.

    raster_settings_list  sequence of GaussianRasterizationSettings that agree in image size,
                          sh_degree and scale_modifier (cameras, fov and bg may differ)
    means2D               (B, P, 3) zeros whose .grad receives the per-view screen-space gradient
                          (may be None under no_grad)
    activation_flags      GRAD_SCALE_TRUE_DERIVATIVE and / or ACT_* bits: `opacities` / `scales` / `rotations` are the model's RAW parameters
                          (`_opacity` logits, `_scaling` log-scales, un-normalised `_rotation`) and
                          sigmoid / exp / normalize (scene/gaussian_model.py:95-115) run inside the
                          per-Gaussian kernels, forward and backward; gradients are w.r.t. the raw tensors
    Returns color (B,3,H,W), radii (B,P) int32, depth (B,1,H,W), alpha (B,1,H,W); every view is
    bit-identical to a separate single-view call, parameter gradients are the sum over the views."""
    rsl = list(raster_settings_list)
    if not rsl:
        raise ValueError("rasterize_gaussians_batch needs at least one view")
    r0 = rsl[0]
    for rs in rsl[1:]:
        if (int(rs.image_height), int(rs.image_width), int(rs.sh_degree), float(rs.scale_modifier)) != \
                (int(r0.image_height), int(r0.image_width), int(r0.sh_degree), float(r0.scale_modifier)):
            raise ValueError("all views of a batch must share image size, sh_degree and scale_modifier")
    B = len(rsl)
    dev = means3D.device
    if B == 1:          # (views of the caller's tensors: no stack kernels in front of a single view)
        vm, pm = r0.viewmatrix.to(dev, torch.float32).reshape(1, 4, 4), r0.projmatrix.to(dev, torch.float32).reshape(1, 4, 4)
        cp, bg = r0.campos.to(dev, torch.float32).reshape(1, 3), r0.bg.to(dev, torch.float32).reshape(1, 3)
    else:
        vm = torch.stack([rs.viewmatrix.to(dev, torch.float32).reshape(4, 4) for rs in rsl])
        pm = torch.stack([rs.projmatrix.to(dev, torch.float32).reshape(4, 4) for rs in rsl])
        cp = torch.stack([rs.campos.to(dev, torch.float32).reshape(3) for rs in rsl])
        bg = torch.stack([rs.bg.to(dev, torch.float32).reshape(3) for rs in rsl])
    if means2D is None:
        means2D = means3D.new_zeros((0,))
    want_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad
        for t in (means3D, means2D, opacities, sh, colors_precomp, scales, rotations, cov3Ds_precomp))
    return _lib.load_binding().rasterize_batch(
        means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, bg, vm, pm, cp,
        int(r0.image_height), int(r0.image_width), [float(rs.tanfovx) for rs in rsl],
        [float(rs.tanfovy) for rs in rsl], float(r0.scale_modifier), int(r0.sh_degree),
        bool(r0.prefiltered), any(bool(rs.debug) for rs in rsl), want_grad, int(activation_flags))


class packed_gradients:
    """Context manager of the view-parallel step (view_parallel.py): a rasterizer backward that runs inside it writes the
    per-Gaussian gradients as ONE (P, 15 + 3M) pack - [means3D 3 | means2D 3, summed over the call's views | sh 3M |
    opacity 1 | scales 3 | rotations 4 | radii 1, max over the views] - straight from its last kernel
    (include/hgs_rast.h: hgs_backward_batch_packed; the tensor the rank all-gathers) and hands autograd strided VIEWS of
    it; `.take()` returns the pack (None if the backward was not eligible - colours / covariances precomputed, or no
    backward ran: the caller then packs the six tensors itself)."""

    def __enter__(self):
        _lib.load_binding().set_packed_backward(True)
        return self

    def __exit__(self, *exc):
        _lib.load_binding().set_packed_backward(False)
        return False

    @staticmethod
    def take():
        return _lib.load_binding().take_packed()


class _RasterizeGaussians:
    """Name kept for callers that reach for upstream's autograd.Function directly:
    `_RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales,
    rotations, cov3Ds_precomp, raster_settings)`.  The node itself is the C++
    `torch::autograd::Function` of csrc/torch_binding.cpp."""

    @staticmethod
    def apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
              raster_settings):
        return rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales,
                                   rotations, cov3Ds_precomp, raster_settings)


# --------------------------------------------------------------------------- the module

class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Frustum test (replaces `_C.mark_visible`)."""
        rs = self.raster_settings
        return _lib.load_binding().mark_visible(
            positions, rs.bg, rs.viewmatrix, rs.projmatrix, rs.campos, int(rs.image_height),
            int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier),
            int(rs.sh_degree), bool(rs.prefiltered), bool(rs.debug))

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None, zero_means2D=False):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or '
                            'precomputed 3D covariance!')
        # upstream turns missing optionals into empty tensors for its `_C`; the binding takes None
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales,
                                   rotations, cov3D_precomp, raster_settings, zero_means2D=zero_means2D)
