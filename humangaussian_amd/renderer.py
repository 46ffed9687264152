"""Mirrors of the reference's two render wrappers around the rasterizer, behaviour-identical
(same argument meaning, same returned dict keys) so callers switch by import only:

* `render()`           <- /root/reference/gaussiansplatting/gaussian_renderer/__init__.py:18-104
                          (called per view by threestudio/systems/GaussianDreamer.py:244-248)
* `Renderer.render()`  <- /root/reference/gs_renderer.py:923-1028 (animation.py:477-484)

`viewpoint_camera` / `pc` / `pipe` are duck-typed exactly like the reference objects
(`scene/cameras.py:17-67`, `scene/gaussian_model.py:95-118`, `arguments/__init__.py:63-68`).
"""
from __future__ import annotations

import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

_SH_C0 = 0.28209479177387814
_SH_C1 = 0.4886025119029199
_SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792,
          0.5462742152960396)
_SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
          -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


def _eval_sh_python(deg, sh, dirs):
    """Real SH basis up to degree 3 in plain torch, for the `convert_SHs_python` branch
    (what `gaussiansplatting/utils/sh_utils.py:57-112` computes there).  sh: (..., 3, K)
    coefficients, dirs: (..., 3) unit vectors -> (..., 3) colours before the +0.5 shift."""
    assert 0 <= deg <= 3 and sh.shape[-1] >= (deg + 1) ** 2
    res = _SH_C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        res = res - _SH_C1 * y * sh[..., 1] + _SH_C1 * z * sh[..., 2] - _SH_C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + _SH_C2[0] * xy * sh[..., 4] + _SH_C2[1] * yz * sh[..., 5]
                   + _SH_C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] + _SH_C2[3] * xz * sh[..., 7]
                   + _SH_C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                res = (res + _SH_C3[0] * y * (3 * xx - yy) * sh[..., 9] + _SH_C3[1] * xy * z * sh[..., 10]
                       + _SH_C3[2] * y * (4 * zz - xx - yy) * sh[..., 11]
                       + _SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                       + _SH_C3[4] * x * (4 * zz - xx - yy) * sh[..., 13]
                       + _SH_C3[5] * z * (xx - yy) * sh[..., 14] + _SH_C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return res


def _tan_half(fov):
    # the reference calls math.tan on a 0-dim device tensor (D2H sync per view,
    # gaussian_renderer/__init__.py:33-34); accept both floats and tensors
    return math.tan(float(fov) * 0.5)


# The reference creates `screenspace_points = torch.zeros_like(xyz, requires_grad=True) + 0` per view
# (gaussian_renderer/__init__.py:26): a fill AND an add kernel in front of every forward.  The rasterizer only routes
# dL/dmeans2D into the tensor's `.grad` (GaussianDreamer.py:385-387), but the reference's own 3DGS trainer reads the VALUES
# too (`gaussiansplatting/train.py:113` -> `gaussian_model.py:436`), so a drop-in hands out ZEROS like the reference - but
# without either launch: the leaf comes from `torch.empty` and the forward's per-Gaussian kernel writes its zeros (ABI v16,
# `_screenspace_points` below; before v16: one torch fill kernel, 2.6 us of a 158 us step with its kernel boundary).
# ZERO_SCREENSPACE_POINTS = False hands out an uninitialised leaf (for loops that never read the values).  bench.py's step
# builds its leaf the same way as render() and says so in its workload string.
ZERO_SCREENSPACE_POINTS = True


def _screenspace_points(xyz, views=None):
    """-> (leaf, zero_in_kernel).  With ZERO_SCREENSPACE_POINTS the zeros are written by the forward's own per-Gaussian
    kernel (ABI v16: `rasterize_gaussians(..., zero_means2D=True)` / the ZERO_MEANS2D flag of the batched call) into
    storage that comes from `torch.empty`: the values behind the call are the reference's, the fill launch is gone.
    (A leaf the kernel cannot write - not fp32 on the device - is filled by torch.zeros as before.)"""
    shape = tuple(xyz.shape) if views is None else (int(views),) + tuple(xyz.shape)
    in_kernel = ZERO_SCREENSPACE_POINTS and xyz.dtype == torch.float32 and xyz.is_cuda
    make = torch.zeros if (ZERO_SCREENSPACE_POINTS and not in_kernel) else torch.empty
    return make(shape, dtype=xyz.dtype, device=xyz.device).requires_grad_(True), in_kernel


# FUSE_ACTIVATIONS = True: `render()` hands the model's RAW parameters (`pc._opacity`, `pc._scaling`, `pc._rotation`) to the
# rasterizer, which applies sigmoid / exp / normalize (scene/gaussian_model.py:95-115) inside its per-Gaussian kernels,
# forward and backward: three elementwise launches and three autograd nodes fewer per view in each direction.  Off by
# default: the un-fused path evaluates `pc.get_*` exactly like the reference (results agree to rounding - expf vs
# torch.exp - not bit for bit); a model whose activations are NOT the reference's must leave it off.
FUSE_ACTIVATIONS = False


def _render_fused(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, screenspace_points, settings,
                  zero_in_kernel=False):
    from .rasterizer import (ACT_OPACITY_SIGMOID, ACT_ROTATION_NORMALIZE, ACT_SCALE_EXP, ZERO_MEANS2D,
                             rasterize_gaussians_batch)
    f = lambda t: None if t is None else t.float()  # noqa: E731
    shs = None
    if override_color is None:
        # (`get_features` = cat(_features_dc, _features_rest), gaussian_model.py:108-111: with sh_degree 0 - HumanGaussian's
        # setting - the rest is empty and the cat a pure copy, forward and backward: the dc tensor goes in as it is)
        rest = getattr(pc, "_features_rest", None)
        shs = pc._features_dc if rest is not None and rest.shape[1] == 0 and hasattr(pc, "_features_dc") else pc.get_features
    image, radii, depth, alpha = rasterize_gaussians_batch(
        f(pc.get_xyz), f(screenspace_points).unsqueeze(0), f(shs), f(override_color), f(pc._opacity), f(pc._scaling),
        f(pc._rotation), None, [settings],
        activation_flags=ACT_OPACITY_SIGMOID | ACT_SCALE_EXP | ACT_ROTATION_NORMALIZE | (ZERO_MEANS2D if zero_in_kernel else 0))
    # (squeeze, not [0]: a select's backward zero-fills and copies a full-size gradient per output)
    return image.squeeze(0), radii.squeeze(0), depth.squeeze(0), alpha.squeeze(0)


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0,
           override_color=None, fuse_activations=None):
    """Render one view.  Returns the reference's dict: render, viewspace_points,
    visibility_filter, radii, depth_3dgs, alpha_3dgs.  `fuse_activations` (default: the module's FUSE_ACTIVATIONS): see there."""
    screenspace_points, zero_in_kernel = _screenspace_points(pc.get_xyz)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=_tan_half(viewpoint_camera.FoVx),
        tanfovy=_tan_half(viewpoint_camera.FoVy),
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=bool(getattr(pipe, "debug", False)),
    )
    fuse = FUSE_ACTIVATIONS if fuse_activations is None else bool(fuse_activations)
    if fuse and not getattr(pipe, "compute_cov3D_python", False) and not getattr(pipe, "convert_SHs_python", False) and \
            all(hasattr(pc, a) for a in ("_opacity", "_scaling", "_rotation")):
        rendered_image, radii, depth, alpha = _render_fused(viewpoint_camera, pc, pipe, bg_color, scaling_modifier,
                                                            override_color, screenspace_points, raster_settings, zero_in_kernel)
        return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
                "radii": radii, "depth_3dgs": depth, "alpha_3dgs": alpha}
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    means3D = pc.get_xyz
    opacity = pc.get_opacity
    scales = rotations = cov3D_precomp = None
    if getattr(pipe, "compute_cov3D_python", False):
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation

    shs = colors_precomp = None
    if override_color is None:
        if getattr(pipe, "convert_SHs_python", False):
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
            dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(_eval_sh_python(pc.active_sh_degree, shs_view, dir_pp) + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    f = lambda t: None if t is None else t.float()  # noqa: E731  (AMP: kernels are fp32)
    rendered_image, radii, depth, alpha = rasterizer(
        means3D=f(means3D), means2D=f(screenspace_points), shs=f(shs),
        colors_precomp=f(colors_precomp), opacities=f(opacity), scales=f(scales),
        rotations=f(rotations), cov3D_precomp=f(cov3D_precomp), zero_means2D=zero_in_kernel)

    return {"render": rendered_image,
            "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0,
            "radii": radii,
            "depth_3dgs": depth,
            "alpha_3dgs": alpha}


class Renderer:
    """`gs_renderer.Renderer.render` mirror: takes any object exposing the reference
    GaussianModel getters as `gaussians`."""

    def __init__(self, gaussians, white_background: bool = True, device="cuda"):
        self.gaussians = gaussians
        self.white_background = white_background
        self.bg_color = torch.tensor([1, 1, 1] if white_background else [0, 0, 0],
                                     dtype=torch.float32, device=device)

    def render(self, viewpoint_camera, scaling_modifier=1.0, bg_color=None, override_color=None,
               compute_cov3D_python=False, convert_SHs_python=False, fuse_activations=None):
        class _Pipe:
            pass
        pipe = _Pipe()
        pipe.compute_cov3D_python = compute_cov3D_python
        pipe.convert_SHs_python = convert_SHs_python
        pipe.debug = False
        out = render(viewpoint_camera, self.gaussians, pipe,
                     self.bg_color if bg_color is None else bg_color, scaling_modifier,
                     override_color, fuse_activations=fuse_activations)
        return {"image": out["render"].clamp(0, 1),     # gs_renderer.py:1017
                "depth": out["depth_3dgs"],
                "alpha": out["alpha_3dgs"],
                "viewspace_points": out["viewspace_points"],
                "visibility_filter": out["visibility_filter"],
                "radii": out["radii"]}


class HostCamera:
    """What `render()` / `render_views()` read from a camera, built WITHOUT device work."""

    def __init__(self, H, W, fovx, fovy, view, full, center):
        self.image_height, self.image_width = int(H), int(W)
        self.FoVx, self.FoVy = float(fovx), float(fovy)
        self.world_view_transform, self.full_proj_transform, self.camera_center = view, full, center


def cameras_from_c2w(c2w, fovy, image_height: int, image_width: int, device="cuda", znear: float = 0.01,
                     zfar: float = 100.0):
    """The B cameras of a training step from their camera-to-world matrices, computed on the HOST with
    ONE upload of B x 35 floats.  The reference builds every view's `Camera` on the device
    (/root/reference/gaussiansplatting/scene/cameras.py:22-53: two `torch.inverse()` launches, a dozen elementwise
    kernels and a device-to-host `tan` per view; utils/graphics_utils.py:73-99 for the projection); the arithmetic
    is restated here in float64 numpy (pinned to the reference class by tests/golden/reference_helpers.npz through
    synth.camera_from_c2w, which shares it).  c2w: (B,4,4) array-like, fovy: radians, scalar or (B,)."""
    import math
    import numpy as np
    c2w = np.asarray(c2w.detach().cpu() if isinstance(c2w, torch.Tensor) else c2w, dtype=np.float64).reshape(-1, 4, 4)
    B = c2w.shape[0]
    fov = np.broadcast_to(np.asarray(fovy.detach().cpu() if isinstance(fovy, torch.Tensor) else fovy, dtype=np.float64).reshape(-1), (B,))
    H, W = int(image_height), int(image_width)
    pack = np.zeros((B, 35), np.float32)
    fovs = []
    for b in range(B):
        fy = float(fov[b])
        focal = H / (2.0 * math.tan(fy / 2.0))
        fx = 2.0 * math.atan(W / (2.0 * focal))                          # cameras.py:22 (fov2focal / focal2fov)
        w2c = np.linalg.inv(c2w[b])
        w2c[1:3, :3] *= -1                                               # cameras.py:28-29
        w2c[:3, 3] *= -1
        V = w2c.T                                                        # cameras.py:50
        ty, tx = math.tan(fy / 2.0), math.tan(fx / 2.0)
        Pm = np.zeros((4, 4))
        Pm[0, 0], Pm[1, 1], Pm[3, 2] = 1.0 / tx, 1.0 / ty, 1.0           # graphics_utils.py:73-93
        Pm[2, 2], Pm[2, 3] = zfar / (zfar - znear), -(zfar * znear) / (zfar - znear)
        pack[b, :16] = V.reshape(-1)
        pack[b, 16:32] = (V @ Pm.T).reshape(-1)                          # cameras.py:52
        pack[b, 32:35] = np.linalg.inv(V)[3, :3]                         # cameras.py:53
        fovs.append((fx, fy))
    dev = torch.from_numpy(pack).to(device, non_blocking=True)
    return [HostCamera(H, W, fovs[b][0], fovs[b][1], dev[b, :16].view(4, 4), dev[b, 16:32].view(4, 4), dev[b, 32:35])
            for b in range(B)]


def render_views(viewpoint_cameras, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0,
                 override_color=None, fuse_activations=False):
    """All views of one training step in ONE rasterize call - the batched form of the loop
    `for id in range(B): render(Camera(...), gaussian, pipe, bg)` at
    /root/reference/threestudio/systems/GaussianDreamer.py:244-266.

    The model's activations (`get_opacity / get_scaling / get_rotation / get_features`,
    scene/gaussian_model.py:95-115) are evaluated once for the batch instead of once per view, the
    per-view launch sets collapse into one, and parameter gradients arrive summed over the views.
    Every view's image / depth / alpha / radii is bit-identical to `render()` of that view.
    fuse_activations=True additionally hands the RAW parameters (`pc._opacity`, `pc._scaling`,
    `pc._rotation`) to the rasterizer, which applies sigmoid / exp / normalize inside its
    per-Gaussian kernels (forward and backward): six elementwise launches fewer per step; results
    then agree with the un-fused path to rounding (expf vs torch.exp), not bit for bit.

    `viewpoint_cameras`: reference `Camera` / `MiniCam` objects, or the `HostCamera`s of `cameras_from_c2w()` (host-side
    matrices, one upload per step instead of the reference's per-view device inversions).

    Returns the reference's dict with a leading view axis:
      render (B,3,H,W), depth_3dgs (B,1,H,W), alpha_3dgs (B,1,H,W), radii (B,P),
      visibility_filter (B,P), viewspace_points (B,P,3)  [its .grad[b] is view b's screen-space
      gradient: GaussianDreamer.py:385-387 sums them],
    plus what the caller accumulates by hand over the loop (GaussianDreamer.py:253-256,289):
      radii_max (P,) = max over views, visibility_any (P,) = radii_max > 0."""
    from .rasterizer import rasterize_gaussians_batch
    cams = list(viewpoint_cameras)
    xyz = pc.get_xyz
    B, P = len(cams), xyz.shape[0]
    screenspace_points, zero_in_kernel = _screenspace_points(xyz, views=B)
    bg_color = bg_color.to(xyz.device)
    settings = [GaussianRasterizationSettings(
        image_height=int(c.image_height), image_width=int(c.image_width),
        tanfovx=_tan_half(c.FoVx), tanfovy=_tan_half(c.FoVy),
        bg=bg_color if bg_color.dim() == 1 else bg_color[i], scale_modifier=scaling_modifier,
        viewmatrix=c.world_view_transform, projmatrix=c.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=c.camera_center, prefiltered=False,
        debug=bool(getattr(pipe, "debug", False))) for i, c in enumerate(cams)]

    from .rasterizer import ZERO_MEANS2D
    act = ZERO_MEANS2D if zero_in_kernel else 0
    fuse = bool(fuse_activations) and not getattr(pipe, "compute_cov3D_python", False) and \
        all(hasattr(pc, a) for a in ("_opacity", "_scaling", "_rotation"))
    scales = rotations = cov3D_precomp = None
    if fuse:
        from .rasterizer import ACT_OPACITY_SIGMOID, ACT_ROTATION_NORMALIZE, ACT_SCALE_EXP
        act |= ACT_OPACITY_SIGMOID | ACT_SCALE_EXP | ACT_ROTATION_NORMALIZE
        opacity, scales, rotations = pc._opacity, pc._scaling, pc._rotation
    else:
        opacity = pc.get_opacity
        if getattr(pipe, "compute_cov3D_python", False):
            cov3D_precomp = pc.get_covariance(scaling_modifier)
        else:
            scales, rotations = pc.get_scaling, pc.get_rotation
    shs = colors_precomp = None
    if override_color is not None:
        colors_precomp = override_color
    elif getattr(pipe, "convert_SHs_python", False):
        raise ValueError("convert_SHs_python gives view-dependent colours: render those views with render()")
    else:
        shs = pc.get_features

    f = lambda t: None if t is None else t.float()  # noqa: E731  (AMP: kernels are fp32)
    image, radii, depth, alpha = rasterize_gaussians_batch(
        f(xyz), f(screenspace_points), f(shs), f(colors_precomp), f(opacity), f(scales), f(rotations),
        f(cov3D_precomp), settings, activation_flags=act)
    radii_max = radii.max(dim=0).values if B > 0 else radii.new_zeros((P,))
    return {"render": image,
            "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0,
            "radii": radii,
            "depth_3dgs": depth,
            "alpha_3dgs": alpha,
            "radii_max": radii_max,
            "visibility_any": radii_max > 0}
