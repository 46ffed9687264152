"""Pure-PyTorch, autograd-differentiable restatement of the 3DGS tile rasterizer that
HumanGaussian calls through `diff_gaussian_rasterization` (ashawkey fork: RGB + depth +
alpha).  TEST INFRASTRUCTURE ONLY - see oracle/__init__.py ("parity unpinned").

What each function follows (reference call sites; the arithmetic itself is un-vendored,
SURVEY.md fact 1 and Appendix A):

* `preprocess`     - stage F1.  Conventions pinned by in-tree helpers:
                     quaternion -> R  `gaussiansplatting/utils/general_utils.py:78-99`,
                     Sigma = (R S)(R S)^T  `general_utils.py:101-110` +
                     `scene/gaussian_model.py:27-31`, 6-float packing
                     `general_utils.py:64-76`, SH basis `utils/sh_utils.py:57-112`,
                     colour rule (+0.5, clamp>=0) and view direction
                     `gaussian_renderer/__init__.py:73-78`, matrix conventions
                     `scene/cameras.py:50-53`, `utils/graphics_utils.py:73-93`.
* `bin_and_sort`   - stages F2-F5: key = (tile, depth bits), ties by index.
* `_blend_tile`    - stage F6 with the backward quirks of A.6 expressed as
                     detach()/mask tricks so that autograd reproduces B1-B3.
* `rasterize`      - the `_C.rasterize_gaussians` boundary called at
                     `gaussian_renderer/__init__.py:86-94` / `gs_renderer.py:1006-1015`.

All per-Gaussian arithmetic is written component-by-component (no matmul) in a fixed
left-to-right order; the HIP preprocess kernel is written in the same order with FP
contraction off, so in float32 the per-Gaussian quantities (and therefore radii / tile
rects / list membership) agree bit-for-bit and parity tests do not flake on ceil()/==
decisions.  Run it in float64 for reference-quality gradients.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import numpy as np
import torch

TILE = 16
NEAR_Z = 0.2
LOWPASS = 0.3
ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.99
T_EPS = 1e-4

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
         0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)


class OracleSettings(NamedTuple):
    """Same 12 fields, same order, as the reference's GaussianRasterizationSettings
    (call site `gaussian_renderer/__init__.py:36-49`)."""
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


# --------------------------------------------------------------------------- preprocess

# The fork's computeCov3D backward forms `s = mod * scale` and returns dL/ds as `dL_dscale` - the factor `mod` of the
# chain rule is missing [UPSTREAM-KNOWLEDGE: cuda_rasterizer/backward.cu, computeCov3D: `dL_dscale->x = dot(Rt[0],
# dL_dMt[0])`].  True (the default) mimics that: value mod * scale, derivative 1 w.r.t. scale.  False = the true
# derivative (the HIP library's HGS_GRAD_SCALE_TRUE_DERIVATIVE).  Identical at scale_modifier == 1, the only value the
# reference passes (gaussian_renderer/__init__.py:18, gs_renderer.py:925).
FORK_SCALE_GRADIENT = True


def _cov3d_from_scale_rot(scales, rots, mod):
    """Sigma = R S^2 R^T, S = diag(mod*scale); quaternion (w,x,y,z) used AS GIVEN
    (no renormalisation: SURVEY A.6 'rotation').  Returns 6 columns xx,xy,xz,yy,yz,zz."""
    if FORK_SCALE_GRADIENT and mod != 1.0 and scales.requires_grad:
        scales = scales + (mod * scales - scales).detach()        # value mod * scale, d/dscale = 1
        mod = 1.0
    sx, sy, sz = mod * scales[:, 0], mod * scales[:, 1], mod * scales[:, 2]
    r, x, y, z = rots[:, 0], rots[:, 1], rots[:, 2], rots[:, 3]
    R00 = 1.0 - 2.0 * (y * y + z * z)
    R01 = 2.0 * (x * y - r * z)
    R02 = 2.0 * (x * z + r * y)
    R10 = 2.0 * (x * y + r * z)
    R11 = 1.0 - 2.0 * (x * x + z * z)
    R12 = 2.0 * (y * z - r * x)
    R20 = 2.0 * (x * z - r * y)
    R21 = 2.0 * (y * z + r * x)
    R22 = 1.0 - 2.0 * (x * x + y * y)
    L00, L01, L02 = R00 * sx, R01 * sy, R02 * sz
    L10, L11, L12 = R10 * sx, R11 * sy, R12 * sz
    L20, L21, L22 = R20 * sx, R21 * sy, R22 * sz
    c0 = L00 * L00 + L01 * L01 + L02 * L02
    c1 = L00 * L10 + L01 * L11 + L02 * L12
    c2 = L00 * L20 + L01 * L21 + L02 * L22
    c3 = L10 * L10 + L11 * L11 + L12 * L12
    c4 = L10 * L20 + L11 * L21 + L12 * L22
    c5 = L20 * L20 + L21 * L21 + L22 * L22
    return c0, c1, c2, c3, c4, c5


def _eval_sh(deg, sh, dx, dy, dz):
    """sh: (P, M, 3); unit direction components.  Basis/signs: sh_utils.py:74-100."""
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dx[:, None], dy[:, None], dz[:, None]
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9]
                       + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13]
                       + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return res


def preprocess(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
               cov3D_precomp, settings: OracleSettings, dtype=torch.float32):
    """Stage F1.  Returns a dict of per-Gaussian tensors; culled Gaussians have
    radii == 0 and tiles_touched == 0 (their other fields are unspecified)."""
    dev = means3D.device
    cast = lambda t: None if t is None else t.to(dtype)  # noqa: E731
    means3D, shs, colors_precomp = cast(means3D), cast(shs), cast(colors_precomp)
    opacities, scales, rotations = cast(opacities), cast(scales), cast(rotations)
    cov3D_precomp = cast(cov3D_precomp)
    V = settings.viewmatrix.to(dtype).to(dev)
    PM = settings.projmatrix.to(dtype).to(dev)
    campos = settings.campos.to(dtype).to(dev)
    H, W = int(settings.image_height), int(settings.image_width)
    P = means3D.shape[0]
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    # scalars are formed the way the host code of the rasterizer forms them: the tangents
    # arrive as fp32, focal = W / (2 * tan) and the 1.3x frustum limit are fp32 products
    ft = np.float32 if dtype == torch.float32 else np.float64
    tanx, tany = ft(np.float32(settings.tanfovx)), ft(np.float32(settings.tanfovy))
    fx = float(ft(W) / (ft(2.0) * tanx))
    fy = float(ft(H) / (ft(2.0) * tany))
    limx, limy = float(ft(np.float32(1.3)) * tanx), float(ft(np.float32(1.3)) * tany)
    tanx, tany = float(tanx), float(tany)

    x, y, z = means3D[:, 0], means3D[:, 1], means3D[:, 2]
    tx0 = V[0, 0] * x + V[1, 0] * y + V[2, 0] * z + V[3, 0]
    ty0 = V[0, 1] * x + V[1, 1] * y + V[2, 1] * z + V[3, 1]
    tz = V[0, 2] * x + V[1, 2] * y + V[2, 2] * z + V[3, 2]
    hx = PM[0, 0] * x + PM[1, 0] * y + PM[2, 0] * z + PM[3, 0]
    hy = PM[0, 1] * x + PM[1, 1] * y + PM[2, 1] * z + PM[3, 1]
    hw = PM[0, 3] * x + PM[1, 3] * y + PM[2, 3] * z + PM[3, 3]
    pw = 1.0 / (hw + 1e-7)
    projx, projy = hx * pw, hy * pw
    if means2D is not None:  # dL/dmeans2D is per NDC unit; input is zeros
        projx = projx + means2D[:, 0].to(dtype)
        projy = projy + means2D[:, 1].to(dtype)

    with torch.no_grad():
        in_front = tz > NEAR_Z

    if cov3D_precomp is not None and cov3D_precomp.numel() > 0:
        c0, c1, c2, c3, c4, c5 = (cov3D_precomp[:, i] for i in range(6))
    else:
        c0, c1, c2, c3, c4, c5 = _cov3d_from_scale_rot(
            scales, rotations, float(np.float32(settings.scale_modifier)))

    # EWA projection with the frustum clamp (value AND gradient quirk)
    tz_safe = torch.where(in_front, tz, torch.ones_like(tz))
    txtz, tytz = tx0 / tz_safe, ty0 / tz_safe
    vx = torch.clamp(txtz, -limx, limx) * tz_safe
    vy = torch.clamp(tytz, -limy, limy) * tz_safe
    with torch.no_grad():
        clx = (txtz < -limx) | (txtz > limx)
        cly = (tytz < -limy) | (tytz > limy)
    tx = torch.where(clx, vx.detach(), tx0 + (vx - tx0).detach())
    ty = torch.where(cly, vy.detach(), ty0 + (vy - ty0).detach())

    # NB: torch evaluates `python_scalar / tensor` as reciprocal(tensor) * scalar, which
    # rounds differently from a true division - divide tensor by tensor instead.
    fx_t, fy_t = torch.full_like(tz_safe, fx), torch.full_like(tz_safe, fy)
    J00 = fx_t / tz_safe
    J02 = -(fx_t * tx) / (tz_safe * tz_safe)
    J11 = fy_t / tz_safe
    J12 = -(fy_t * ty) / (tz_safe * tz_safe)
    M00, M01, M02 = (J00 * V[0, 0] + J02 * V[0, 2], J00 * V[1, 0] + J02 * V[1, 2],
                     J00 * V[2, 0] + J02 * V[2, 2])
    M10, M11, M12 = (J11 * V[0, 1] + J12 * V[0, 2], J11 * V[1, 1] + J12 * V[1, 2],
                     J11 * V[2, 1] + J12 * V[2, 2])
    u0 = M00 * c0 + M01 * c1 + M02 * c2
    u1 = M00 * c1 + M01 * c3 + M02 * c4
    u2 = M00 * c2 + M01 * c4 + M02 * c5
    w0 = M10 * c0 + M11 * c1 + M12 * c2
    w1 = M10 * c1 + M11 * c3 + M12 * c4
    w2 = M10 * c2 + M11 * c4 + M12 * c5
    a = (u0 * M00 + u1 * M01 + u2 * M02) + LOWPASS
    b = u0 * M10 + u1 * M11 + u2 * M12
    c = (w0 * M10 + w1 * M11 + w2 * M12) + LOWPASS
    det = a * c - b * b
    with torch.no_grad():
        det_ok = det != 0
    det_inv = 1.0 / torch.where(det_ok, det, torch.ones_like(det))
    conA, conB, conC = c * det_inv, -b * det_inv, a * det_inv

    with torch.no_grad():
        mid = 0.5 * (a + c)
        root = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        lam = torch.maximum(mid + root, mid - root)
        radius = torch.ceil(3.0 * torch.sqrt(lam))
        radius = torch.nan_to_num(radius, nan=0.0, posinf=0.0, neginf=0.0)

    m2x = ((projx + 1.0) * W - 1.0) * 0.5
    m2y = ((projy + 1.0) * H - 1.0) * 0.5

    with torch.no_grad():
        def _tile(v, lim):  # C int truncation toward zero, then clamp to [0, lim]
            v = torch.nan_to_num(v / TILE, nan=0.0, posinf=0.0, neginf=0.0)
            return torch.clamp(torch.trunc(v).to(torch.int64), 0, lim)
        rminx = _tile(m2x - radius, gx)
        rminy = _tile(m2y - radius, gy)
        rmaxx = _tile(m2x + radius + (TILE - 1), gx)
        rmaxy = _tile(m2y + radius + (TILE - 1), gy)
        tiles = (rmaxx - rminx) * (rmaxy - rminy)
        visible = in_front & det_ok & (tiles > 0)
        tiles = torch.where(visible, tiles, torch.zeros_like(tiles))
        radii = torch.where(visible, radius.to(torch.int64),
                            torch.zeros_like(tiles)).to(torch.int32)

    # colour
    if colors_precomp is not None and colors_precomp.numel() > 0:
        rgb = colors_precomp
        clamped = torch.zeros(P, 3, dtype=torch.bool, device=dev)
    else:
        ddx, ddy, ddz = x - campos[0], y - campos[1], z - campos[2]
        n = torch.sqrt(ddx * ddx + ddy * ddy + ddz * ddz)
        raw = _eval_sh(int(settings.sh_degree), shs, ddx / n, ddy / n, ddz / n) + 0.5
        with torch.no_grad():
            clamped = raw < 0
        rgb = torch.clamp_min(raw, 0.0)

    return dict(
        mean2D=torch.stack([m2x, m2y], 1), depth=tz, conic=torch.stack([conA, conB, conC], 1),
        opacity=opacities.reshape(-1), rgb=rgb, radii=radii, tiles_touched=tiles,
        rect=torch.stack([rminx, rminy, rmaxx, rmaxy], 1), visible=visible,
        clamped=clamped, cov2D=torch.stack([a, b, c], 1),
        cov3D=torch.stack([c0, c1, c2, c3, c4, c5], 1), grid=(gx, gy),
    )


def mark_visible(means3D, settings: OracleSettings):
    """`GaussianRasterizer.markVisible` (frustum test only)."""
    V = settings.viewmatrix.to(means3D.dtype)
    x, y, z = means3D[:, 0], means3D[:, 1], means3D[:, 2]
    tz = V[0, 2] * x + V[1, 2] * y + V[2, 2] * z + V[3, 2]
    return tz > NEAR_Z


# ------------------------------------------------------------------------------ binning

@torch.no_grad()
def bin_and_sort(pre):
    """Stages F2-F5.  Returns (gauss_idx_sorted[R], tile_id_sorted[R], ranges[T,2]).
    Order inside a tile: ascending fp32 depth, ties by ascending Gaussian index
    (stable radix sort over keys emitted in index order)."""
    gx, gy = pre["grid"]
    T = gx * gy
    tt = pre["tiles_touched"]
    idx = torch.nonzero(tt > 0).reshape(-1)
    ranges = torch.zeros(T, 2, dtype=torch.int64)
    if idx.numel() == 0:
        return idx, idx.clone(), ranges
    cnt = tt[idx]
    rep = torch.repeat_interleave(idx, cnt)                     # Gaussian per entry
    start = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(rep.numel()) - torch.repeat_interleave(start, cnt)
    rect = pre["rect"][rep]
    wdt = rect[:, 2] - rect[:, 0]
    tyy = rect[:, 1] + local // wdt                              # y outer, x inner
    txx = rect[:, 0] + local % wdt
    tile = tyy * gx + txx
    depth32 = pre["depth"].detach().to(torch.float32)[rep]
    # lexicographic (tile, depth, idx): entries are already in idx order
    o1 = torch.argsort(depth32, stable=True)
    o2 = torch.argsort(tile[o1], stable=True)
    order = o1[o2]
    g_sorted, t_sorted = rep[order], tile[order]
    counts = torch.bincount(t_sorted, minlength=T)
    ends = torch.cumsum(counts, 0)
    ranges[:, 0] = ends - counts
    ranges[:, 1] = ends
    return g_sorted, t_sorted, ranges


# ------------------------------------------------------------------------------ blending

FRAGILE_EPS_ALPHA = 2e-5   # relative distance of op*G from 1/255 below which fp32 may decide differently
FRAGILE_EPS_T = 2e-4       # relative distance of test_T from 1e-4 (T is a product of hundreds of factors)
FRAGILE_EPS_POWER = 2e-6   # absolute distance of the exponent from 0


def _blend_tile(pxf, pyf, xy, conic, opac, rgb, depth, chunk=4096, fragile_out=None):
    """Stage F6 for one tile.  pxf/pyf: (Npix,) pixel centres; the rest are the tile's
    depth-sorted Gaussians.  Returns C (Npix,3), D, Wt, T_final (Npix,), n_contrib.
    `fragile_out` (a list) receives a (Npix,) bool mask of pixels where one of the three hard
    decisions of SURVEY A.5 (power > 0, alpha < 1/255, test_T < 1e-4) sits within rounding
    distance of its threshold, i.e. where two correct fp32 implementations may legitimately
    take different branches ("threshold flips"); parity tests gate every other pixel tightly."""
    npix = pxf.shape[0]
    fragile = torch.zeros(npix, dtype=torch.bool)
    fragile_entries = torch.zeros(xy.shape[0], dtype=torch.bool)   # entries involved in such a decision
    dt = xy.dtype
    T_run = torch.ones(npix, dtype=dt)
    C = torch.zeros(npix, 3, dtype=dt)
    D = torch.zeros(npix, dtype=dt)
    Wt = torch.zeros(npix, dtype=dt)
    n_contrib = torch.zeros(npix, dtype=torch.int64)
    done = torch.zeros(npix, dtype=torch.bool)
    n = xy.shape[0]
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        dx = xy[None, s:e, 0] - pxf[:, None]
        dy = xy[None, s:e, 1] - pyf[:, None]
        A, B, Cc = conic[None, s:e, 0], conic[None, s:e, 1], conic[None, s:e, 2]
        power = -0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy
        araw = opac[None, s:e] * torch.exp(power)
        alpha = araw + (torch.clamp(araw, max=ALPHA_MAX) - araw).detach()  # straight-through
        with torch.no_grad():
            keep = (power <= 0) & (alpha >= ALPHA_MIN) & (~done[:, None])
        a_eff = torch.where(keep, alpha, torch.zeros_like(alpha))
        one_m = 1.0 - a_eff
        # running transmittance INCLUDING this Gaussian (= upstream's test_T)
        Tincl = T_run[:, None] * torch.cumprod(one_m, dim=1)
        Tbefore = torch.cat([T_run[:, None], Tincl[:, :-1]], dim=1)
        with torch.no_grad():
            stop = keep & (Tincl < T_EPS)
            stopped = torch.cumsum(stop.to(torch.int32), dim=1) > 0   # from the stopper on
            live = keep & ~stopped
            if fragile_out is not None:
                reach = (~done[:, None]) & ~(stopped & ~stop)             # entries up to and including the stopper
                near_a = (power <= FRAGILE_EPS_POWER) & ((araw / ALPHA_MIN - 1.0).abs() < FRAGILE_EPS_ALPHA)
                near_p = power.abs() < FRAGILE_EPS_POWER
                near_t = (alpha >= ALPHA_MIN * (1 - FRAGILE_EPS_ALPHA)) & ((Tincl / T_EPS - 1.0).abs() < FRAGILE_EPS_T)
                hit = reach & (near_a | near_p | near_t)
                fragile |= hit.any(dim=1)
                fragile_entries[s:e] |= hit.any(dim=0)
            pos = torch.arange(s + 1, e + 1)[None, :].expand(npix, -1)
            last = torch.where(live, pos, torch.zeros_like(pos)).amax(dim=1)
            n_contrib = torch.maximum(n_contrib, last)
        w = torch.where(live, a_eff * Tbefore, torch.zeros_like(a_eff))
        C = C + w @ rgb[s:e]
        D = D + w @ depth[s:e]
        Wt = Wt + w.sum(dim=1)
        # transmittance after the last LIVE Gaussian of this chunk
        T_run = T_run * torch.cumprod(torch.where(live, one_m, torch.ones_like(one_m)),
                                      dim=1)[:, -1]
        with torch.no_grad():
            done = done | stopped[:, -1]
    if fragile_out is not None:
        fragile_out.append(fragile)
        fragile_out.append(fragile_entries)
    return C, D, Wt, T_run, n_contrib


def rasterize(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
              cov3D_precomp, settings: OracleSettings, dtype=torch.float32,
              return_aux: bool = False, tile_stride: int = 1):
    """Full forward (differentiable).  Returns color (3,H,W), radii (P,) int32,
    depth (1,H,W), alpha (1,H,W) [+ aux dict].  tile_stride > 1 blends only every
    tile_stride-th non-empty tile (the rest stay background): a bounded SAMPLE of the
    workload for timing the CPU baseline, never used for parity."""
    H, W = int(settings.image_height), int(settings.image_width)
    bg = settings.bg.to(dtype).reshape(3)
    P = means3D.shape[0]
    color = bg[:, None, None].expand(3, H, W).clone()
    depth = torch.zeros(1, H, W, dtype=dtype)
    alpha = torch.zeros(1, H, W, dtype=dtype)
    n_contrib_img = torch.zeros(H, W, dtype=torch.int64)
    final_T = torch.ones(H, W, dtype=dtype)
    fragile_img = torch.zeros(H, W, dtype=torch.bool)
    if P == 0:
        radii = torch.zeros(0, dtype=torch.int32)
        out = (color, radii, depth, alpha)
        return out + ({"n_contrib": n_contrib_img, "final_T": final_T},) if return_aux else out

    pre = preprocess(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                     cov3D_precomp, settings, dtype)
    g_sorted, _, ranges = bin_and_sort(pre)
    gx, gy = pre["grid"]
    color_parts, depth_parts, alpha_parts = [], [], []
    active = torch.nonzero(ranges[:, 1] > ranges[:, 0]).reshape(-1).tolist()
    n_active_total = len(active)
    if tile_stride > 1:
        active = active[::tile_stride]
    colors_out = color
    for t in active:
        s, e = int(ranges[t, 0]), int(ranges[t, 1])
        gi = g_sorted[s:e]
        ty_, tx_ = divmod(t, gx)
        x0, y0 = tx_ * TILE, ty_ * TILE
        x1, y1 = min(x0 + TILE, W), min(y0 + TILE, H)
        ys, xs = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
        pxf, pyf = xs.reshape(-1).to(dtype), ys.reshape(-1).to(dtype)
        fr = [] if return_aux else None
        C, D, Wt, Tf, nc = _blend_tile(pxf, pyf, pre["mean2D"][gi], pre["conic"][gi],
                                       pre["opacity"][gi], pre["rgb"][gi], pre["depth"][gi], fragile_out=fr)
        if fr:
            fragile_img[y0:y1, x0:x1] = fr[0].reshape(y1 - y0, x1 - x0)
        Cb = C + Tf[:, None] * bg[None, :]
        color_parts.append((y0, y1, x0, x1, Cb))
        depth_parts.append(D)
        alpha_parts.append(Wt)
        n_contrib_img[y0:y1, x0:x1] = nc.reshape(y1 - y0, x1 - x0)
        final_T[y0:y1, x0:x1] = Tf.detach().reshape(y1 - y0, x1 - x0)
    # assemble without in-place writes on graph tensors
    if active:
        canvas_c = [[None] * gx for _ in range(gy)]
        canvas_d = [[None] * gx for _ in range(gy)]
        canvas_a = [[None] * gx for _ in range(gy)]
        for (y0, y1, x0, x1, Cb), D, Wt in zip(color_parts, depth_parts, alpha_parts):
            hh, ww = y1 - y0, x1 - x0
            canvas_c[y0 // TILE][x0 // TILE] = Cb.t().reshape(3, hh, ww)
            canvas_d[y0 // TILE][x0 // TILE] = D.reshape(1, hh, ww)
            canvas_a[y0 // TILE][x0 // TILE] = Wt.reshape(1, hh, ww)
        rows_c, rows_d, rows_a = [], [], []
        for j in range(gy):
            hh = min(TILE, H - j * TILE)
            rc, rd, ra = [], [], []
            for i in range(gx):
                ww = min(TILE, W - i * TILE)
                if canvas_c[j][i] is None:
                    rc.append(bg[:, None, None].expand(3, hh, ww))
                    rd.append(torch.zeros(1, hh, ww, dtype=dtype))
                    ra.append(torch.zeros(1, hh, ww, dtype=dtype))
                else:
                    rc.append(canvas_c[j][i]); rd.append(canvas_d[j][i]); ra.append(canvas_a[j][i])
            rows_c.append(torch.cat(rc, 2)); rows_d.append(torch.cat(rd, 2)); rows_a.append(torch.cat(ra, 2))
        colors_out = torch.cat(rows_c, 1)
        depth = torch.cat(rows_d, 1)
        alpha = torch.cat(rows_a, 1)
    out = (colors_out, pre["radii"], depth, alpha)
    if return_aux:
        return out + ({"n_contrib": n_contrib_img, "final_T": final_T, "pre": pre, "fragile": fragile_img,
                       "ranges": ranges, "g_sorted": g_sorted,
                       "active_tiles": n_active_total, "blended_tiles": len(active)},)
    return out


# ------------------------------------------------------------- streamed forward + backward

def forward_backward(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                     settings: OracleSettings, dL_dcolor, dL_ddepth, dL_dalpha,
                     dtype=torch.float64, want_means2D=True):
    """Forward AND backward of `rasterize` with memory bounded by ONE tile: the blend graph
    of each tile is built, back-propagated into per-Gaussian leaf copies of the preprocess
    outputs and dropped; the accumulated per-Gaussian gradients are then chained through the
    preprocess graph once.  Mathematically identical to `rasterize(...)` + autograd (same
    functions, same quirks), but usable at BASELINE.json's config 4 (500k Gaussians, lists of
    10^4 entries) in float64.  `fragile` / `flip_gaussians`: pixels / Gaussians involved in a hard
    decision that sits within rounding distance of its threshold (see `_blend_tile`).
    Returns dict(color, radii, depth, alpha, n_contrib, fragile, flip_gaussians,
    grads={means3D, means2D, shs|colors_precomp, opacities, scales, rotations|cov3D_precomp})."""
    H, W = int(settings.image_height), int(settings.image_width)
    bg = settings.bg.to(dtype).reshape(3)
    P = means3D.shape[0]
    leaf = lambda t: None if t is None else t.detach().to(dtype).clone().requires_grad_(True)  # noqa: E731
    ins = dict(means3D=leaf(means3D), shs=leaf(shs), colors_precomp=leaf(colors_precomp),
               opacities=leaf(opacities), scales=leaf(scales), rotations=leaf(rotations),
               cov3D_precomp=leaf(cov3D_precomp))
    m2 = torch.zeros(P, 3, dtype=dtype, requires_grad=True) if want_means2D else None
    pre = preprocess(ins["means3D"], m2, ins["shs"], ins["colors_precomp"], ins["opacities"],
                     ins["scales"], ins["rotations"], ins["cov3D_precomp"], settings, dtype)
    g_sorted, _, ranges = bin_and_sort(pre)
    gx, gy = pre["grid"]
    names = ("mean2D", "conic", "opacity", "rgb", "depth")
    cut = {k: pre[k].detach().clone().requires_grad_(True) for k in names}
    color = bg[:, None, None].expand(3, H, W).clone()
    depth = torch.zeros(1, H, W, dtype=dtype)
    alpha = torch.zeros(1, H, W, dtype=dtype)
    n_contrib = torch.zeros(H, W, dtype=torch.int64)
    fragile = torch.zeros(H, W, dtype=torch.bool)
    flip_gaussians = torch.zeros(P, dtype=torch.bool)
    gc, gd, ga = dL_dcolor.to(dtype), dL_ddepth.to(dtype), dL_dalpha.to(dtype)
    active = torch.nonzero(ranges[:, 1] > ranges[:, 0]).reshape(-1).tolist()
    for t in active:
        s, e = int(ranges[t, 0]), int(ranges[t, 1])
        gi = g_sorted[s:e]
        ty_, tx_ = divmod(t, gx)
        x0, y0 = tx_ * TILE, ty_ * TILE
        x1, y1 = min(x0 + TILE, W), min(y0 + TILE, H)
        ys, xs = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
        pxf, pyf = xs.reshape(-1).to(dtype), ys.reshape(-1).to(dtype)
        fr = []
        C, D, Wt, Tf, nc = _blend_tile(pxf, pyf, cut["mean2D"][gi], cut["conic"][gi], cut["opacity"][gi],
                                       cut["rgb"][gi], cut["depth"][gi], fragile_out=fr)
        Cb = C + Tf[:, None] * bg[None, :]
        hh, ww = y1 - y0, x1 - x0
        loss = ((Cb.t().reshape(3, hh, ww) * gc[:, y0:y1, x0:x1]).sum()
                + (D.reshape(hh, ww) * gd[0, y0:y1, x0:x1]).sum()
                + (Wt.reshape(hh, ww) * ga[0, y0:y1, x0:x1]).sum())
        loss.backward()
        with torch.no_grad():
            color[:, y0:y1, x0:x1] = Cb.t().reshape(3, hh, ww)
            depth[0, y0:y1, x0:x1] = D.reshape(hh, ww)
            alpha[0, y0:y1, x0:x1] = Wt.reshape(hh, ww)
            n_contrib[y0:y1, x0:x1] = nc.reshape(hh, ww)
            fragile[y0:y1, x0:x1] = fr[0].reshape(hh, ww)
            flip_gaussians[gi[fr[1]]] = True
    heads, hgrads = [], []
    for k in names:
        if cut[k].grad is not None and pre[k].requires_grad:
            heads.append(pre[k]); hgrads.append(cut[k].grad)
    if heads:
        torch.autograd.backward(heads, hgrads)
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in ins.items() if v is not None}
    if m2 is not None:
        grads["means2D"] = m2.grad if m2.grad is not None else torch.zeros_like(m2)
    return dict(color=color, radii=pre["radii"], depth=depth, alpha=alpha, n_contrib=n_contrib,
                fragile=fragile, flip_gaussians=flip_gaussians, grads=grads, num_rendered=int(g_sorted.numel()),
                max_list=int((ranges[:, 1] - ranges[:, 0]).max()) if len(active) else 0)
