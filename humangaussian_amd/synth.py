"""Synthetic inputs for benchmarks and tests (no datasets, no SMPL-X files on the box).

* `human_points`     - area-uniform samples on the reference's `load/shapes/human.obj`, normalised as
  `threestudio/utils/poser.py:337-357` (1.20 x 0.30 x 1.56, z-up, area 1.51): the cloud SURVEY.md 8(d)
  prescribes, standing in for `pcb()` (`threestudio/systems/GaussianDreamer.py:220-232`); the mesh is a LOCAL
  asset (humangaussian_amd/data/human_mesh.npz, built from the reference tree where it exists, not redistributed).
* `humanoid_points`  - the same extents as an analytic capsule humanoid (source="capsule": the fallback where the
  asset is missing); `humanoid_mesh` - the same primitives as a triangle mesh (the animation leg's fallback).
* `init_cloud`       - Gaussian parameters as `GaussianModel.create_from_pcd` makes them
  (`gaussiansplatting/scene/gaussian_model.py:124-147`: isotropic scale from the mean 3-NN
  distance, opacity 0.1, identity rotation, colour 0.5), or a randomised "mid-training"
  variant.
* `orbit_camera`     - c2w as `threestudio/data/uncond.py:378-495` builds it, then the
  matrices exactly as `gaussiansplatting/scene/cameras.py:22-53` +
  `utils/graphics_utils.py:73-99` derive them (analytically on the host).
All numpy / CPU torch; callers move tensors to the device.
"""
from __future__ import annotations

import math
import os
from typing import NamedTuple

import numpy as np
import torch

SH_C0 = 0.28209479177387814


# ------------------------------------------------------------------------------ cloud

def humanoid_points(n: int, seed: int = 0) -> np.ndarray:
    """(n,3) float32 points, z-up, centred, height ~1.56, arm span ~1.20, depth ~0.30."""
    rng = np.random.default_rng(seed)
    # (kind, centre/endpoints, radii) ; cylinders sampled on the lateral surface,
    # ellipsoids/spheres on the surface; weights = approximate areas
    parts = [
        ("cyl", (-0.09, 0.0, -0.78), (-0.09, 0.0, -0.02), (0.065, 0.065)),   # left leg
        ("cyl", (0.09, 0.0, -0.78), (0.09, 0.0, -0.02), (0.065, 0.065)),     # right leg
        ("cyl", (0.0, 0.0, -0.02), (0.0, 0.0, 0.53), (0.16, 0.10)),          # torso (elliptic)
        ("cyl", (0.17, 0.0, 0.48), (0.60, 0.0, 0.42), (0.04, 0.04)),         # arms
        ("cyl", (-0.17, 0.0, 0.48), (-0.60, 0.0, 0.42), (0.04, 0.04)),
        ("sph", (0.0, 0.0, 0.66), None, (0.10, 0.12)),                       # head
    ]
    areas = []
    for kind, a, b, r in parts:
        if kind == "cyl":
            L = np.linalg.norm(np.subtract(b, a))
            per = 2 * math.pi * math.sqrt((r[0] ** 2 + r[1] ** 2) / 2)
            areas.append(per * L)
        else:
            areas.append(4 * math.pi * r[0] * r[1])
    areas = np.asarray(areas)
    counts = rng.multinomial(n, areas / areas.sum())
    out = []
    for (kind, a, b, r), c in zip(parts, counts):
        if c == 0:
            continue
        th = rng.uniform(0, 2 * math.pi, c)
        if kind == "cyl":
            a, b = np.asarray(a), np.asarray(b)
            axis = b - a
            L = np.linalg.norm(axis)
            axis = axis / L
            ref = np.array([0.0, 1.0, 0.0])
            e1 = np.cross(axis, ref); e1 /= np.linalg.norm(e1)
            e2 = np.cross(axis, e1)
            u = rng.uniform(0, 1, c)
            pts = (a[None] + u[:, None] * L * axis[None]
                   + r[0] * np.cos(th)[:, None] * e1[None] + r[1] * np.sin(th)[:, None] * e2[None])
        else:
            zc = rng.uniform(-1, 1, c)
            rr = np.sqrt(1 - zc * zc)
            pts = np.asarray(a)[None] + np.stack(
                [r[0] * rr * np.cos(th), r[0] * rr * np.sin(th), r[1] * zc], 1)
        out.append(pts)
    pts = np.concatenate(out, 0)
    rng.shuffle(pts, axis=0)          # the real cloud has no spatial order in index space
    return pts.astype(np.float32)


def humanoid_mesh(around: int = 24, along: int = 16):
    """The capsule humanoid of `humanoid_points` as a closed-enough triangle mesh (tubes of elliptic cross-section, a
    lat-long ellipsoid for the head): (vertices (V,3) float32, faces (F,3) int32).  The animation leg's body mesh where the
    reference's human.obj asset is missing - same extents, z-up, same joint centres for `animation.MotionDriver`."""
    tubes = [((-0.09, 0.0, -0.78), (-0.09, 0.0, -0.02), (0.065, 0.065)), ((0.09, 0.0, -0.78), (0.09, 0.0, -0.02), (0.065, 0.065)),
             ((0.0, 0.0, -0.02), (0.0, 0.0, 0.53), (0.16, 0.10)),
             ((0.17, 0.0, 0.48), (0.60, 0.0, 0.42), (0.04, 0.04)), ((-0.17, 0.0, 0.48), (-0.60, 0.0, 0.42), (0.04, 0.04))]
    V, F = [], []

    def grid(pts, rows, cols, wrap):               # pts: (rows, cols, 3) -> quads split into two triangles
        base = sum(len(v) for v in V)
        V.append(pts.reshape(-1, 3))
        for r in range(rows - 1):
            for c in range(cols if wrap else cols - 1):
                i0, i1 = base + r * cols + c, base + r * cols + (c + 1) % cols
                j0, j1 = i0 + cols, i1 + cols
                F.append((i0, i1, j1)); F.append((i0, j1, j0))

    th = np.linspace(0.0, 2.0 * math.pi, around, endpoint=False)
    for a, b, r in tubes:
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        axis = (b - a) / np.linalg.norm(b - a)
        e1 = np.cross(axis, [0.0, 1.0, 0.0]); e1 /= np.linalg.norm(e1)
        e2 = np.cross(axis, e1)
        u = np.linspace(0.0, 1.0, along)
        ring = r[0] * np.cos(th)[:, None] * e1[None] + r[1] * np.sin(th)[:, None] * e2[None]
        grid(a[None, None] + u[:, None, None] * (b - a)[None, None] + ring[None], along, around, True)
    ph = np.linspace(0.02, math.pi - 0.02, along)
    head = np.stack([0.10 * np.sin(ph)[:, None] * np.cos(th)[None], 0.10 * np.sin(ph)[:, None] * np.sin(th)[None],
                     0.12 * np.cos(ph)[:, None] * np.ones_like(th)[None]], -1) + np.array([0.0, 0.0, 0.66])
    grid(head, along, around, True)
    return np.concatenate(V, 0).astype(np.float32), np.asarray(F, np.int32)


def human_mesh():
    """(vertices (V,3) float32, faces (F,3) int32, label): the reference's normalised human.obj where the local asset
    exists (humangaussian_amd/data), else the procedural capsule mesh - the label says which."""
    from . import data
    if data.have_human_mesh():
        m = np.load(data.HUMAN_MESH)
        return m["vertices"].astype(np.float32), m["faces"].astype(np.int32), "human_obj"
    v, f = humanoid_mesh()
    return v, f, "capsule"


def human_points(n: int, seed: int = 0) -> np.ndarray:
    """(n,3) float32 points sampled AREA-UNIFORMLY on the reference's `load/shapes/human.obj`, normalised as the reference
    normalises its body mesh (threestudio/utils/poser.py:337-357 with `scale(-10)`: extent 1.20 x 0.30 x 1.56, z-up,
    area 1.51) - the cloud SURVEY.md 8(d) prescribes for every benchmark configuration, standing in for
    `skel.sample_smplx_points` (GaussianDreamer.py:220-232).  The mesh is the LOCAL asset
    humangaussian_amd/data/human_mesh.npz (data/make_human_mesh.py; not redistributed); seeded `numpy.random.default_rng(seed)`."""
    from . import data
    if not data.have_human_mesh():
        raise FileNotFoundError(f"{data.HUMAN_MESH} is missing (generate it with humangaussian_amd/data/make_human_mesh.py "
                                f"where /root/reference exists), or ask for source='capsule' / 'auto'")
    m = np.load(data.HUMAN_MESH)
    v, f = m["vertices"].astype(np.float64), m["faces"]
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    rng = np.random.default_rng(seed)
    tri = np.searchsorted(np.cumsum(area) / area.sum(), rng.uniform(0, 1, n), side="right").clip(0, len(f) - 1)
    r1, r2 = np.sqrt(rng.uniform(0, 1, n)), rng.uniform(0, 1, n)
    w0, w1, w2 = 1.0 - r1, r1 * (1.0 - r2), r1 * r2                   # uniform on the triangle
    pts = w0[:, None] * a[tri] + w1[:, None] * b[tri] + w2[:, None] * c[tri]
    return pts.astype(np.float32)                                     # (drawn independently: no spatial order in index space)


def body_points(n: int, seed: int = 0, source: str = "auto") -> np.ndarray:
    """`human_points` where the local mesh asset exists, `humanoid_points` otherwise (or as `source` says)."""
    return human_points(n, seed) if resolve_cloud_source(source) == "human_obj" else humanoid_points(n, seed)


CLOUD_SOURCES = ("auto", "human_obj", "capsule")


def resolve_cloud_source(source: str = "auto") -> str:
    """"auto" -> "human_obj" where the local mesh asset exists, else "capsule" (callers report the resolved name)."""
    if source != "auto":
        return source
    from . import data
    return "human_obj" if data.have_human_mesh() else "capsule"


def mean_knn_dist2(points: np.ndarray, k: int = 3) -> np.ndarray:
    """Mean squared distance to the k nearest neighbours (what simple_knn.distCUDA2
    returns, `submodules/simple-knn/simple_knn.cu:147-183`)."""
    from scipy.spatial import cKDTree
    d, _ = cKDTree(points).query(points, k=k + 1)
    return (d[:, 1:] ** 2).mean(1).astype(np.float32)


class Cloud(NamedTuple):
    means3D: torch.Tensor     # (P,3)
    shs: torch.Tensor         # (P,M,3)
    opacities: torch.Tensor   # (P,1)  post-sigmoid
    scales: torch.Tensor      # (P,3)  post-exp
    rotations: torch.Tensor   # (P,4)  post-normalise (w,x,y,z)
    sh_degree: int


def init_cloud(n: int, sh_degree: int = 0, variant: str = "mid", seed: int = 0, source: str = "auto") -> Cloud:
    """source "human_obj" (SURVEY.md 8(d)'s cloud, `human_points`: needs the local mesh asset), "capsule"
    (`humanoid_points`: the analytic stand-in, same extents, workload shape within 2 %), or "auto" (the default):
    human_obj where the asset exists, capsule otherwise - `resolve_cloud_source` tells a caller which."""
    rng = np.random.default_rng(seed + 1)
    source = resolve_cloud_source(source)
    if source not in ("human_obj", "capsule"):
        raise ValueError(source)
    pts = body_points(n, seed, source)
    M = (sh_degree + 1) ** 2
    d2 = np.maximum(mean_knn_dist2(pts), 1e-7)
    log_scale = np.log(np.sqrt(d2))[:, None].repeat(3, 1)
    shs = np.zeros((n, M, 3), np.float32)
    shs[:, 0, :] = (0.5 - 0.5) / SH_C0                      # RGB2SH(0.5)
    if variant == "init":
        opac = np.full((n, 1), 0.1, np.float32)
        quat = np.zeros((n, 4), np.float32); quat[:, 0] = 1.0
    elif variant == "mid":
        shs[:, 0, :] += rng.normal(0, 0.3, (n, 3)) / 1.0
        if M > 1:
            shs[:, 1:, :] = rng.normal(0, 0.1, (n, M - 1, 3))
        opac = rng.uniform(0.05, 0.95, (n, 1)).astype(np.float32)
        log_scale = log_scale + rng.normal(0, 0.3, (n, 3))
        quat = rng.normal(0, 1, (n, 4)).astype(np.float32)
        quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    else:
        raise ValueError(variant)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))  # noqa: E731
    return Cloud(t(pts), t(shs), t(opac), t(np.exp(log_scale)), t(quat), sh_degree)


# ----------------------------------------------------------------------------- cameras

class Cam(NamedTuple):
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor   # (4,4) = w2c^T
    full_proj_transform: torch.Tensor    # (4,4) = V @ P^T
    camera_center: torch.Tensor          # (3,)
    c2w: torch.Tensor                    # (4,4)


def c2w_orbit(elev_deg: float, azim_deg: float, dist: float, center=(0.0, 0.0, 0.0)) -> np.ndarray:
    el, az = math.radians(elev_deg), math.radians(azim_deg)
    ctr = np.asarray(center, np.float64)
    pos = ctr + dist * np.array([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az),
                                 math.sin(el)])
    up = np.array([0.0, 0.0, 1.0])
    look = ctr - pos; look /= np.linalg.norm(look)
    right = np.cross(look, up); right /= np.linalg.norm(right)
    up2 = np.cross(right, look); up2 /= np.linalg.norm(up2)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up2, -look, pos
    return c2w


def camera_from_c2w(c2w: np.ndarray, fovy: float, H: int, W: int, znear=0.01, zfar=100.0) -> Cam:
    focal = H / (2.0 * math.tan(fovy / 2.0))
    fovx = 2.0 * math.atan(W / (2.0 * focal))
    w2c = np.linalg.inv(c2w)
    w2c[1:3, :3] *= -1          # the reference's "rectify" step (cameras.py:28-29)
    w2c[:3, 3] *= -1
    V = w2c.T
    ty, tx = math.tan(fovy / 2.0), math.tan(fovx / 2.0)
    Pm = np.zeros((4, 4))
    Pm[0, 0] = 1.0 / tx
    Pm[1, 1] = 1.0 / ty
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    full = V @ Pm.T
    center = np.linalg.inv(V)[3, :3]
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))  # noqa: E731
    return Cam(H, W, fovx, fovy, f32(V), f32(full), f32(center), f32(c2w))


def orbit_camera(elev_deg, azim_deg, dist, fovy_deg, H, W, center=(0.0, 0.0, 0.0)) -> Cam:
    return camera_from_c2w(c2w_orbit(elev_deg, azim_deg, dist, center), math.radians(fovy_deg), H, W)


def random_cameras(n: int, H: int, W: int, seed: int = 0, stratified: bool = True):
    """`configs/test.yaml` ranges: fovy 40-70 deg, distance 1.5-2.0, elevation -30..30,
    azimuth stratified over the batch like `uncond.py:353-361`."""
    rng = np.random.default_rng(seed + 7)
    cams = []
    for i in range(n):
        if stratified:
            az = (rng.uniform() + i) / n * 360.0 - 180.0
        else:
            az = rng.uniform(-180, 180)
        cams.append(orbit_camera(rng.uniform(-30, 30), az, rng.uniform(1.5, 2.0),
                                 rng.uniform(40, 70), H, W))
    return cams
