"""Shared scene builders for the tests (CPU tensors; callers move them to the device)."""
import math

import numpy as np
import torch

from humangaussian_amd import synth
from oracle import OracleSettings


def make_scene(P=64, sh_degree=0, M=None, seed=0, H=48, W=64, spread=0.35, scale=0.05,
               dist=2.0, fovy=50.0, elev=10.0, azim=30.0, bg=(0.1, 0.2, 0.3)):
    """Small random scene around the origin seen by one orbit camera."""
    g = torch.Generator().manual_seed(seed)
    M = (sh_degree + 1) ** 2 if M is None else M
    means = (torch.rand(P, 3, generator=g) - 0.5) * 2 * spread
    scales = scale * torch.exp(0.5 * torch.randn(P, 3, generator=g))
    rots = torch.nn.functional.normalize(torch.randn(P, 4, generator=g), dim=1)
    opac = 0.05 + 0.9 * torch.rand(P, 1, generator=g)
    shs = torch.zeros(P, M, 3)
    shs[:, 0] = torch.randn(P, 3, generator=g) * 0.8
    if M > 1:
        shs[:, 1:] = torch.randn(P, M - 1, 3, generator=g) * 0.3
    cam = synth.orbit_camera(elev, azim, dist, fovy, H, W)
    return dict(means3D=means, scales=scales, rotations=rots, opacities=opac, shs=shs,
                cam=cam, sh_degree=sh_degree, bg=torch.tensor(bg, dtype=torch.float32))


def oracle_settings(scene, scale_modifier=1.0):
    cam = scene["cam"]
    return OracleSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5),
                          math.tan(cam.FoVy * 0.5), scene["bg"], scale_modifier,
                          cam.world_view_transform, cam.full_proj_transform,
                          scene["sh_degree"], cam.camera_center, False, False)


def cov3d_from(scene, mod=1.0):
    """Packed 6-float covariance like GaussianModel.get_covariance."""
    s, q = scene["scales"] * mod, scene["rotations"]
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    L = R * s[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1)
