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


# ----------------------------------------------------------------- full-size parity helper
PARITY_LOG = []


def _dump_parity_log():
    import json
    import os
    path = os.environ.get("HGS_PARITY_STATS")
    if path and PARITY_LOG:
        with open(path, "w") as f:
            json.dump(PARITY_LOG, f, indent=1)


import atexit  # noqa: E402

atexit.register(_dump_parity_log)


def check_against_fp64_oracle(name, cloud, settings_fp32, hip, grads, img_tol=1e-4, grad_tol=1e-3,
                              cos_tol=1e-6, threads=None, minority_caps=True, gate_flip_images=True, raise_on_failure=True,
                              grad_tol_by_key=None):
    """BASELINE.json's parity bar at full size.  `hip` = (color, radii, depth, alpha, grads dict)
    on the CPU; `grads` = the three incoming gradients (or None: forward only).

    * radii: EXACT against the fp32 oracle (bit-identical per-Gaussian arithmetic);
    * images vs the fp32 oracle: every pixel <= img_tol (depth: relative to the scene's largest
      depth) except THRESHOLD-FLIP pixels, which are identified explicitly, not by a budget: the
      oracle flags a pixel when one of the hard decisions of the blend (alpha >= 1/255,
      power <= 0, test_T >= 1e-4) lies within rounding distance of its threshold, so that two
      correct fp32 evaluations (exp2-folded conic here, exp there) may take different branches;
      flagged pixels must stay below two minimal contributions (2/255) and stay a tiny minority;
    * gradients, per tensor and per Gaussian, relative to max |g64|:
        A. vs the fp32 oracle <= grad_tol          (parity in the arithmetic the reference uses)
        B. vs the fp64 oracle <= max(grad_tol, 1.25 x the fp32 oracle's own distance to fp64)
           (flagged flip Gaussians: max(10 x that, 1.25 x the fp32 oracle's own distance on the flagged ones))
      and cosine(g, g64) >= 1 - max(cos_tol, 2 x (1 - cosine(g32, g64))).  B's second term exists
      because fp32 itself - the oracle included, and upstream's fp32 CUDA kernels with it - is not
      within 1e-3 of the exact gradient on every workload: at configs[3] (500k sub-pixel Gaussians,
      the 0.3 px low-pass dominates cov2D) the conic chain amplifies rounding to ~2e-2 of max|g|;
      at configs[1] it is 9e-4.  Both distances are recorded.
      The hard alpha threshold makes the gradient DISCONTINUOUS: one flipped (pixel, Gaussian)
      pair moves that Gaussian's gradient by up to |dL/dalpha| / 255 * |conic d| * W/2 - a few
      1e-3 of max|g| here - so the Gaussians the oracle flagged as involved in a threshold decision
      (fp32 or fp64 run; a few hundred of 10^5) are gated at 10 x the bound instead and counted.
    The images are gated against the fp32 oracle because fp32 itself (any implementation, the
    oracle included) is not within 1e-4 of an fp64 evaluation on every pixel of a million:
    ill-conditioned conics carry alpha errors of ~1e-4 relative; how far the fp32 oracle and this
    implementation each are from fp64 is recorded in the statistics (informational).
    Returns the statistics (also appended to PARITY_LOG / $HGS_PARITY_STATS)."""
    import os
    import oracle
    if not torch.is_grad_enabled():          # the oracle differentiates with autograd
        with torch.enable_grad():
            return check_against_fp64_oracle(name, cloud, settings_fp32, hip, grads, img_tol, grad_tol, cos_tol, threads,
                                             minority_caps, gate_flip_images, raise_on_failure, grad_tol_by_key)
    torch.set_num_threads(threads or max(1, min(os.cpu_count() or 1, 64)))
    c, r, d, a, g = hip
    st = settings_fp32
    H, W = int(st.image_height), int(st.image_width)
    zero = [torch.zeros(3, H, W), torch.zeros(1, H, W), torch.zeros(1, H, W)]
    args = (cloud.means3D, cloud.shs, None, cloud.opacities, cloud.scales, cloud.rotations, None, st)
    o32 = oracle.forward_backward(*args, *(grads if grads is not None else zero), dtype=torch.float32,
                                  want_means2D=grads is not None)
    assert torch.equal(r, o32["radii"]), f"{name}: radii differ from the fp32 oracle"
    o64 = oracle.forward_backward(*args, *(grads if grads is not None else zero), dtype=torch.float64,
                                  want_means2D=grads is not None)
    fragile = o32["fragile"]
    failures = []

    def gate(ok, *what):
        if not ok:
            failures.append(what)

    stats = {"case": name, "P": int(cloud.means3D.shape[0]), "num_rendered": o32["num_rendered"],
             "longest_tile_list": o32["max_list"], "flagged_flip_pixels": int(fragile.sum()),
             "n_contrib_fp32_vs_fp64_oracle_differs": int((o32["n_contrib"] != o64["n_contrib"]).sum())}
    dmax = float(o32["depth"].max())
    for key, got, r32, r64, scale in (("color", c, o32["color"], o64["color"], 1.0),
                                      ("alpha", a, o32["alpha"], o64["alpha"], 1.0),
                                      ("depth", d, o32["depth"], o64["depth"], max(dmax, 1.0))):
        err = (got.double() - r32.double()).abs().reshape(-1, H, W).amax(dim=0)
        solid, flips = err[~fragile], err[fragile]
        stats[f"{key}_max_err_nonflip"] = float(solid.max()) if solid.numel() else 0.0
        stats[f"{key}_flipped_pixels_above_tol"] = int((flips > img_tol * scale).sum())
        stats[f"{key}_max_err_flip"] = float(flips.max()) if flips.numel() else 0.0
        e64 = (got.double() - r64).abs().reshape(-1, H, W).amax(dim=0)
        o64e = (r32.double() - r64).abs().reshape(-1, H, W).amax(dim=0)
        stats[f"{key}_vs_fp64_pixels_above_tol"] = int((e64 > img_tol * scale).sum())
        stats[f"{key}_fp32oracle_vs_fp64_pixels_above_tol"] = int((o64e > img_tol * scale).sum())
        gate(stats[f"{key}_max_err_nonflip"] <= img_tol * scale, key, "non-flip pixel above tolerance")
        gate(not gate_flip_images or stats[f"{key}_max_err_flip"] <= (2.1 / 255.0) * scale + img_tol * scale, key, "flip pixel above 2/255")
    # (minority_caps=False: cameras inside the cloud, tests/test_gpu_random_cameras.py - a sizeable share of their pixels sits
    #  within rounding distance of a threshold; the count is recorded, the other gates stay)
    gate(not minority_caps or stats["flagged_flip_pixels"] <= 0.002 * H * W, "too many flagged pixels")     # the exemption stays a tiny minority
    if grads is not None:
        Pn = int(cloud.means3D.shape[0])
        flipg = o32["flip_gaussians"] | o64["flip_gaussians"]
        stats["flagged_flip_gaussians"] = int(flipg.sum())
        gate(not minority_caps or stats["flagged_flip_gaussians"] <= 0.02 * Pn, "too many flagged Gaussians")
        for k, ref in o64["grads"].items():
            if k not in g:
                continue
            got = g[k].double().reshape(ref.shape)
            scale = max(float(ref.abs().max()), 1e-300)
            err = (got - ref).abs().reshape(Pn, -1).amax(dim=1)
            cos = float(torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0))
            r32 = o32["grads"][k].double().reshape(ref.shape)
            e32 = (r32 - ref).abs().reshape(Pn, -1).amax(dim=1)             # fp32 oracle vs fp64
            eA = (got - r32).abs().reshape(Pn, -1).amax(dim=1)              # this implementation vs fp32 oracle
            cos32 = float(torch.nn.functional.cosine_similarity(r32.flatten(), ref.flatten(), dim=0))
            nf = ~flipg
            has_flip = stats["flagged_flip_gaussians"] > 0
            st_ = {"vs_fp64_nonflip": float(err[nf].max()) / scale,
                   "vs_fp64_flip": float(err[flipg].max()) / scale if has_flip else 0.0,
                   "vs_fp32oracle_nonflip": float(eA[nf].max()) / scale,
                   "vs_fp32oracle_flip": float(eA[flipg].max()) / scale if has_flip else 0.0,
                   "fp32oracle_vs_fp64_nonflip": float(e32[nf].max()) / scale,
                   "fp32oracle_vs_fp64_flip": float(e32[flipg].max()) / scale if has_flip else 0.0,
                   "flip_gaussians_above_tol": int((err[flipg] > grad_tol * scale).sum()),
                   "1-cos": 1.0 - cos, "fp32oracle_1-cos": 1.0 - cos32}
            for kk, vv in st_.items():
                stats[f"grad_{k}_{kk}"] = vv
            gtol = (grad_tol_by_key or {}).get(k, grad_tol)      # (per-tensor bound: tests/test_gpu_random_cameras.py)
            boundB = max(gtol, 1.25 * st_["fp32oracle_vs_fp64_nonflip"])
            gate(st_["vs_fp32oracle_nonflip"] <= gtol, k, "A: vs fp32 oracle", st_["vs_fp32oracle_nonflip"])
            gate(st_["vs_fp32oracle_flip"] <= 10 * gtol, k, "A: vs fp32 oracle (flip Gaussians)", st_["vs_fp32oracle_flip"])
            gate(st_["vs_fp64_nonflip"] <= boundB, k, "B: vs fp64 oracle", st_["vs_fp64_nonflip"], boundB)
            # (a flipped threshold decision moves a Gaussian's gradient by a finite step: where the fp32 ORACLE's own step
            #  against fp64 exceeds ten bounds - 1.14e-2 on one Gaussian of configs[2]'s view 5 on the human.obj cloud - the
            #  gate is 1.25 x that step, as for the non-flip Gaussians; gate A above still pins us to the fp32 oracle)
            gate(st_["vs_fp64_flip"] <= max(10 * boundB, 1.25 * st_["fp32oracle_vs_fp64_flip"]), k,
                 "B: vs fp64 oracle (flip Gaussians)", st_["vs_fp64_flip"], st_["fp32oracle_vs_fp64_flip"])
            gate(1.0 - cos <= max(cos_tol, 2.0 * (1.0 - cos32), gtol * gtol), k, "cosine", cos, cos32)
    stats["failures"] = [list(map(str, f)) for f in failures]
    PARITY_LOG.append(stats)
    if raise_on_failure:
        print("PARITY", stats)
        assert not failures, (name, failures, stats)
    return stats
