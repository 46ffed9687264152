"""Randomized check of the two result-relevant CUTS (task 7): the tile rect cut to the box of the
alpha >= 1/255 ellipse (`hgs_alpha_rect`: which (Gaussian, tile) pairs get list entries at all) and the 16-bit cell
masks (`hgs_cell_mask`: which 4x4 cells a record is blended in).  Both are proven conservative on the CPU
(tests/test_cellmask_cpu.py); here 200 random cameras - wide, zoomed-in (huge radii, frustum clamp), grazing, partly
behind the near plane - render a 5k-Gaussian cloud on the GPU and EVERY pixel is compared with the fp32 oracle, which
applies neither cut: a too-tight rect or mask drops contributions and shows up in n_contrib (exact list positions) and
in the images."""
import math

import numpy as np
import pytest
import torch

from abi_runner import RawCall
from helpers import make_scene
from humangaussian_amd import synth
from test_gpu_parity import oracle_forward

pytestmark = pytest.mark.gpu

N_CAMERAS = 200
H = W = 80            # 5 x 5 tiles


def _camera(rng):
    kind = rng.integers(0, 4)
    if kind == 0:      # ordinary orbit
        elev, dist, fov = rng.uniform(-40, 40), rng.uniform(1.2, 3.0), rng.uniform(35, 75)
    elif kind == 1:    # zoomed in: large radii, Gaussians outside 1.3 tan(fov)
        elev, dist, fov = rng.uniform(-30, 30), rng.uniform(0.9, 1.6), rng.uniform(8, 25)
    elif kind == 2:    # inside the cloud: many Gaussians behind / at the near plane, huge footprints
        elev, dist, fov = rng.uniform(-60, 60), rng.uniform(0.15, 0.6), rng.uniform(50, 100)
    else:              # grazing / top-down with a wide lens
        elev, dist, fov = rng.choice([-1, 1]) * rng.uniform(70, 88), rng.uniform(1.0, 2.5), rng.uniform(60, 110)
    return synth.orbit_camera(float(elev), float(rng.uniform(0, 360)), float(dist), float(fov), H, W)


@pytest.mark.timeout(1500)
def test_random_cameras_n_contrib_and_images_match_the_uncut_oracle():
    rng = np.random.default_rng(2024)
    base = make_scene(P=5000, sh_degree=0, seed=11, H=H, W=W, spread=0.45, scale=0.03)
    g = torch.Generator().manual_seed(3)
    # a mix of shapes: small blobs, long needles, flat discs; faint and opaque
    s = base["scales"]
    kind = torch.randint(0, 3, (5000,), generator=g)
    s[kind == 1, 0] *= 12.0
    s[kind == 2, :2] *= 5.0
    base["opacities"] = torch.where(torch.rand(5000, 1, generator=g) < 0.3, 0.004 + 0.03 * torch.rand(5000, 1, generator=g),
                                    0.05 + 0.9 * torch.rand(5000, 1, generator=g))
    worst_img, bad_pixels, entries, entries_upstream = 0.0, 0, 0, 0
    for k in range(N_CAMERAS):
        sc = dict(base)
        sc["cam"] = _camera(rng)
        rc = RawCall(sc, capacity=1 << 19)
        assert rc.forward() == 0 and not rc.status[4], k
        oc, orad, od, oa, aux, _ = oracle_forward(sc)
        assert torch.equal(rc.radii.cpu(), orad), k
        ncon = np.frombuffer(rc.img[: H * W * 4].cpu().numpy().tobytes(), dtype=np.uint32).reshape(H, W)
        ref = aux["n_contrib"].numpy().astype(np.uint32)
        same = ncon == ref
        # threshold flips (alpha within rounding of 1/255, T of 1e-4: exp2-folded conic here, exp there) move the LAST
        # contributor of a pixel; a dropped entry would shift whole cells
        bad = int((~same).sum())
        bad_pixels += bad
        assert bad <= 4, (k, bad)
        err = torch.maximum((rc.color.cpu() - oc).abs().amax(0), torch.maximum((rc.alpha.cpu() - oa).abs()[0],
                            (rc.depth.cpu() - od).abs()[0] / max(1.0, float(od.max()))))
        e_same = float(err[torch.from_numpy(same)].max()) if same.any() else 0.0
        worst_img = max(worst_img, e_same)
        assert e_same <= 1e-4, (k, e_same)
        assert float(err.max()) <= 2.0 / 255.0, (k, float(err.max()))
        entries += rc.status[0]
        entries_upstream += int((aux["ranges"][:, 1] - aux["ranges"][:, 0]).sum())
    assert bad_pixels <= N_CAMERAS // 2, bad_pixels            # flips are rare
    assert entries < 0.97 * entries_upstream                    # and the rect cut did remove entries
    print(f"random cameras: worst image error on matching pixels {worst_img:.2e}, {bad_pixels} flip pixels of "
          f"{N_CAMERAS * H * W}, entries {entries} vs upstream {entries_upstream}")
