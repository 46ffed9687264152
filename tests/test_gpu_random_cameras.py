"""Randomized check of the two result-relevant CUTS (task 7): the tile rect cut to the box of the
alpha >= 1/255 ellipse (`hgs_alpha_rect`: which (Gaussian, tile) pairs get list entries at all) and the 16-bit cell
masks (`hgs_cell_mask`: which 4x4 cells a record is blended in).  Both are proven conservative on the CPU
(tests/test_cellmask_cpu.py).  Here 200 random cameras - wide, zoomed-in (huge radii, frustum clamp), inside the cloud
(near plane, footprints of thousands of pixels), grazing - render a 5k-Gaussian cloud of blobs, needles and discs:

  1. the library is built a second time WITHOUT the two cuts (-DHGS_DEBUG_NO_CUTS: every entry upstream would make, every
     cell) and both builds render all cameras: a dropped contribution cannot hide behind an oracle's tolerance - the
     images must agree to re-association noise (2e-5: lists of thousands of records, and which cell lists take the
     four-records-per-iteration mode depends on their lengths) while the entry count drops by a third;
  2. sanity against the fp32 oracle (which applies neither cut) on the first 40 cameras: at least 99.5 % of the pixels
     of every image within 1e-4 (the rest: threshold flips - inside-the-cloud cameras evaluate `power` as a difference
     of terms of 1e3..1e5, where the exp2-folded conic here and exp there round differently; the full-size tests gate
     those pixel by pixel against fp64 with the oracle's fragile flags).
(n_contrib is not comparable with the oracle here: it counts positions in the tile's list, and this implementation's
lists are the shorter ones.)"""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import make_scene
from humangaussian_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


def _scene():
    base = make_scene(P=5000, sh_degree=0, seed=11, H=H, W=W, spread=0.45, scale=0.03)
    g = torch.Generator().manual_seed(3)
    s = base["scales"]                      # a mix of shapes: small blobs, long needles, flat discs; faint and opaque
    kind = torch.randint(0, 3, (5000,), generator=g)
    s[kind == 1, 0] *= 12.0
    s[kind == 2, :2] *= 5.0
    base["opacities"] = torch.where(torch.rand(5000, 1, generator=g) < 0.3, 0.004 + 0.03 * torch.rand(5000, 1, generator=g),
                                    0.05 + 0.9 * torch.rand(5000, 1, generator=g))
    return base


def _render_all(out_path):
    """(worker, also run as a script under LD_PRELOAD of the no-cuts build) all cameras through the raw C ABI."""
    from abi_runner import RawCall
    rng = np.random.default_rng(2024)
    base = _scene()
    res = []
    for k in range(N_CAMERAS):
        sc = dict(base)
        sc["cam"] = _camera(rng)
        rc = RawCall(sc, capacity=1 << 19)
        assert rc.forward() == 0 and not rc.status[4], k
        res.append((rc.color.cpu(), rc.depth.cpu(), rc.alpha.cpu(), rc.radii.cpu(), rc.status[0]))
    torch.save(res, out_path)


@pytest.mark.timeout(1500)
def test_cuts_change_nothing_and_images_match_the_oracle(tmp_path):
    from test_gpu_parity import oracle_forward
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    assert os.path.exists(hipcc), "the no-cuts comparison build needs hipcc on the GPU box"
    # ---- the library without the two cuts (test-only build flag; same sources, same flags otherwise)
    csrc = os.path.join(ROOT, "humangaussian_amd", "csrc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-DHGS_DEBUG_NO_CUTS"]
    procs = [subprocess.Popen([hipcc] + flags + ["-c", os.path.join(csrc, src), "-o", str(tmp_path / (src + ".o"))])
             for src in ("api.hip", "render_bwd.hip")]
    assert all(p.wait() == 0 for p in procs)
    nocuts = str(tmp_path / "libhgs_rast.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", str(tmp_path / "api.hip.o"),
                           str(tmp_path / "render_bwd.hip.o"), "-o", nocuts])
    env = dict(os.environ, LD_PRELOAD=nocuts, HGS_LIB=nocuts, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]))
    code = "import test_gpu_random_cameras as T; T._render_all(%r)" % str(tmp_path / "nocuts.pt")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    _render_all(str(tmp_path / "cuts.pt"))
    a, b = torch.load(tmp_path / "cuts.pt"), torch.load(tmp_path / "nocuts.pt")
    worst, ent, ent_nocut = 0.0, 0, 0
    for k, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x[3], y[3]), k                                   # radii
        for i in range(3):
            d = float((x[i] - y[i]).abs().max()) / (max(1.0, float(y[1].max())) if i == 1 else 1.0)
            worst = max(worst, d)
            assert d <= 2e-5, (k, i, d)
        ent += x[4]; ent_nocut += y[4]
    assert ent < 0.8 * ent_nocut, (ent, ent_nocut)                          # the rect cut is worth something
    # ---- sanity against the fp32 oracle
    rng = np.random.default_rng(2024)
    base = _scene()
    worst_frac = 0.0
    for k in range(40):
        sc = dict(base)
        sc["cam"] = _camera(rng)
        oc, orad, od, oa, aux, _ = oracle_forward(sc)
        assert torch.equal(a[k][3], orad), k                                # radii: the fp32 decision sequence, exact
        err = torch.maximum((a[k][0] - oc).abs().amax(0), torch.maximum(
            (a[k][2] - oa).abs()[0], (a[k][1] - od).abs()[0] / max(1.0, float(od.max()))))
        assert torch.isfinite(err).all(), k
        frac = float((err > 1e-4).float().mean())
        worst_frac = max(worst_frac, frac)
        assert frac <= 0.005, (k, frac, float(err.max()))
    print(f"random cameras: cuts vs no cuts {worst:.1e} (entries {ent} vs {ent_nocut}); vs fp32 oracle: at most "
          f"{100 * worst_frac:.2f} % of an image's pixels above 1e-4")
