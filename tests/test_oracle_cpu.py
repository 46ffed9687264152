"""CPU tests (no GPU): pin the oracle.

* against the golden vectors produced by the REFERENCE's own PyTorch helpers
  (tests/golden/reference_helpers.npz <- tests/golden/make_golden.py, run where
  /root/reference exists): SH basis, covariance packing, quaternion convention, camera /
  projection matrices;
* fp64 gradcheck: the oracle's backward is the derivative of its forward;
* the quirk list of SURVEY.md A.5/A.6 and structural invariants.
"""
import math
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import gs_oracle
from helpers import cov3d_from, make_scene, oracle_settings
from humangaussian_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ref():
    return np.load(os.path.join(GOLD, "reference_helpers.npz"))


# ------------------------------------------------------------ reference golden vectors

@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_basis_matches_reference_eval_sh(ref, deg):
    sh = torch.from_numpy(ref["sh_coeffs"]).double()
    d = torch.from_numpy(ref["sh_dirs"]).double()
    got = gs_oracle._eval_sh(deg, sh, d[:, 0], d[:, 1], d[:, 2])
    assert np.abs(got.numpy() - ref[f"sh_eval_deg{deg}"]).max() < 2e-6


def test_rgb2sh_constant(ref):
    assert abs((0.5 - 0.5) / gs_oracle.SH_C0 - float(ref["rgb2sh_half"])) < 1e-7
    assert abs(synth.SH_C0 - gs_oracle.SH_C0) == 0


@pytest.mark.parametrize("mod", [1.0, 0.7])
def test_covariance_matches_reference_build_covariance(ref, mod):
    s = torch.from_numpy(ref["cov_scales"]).double()
    q = torch.from_numpy(ref["cov_rots"]).double()
    got = torch.stack(gs_oracle._cov3d_from_scale_rot(s, q, mod), 1).numpy()
    want = ref[f"cov_mod{mod}"]
    assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
    # the test helper used to feed cov3D_precomp follows the same packing
    sc = {"scales": s.float(), "rotations": q.float()}
    assert np.abs(cov3d_from(sc, mod).numpy() - want).max() <= 2e-6


def test_camera_matrices_match_reference_camera_class(ref):
    for i in range(ref["cam_c2w"].shape[0]):
        H, W = (int(v) for v in ref["cam_hw"][i])
        cam = synth.camera_from_c2w(ref["cam_c2w"][i].astype(np.float64), float(ref["cam_fovy"][i]), H, W)
        assert abs(cam.FoVx - float(ref["cam_fovx"][i])) < 1e-6
        assert np.abs(cam.world_view_transform.numpy() - ref["cam_V"][i]).max() < 2e-6
        assert np.abs(cam.full_proj_transform.numpy() - ref["cam_full"][i]).max() < 5e-6
        assert np.abs(cam.camera_center.numpy() - ref["cam_center"][i]).max() < 2e-6


def test_camera_convention_pixel_mapping():
    """camera space x right / y down / z forward, world z up,
    pix = f * x / z + W/2 - 0.5."""
    cam = synth.orbit_camera(10.0, 30.0, 1.75, 55.0, 1024, 1024)
    sc = make_scene(P=2, H=1024, W=1024)
    sc["cam"] = cam
    sc["means3D"] = torch.tensor([[0.0, 0.0, 0.0], [0.0, 0.0, 0.7]])
    pre = oracle.preprocess(sc["means3D"], None, sc["shs"], None, sc["opacities"], sc["scales"],
                            sc["rotations"], None, oracle_settings(sc))
    assert abs(float(pre["depth"][0]) - 1.75) < 1e-5
    assert torch.allclose(pre["mean2D"][0], torch.tensor([511.5, 511.5]), atol=1e-2)
    assert float(pre["mean2D"][1, 1]) < 200.0          # +z world is up = smaller pixel y


# ------------------------------------------------------------------- oracle self-tests

def test_gradcheck_fp64():
    sc = make_scene(P=10, sh_degree=2, H=32, W=32, spread=0.25, scale=0.08, seed=4)
    st = oracle_settings(sc)
    g = torch.Generator().manual_seed(0)
    wc, wd, wa = (torch.randn(s, generator=g, dtype=torch.float64) for s in ((3, 32, 32), (1, 32, 32), (1, 32, 32)))

    def f(m, s, q, o, sh):
        c, _, d, a = oracle.rasterize(m, None, sh, None, o, s, q, None, st, dtype=torch.float64)
        return (c * wc).sum() + (d * wd).sum() + (a * wa).sum()

    ins = [sc[k].double().requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs")]
    assert torch.autograd.gradcheck(f, ins, eps=1e-6, atol=1e-5, rtol=1e-4)


def test_gradcheck_fp64_precomputed_inputs():
    sc = make_scene(P=8, H=32, W=32, spread=0.25, scale=0.08, seed=6)
    st = oracle_settings(sc)
    cov = cov3d_from(sc).double().requires_grad_(True)
    col = torch.rand(8, 3, dtype=torch.float64, requires_grad=True)
    g = torch.Generator().manual_seed(1)
    wc = torch.randn(3, 32, 32, generator=g, dtype=torch.float64)

    def f(m, o, c3, cp):
        c, _, d, a = oracle.rasterize(m, None, None, cp, o, None, None, c3, st, dtype=torch.float64)
        return (c * wc).sum() + d.sum() - a.sum()

    ins = [sc["means3D"].double().requires_grad_(True), sc["opacities"].double().requires_grad_(True), cov, col]
    assert torch.autograd.gradcheck(f, ins, eps=1e-6, atol=1e-5, rtol=1e-4)


def test_means2d_gradient_is_in_ndc_units():
    """dL/dmeans2D[:, :2] = dL/d(pixel mean) * (0.5 W, 0.5 H), z = 0."""
    sc = make_scene(P=30, H=48, W=64, seed=8)
    st = oracle_settings(sc)
    m2d = torch.zeros(30, 3, dtype=torch.float64, requires_grad=True)
    c, _, d, a = oracle.rasterize(sc["means3D"].double(), m2d, sc["shs"].double(), None,
                                  sc["opacities"].double(), sc["scales"].double(),
                                  sc["rotations"].double(), None, st, dtype=torch.float64)
    (c.sum() + d.sum()).backward()
    g = m2d.grad
    assert float(g[:, 2].abs().max()) == 0.0 and float(g[:, :2].abs().max()) > 0
    # finite difference of the pixel-space mean by 1e-4 px in x
    eps = 1e-4
    k = int(g[:, 0].abs().argmax())
    shift = torch.zeros(30, 3, dtype=torch.float64)
    shift[k, 0] = eps / (0.5 * 64)
    outs = []
    for sgn in (+1, -1):
        c2, _, d2, _ = oracle.rasterize(sc["means3D"].double(), sgn * shift, sc["shs"].double(), None,
                                        sc["opacities"].double(), sc["scales"].double(),
                                        sc["rotations"].double(), None, st, dtype=torch.float64)
        outs.append(float(c2.sum() + d2.sum()))
    fd_pix = (outs[0] - outs[1]) / (2 * eps)
    assert abs(fd_pix * 0.5 * 64 - float(g[k, 0])) <= 1e-4 * abs(float(g[k, 0])) + 1e-9


def test_blend_semantics_sequential_reference():
    """The vectorised tile blend equals a literal per-pixel loop of SURVEY A.5 (skip rules,
    the terminating Gaussian is not blended, n_contrib = last contributor)."""
    g = torch.Generator().manual_seed(3)
    n = 300
    xy = torch.rand(n, 2, generator=g) * 16
    conic = torch.stack([0.05 + torch.rand(n, generator=g) * 0.3, (torch.rand(n, generator=g) - 0.5) * 0.05,
                         0.05 + torch.rand(n, generator=g) * 0.3], 1)
    opac = 0.3 + 0.69 * torch.rand(n, generator=g)
    rgb = torch.rand(n, 3, generator=g)
    dep = torch.sort(1 + torch.rand(n, generator=g))[0]
    ys, xs = torch.meshgrid(torch.arange(16), torch.arange(16), indexing="ij")
    pxf, pyf = xs.reshape(-1).float(), ys.reshape(-1).float()
    C, D, Wt, Tf, nc = gs_oracle._blend_tile(pxf, pyf, xy, conic, opac, rgb, dep, chunk=64)
    stops = 0
    for p in range(0, 256, 7):
        T, c, d, w, last = 1.0, np.zeros(3), 0.0, 0.0, 0
        for j in range(n):
            dx, dy = float(xy[j, 0] - pxf[p]), float(xy[j, 1] - pyf[p])
            power = -0.5 * (float(conic[j, 0]) * dx * dx + float(conic[j, 2]) * dy * dy) - float(conic[j, 1]) * dx * dy
            if power > 0:
                continue
            al = min(0.99, float(opac[j]) * math.exp(power))
            if al < 1 / 255:
                continue
            tt = T * (1 - al)
            if tt < 1e-4:
                stops += 1
                break
            c += rgb[j].numpy() * al * T; d += float(dep[j]) * al * T; w += al * T
            T = tt; last = j + 1
        assert abs(T - float(Tf[p])) < 1e-5 and int(nc[p]) == last
        assert np.abs(c - C[p].numpy()).max() < 1e-5 and abs(d - float(D[p])) < 1e-4 and abs(w - float(Wt[p])) < 1e-5
    assert stops > 5       # early termination really exercised


def test_structural_invariants_and_culling():
    sc = make_scene(P=400, sh_degree=1, H=64, W=80, seed=12, spread=0.5)
    sc["means3D"][:40] += torch.tensor(sc["cam"].camera_center) * 2.0     # behind the camera
    st = oracle_settings(sc)
    c, r, d, a, aux = oracle.rasterize(sc["means3D"], None, sc["shs"], None, sc["opacities"], sc["scales"],
                                       sc["rotations"], None, st, return_aux=True)
    assert r.dtype == torch.int32 and int(r[:40].abs().sum()) == 0 and int((r[40:] > 0).sum()) > 300
    assert torch.equal(oracle.mark_visible(sc["means3D"], st), aux["pre"]["depth"] > 0.2)
    assert float((a[0] - (1 - aux["final_T"])).abs().max()) < 1e-5         # alpha = 1 - T
    assert float(a.min()) >= 0 and float(a.max()) <= 1 + 1e-6 and float(d.min()) >= 0
    # colour = blended + T * bg
    st0 = st._replace(bg=torch.zeros(3))
    c0 = oracle.rasterize(sc["means3D"], None, sc["shs"], None, sc["opacities"], sc["scales"],
                          sc["rotations"], None, st0)[0]
    assert float((c - (c0 + aux["final_T"][None] * st.bg[:, None, None])).abs().max()) < 1e-6
    # list order inside every tile: depth ascending, ties by index
    pre, gs, rng = aux["pre"], aux["g_sorted"], aux["ranges"]
    for t in torch.nonzero(rng[:, 1] > rng[:, 0]).flatten().tolist()[:50]:
        ids = gs[rng[t, 0]: rng[t, 1]]
        dd = pre["depth"][ids]
        assert bool((dd[1:] >= dd[:-1]).all())
        same = dd[1:] == dd[:-1]
        assert bool((ids[1:][same] > ids[:-1][same]).all())


def test_depth_ties_resolve_by_index():
    sc = make_scene(P=40, H=32, W=32, seed=13, spread=0.05, scale=0.03)
    sc["means3D"][20:] = sc["means3D"][:20]                    # clones, as densify_and_clone makes
    st = oracle_settings(sc)
    *_, aux = oracle.rasterize(sc["means3D"], None, sc["shs"], None, sc["opacities"], sc["scales"],
                               sc["rotations"], None, st, return_aux=True)
    rng, gs = aux["ranges"], aux["g_sorted"]
    t = int((rng[:, 1] - rng[:, 0]).argmax())
    ids = gs[rng[t, 0]: rng[t, 1]].tolist()
    for i in range(20):
        if i in ids and i + 20 in ids:
            assert ids.index(i) + 1 == ids.index(i + 20)


def test_golden_scene_regenerates():
    """The committed oracle fixture is what today's oracle produces (guards silent drift)."""
    z = np.load(os.path.join(GOLD, "oracle_scene.npz"))
    sc = make_scene(P=160, sh_degree=2, seed=2024, H=48, W=64, spread=0.3)
    for k in ("means3D", "shs", "opacities", "scales", "rotations"):
        assert np.array_equal(sc[k].numpy(), z[f"in_{k}"])
    c, r, d, a = oracle.rasterize(sc["means3D"], None, sc["shs"], None, sc["opacities"], sc["scales"],
                                  sc["rotations"], None, oracle_settings(sc), dtype=torch.float64)
    assert np.abs(c.numpy() - z["color"]).max() < 1e-9 and np.abs(d.numpy() - z["depth"]).max() < 1e-9
    assert np.array_equal(r.numpy(), z["radii"])


def test_streamed_forward_backward_equals_autograd_of_rasterize():
    """oracle.forward_backward (memory bounded by one tile; used for the full-size config-4 parity
    test) is the same function as rasterize() + autograd: identical outputs and gradients."""
    sc = make_scene(P=300, sh_degree=2, seed=11, H=80, W=112, spread=0.3, scale=0.06)
    st = oracle_settings(sc)
    gen = torch.Generator().manual_seed(4)
    gc, gd, ga = (torch.randn(s, generator=gen, dtype=torch.float64) for s in ((3, 80, 112), (1, 80, 112), (1, 80, 112)))
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    ins = {k: sc[k].double().requires_grad_(True) for k in names}
    m2 = torch.zeros(300, 3, dtype=torch.float64, requires_grad=True)
    c, r, d, a, aux = oracle.rasterize(ins["means3D"], m2, ins["shs"], None, ins["opacities"], ins["scales"],
                                       ins["rotations"], None, st, dtype=torch.float64, return_aux=True)
    ((c * gc).sum() + (d * gd).sum() + (a * ga).sum()).backward()
    out = oracle.forward_backward(sc["means3D"], sc["shs"], None, sc["opacities"], sc["scales"], sc["rotations"],
                                  None, st, gc, gd, ga, dtype=torch.float64)
    assert torch.equal(out["radii"], r) and torch.equal(out["n_contrib"], aux["n_contrib"])
    for got, ref in ((out["color"], c), (out["depth"], d), (out["alpha"], a)):
        assert float((got - ref.detach()).abs().max()) < 1e-12
    for k in names:
        ref = ins[k].grad
        assert float((out["grads"][k] - ref).abs().max()) <= 1e-10 * max(1.0, float(ref.abs().max())), k
    assert float((out["grads"]["means2D"] - m2.grad).abs().max()) <= 1e-10 * float(m2.grad.abs().max())
    # the fragile mask flags only a small minority of pixels on a generic scene
    assert out["fragile"].dtype == torch.bool and float(out["fragile"].float().mean()) < 0.02


def test_fork_scale_gradient_switch_drops_exactly_the_modifier_factor(monkeypatch):
    """scale_modifier != 1: the fork's dL/dscale is dL/d(mod * scale) - the modifier's factor is missing (SURVEY A.6 list,
    oracle/gs_oracle.py::FORK_SCALE_GRADIENT) - every other gradient and every output is the same either way."""
    sc = make_scene(P=60, sh_degree=1, seed=5)
    st = oracle_settings(sc, 1.3)
    g = torch.Generator().manual_seed(2)
    w = [torch.randn(s, generator=g, dtype=torch.float64) for s in ((3, 48, 64), (1, 48, 64), (1, 48, 64))]

    def run(fork):
        monkeypatch.setattr(gs_oracle, "FORK_SCALE_GRADIENT", fork)
        ins = {k: sc[k].double().clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
        c, r, d, a = oracle.rasterize(ins["means3D"], None, ins["shs"], None, ins["opacities"], ins["scales"],
                                      ins["rotations"], None, st, dtype=torch.float64)
        gl = torch.autograd.grad((c * w[0]).sum() + (d * w[1]).sum() + (a * w[2]).sum(), list(ins.values()))
        return (c.detach(), d.detach(), a.detach()), dict(zip(ins, gl))
    out_f, g_f = run(True)
    out_t, g_t = run(False)
    for x, y in zip(out_f, out_t):
        assert torch.equal(x, y)
    for k in g_f:
        if k != "scales":
            assert torch.allclose(g_f[k], g_t[k], rtol=1e-12, atol=0)
    assert float(g_t["scales"].abs().max()) > 0
    assert torch.allclose(g_f["scales"] * float(np.float32(1.3)), g_t["scales"], rtol=1e-12, atol=1e-300)   # (the modifier is an fp32 scalar)
