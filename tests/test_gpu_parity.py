"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical
seeded inputs.  Tolerances are BASELINE.json's: |dRGB|, |dalpha| <= 1e-4, depth <= 1e-4 x
max depth, gradients <= 1e-3 x max|oracle gradient| per tensor; radii / tile rects exact."""
import math

import numpy as np
import pytest
import torch

import oracle
from abi_runner import RawCall
from helpers import cov3d_from, make_scene, oracle_settings

pytestmark = pytest.mark.gpu

IMG_TOL = 1e-4
GRAD_TOL = 1e-3


def oracle_forward(scene, dtype=torch.float32, mod=1.0, colors_precomp=None, cov3D=None,
                   grads=None, sh_degree=None):
    st = oracle_settings(scene, mod)
    if sh_degree is not None:
        st = st._replace(sh_degree=sh_degree)
    names = ["means3D", "opacities"]
    ins = {"means3D": scene["means3D"], "opacities": scene["opacities"]}
    if colors_precomp is None:
        ins["shs"] = scene["shs"]
    else:
        ins["colors_precomp"] = colors_precomp
    if cov3D is None:
        ins["scales"], ins["rotations"] = scene["scales"], scene["rotations"]
    else:
        ins["cov3D_precomp"] = cov3D
    ins = {k: v.to(dtype).clone().requires_grad_(grads is not None) for k, v in ins.items()}
    P = scene["means3D"].shape[0]
    m2d = torch.zeros(P, 3, dtype=dtype, requires_grad=grads is not None)
    out = oracle.rasterize(ins["means3D"], m2d, ins.get("shs"), ins.get("colors_precomp"),
                           ins["opacities"], ins.get("scales"), ins.get("rotations"),
                           ins.get("cov3D_precomp"), st, dtype=dtype, return_aux=True)
    color, radii, depth, alpha, aux = out
    g = None
    if grads is not None:
        loss = sum((o * w.to(dtype)).sum() for o, w in zip((color, depth, alpha), grads) if w is not None)
        tens = list(ins.values()) + [m2d]
        gl = torch.autograd.grad(loss, tens, allow_unused=True)
        g = {k: (torch.zeros_like(t) if x is None else x) for k, x, t in zip(list(ins) + ["means2D"], gl, tens)}
    return color.detach(), radii, depth.detach(), alpha.detach(), aux, g


def check_images(rc, ocolor, odepth, oalpha):
    dmax = max(1.0, float(odepth.max()))
    assert torch.isfinite(rc.color).all() and torch.isfinite(rc.depth).all() and torch.isfinite(rc.alpha).all()
    ec = (rc.color.cpu() - ocolor.float()).abs().max().item()
    ed = (rc.depth.cpu() - odepth.float()).abs().max().item()
    ea = (rc.alpha.cpu() - oalpha.float()).abs().max().item()
    assert ec <= IMG_TOL, f"color err {ec}"
    assert ea <= IMG_TOL, f"alpha err {ea}"
    assert ed <= IMG_TOL * dmax, f"depth err {ed}"


def check_grads(got, ref):
    for k, r in ref.items():
        gk = got[k]
        assert gk is not None, k
        assert torch.isfinite(gk).all(), f"non-finite gradient {k}"
        r = r.float().reshape(gk.shape)
        scale = max(r.abs().max().item(), 1e-12)
        err = (gk - r).abs().max().item()
        assert err <= GRAD_TOL * scale, f"grad {k}: err {err:.3e} vs scale {scale:.3e}"
        if r.abs().max() > 0:
            cos = torch.nn.functional.cosine_similarity(gk.double().flatten(), r.double().flatten(), dim=0)
            assert cos > 1 - 1e-5, f"grad {k}: cosine {cos}"


def rand_grads(H, W, seed=1, with_alpha=True):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g),
            torch.randn(1, H, W, generator=g) if with_alpha else None)


# ----------------------------------------------------------------------------- forward

@pytest.mark.parametrize("deg,M", [(0, 1), (1, 4), (2, 9), (3, 16), (0, 16), (2, 16)])
def test_forward_sh_degrees(deg, M):
    sc = make_scene(P=300, sh_degree=deg, M=M, seed=deg + 10 * M, H=64, W=80)
    rc = RawCall(sc)
    assert rc.forward() == 0 and not rc.status[4]
    oc, orad, od, oa, aux, _ = oracle_forward(sc)
    assert torch.equal(rc.radii.cpu(), orad)
    check_images(rc, oc, od, oa)
    assert 0 < rc.status[0] <= int(aux["pre"]["tiles_touched"].sum())     # list entries: upstream's minus the provably empty ones


def test_preprocess_records_bitwise():
    """Per-Gaussian quantities agree with the fp32 oracle bit-for-bit (same op order,
    contraction off) - so ceil()/==/int() decisions can never flake."""
    sc = make_scene(P=2000, sh_degree=3, seed=3, H=128, W=96, spread=0.6)
    rc = RawCall(sc, capacity=1 << 17)
    assert rc.forward() == 0 and not rc.status[4]
    rec = rc.geom_records()
    pre = oracle.preprocess(sc["means3D"], None, sc["shs"], None, sc["opacities"], sc["scales"],
                            sc["rotations"], None, oracle_settings(sc))
    vis = pre["visible"].numpy()
    assert vis.sum() > 1000
    assert np.array_equal(rec["radius"], pre["radii"].numpy())
    for name, ref in (("mx", pre["mean2D"][:, 0]), ("my", pre["mean2D"][:, 1]),
                      ("ca", pre["conic"][:, 0]), ("cb", pre["conic"][:, 1]),
                      ("cc", pre["conic"][:, 2]), ("depth", pre["depth"])):
        a, b = rec[name][vis], ref.numpy()[vis]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name
    col = np.stack([rec["r"], rec["g"], rec["b"]], 1)[vis]
    assert np.abs(col - pre["rgb"].numpy()[vis]).max() <= 1e-6
    # Tile rect: upstream's rect cut down to the tiles the box of the alpha >= 1/255 ellipse can touch (preprocess.hip).
    # It must stay inside upstream's rect and keep every tile that holds a live pixel centre (fp64 ellipse, no margin).
    rect = pre["rect"].numpy()
    lo_x, lo_y = (rec["rect_lo"] & 0xFFFF).astype(np.int64), (rec["rect_lo"] >> 16).astype(np.int64)
    hi_x, hi_y = (rec["rect_hi"] & 0xFFFF).astype(np.int64), (rec["rect_hi"] >> 16).astype(np.int64)
    nonempty = vis & (hi_x > lo_x) & (hi_y > lo_y)
    assert (lo_x[nonempty] >= rect[nonempty, 0]).all() and (lo_y[nonempty] >= rect[nonempty, 1]).all()
    assert (hi_x[nonempty] <= rect[nonempty, 2]).all() and (hi_y[nonempty] <= rect[nonempty, 3]).all()
    m = pre["mean2D"].numpy().astype(np.float64)
    con = pre["conic"].numpy().astype(np.float64)
    op = pre["opacity"].numpy().astype(np.float64).reshape(-1)
    tau = 2 * np.log(np.maximum(255 * op, 1e-30))
    det = con[:, 0] * con[:, 2] - con[:, 1] ** 2
    ex = np.sqrt(np.maximum(tau, 0) * con[:, 2] / det)
    ey = np.sqrt(np.maximum(tau, 0) * con[:, 0] / det)
    need = vis & (255 * op >= 1.0)
    # pixel-centre span of the ellipse's box, clipped to upstream's rect, in tiles
    nx0 = np.maximum(np.ceil((m[:, 0] - ex - 15) / 16), rect[:, 0]); nx1 = np.minimum(np.floor((m[:, 0] + ex) / 16) + 1, rect[:, 2])
    ny0 = np.maximum(np.ceil((m[:, 1] - ey - 15) / 16), rect[:, 1]); ny1 = np.minimum(np.floor((m[:, 1] + ey) / 16) + 1, rect[:, 3])
    need &= (nx1 > nx0) & (ny1 > ny0)
    assert (lo_x[need] <= nx0[need]).all() and (hi_x[need] >= nx1[need]).all()
    assert (lo_y[need] <= ny0[need]).all() and (hi_y[need] >= ny1[need]).all()
    # ... and it is tight: at most one tile of slack per side, only from the safety margins
    assert ((nx0[need] - lo_x[need]) <= 1).all() and ((hi_x[need] - nx1[need]) <= 1).all()
    # entry ids: `offset` = exclusive prefix of the kept tiles inside the Gaussian's 256-chunk (the chunk
    # bases are bump-allocated on the device; the gradient tests prove the ranges tile [0, R))
    tt = np.where(vis, (hi_x - lo_x).clip(0) * (hi_y - lo_y).clip(0), 0)
    for c0 in range(0, len(tt), 256):
        seg = tt[c0:c0 + 256]
        hit = seg > 0                         # culled Gaussians keep offset 0
        assert np.array_equal(rec["offset"][c0:c0 + 256][hit], (np.cumsum(seg) - seg).astype(np.uint32)[hit])
    assert int(tt.sum()) == rc.status[0] <= int(pre["tiles_touched"].numpy().sum())
    assert rc.status[0] < 0.95 * int(pre["tiles_touched"].numpy().sum())          # the cut is worth something


@pytest.mark.parametrize("H,W", [(16, 16), (17, 33), (100, 60), (1, 1)])
def test_forward_ragged_image_sizes(H, W):
    sc = make_scene(P=150, seed=H * 100 + W, H=H, W=W)
    rc = RawCall(sc)
    assert rc.forward() == 0
    oc, orad, od, oa, _, _ = oracle_forward(sc)
    assert torch.equal(rc.radii.cpu(), orad)
    check_images(rc, oc, od, oa)


def test_forward_precomputed_inputs_and_scale_modifier():
    sc = make_scene(P=250, sh_degree=0, seed=5)
    colors = torch.rand(250, 3)
    cov = cov3d_from(sc, mod=0.7)
    rc = RawCall(sc, colors_precomp=colors, cov3D_precomp=cov)
    assert rc.forward() == 0
    oc, orad, od, oa, _, _ = oracle_forward(sc, colors_precomp=colors, cov3D=cov)
    assert torch.equal(rc.radii.cpu(), orad)
    check_images(rc, oc, od, oa)
    rc2 = RawCall(sc, scale_modifier=0.7)
    assert rc2.forward() == 0
    oc2, orad2, od2, oa2, _, _ = oracle_forward(sc, mod=0.7)
    assert torch.equal(rc2.radii.cpu(), orad2)
    check_images(rc2, oc2, od2, oa2)


def test_edge_cases_empty_culled_single():
    sc = make_scene(P=1, seed=1)
    for P in (0, 1):
        s2 = dict(sc)
        for k in ("means3D", "scales", "rotations", "opacities", "shs"):
            s2[k] = sc[k][:P]
        rc = RawCall(s2)
        assert rc.forward() == 0
        oc, orad, od, oa, _, _ = oracle_forward(s2)
        check_images(rc, oc, od, oa)
        assert torch.equal(rc.radii.cpu(), orad)
    # everything behind the camera -> background only, radii 0, R = 0
    s3 = make_scene(P=100, seed=2)
    s3["means3D"] = s3["means3D"] + torch.tensor(s3["cam"].camera_center) * 2.0
    rc = RawCall(s3)
    assert rc.forward() == 0
    assert rc.status[0] == 0 and int(rc.radii.abs().sum()) == 0
    assert torch.allclose(rc.color.cpu(), s3["bg"][:, None, None].expand_as(rc.color.cpu()))
    assert float(rc.alpha.abs().max()) == 0.0 and float(rc.depth.abs().max()) == 0.0
    g = rc.backward(*rand_grads(rc.H, rc.W))
    for k, v in g.items():
        if v is not None:
            assert float(v.abs().max()) == 0.0, k


def test_capacity_overflow_is_reported_and_retry_works():
    sc = make_scene(P=400, seed=7)
    rc = RawCall(sc, capacity=16)
    assert rc.forward() == 0
    assert rc.status[4] != 0 and rc.status[0] > 16          # overflow flagged, R reported
    rc2 = RawCall(sc, capacity=rc.status[0])                 # exact fit
    assert rc2.forward() == 0 and rc2.status[4] == 0
    oc, orad, od, oa, _, _ = oracle_forward(sc)
    check_images(rc2, oc, od, oa)


@pytest.mark.parametrize("n", [700, 1500, 5000, 18000])
def test_long_tile_lists_sort_classes_and_early_termination(n):
    """Many Gaussians stacked over a few tiles: exercises the medium / large / global sort
    classes, multi-bucket state, depth ties (duplicated points) and T < 1e-4 termination."""
    g = torch.Generator().manual_seed(n)
    sc = make_scene(P=n, seed=n, H=32, W=48, spread=0.02, scale=0.01, dist=2.0)
    sc["means3D"][: n // 4] = sc["means3D"][n // 4: 2 * (n // 4)]      # exact depth ties
    sc["opacities"] = 0.02 + 0.5 * torch.rand(n, 1, generator=g)
    rc = RawCall(sc, capacity=max(8 * n, 1 << 16))
    assert rc.forward() == 0 and not rc.status[4]
    oc, orad, od, oa, aux, _ = oracle_forward(sc)
    assert torch.equal(rc.radii.cpu(), orad)
    assert int((aux["ranges"][:, 1] - aux["ranges"][:, 0]).max()) > min(n, 16384) * 0.5
    check_images(rc, oc, od, oa)
    ncon = np.frombuffer(rc.img[: rc.H * rc.W * 4].cpu().numpy().tobytes(), dtype=np.uint32).reshape(rc.H, rc.W)
    assert np.array_equal(ncon, aux["n_contrib"].numpy().astype(np.uint32))


@pytest.mark.parametrize("case,n", [("equal_depths", 1500), ("outlier", 1500), ("equal_depths", 9000)])
def test_sort_with_degenerate_depth_distributions(case, n):
    """The rank sort buckets a tile's keys by depth; a bucket of more than HGS_RANK_BUCKET_MAX (192) keys sends the tile
    through the bitonic network instead.  `equal_depths`: 6 distinct positions x n / 6 copies (exact ties, broken by the
    Gaussian index like upstream's stable radix sort); `outlier`: one far Gaussian stretches the depth range so that
    the body of the list shares a few buckets; n = 9000: the same in the large class (hgs_k_sort_large, 4097..16384 entries).
    Order-sensitive outputs (n_contrib exact, images) against the oracle."""
    g = torch.Generator().manual_seed(17)
    sc = make_scene(P=n, seed=123, H=32, W=48, spread=0.02, scale=0.01, dist=2.0)
    if case == "equal_depths":
        sc["means3D"] = sc["means3D"][torch.arange(n) % 6].clone()          # 6 distinct positions x n / 6 copies
    else:
        # one Gaussian far behind the cloud, on the camera axis (still inside the frustum, large enough to reach every tile)
        cam_c = torch.as_tensor(sc["cam"].camera_center, dtype=torch.float32)
        sc["means3D"][7] = cam_c + (sc["means3D"].mean(0) - cam_c) * 600.0
        sc["scales"][7] = 30.0
    sc["opacities"] = 0.02 + 0.3 * torch.rand(n, 1, generator=g)
    rc = RawCall(sc, capacity=max(8 * n, 1 << 16))
    assert rc.forward() == 0 and not rc.status[4]
    oc, orad, od, oa, aux, _ = oracle_forward(sc)
    assert torch.equal(rc.radii.cpu(), orad)
    assert rc.status[6] > (4096 if n > 4096 else n // 3)               # a list long enough to overflow a bucket (n = 9000: of the large class)
    check_images(rc, oc, od, oa)
    ncon = np.frombuffer(rc.img[: rc.H * rc.W * 4].cpu().numpy().tobytes(), dtype=np.uint32).reshape(rc.H, rc.W)
    assert np.array_equal(ncon, aux["n_contrib"].numpy().astype(np.uint32))
    grads = rand_grads(32, 48, seed=5)
    got = rc.backward(*grads)
    *_, ref = oracle_forward(sc, dtype=torch.float64, grads=grads)
    check_grads(got, ref)


@pytest.mark.parametrize("side", [1936, 2048, 2080])
def test_very_large_images_bin_paths(side):
    """Tile-count regimes of the binning stage: T = 121^2 = 14641 (LDS histograms, scan without the
    on-chip count copy), T = 128^2 = 16384 (largest LDS-histogram case), T = 130^2 = 16900
    (global-atomic fallback: hgs_k_preprocess_fwd_ga / hgs_k_fill_ga, single-workgroup scan).
    Few Gaussians so that the oracle stays fast; forward and backward are compared."""
    sc = make_scene(P=200, sh_degree=1, seed=side, H=side, W=side, spread=0.5, scale=0.002)
    rc = RawCall(sc, capacity=1 << 17)
    assert rc.forward() == 0 and not rc.status[4]
    oc, orad, od, oa, aux, _ = oracle_forward(sc)
    assert torch.equal(rc.radii.cpu(), orad)
    assert 0 < rc.status[0] <= int((aux["ranges"][:, 1] - aux["ranges"][:, 0]).sum())
    check_images(rc, oc, od, oa)
    grads = rand_grads(side, side, seed=3)
    got = rc.backward(*grads)
    *_, ref = oracle_forward(sc, dtype=torch.float64, grads=grads)
    check_grads(got, ref)


# ---------------------------------------------------------------------------- backward

@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_backward_vs_fp64_oracle(deg):
    sc = make_scene(P=220, sh_degree=deg, seed=20 + deg, H=64, W=64)
    rc = RawCall(sc)
    assert rc.forward() == 0
    grads = rand_grads(64, 64, seed=deg)
    got = rc.backward(*grads)
    *_, ref = oracle_forward(sc, dtype=torch.float64, grads=grads)
    check_grads(got, ref)


def test_backward_precomputed_inputs_and_scale_modifier():
    sc = make_scene(P=180, sh_degree=0, seed=31)
    colors = torch.rand(180, 3)
    cov = cov3d_from(sc)
    rc = RawCall(sc, colors_precomp=colors, cov3D_precomp=cov)
    assert rc.forward() == 0
    grads = rand_grads(rc.H, rc.W, seed=3)
    got = rc.backward(*grads)
    *_, ref = oracle_forward(sc, dtype=torch.float64, colors_precomp=colors, cov3D=cov, grads=grads)
    check_grads(got, ref)
    rc2 = RawCall(sc, scale_modifier=1.3)
    assert rc2.forward() == 0
    got2 = rc2.backward(*grads)
    *_, ref2 = oracle_forward(sc, dtype=torch.float64, mod=1.3, grads=grads)
    check_grads(got2, ref2)


def test_backward_multibucket_termination_and_bg():
    """Lists of several buckets with early termination, non-zero background, and only some
    of the three output gradients present."""
    n = 1200
    g = torch.Generator().manual_seed(5)
    sc = make_scene(P=n, seed=77, H=32, W=32, spread=0.03, scale=0.012, bg=(0.9, 0.5, 0.2))
    sc["opacities"] = 0.05 + 0.6 * torch.rand(n, 1, generator=g)
    rc = RawCall(sc)
    assert rc.forward() == 0
    for grads in (rand_grads(32, 32, 9), (rand_grads(32, 32, 10)[0], None, None),
                  (None, rand_grads(32, 32, 11)[1], None)):
        got = rc.backward(*grads)
        *_, ref = oracle_forward(sc, dtype=torch.float64, grads=grads)
        check_grads(got, ref)


def test_backward_frustum_clamp_quirk():
    """Gaussians beyond 1.3 x tan(fov) take the clamped-Jacobian branch."""
    sc = make_scene(P=120, seed=41, H=48, W=48, spread=1.6, scale=0.5, fovy=25.0, dist=1.2)
    rc = RawCall(sc)
    assert rc.forward() == 0
    assert (rc.geom_records()["flags"] != 0).sum() > 3
    grads = rand_grads(48, 48, seed=2)
    got = rc.backward(*grads)
    oc, orad, od, oa, _, ref = oracle_forward(sc, dtype=torch.float64, grads=grads)
    assert torch.equal(rc.radii.cpu(), oracle_forward(sc)[1])
    check_grads(got, ref)


def test_backward_is_deterministic_and_linear():
    sc = make_scene(P=600, sh_degree=1, seed=51, H=64, W=64, spread=0.2)
    rc = RawCall(sc)
    assert rc.forward() == 0
    g1, g2 = rand_grads(64, 64, 1), rand_grads(64, 64, 2)
    a = rc.backward(*g1)
    b = rc.backward(*g1)
    for k in a:
        if a[k] is not None:
            assert torch.equal(a[k], b[k]), f"{k} not bitwise reproducible"
    c = rc.backward(*g2)
    s = rc.backward(*[x + y for x, y in zip(g1, g2)])
    for k in a:
        if a[k] is not None:
            ref = a[k] + c[k]
            assert (s[k] - ref).abs().max() <= 1e-4 * max(1e-9, ref.abs().max()), k


def test_sort_class_hint_violation_is_reported_and_valid_hint_is_exact():
    sc = make_scene(P=3000, seed=9, H=32, W=32, spread=0.02, scale=0.01)
    rc = RawCall(sc, capacity=1 << 17, max_tile_hint=1024)
    assert rc.forward() == 0
    assert rc.status[4] & 2 and rc.status[6] > 1024        # longest list reported
    rc2 = RawCall(sc, capacity=1 << 17, max_tile_hint=rc.status[6])
    assert rc2.forward() == 0 and rc2.status[4] == 0
    oc, orad, od, oa, _, _ = oracle_forward(sc)
    check_images(rc2, oc, od, oa)


def test_long_list_paths_agree_bitwise():
    """The same deep lists (> 1024 entries: several pixel-state segments per cell list, four-records-per-iteration
    forward for the long cells) with and without the caller's longest-list hint: the hint only decides which sort-class
    kernels are launched, so outputs and gradients must agree bit for bit, and both match the oracle."""
    sc = make_scene(P=2600, seed=77, H=32, W=32, spread=0.02, scale=0.01, dist=2.0)
    g = torch.Generator().manual_seed(5)
    sc["opacities"] = 0.01 + 0.05 * torch.rand(2600, 1, generator=g)      # deep lists, late termination
    a = RawCall(sc, capacity=1 << 17, max_tile_hint=0)
    assert a.forward() == 0 and a.status[4] == 0
    longest = a.status[6]
    assert 1024 < longest <= 3072, longest                  # several segments, recompute path allowed
    b = RawCall(sc, capacity=1 << 17, max_tile_hint=longest)
    assert b.forward() == 0 and b.status[4] == 0
    for x, y in ((a.color, b.color), (a.depth, b.depth), (a.alpha, b.alpha)):
        assert torch.equal(x, y)
    oc, orad, od, oa, _, _ = oracle_forward(sc)
    check_images(b, oc, od, oa)
    grads = rand_grads(32, 32, seed=8)
    ga, gb = a.backward(*grads), b.backward(*grads)
    for k in ga:
        if ga[k] is not None:
            assert torch.equal(ga[k], gb[k]), k


def test_backward_without_host_status_matches():
    """hgs_backward(status=NULL): capacity-bounded grid + device-side status."""
    sc = make_scene(P=900, sh_degree=1, seed=61, H=48, W=64, spread=0.1)
    rc = RawCall(sc, capacity=1 << 16)
    assert rc.forward() == 0 and not rc.status[4]
    grads = rand_grads(48, 64, seed=4)
    a = rc.backward(*grads, use_status=True)
    b = rc.backward(*grads, use_status=False)
    for k in a:
        if a[k] is not None:
            assert torch.equal(a[k], b[k]), k


def test_status_via_mapped_pinned_memory_equals_copied_status():
    sc = make_scene(P=500, seed=81, H=48, W=48)
    a = RawCall(sc, mapped=0); assert a.forward() == 0
    b = RawCall(sc, mapped=1); assert b.forward() == 0
    # (word 2 = num_pairs: known only once the sort has run - 0 in the copy taken behind the fill stage, published into a
    #  mapped mirror by the blend forward)
    assert a.status[:2] + a.status[3:] == b.status[:2] + b.status[3:] and a.status[0] > 0
    assert a.status[2] == 0 and b.status[2] > 0


def test_pair_count_is_published_and_sizes_the_backward_scratch():
    """hgs_status.num_pairs: 0 in what the status copy delivers (the sort has not run then), the (entry, cell) pair total
    in a MAPPED host mirror once the blend forward has started.  A scratch of hgs_bwd_scratch_bytes_pairs(R, num_pairs)
    - 48 B per entry + 40 B per pair instead of 16 pairs per entry - gives the same gradients bit for bit and nothing is
    written behind it."""
    sc = make_scene(P=4000, sh_degree=1, seed=11, H=160, W=208, spread=0.5)
    rc0 = RawCall(sc, capacity=1 << 17, mapped=0)
    assert rc0.forward() == 0 and not rc0.status[4]
    assert rc0.status[2] == 0                                   # copied behind the fill stage: not known yet
    rc = RawCall(sc, capacity=1 << 17, mapped=1)
    assert rc.forward() == 0 and not rc.status[4]
    R, pairs = rc.status[0], rc.status[2]
    assert R == rc0.status[0] and 0 < pairs <= 16 * R
    lib = rc.lib
    assert lib.hgs_bwd_scratch_bytes_pairs(R, pairs) < lib.hgs_bwd_scratch_bytes(R)
    assert lib.hgs_bwd_scratch_bytes_pairs(R, 0) == lib.hgs_bwd_scratch_bytes(R)
    gc, gd, ga = rand_grads(160, 208, seed=5)
    g_worst = rc.backward(gc, gd, ga)
    g_pairs = rc.backward(gc, gd, ga, pairs_scratch=True)
    for k, a in g_worst.items():
        if a is not None:
            assert torch.equal(a.view(torch.int32), g_pairs[k].view(torch.int32)), k


def test_understated_pair_count_poisons_the_gradients_and_stays_in_bounds():
    """ABI v14: status->num_pairs DECLARES the pair rows the scratch holds.  A count below what the forward left on the
    device (a stale status slot, a status of another call) must not overrun the scratch: the blend backward writes no
    pair row, the reduction poisons the gradient rows, every gradient that depends on a tile entry is NaN - loud, in bounds."""
    sc = make_scene(P=3000, sh_degree=0, seed=12, H=128, W=160, spread=0.5)
    rc = RawCall(sc, capacity=1 << 16, mapped=1)
    assert rc.forward() == 0 and not rc.status[4]
    R, pairs = rc.status[0], rc.status[2]
    assert R > 0 and pairs > R
    gc, gd, ga = rand_grads(128, 160, seed=6)
    good = rc.backward(gc, gd, ga, pairs_scratch=True)
    assert all(torch.isfinite(v).all() for v in good.values() if v is not None)
    rc.status[2] = pairs // 2                       # understate: the scratch (and its guard region) is sized by it
    bad = rc.backward(gc, gd, ga, pairs_scratch=True)
    vis = rc.radii.cpu() > 0
    assert vis.any() and torch.isnan(bad["means3D"][vis]).any() and torch.isnan(bad["opacities"][vis]).any()
    rc.status[2] = pairs                            # the true count again: same bits as before
    again = rc.backward(gc, gd, ga, pairs_scratch=True)
    for k, a in good.items():
        if a is not None:
            assert torch.equal(a.view(torch.int32), again[k].view(torch.int32)), k
