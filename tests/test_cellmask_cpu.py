"""The per-cell cull mask the sort kernel computes (csrc/cellmask.h, the SAME source compiled for the host):
a cleared bit must never hide a live (pixel, Gaussian) pair of the blend (power <= 0 and
alpha >= 1/255), and the mask should be tight (close to the exact ellipse-vs-rectangle test)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path_factory, name, flags=()):
    so = str(tmp_path_factory.mktemp("cellmask") / f"cellmask_host_{name}.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-Wno-unknown-pragmas", *flags,
                           os.path.join(ROOT, "tests", "cellmask_host.cpp"), "-o", so])
    L = ctypes.CDLL(so)
    L.hgs_cell_mask_host.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 9
    L.hgs_alpha_rect_host.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 7
    return L


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    return _build(tmp_path_factory, "exact")


# The device build of the mask takes sqrt / reciprocal / log from the raw 1-ulp instructions (cellmask.h: HGS_CM_*): host
# builds whose three results are pushed 3 ulp (3.6e-7 relative) down / up, in every combination of directions, must pass
# the same brute-force check - the margins of the test have to carry any such error, whatever its sign.
_SKEWS = [(a, b, c) for a in (-1, 1) for b in (-1, 1) for c in (-1, 1)]


@pytest.fixture(scope="module", params=_SKEWS, ids=lambda t: "sqrt%+d_rcp%+d_ln%+d" % t)
def skewed_lib(request, tmp_path_factory):
    f = lambda sgn: "(1.0f%s3.6e-7f)" % ("+" if sgn > 0 else "-")  # noqa: E731
    a, b, c = request.param
    return _build(tmp_path_factory, "skew%d%d%d" % (a + 1, b + 1, c + 1),
                  ["-DHGS_CM_SKEW_SQRT=" + f(a), "-DHGS_CM_SKEW_RCP=" + f(b), "-DHGS_CM_SKEW_LN=" + f(c)])


def _masks(lib, mx, my, ca, cb, cc, op, x0, y0):
    arrs = [np.ascontiguousarray(a, np.float32) for a in (mx, my, ca, cb, cc, op, x0, y0)]
    out = np.zeros(len(arrs[0]), np.uint32)
    lib.hgs_cell_mask_host(len(out), *[a.ctypes.data for a in arrs], out.ctypes.data)
    return out


def _live_cells(mx, my, ca, cb, cc, op, x0, y0):
    """bit c set iff some pixel of cell c passes the blend's test, in the kernel's fp32 arithmetic
    (exp2-folded conic, hgs_eval_alpha) - evaluated with a small slack towards 'live'."""
    f = np.float32
    lx = np.arange(256) % 16
    ly = np.arange(256) // 16
    px = (x0[:, None] + lx[None]).astype(f)
    py = (y0[:, None] + ly[None]).astype(f)
    log2e = f(1.4426950408889634)
    qa, qb, qc = (f(-0.5) * ca * log2e).astype(f), (-cb * log2e).astype(f), (f(-0.5) * cc * log2e).astype(f)
    dx, dy = (mx[:, None] - px).astype(f), (my[:, None] - py).astype(f)
    m2 = (qa[:, None] * dx + qb[:, None] * dy).astype(f)
    p2 = (dx * m2 + (qc[:, None] * dy) * dy).astype(f)
    alpha = np.minimum(f(0.99), op[:, None] * np.exp2(p2.astype(np.float64)))
    live = (p2 <= 1e-6) & (alpha >= (1.0 / 255.0) * (1 - 1e-5))
    cell = (ly // 4) * 4 + lx // 4
    out = np.zeros(len(mx), np.uint32)
    for c in range(16):
        out |= (live[:, cell == c].any(1).astype(np.uint32) << c)
    return out, live


def _random_entries(n, seed, s1_max=40.0, aniso_min=0.02):
    rng = np.random.default_rng(seed)
    # covariance = R diag(s1^2, s2^2) R^T + 0.3 I (the rasterizer's low-pass), conic = its inverse
    s1 = np.exp(rng.uniform(np.log(0.05), np.log(s1_max), n))
    s2 = s1 * np.exp(rng.uniform(np.log(aniso_min), 0.0, n))
    th = rng.uniform(0, np.pi, n)
    c, s = np.cos(th), np.sin(th)
    a = c * c * s1 ** 2 + s * s * s2 ** 2 + 0.3
    b = c * s * (s1 ** 2 - s2 ** 2)
    d = s * s * s1 ** 2 + c * c * s2 ** 2 + 0.3
    det = a * d - b * b
    ca, cb, cc = d / det, -b / det, a / det          # (float64 here; the kernel sees their fp32 roundings)
    op = np.where(rng.uniform(size=n) < 0.1, rng.uniform(0.0, 0.01, n), rng.uniform(0.004, 1.0, n))
    x0 = 16.0 * rng.integers(0, 64, n)
    y0 = 16.0 * rng.integers(0, 64, n)
    r = 3.0 * np.sqrt(np.maximum(a, d)) + 4
    mx = x0 + 8 + rng.uniform(-1, 1, n) * (8 + r)
    my = y0 + 8 + rng.uniform(-1, 1, n) * (8 + r)
    return [v.astype(np.float32) for v in (mx, my, ca, cb, cc, op, x0, y0)]


def test_cell_mask_never_hides_a_live_pair_and_is_tight(lib):
    tot_kept = tot_live = 0
    for seed in range(4):
        e = _random_entries(60000, seed)
        m = _masks(lib, *e)
        live_cells, _ = _live_cells(*e)
        hidden = live_cells & ~m
        assert not hidden.any(), (int(np.count_nonzero(hidden)), [v[np.nonzero(hidden)[0][:3]] for v in e])
        tot_kept += sum(int(np.count_nonzero(m & (1 << c))) for c in range(16))
        tot_live += sum(int(np.count_nonzero(live_cells & (1 << c))) for c in range(16))
    # cells with a live PIXEL CENTRE are a subset of cells meeting the ellipse; the mask may keep a few more
    assert tot_kept <= 1.35 * tot_live, (tot_kept, tot_live)


def test_cell_mask_edge_cases(lib):
    f = np.float32
    one = lambda v: np.array([v], f)  # noqa: E731
    # opacity below 1/255: nothing can blend
    assert _masks(lib, one(8), one(8), one(1), one(0), one(1), one(0.003), one(0), one(0))[0] == 0
    # degenerate conic: never culled
    assert _masks(lib, one(8), one(8), one(1), one(1), one(1), one(0.5), one(0), one(0))[0] == 0xFFFF
    # a huge Gaussian covers every cell
    assert _masks(lib, one(100), one(-50), one(1e-4), one(0), one(1e-4), one(0.9), one(0), one(0))[0] == 0xFFFF
    # a tiny opaque Gaussian in the centre of cell 5 (pixels 4..7 x 4..7) touches only that cell
    assert _masks(lib, one(5.5), one(5.5), one(3.3), one(0), one(3.3), one(0.9), one(0), one(0))[0] == 1 << 5
    # far away: nothing
    assert _masks(lib, one(500), one(500), one(1), one(0), one(1), one(0.9), one(0), one(0))[0] == 0


def test_cell_mask_with_huge_elongated_gaussians(lib):
    """Zoomed-in cameras: sigma_major up to 3000 px at anisotropies down to 1e-3.  ca cc - cb^2 cancels in plain fp32
    there (cleared bits hid pixels with alpha up to 0.0085); the mask uses a compensated determinant and
    refuses to cull when even that is within rounding of zero."""
    for seed in range(3):
        e = _random_entries(40000, 100 + seed, s1_max=3000.0, aniso_min=1e-3)
        m = _masks(lib, *e)
        live_cells, _ = _live_cells(*e)
        hidden = live_cells & ~m
        assert not hidden.any(), (int(np.count_nonzero(hidden)), [v[np.nonzero(hidden)[0][:3]] for v in e])


def test_cell_mask_carries_three_ulp_of_error_in_its_sqrt_rcp_and_log(skewed_lib):
    """What the device's raw v_sqrt_f32 / v_rcp_f32 / v_log_f32 may do to the mask (each within 1 ulp, either sign): the
    brute-force check on ordinary and on huge elongated Gaussians with those results skewed by 3 ulp."""
    for seed, (s1_max, aniso) in ((300, (40.0, 0.02)), (301, (3000.0, 1e-3))):
        e = _random_entries(40000, seed, s1_max=s1_max, aniso_min=aniso)
        m = _masks(skewed_lib, *e)
        live_cells, _ = _live_cells(*e)
        hidden = live_cells & ~m
        assert not hidden.any(), (int(np.count_nonzero(hidden)), [v[np.nonzero(hidden)[0][:3]] for v in e])


def test_alpha_rect_keeps_every_tile_with_a_live_pixel(lib):
    """The cut that decides which (Gaussian, tile) pairs get list entries at all (hgs_alpha_rect, used by
    hgs_k_preprocess_fwd): starting from a generous rect around the Gaussian, every tile that holds a live pixel must
    survive - checked tile by tile with the kernel's own alpha arithmetic, for ordinary and for huge elongated Gaussians."""
    f = np.float32
    for seed, (s1_max, aniso) in enumerate(((40.0, 0.02), (3000.0, 1e-3))):
        mx, my, ca, cb, cc, op, x0, y0 = _random_entries(20000, 200 + seed, s1_max, aniso)
        n = len(mx)
        # the tile (x0, y0) is the one under test; hand the cut a rect of 5 x 5 tiles around it
        t0x, t0y = (x0 / 16).astype(np.int32), (y0 / 16).astype(np.int32)
        rect = np.stack([t0x - 2, t0y - 2, t0x + 3, t0y + 3], 1).astype(np.int32).copy()
        arrs = [np.ascontiguousarray(a, f) for a in (mx, my, ca, cb, cc, op)]
        lib.hgs_alpha_rect_host(n, *[a.ctypes.data for a in arrs], rect.ctypes.data)
        kept = (rect[:, 0] <= t0x) & (t0x < rect[:, 2]) & (rect[:, 1] <= t0y) & (t0y < rect[:, 3])
        live_cells, _ = _live_cells(mx, my, ca, cb, cc, op, x0, y0)
        lost = (live_cells != 0) & ~kept
        assert not lost.any(), (int(lost.sum()), [v[np.nonzero(lost)[0][:3]] for v in (mx, my, ca, cb, cc, op, x0, y0)])
        if s1_max < 100:
            assert kept.sum() <= 1.6 * (live_cells != 0).sum() + 0.02 * n          # and it cuts: few tiles kept without a live pixel
