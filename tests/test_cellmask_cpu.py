"""The per-cell cull mask the sort kernel computes (csrc/cellmask.h, the SAME source compiled for the host):
a cleared bit must never hide a live (pixel, Gaussian) pair of the blend (power <= 0 and
alpha >= 1/255), and the mask should be tight (close to the exact ellipse-vs-rectangle test)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("cellmask") / "cellmask_host.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-Wno-unknown-pragmas",
                           os.path.join(ROOT, "tests", "cellmask_host.cpp"), "-o", so])
    L = ctypes.CDLL(so)
    L.hgs_cell_mask_host.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 9
    L.hgs_tile_hit_host.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 9
    return L


def _masks(lib, mx, my, ca, cb, cc, op, x0, y0):
    arrs = [np.ascontiguousarray(a, np.float32) for a in (mx, my, ca, cb, cc, op, x0, y0)]
    out = np.zeros(len(arrs[0]), np.uint32)
    lib.hgs_cell_mask_host(len(out), *[a.ctypes.data for a in arrs], out.ctypes.data)
    return out


def _tile_hits(lib, mx, my, ca, cb, cc, op, x0, y0):
    arrs = [np.ascontiguousarray(a, np.float32) for a in (mx, my, ca, cb, cc, op, x0, y0)]
    out = np.zeros(len(arrs[0]), np.uint32)
    lib.hgs_tile_hit_host(len(out), *[a.ctypes.data for a in arrs], out.ctypes.data)
    return out.astype(bool)


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


def _random_entries(n, seed):
    rng = np.random.default_rng(seed)
    # covariance = R diag(s1^2, s2^2) R^T + 0.3 I (the rasterizer's low-pass), conic = its inverse
    s1 = np.exp(rng.uniform(np.log(0.05), np.log(40.0), n))
    s2 = s1 * np.exp(rng.uniform(np.log(0.02), 0.0, n))
    th = rng.uniform(0, np.pi, n)
    c, s = np.cos(th), np.sin(th)
    a = c * c * s1 ** 2 + s * s * s2 ** 2 + 0.3
    b = c * s * (s1 ** 2 - s2 ** 2)
    d = s * s * s1 ** 2 + c * c * s2 ** 2 + 0.3
    det = a * d - b * b
    ca, cb, cc = d / det, -b / det, a / det
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


def test_tile_hit_is_a_superset_of_the_cell_mask_and_of_every_live_pixel(lib):
    """The binning stage drops a (Gaussian, tile) pair when hgs_tile_hit fails: it must hold whenever a cell bit is
    set (the blend kernels walk cell lists built from the kept entries) and whenever any pixel of the tile is live."""
    kept = live_any = n = 0
    for seed in range(4):
        e = _random_entries(60000, 10 + seed)
        hit = _tile_hits(lib, *e)
        m = _masks(lib, *e)
        live_cells, _ = _live_cells(*e)
        assert not ((m != 0) & ~hit).any()
        assert not ((live_cells != 0) & ~hit).any()
        kept += int(hit.sum()); live_any += int((live_cells != 0).sum()); n += len(hit)
    assert kept <= 1.15 * live_any + 0.01 * n, (kept, live_any, n)      # tight: few tiles kept without a live pixel
