"""CPU restatement (numpy, fp32 where the device uses fp32) of the grid search of csrc/knn.hip - grid from the bounding box,
cell of a point, shells of cells, the `reach` stop rule - checked against the brute force on random, surface, planar,
collinear, duplicated and outlier-stretched clouds: the stop rule must never end a search before the true three nearest
neighbours have been seen.  (The HIP kernels themselves run in tests/test_gpu_api_contract.py against a k-d tree and against
the brute-force kernel, bit for bit.)"""
import numpy as np
import pytest

F = np.float32
CELL_MAX = 4096


def grid_setup(pts, nc_max):
    P = len(pts)
    lo = pts.min(0).astype(F)
    ext = (pts.max(0).astype(F) - lo).astype(F)
    emax = F(ext.max())
    h, g = F(1.0), np.ones(3, np.int64)
    if emax > 0 and P > 8:
        floor_ext = F(emax * F(1e-3))
        vol = F(np.prod(np.maximum(ext, floor_ext).astype(F)))
        h = F(np.cbrt(F(2.0) * vol / F(P)))
        for _ in range(64):
            c = np.floor(ext / h).astype(F) + F(1.0)
            g = np.clip(c, 1, 4096).astype(np.int64)
            if int(np.prod(g)) <= nc_max and np.all(ext / h < 4095.0):
                break
            h = F(h * F(1.26))
    return lo, h, F(1.0) / h, g


def cell_of(p, lo, inv_h, g):
    c = np.floor(((p - lo).astype(F) * inv_h).astype(F))
    return np.minimum(np.maximum(c, 0), (g - 1).astype(F)).astype(np.int64)


def grid_knn(pts):
    pts = pts.astype(F)
    P = len(pts)
    nc_max = min(max(64, 2 * P), 1 << 22)
    lo, h, inv_h, g = grid_setup(pts, nc_max)
    cells = np.stack([cell_of(p, lo, inv_h, g) for p in pts])
    key = (cells[:, 2] * g[1] + cells[:, 1]) * g[0] + cells[:, 0]
    buckets = {}
    for i, k in enumerate(key):
        buckets.setdefault(int(k), []).append(i)
    if max(len(v) for v in buckets.values()) > CELL_MAX:
        return None
    out = np.zeros(P, F)
    visited_total = 0
    for i in range(P):
        me, (cx, cy, cz) = pts[i], cells[i]
        best = [F(3.4e38)] * 3
        for r in range(0, int(g.max()) + 1):
            z0, z1, y0, y1 = max(cz - r, 0), min(cz + r, g[2] - 1), max(cy - r, 0), min(cy + r, g[1] - 1)
            x0, x1 = max(cx - r, 0), min(cx + r, g[0] - 1)
            for z in range(z0, z1 + 1):
                for y in range(y0, y1 + 1):
                    if abs(z - cz) == r or abs(y - cy) == r:
                        xs = range(x0, x1 + 1)
                    else:
                        xs = [x for x in (cx - r, cx + r) if 0 <= x <= g[0] - 1]
                        if r == 0:
                            xs = xs[:1]
                    for x in xs:
                        for j in buckets.get(int((z * g[1] + y) * g[0] + x), ()):
                            visited_total += 1
                            if j == i:
                                continue
                            d = pts[j] - me
                            best = sorted(best + [F(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])])[:3]
            if x0 == 0 and y0 == 0 and z0 == 0 and x1 == g[0] - 1 and y1 == g[1] - 1 and z1 == g[2] - 1:
                break
            reach = F(3.4e38)
            for a, c in enumerate((cx, cy, cz)):
                rel = F(me[a] - lo[a])                      # grid-relative, like the cell assignment (knn.hip)
                if c - r > 0:
                    reach = min(reach, F(rel - F(F(c - r) * h)))
                if c + r < g[a] - 1:
                    reach = min(reach, F(F(F(c + r + 1) * h) - rel))
            reach = max(F(reach - F(1e-3) * h), F(0))
            if best[2] <= reach * reach:
                break
        out[i] = (best[0] + best[1] + best[2]) / F(3.0)
    return out, visited_total / P, g


def brute(pts):
    p = pts.astype(F)
    d = ((p[:, None, :] - p[None, :, :]) ** 2).astype(F)
    d = (d[..., 0] + d[..., 1] + d[..., 2]).astype(F)
    np.fill_diagonal(d, np.inf)
    s = np.sort(d, 1)[:, :3].astype(F)
    return ((s[:, 0] + s[:, 1]) + s[:, 2]) / F(3.0)


def clouds():
    rng = np.random.default_rng(0)
    yield "gauss", rng.normal(0, 0.4, (700, 3))
    th, u = rng.uniform(0, 2 * np.pi, 900), rng.uniform(-1, 1, 900)
    yield "sphere_surface", np.stack([np.sqrt(1 - u * u) * np.cos(th), np.sqrt(1 - u * u) * np.sin(th), u], 1) * 0.5
    yield "planar", np.concatenate([rng.uniform(-1, 1, (600, 2)), np.zeros((600, 1))], 1)
    yield "collinear", np.stack([rng.uniform(-3, 3, 300), np.zeros(300), np.zeros(300)], 1)
    dup = rng.normal(0, 0.2, (400, 3))
    dup[5] = dup[6]
    dup[50] = dup[51] = dup[52] = dup[53]
    yield "duplicates", dup
    out = rng.normal(0, 0.01, (500, 3))
    out[0] = (50.0, 0, 0)
    out[1] = (-50.0, 0.3, 0)
    yield "outliers", out                      # the box is stretched: almost everything falls into a few cells
    yield "tiny", rng.normal(0, 1, (5, 3))
    yield "all_equal", np.ones((40, 3)) * 0.25
    clus = np.concatenate([rng.normal((0, 0, 0), 0.01, (300, 3)), rng.normal((1, 1, 1), 0.3, (300, 3))])
    yield "two_scales", clus
    # clouds translated far from the world origin: ulp(|coord|) is no longer small against 1e-3 h
    far = rng.normal(0, 0.4, (600, 3)).astype(np.float32)
    yield "translated_1e3", far + np.float32(1.0e3)
    yield "translated_1e4", far + np.array([1.0e4, -1.0e4, 3.0e3], np.float32)


@pytest.mark.parametrize("name,pts", list(clouds()), ids=[n for n, _ in clouds()])
def test_grid_search_is_exact(name, pts):
    res = grid_knn(pts)
    assert res is not None
    got, visited, g = res
    want = brute(pts)
    assert np.array_equal(got, want), (name, np.abs(got - want).max(), g)


def test_grid_search_looks_at_a_small_neighbourhood_on_a_surface_cloud():
    rng = np.random.default_rng(1)
    th, u = rng.uniform(0, 2 * np.pi, 3000), rng.uniform(-1, 1, 3000)
    pts = np.stack([np.sqrt(1 - u * u) * np.cos(th), np.sqrt(1 - u * u) * np.sin(th), u], 1) * 0.5
    got, visited, g = grid_knn(pts)
    assert np.array_equal(got, brute(pts))
    assert visited < 0.05 * len(pts), visited          # candidates per point: a few dozen, not P
