"""Work model of the blend kernels on the bench scene (CPU, numpy): iterations per scheme, chain lengths,
work behind terminated pixels.  Usage: python tools/work_stats.py [P] [variant] [azim]"""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import oracle
from oracle import gs_oracle as go
from humangaussian_amd import synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
variant = sys.argv[2] if len(sys.argv) > 2 else "mid"
azim = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
cloud = synth.init_cloud(P, 0, variant, seed=0)
cam = synth.orbit_camera(10.0, azim, 1.75, 55.0, 1024, 1024)
st = oracle.OracleSettings(1024, 1024, math.tan(cam.FoVx * .5), math.tan(cam.FoVy * .5), torch.zeros(3), 1.0,
                           cam.world_view_transform, cam.full_proj_transform, 0, cam.camera_center, False, False)
with torch.no_grad():
    pre = go.preprocess(cloud.means3D, None, cloud.shs, None, cloud.opacities, cloud.scales, cloud.rotations, None, st)
    g_sorted, t_sorted, ranges = go.bin_and_sort(pre)
R = g_sorted.numel()
gx = pre["grid"][0]
M2 = pre["mean2D"].numpy().astype(np.float64); CON = pre["conic"].numpy().astype(np.float64); OP = pre["opacity"].numpy().astype(np.float64)
gs = g_sorted.numpy(); rg = ranges.numpy()
lx = np.arange(256) % 16; ly = np.arange(256) // 16
cell_of_px = (ly // 4) * 4 + (lx // 4)            # 16 cells, row-major 4x4
quad_of_cell = np.array([(c // 4 // 2) * 2 + (c % 4) // 2 for c in range(16)])
quad_of_px = (ly // 8) * 2 + lx // 8

def cell_masks(m, ca, cb, cc, op, x0t, y0t, cw, ch):
    tau = 2 * np.log(np.maximum(255 * op, 1.0)); vis = 255 * op >= 1.0
    def qf(pxx, pyy):
        ddx = pxx - m[:, 0]; ddy = pyy - m[:, 1]
        return ca * ddx * ddx + 2 * cb * ddx * ddy + cc * ddy * ddy
    out = []
    for cy in range(16 // ch):
        for cx in range(16 // cw):
            x0 = x0t + cx * cw; y0 = y0t + cy * ch; x1 = x0 + cw - 1; y1 = y0 + ch - 1
            best = qf(np.clip(m[:, 0], x0, x1), np.clip(m[:, 1], y0, y1))
            for xe in (x0, x1):
                best = np.minimum(best, qf(xe, np.clip(m[:, 1] - (cb / cc) * (xe - m[:, 0]), y0, y1)))
            for ye in (y0, y1):
                best = np.minimum(best, qf(np.clip(m[:, 0] - (cb / ca) * (ye - m[:, 1]), x0, x1), ye))
            out.append(vis & (best <= tau))
    return np.stack(out, 1)    # (n, ncell)

tot = dict(n=0, n_tilekept=0, live=0, live_term=0, it_cur=0, it_cur_trim=0, cellpairs=0, cellpairs_trim=0)
for B in (64, 128, 256, 10**9):
    tot[f"it_rows_B{B}"] = 0
tot["it_rows_seg256_notrim"] = 0
chains_cur, chains_rows, tiles_n = [], [], []
cell_len, cell_len_trim = [], []
for t in np.nonzero(rg[:, 1] > rg[:, 0])[0]:
    s, e = rg[t]; gi = gs[s:e]; n = e - s
    m = M2[gi]; ca, cb, cc = CON[gi, 0], CON[gi, 1], CON[gi, 2]; op = OP[gi]
    x0t = (t % gx) * 16.0; y0t = (t // gx) * 16.0
    px = x0t + lx; py = y0t + ly
    dx = m[:, 0:1] - px[None]; dy = m[:, 1:2] - py[None]
    power = -0.5 * (ca[:, None] * dx * dx + cc[:, None] * dy * dy) - cb[:, None] * dx * dy
    alpha = np.minimum(0.99, op[:, None] * np.exp(power))
    live = (power <= 0) & (alpha >= 1 / 255.)
    a = np.where(live, alpha, 0.0)
    Tincl = np.cumprod(1 - a, 0)
    stop = live & (Tincl < 1e-4)
    stopped = np.cumsum(stop, 0) > 0
    liveT = live & ~stopped
    pos = np.arange(1, n + 1)[:, None]
    ncontrib = np.where(liveT, pos, 0).max(0)            # (256,)
    cm = cell_masks(m, ca, cb, cc, op, x0t, y0t, 4, 4)   # (n,16)
    qm = cell_masks(m, ca, cb, cc, op, x0t, y0t, 8, 8)   # (n,4)
    tot["n"] += n; tot["n_tilekept"] += cm.any(1).sum(); tot["live"] += live.sum(); tot["live_term"] += liveT.sum()
    # current scheme: per quadrant kept records, trimmed at quadrant's deepest pixel
    ch_cur = 0
    for q in range(4):
        mx = ncontrib[quad_of_px == q].max()
        k = qm[:, q].sum(); kt = qm[:mx, q].sum()
        tot["it_cur"] += k; tot["it_cur_trim"] += kt; ch_cur = max(ch_cur, kt)
    chains_cur.append(ch_cur)
    # rows scheme
    cmax = np.array([ncontrib[cell_of_px == c].max() for c in range(16)])
    cmt = cm & (np.arange(n)[:, None] < cmax[None, :])
    tot["cellpairs"] += cm.sum(); tot["cellpairs_trim"] += cmt.sum()
    cell_len += list(cm.sum(0)); cell_len_trim += list(cmt.sum(0))
    ch_rows = 0
    for q in range(4):
        cells = np.nonzero(quad_of_cell == q)[0]
        sub = cmt[:, cells]
        for B in (64, 128, 256, 10**9):
            if B > n: it = sub.sum(0).max()
            else:
                nb = (n + B - 1) // B
                pad = np.zeros((nb * B, 4), bool); pad[:n] = sub
                it = pad.reshape(nb, B, 4).sum(1).max(1).sum()
            tot[f"it_rows_B{B}"] += it
            if B == 64: ch_rows = max(ch_rows, it)
        subn = cm[:, cells]
        nb = (n + 255) // 256; pad = np.zeros((nb * 256, 4), bool); pad[:n] = subn
        tot["it_rows_seg256_notrim"] += pad.reshape(nb, 256, 4).sum(1).max(1).sum()
    chains_rows.append(ch_rows); tiles_n.append(n)
print("R", R, {k: int(v) for k, v in tot.items()})
n = tot["n"]
print("per entry: tile-kept %.3f  live px %.1f  live before termination %.1f" % (tot["n_tilekept"] / n, tot["live"] / n, tot["live_term"] / n))
print("iterations per entry: current %.3f (trimmed %.3f); 4-row B64 %.3f B128 %.3f B256 %.3f Binf %.3f; ideal cellpairs/4 %.3f (trim %.3f); seg256 no-trim %.3f" % (
    tot["it_cur"] / n, tot["it_cur_trim"] / n, tot["it_rows_B64"] / n, tot["it_rows_B128"] / n, tot["it_rows_B256"] / n,
    tot[f"it_rows_B{10**9}"] / n, tot["cellpairs"] / 4 / n, tot["cellpairs_trim"] / 4 / n, tot["it_rows_seg256_notrim"] / n))
cc_ = np.array(chains_cur); cr = np.array(chains_rows); tn = np.array(tiles_n)
print("tiles", len(tn), "list n: mean %.0f p50 %d p90 %d max %d" % (tn.mean(), np.median(tn), np.percentile(tn, 90), tn.max()))
print("longest wave chain (iterations): current max %d p90 %d ; rows max %d p90 %d" % (cc_.max(), np.percentile(cc_, 90), cr.max(), np.percentile(cr, 90)))

cl = np.array(cell_len); clt = np.array(cell_len_trim)
for name, x in (("cell list length", cl), ("trimmed", clt)):
    x = x[x > 0]
    print(name, "cells", len(x), "mean %.0f p50 %d p90 %d p99 %d max %d" % (x.mean(), np.median(x), np.percentile(x, 90), np.percentile(x, 99), x.max()))
    for thr in (64, 128, 192, 256, 384):
        print("   > %d: %d cells, %.1f%% of pairs; excess beyond thr %.1f%%" % (thr, (x > thr).sum(), 100 * x[x > thr].sum() / x.sum(), 100 * (x[x > thr] - thr).sum() / x.sum()))
