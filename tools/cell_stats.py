"""Lane utilisation of the blend kernels as a function of the cull granularity (bench scene, CPU).
For cells of cw x ch pixels inside the 16x16 tile: fraction of (entry, cell) pairs kept by the exact
ellipse-vs-rectangle test, live pixels (alpha >= 1/255, power <= 0) per kept pair."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import oracle
from oracle import gs_oracle as go
from humangaussian_amd import synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
deg = 0
cloud = synth.init_cloud(P, deg, sys.argv[2] if len(sys.argv) > 2 else "mid", seed=0)
cam = synth.orbit_camera(10.0, 30.0, 1.75, 55.0, 1024, 1024)
st = oracle.OracleSettings(1024, 1024, math.tan(cam.FoVx * .5), math.tan(cam.FoVy * .5), torch.zeros(3), 1.0,
                           cam.world_view_transform, cam.full_proj_transform, deg, cam.camera_center, False, False)
with torch.no_grad():
    pre = go.preprocess(cloud.means3D, None, cloud.shs, None, cloud.opacities, cloud.scales, cloud.rotations, None, st)
    g_sorted, t_sorted, ranges = go.bin_and_sort(pre)
R = g_sorted.numel(); print("R", R)
rng = np.random.default_rng(0)
sel = rng.choice(R, size=min(R, 40000), replace=False)
g = g_sorted.numpy()[sel]; t = t_sorted.numpy()[sel]
m = pre["mean2D"].numpy()[g].astype(np.float64); con = pre["conic"].numpy()[g].astype(np.float64); op = pre["opacity"].numpy()[g].astype(np.float64)
gx = pre["grid"][0]
tx = (t % gx) * 16.0; ty = (t // gx) * 16.0
ca, cb, cc = con[:, 0], con[:, 1], con[:, 2]
tau = 2 * np.log(np.maximum(255 * op, 1.0))
vis = (255 * op >= 1.0)
# truth per pixel of the tile
px = tx[:, None] + (np.arange(256) % 16)[None, :]; py = ty[:, None] + (np.arange(256) // 16)[None, :]
dx = m[:, 0:1] - px; dy = m[:, 1:2] - py
power = -0.5 * (ca[:, None] * dx * dx + cc[:, None] * dy * dy) - cb[:, None] * dx * dy
alpha = np.minimum(0.99, op[:, None] * np.exp(power))
live = (power <= 0) & (alpha >= 1 / 255.)
print("live pixels per entry: %.1f" % live.sum(1).mean())
def qf(pxx, pyy):
    ddx = pxx - m[:, 0]; ddy = pyy - m[:, 1]
    return ca * ddx * ddx + 2 * cb * ddx * ddy + cc * ddy * ddy
for cw, ch in ((16, 16), (8, 8), (8, 4), (4, 4), (4, 2), (2, 2)):
    kept = 0; lanes = 0; ncell = (16 // cw) * (16 // ch)
    keptmat = []
    for cy in range(16 // ch):
        for cx in range(16 // cw):
            x0 = tx + cx * cw; y0 = ty + cy * ch; x1 = x0 + cw - 1; y1 = y0 + ch - 1
            cxx = np.clip(m[:, 0], x0, x1); cyy = np.clip(m[:, 1], y0, y1)
            best = qf(cxx, cyy)
            for xe in (x0, x1):
                ys = np.clip(m[:, 1] - (cb / cc) * (xe - m[:, 0]), y0, y1); best = np.minimum(best, qf(xe, ys))
            for ye in (y0, y1):
                xs = np.clip(m[:, 0] - (cb / ca) * (ye - m[:, 1]), x0, x1); best = np.minimum(best, qf(xs, ye))
            inside = (m[:, 0] >= x0) & (m[:, 0] <= x1) & (m[:, 1] >= y0) & (m[:, 1] <= y1)
            best = np.where(inside, 0.0, best)
            exact = vis & (best <= tau)
            keptmat.append(exact)
            kept += exact.sum()
    kp = kept / len(sel)
    print(f"cell {cw}x{ch}: kept cells per entry {kp:.2f} of {ncell} ({kp/ncell:.3f}); live px per kept cell {live.sum()/kept:.1f} of {cw*ch} = {live.sum()/kept/(cw*ch):.3f};"
          f" lane-slots per entry {kp*cw*ch:.0f}")
