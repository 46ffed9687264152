"""How tight is the quadrant cull mask?  For the bench scene: per (entry, quadrant) pair compare
(a) the bounding-box test the sort kernel uses, (b) an exact ellipse-vs-rectangle test, (c) ground
truth: any pixel of the quadrant with alpha >= 1/255 (and power <= 0)."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import bench, oracle
from oracle import gs_oracle as go
from humangaussian_amd import synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cloud = synth.init_cloud(P, deg, "mid", seed=0)
cam = synth.orbit_camera(10.0, 30.0, 1.75, 55.0, 1024, 1024)
st = oracle.OracleSettings(1024, 1024, math.tan(cam.FoVx * .5), math.tan(cam.FoVy * .5), torch.zeros(3), 1.0,
                           cam.world_view_transform, cam.full_proj_transform, deg, cam.camera_center, False, False)
with torch.no_grad():
    pre = go.preprocess(cloud.means3D, None, cloud.shs, None, cloud.opacities, cloud.scales, cloud.rotations, None, st)
    g_sorted, t_sorted, ranges = go.bin_and_sort(pre)
R = g_sorted.numel(); print("R", R)
rng = np.random.default_rng(0)
sel = rng.choice(R, size=min(R, 60000), replace=False)
g = g_sorted.numpy()[sel]; t = t_sorted.numpy()[sel]
m = pre["mean2D"].numpy()[g].astype(np.float64); con = pre["conic"].numpy()[g].astype(np.float64); op = pre["opacity"].numpy()[g].astype(np.float64)
gx = pre["grid"][0]
tx = (t % gx) * 16.0; ty = (t // gx) * 16.0
ca, cb, cc = con[:, 0], con[:, 1], con[:, 2]
tau = 2 * np.log(np.maximum(255 * op, 1.0))
det = ca * cc - cb * cb
ex = np.sqrt(tau * cc / det); ey = np.sqrt(tau * ca / det)
tot_b = tot_e = tot_t = 0; act_lanes = 0
for q in range(4):
    x0 = tx + (q & 1) * 8; y0 = ty + (q >> 1) * 8; x1 = x0 + 7; y1 = y0 + 7
    vis = (255 * op >= 1.0)
    bbox = vis & (m[:, 0] + ex >= x0) & (m[:, 0] - ex <= x1) & (m[:, 1] + ey >= y0) & (m[:, 1] - ey <= y1)
    # exact: min of quadratic form over the rectangle
    def qf(px, py):
        dx = px - m[:, 0]; dy = py - m[:, 1]
        return ca * dx * dx + 2 * cb * dx * dy + cc * dy * dy
    cxx = np.clip(m[:, 0], x0, x1); cyy = np.clip(m[:, 1], y0, y1)
    best = qf(cxx, cyy)
    for xe in (x0, x1):
        ys = np.clip(m[:, 1] - (cb / cc) * (xe - m[:, 0]), y0, y1)
        best = np.minimum(best, qf(xe, ys))
    for ye in (y0, y1):
        xs = np.clip(m[:, 0] - (cb / ca) * (ye - m[:, 1]), x0, x1)
        best = np.minimum(best, qf(xs, ye))
    inside = (m[:, 0] >= x0) & (m[:, 0] <= x1) & (m[:, 1] >= y0) & (m[:, 1] <= y1)
    best = np.where(inside, 0.0, best)
    exact = vis & (best <= tau)
    # truth: any pixel with alpha >= 1/255
    px = x0[:, None] + (np.arange(64) % 8)[None, :]; py = y0[:, None] + (np.arange(64) // 8)[None, :]
    dx = m[:, 0:1] - px; dy = m[:, 1:2] - py
    power = -0.5 * (ca[:, None] * dx * dx + cc[:, None] * dy * dy) - cb[:, None] * dx * dy
    alpha = np.minimum(0.99, op[:, None] * np.exp(power))
    live = (power <= 0) & (alpha >= 1 / 255.)
    truth = live.any(1)
    tot_b += bbox.sum(); tot_e += exact.sum(); tot_t += truth.sum(); act_lanes += live.sum()
    assert not (truth & ~exact).any(), "exact test culls a live pair!"
n = 4 * len(sel)
print(f"pairs {n}: bbox {tot_b/n:.3f}  exact-ellipse {tot_e/n:.3f}  truth(any live pixel) {tot_t/n:.3f}")
print(f"live lanes per bbox-kept pair: {act_lanes/tot_b:.1f} of 64; per exact-kept pair {act_lanes/tot_e:.1f}")
