"""How much backward work lies beyond the deepest pixel of a quadrant?  (entries past
max n_contrib of the quadrant's 64 pixels can be dropped from that quadrant's list)"""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import bench, oracle
from oracle import gs_oracle as go
from humangaussian_amd import synth
torch.set_num_threads(16)
P, deg = 100000, 0
cloud = synth.init_cloud(P, deg, "mid", seed=0)
cam = synth.orbit_camera(10.0, 30.0, 1.75, 55.0, 1024, 1024)
st = oracle.OracleSettings(1024, 1024, math.tan(cam.FoVx * .5), math.tan(cam.FoVy * .5), torch.zeros(3), 1.0,
                           cam.world_view_transform, cam.full_proj_transform, deg, cam.camera_center, False, False)
with torch.no_grad():
    out = oracle.rasterize(cloud.means3D, None, cloud.shs, None, cloud.opacities, cloud.scales, cloud.rotations, None, st,
                           return_aux=True)
aux = out[-1]
nc = aux["n_contrib"].numpy().reshape(1024, 1024)
pre = aux["pre"]
g_sorted, t_sorted, ranges = go.bin_and_sort(pre)
ranges = ranges.numpy()
gx = pre["grid"][0]
tot_entries = 0; kept_q = 0; tot_q = 0; tile_max_kept = 0
for t in range(ranges.shape[0]):
    n = ranges[t, 1] - ranges[t, 0]
    if n == 0: continue
    ty, tx = divmod(t, gx)
    blk = nc[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16]
    tot_entries += n
    tile_max_kept += min(n, ((blk.max() + 63) // 64) * 64)      # current rule: whole buckets below tile max
    for q in range(4):
        qm = blk[(q >> 1) * 8:(q >> 1) * 8 + 8, (q & 1) * 8:(q & 1) * 8 + 8].max()
        tot_q += n; kept_q += min(n, qm)
print("entries", tot_entries, " processed today (bucket rule on tile max):", tile_max_kept / tot_entries,
      " with per-quadrant trim:", kept_q / tot_q)
