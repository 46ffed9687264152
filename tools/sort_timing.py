"""Per-phase timestamps of hgs_k_sort_lds (needs a -DHGS_SORT_TIMING variant preloaded)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import bench
from humangaussian_amd import synth
from abi_runner import RawCall
cloud = synth.init_cloud(100000, 0, "mid", seed=0)
cam = bench.camera_for_rank(0)
scene = dict(means3D=cloud.means3D, shs=cloud.shs, opacities=cloud.opacities, scales=cloud.scales,
             rotations=cloud.rotations, bg=torch.zeros(3), cam=cam, sh_degree=0)
C = 1 << 19
rc = RawCall(scene, capacity=C, mapped=0, max_tile_hint=3000)
rc.forward(); rc.forward()
al = lambda n: (n + 255) // 256 * 256
off = al(C * 8) + al(C * 48) + al(((C + 63) // 64) * 6 * 256 * 4)
act = rc.status[1]
tm = rc.bin[off: off + act * 64].cpu().numpy().view(np.uint64).reshape(act, 8).astype(np.int64)
n = tm[:, 4]
d = np.diff(tm[:, :4], axis=1)
print("active tiles", act, "longest", n.max())
for lo, hi in ((1, 64), (65, 256), (257, 1024), (1025, 2048), (2049, 4096)):
    m = (n >= lo) & (n <= hi)
    if m.any():
        print(f"n in [{lo},{hi}]: tiles {m.sum():4d}  load {d[m,0].mean():8.0f}  sort {d[m,1].mean():8.0f}  gather {d[m,2].mean():8.0f}  total {d[m].sum(1).mean():8.0f} ticks (max {d[m].sum(1).max()})")
span = tm[:, 3].max() - tm[:, 0].min()
print("kernel span (same-XCD clocks only comparable):", span)
