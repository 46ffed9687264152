"""Device timeline of the forward's sort and blend kernels (which waves run when, on which XCD).
Needs a -DHGS_TIMELINE build of the library:
  bash tools/mkvariant.sh timeline "-DHGS_TIMELINE -DHGS_BWD_TIMING"
  HGS_LIB=variants/timeline/libhgs_rast.so python tools/timeline.py [views]        (on the GPU box)
Every wave leaves (start, end, HW_ID | XCC_ID << 32, tag) in a static device table; wall_clock64 ticks at
100 MHz.  tag: sort = the tile's list length; blend = list length, or 1 << 32 | segment index for the
segments of long lists."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import torch
from abi_runner import RawCall
from humangaussian_amd import synth

SLOTS = 1 << 17
TICK_US = 0.01
KERNELS = {3: "sort_lds", 4: "render_fwd"}

cloud = synth.init_cloud(100000, 0, "mid", seed=0)
cam = synth.orbit_camera(10.0, 30.0, 1.75, 55.0, 1024, 1024)
scene = dict(means3D=cloud.means3D, shs=cloud.shs, opacities=cloud.opacities, scales=cloud.scales,
             rotations=cloud.rotations, bg=torch.zeros(3), cam=cam, sh_degree=0)
rc = RawCall(scene, capacity=1 << 19, mapped=0)
assert rc.forward() == 0
rc.max_tile_hint = int(rc.status[6] * 1.5 + 64)          # what the torch binding passes once its estimate has settled
for _ in range(5):
    assert rc.forward() == 0
torch.cuda.synchronize()
print("R", rc.status[0], "longest list", rc.status[6], "segments", rc.status[7], "bwd items", rc.status[3])
lib = rc.lib
lib.hgs_debug_timeline_read.restype = ctypes.c_int
lib.hgs_debug_timeline_read.argtypes = [ctypes.c_int, ctypes.c_void_p]
buf = np.zeros((SLOTS, 4), dtype=np.uint64)


def read(kid):
    r = lib.hgs_debug_timeline_read(kid, buf.ctypes.data_as(ctypes.c_void_p))
    assert r == 0, r
    return buf.copy().astype(np.int64)


for kid in list(KERNELS) + [5]:
    read(kid)                          # clear what the warm-up runs left
assert rc.forward() == 0
torch.cuda.synchronize()

for kid, name in KERNELS.items():
    tm = read(kid)
    slot = np.nonzero(tm[:, 0])[0]
    tm = tm[slot]
    t0 = tm[:, 0].min()
    st, en = (tm[:, 0] - t0) * TICK_US, (tm[:, 1] - t0) * TICK_US
    dur = en - st
    tag = tm[:, 3]
    xcc = (tm[:, 2] >> 32) & 0xf
    hw = tm[:, 2] & 0xffffffff
    span = en.max()
    print(f"\n== {name}: {len(slot)} waves, span {span:.1f} us; wave duration us: mean {dur.mean():.2f} "
          f"p50 {np.percentile(dur, 50):.2f} p90 {np.percentile(dur, 90):.2f} p99 {np.percentile(dur, 99):.2f} "
          f"max {dur.max():.2f}; sum {dur.sum() / 1024:.1f} us per SIMD")
    nb = 16
    edges = np.linspace(0, span, nb + 1)
    occ = []
    for a, b in zip(edges[:-1], edges[1:]):
        occ.append(float(np.clip(np.minimum(en, b) - np.maximum(st, a), 0, None).sum() / (b - a) / 1024))
    print("resident waves per SIMD over time (%d bins of %.1f us):" % (nb, span / nb), " ".join("%.2f" % o for o in occ))
    print("starts per bin:", np.histogram(st, bins=edges)[0].tolist())
    print("ends per bin:  ", np.histogram(en, bins=edges)[0].tolist())
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print(f"  xcc {x}: {int(m.sum())} waves, busy {dur[m].sum() / 128:.1f} us/SIMD, last end {en[m].max():.1f} us")
    order = np.argsort(-en)[:12]
    print("last waves to finish (slot, start, dur, tag):",
          [(int(slot[i]), round(float(st[i]), 1), round(float(dur[i]), 1), hex(int(tag[i]))) for i in order])
    order = np.argsort(-dur)[:12]
    print("longest waves (slot, start, dur, tag):",
          [(int(slot[i]), round(float(st[i]), 1), round(float(dur[i]), 1), hex(int(tag[i]))) for i in order])
    short = tag < (1 << 32)
    for lo, hi in ((0, 1), (1, 64), (64, 256), (256, 512), (512, 1024), (1024, 1 << 31)):
        m = short & (tag >= lo) & (tag < hi)
        if m.any():
            print(f"  list length [{lo},{hi}): {int(m.sum())} waves, dur mean {dur[m].mean():.2f} max {dur[m].max():.2f}, "
                  f"start p50 {np.percentile(st[m], 50):.1f}, end max {en[m].max():.1f}")
    m = ~short
    if m.any():
        print(f"  segments of long lists: {int(m.sum())} waves, dur mean {dur[m].mean():.2f} max {dur[m].max():.2f}, "
              f"end max {en[m].max():.1f}")
        for k in sorted(set((tag[m] & 0xffff).tolist())):
            mk = m & ((tag & 0xffff) == k)
            print(f"    segment {k}: {int(mk.sum())} waves, dur mean {dur[mk].mean():.2f} max {dur[mk].max():.2f}")
    # is the dispatch in index order?  (the longest-first schedules rely on it for speed, never for results)
    inv = int((np.diff(st[np.argsort(slot)]) < -1.0).sum())
    print(f"  waves starting > 1 us before a lower-indexed wave: {inv}")

# phases of every tile's sort (wave 0 of its workgroup): key load | network | record gather
ph = read(5)
ph = ph[ph[:, 3] != 0]
n, E = ph[:, 3] & 0xffffffff, ph[:, 3] >> 32
print("\n== sort phases per tile (us): load keys | sorting network | gather records")
for lo, hi in ((1, 64), (64, 256), (256, 512), (512, 1024), (1024, 2048), (2048, 4097)):
    m = (n >= lo) & (n < hi)
    if m.any():
        print(f"  n in [{lo},{hi}): {int(m.sum())} tiles, keys/thread {sorted(set(E[m].tolist()))}: "
              f"load {ph[m, 0].mean() * TICK_US:.2f} | network {ph[m, 1].mean() * TICK_US:.2f} (max {ph[m, 1].max() * TICK_US:.2f}) | "
              f"gather {ph[m, 2].mean() * TICK_US:.2f} (max {ph[m, 2].max() * TICK_US:.2f})")
