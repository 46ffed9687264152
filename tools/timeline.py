"""Device timelines of the cell-row kernels.  Needs a -DHGS_TIMELINE build of the library:
  bash tools/mkvariant.sh timeline "-DHGS_TIMELINE"
  HGS_LIB=variants/timeline/libhgs_rast.so LD_PRELOAD=variants/timeline/libhgs_rast.so python tools/timeline.py   (GPU box)
sort: phases per tile (ranks | records + masks | cell tables | range allocation | cell lists);
forward: per wave start / end; backward: per group of four work items: wall start / end and cycles per phase."""
import ctypes
import faulthandler
import os
import sys

faulthandler.dump_traceback_later(int(os.environ.get('HGS_TL_WATCHDOG', '45')), exit=True)

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import torch
from abi_runner import RawCall
from humangaussian_amd import synth

SLOTS = 1 << 17
TICK_US = 0.01
cloud = synth.init_cloud(100000, 0, "mid", seed=0)
cam = synth.orbit_camera(10.0, 30.0, 1.75, 55.0, 1024, 1024)
scene = dict(means3D=cloud.means3D, shs=cloud.shs, opacities=cloud.opacities, scales=cloud.scales,
             rotations=cloud.rotations, bg=torch.zeros(3), cam=cam, sh_degree=0)
rc = RawCall(scene, capacity=1 << 19, mapped=0)
assert rc.forward() == 0
rc.max_tile_hint = int(rc.status[6] * 1.5 + 64)
for _ in range(5):
    assert rc.forward() == 0
torch.cuda.synchronize()
print("R", rc.status[0], "longest list", rc.status[6])
lib = rc.lib
lib.hgs_debug_timeline_read.restype = ctypes.c_int
lib.hgs_debug_timeline_read.argtypes = [ctypes.c_int, ctypes.c_void_p]
buf = np.zeros((SLOTS, 4), dtype=np.uint64)


def read(kid):
    r = lib.hgs_debug_timeline_read(kid, buf.ctypes.data_as(ctypes.c_void_p))
    assert r == 0, r
    return buf.copy().astype(np.int64)


for kid in (2, 3, 4, 5):
    read(kid)
assert rc.forward() == 0
torch.cuda.synchronize()


def simd_balance(key, dur, en, what):
    """Per PHYSICAL SIMD (XCC id + SE / SH / CU / SIMD fields of HW_ID): how many waves / groups it ran, their summed
    time and when its last one ended - the placement the work tables assume (heavy + light mix per SIMD) made visible."""
    uniq, inv = np.unique(key, return_inverse=True)
    cnt = np.bincount(inv)
    load = np.bincount(inv, weights=dur)
    last = np.zeros(len(uniq))
    np.maximum.at(last, inv, en)
    pc = lambda x: " / ".join("%.1f" % np.percentile(x, q) for q in (0, 10, 50, 90, 100))
    print(f"  physical SIMDs seen: {len(uniq)}; {what} per SIMD min/p10/p50/p90/max: {pc(cnt)}; summed time per SIMD (us): {pc(load)}; "
          f"last end per SIMD (us): {pc(last)}")


def waves(kid, name, items=False):
    tm = read(kid)
    slot = np.nonzero(tm[:, 0])[0]
    tm = tm[slot]
    t0 = tm[:, 0].min()
    st, en = (tm[:, 0] - t0) * TICK_US, (tm[:, 1] - t0) * TICK_US
    dur = en - st
    tag = tm[:, 3]
    span = en.max()
    print(f"\n== {name}: {len(slot)} waves, span {span:.1f} us; wave duration us: mean {dur.mean():.2f} p50 {np.percentile(dur, 50):.2f} "
          f"p90 {np.percentile(dur, 90):.2f} p99 {np.percentile(dur, 99):.2f} max {dur.max():.2f}; sum {dur.sum() / 1024:.1f} us per SIMD")
    nb = 16
    edges = np.linspace(0, span, nb + 1)
    occ = [float(np.clip(np.minimum(en, b) - np.maximum(st, a), 0, None).sum() / (b - a) / 1024) for a, b in zip(edges[:-1], edges[1:])]
    print("resident waves per SIMD over time (%d bins of %.1f us):" % (nb, span / nb), " ".join("%.2f" % o for o in occ))
    simd_balance(((tm[:, 2] >> 32) & 0xf) << 16 | (tm[:, 2] & 0xff30), dur, en, "waves")
    if kid == 3:     # which sort workgroups share a CU (4 waves per workgroup; CU = XCC id + the SE / SH / CU fields of HW_ID)
        cu3 = (((tm[:, 2] >> 32) & 0xf) << 16) | (tm[:, 2] & 0xff00)
        blk3 = slot // 4
        mates = {}
        for kx in np.unique(cu3):
            bs = sorted(set(int(x) for x in blk3[cu3 == kx]))
            for bb in bs:
                mates[bb] = bs
        print("  sort workgroups sharing a CU with workgroups 0..7:", [mates.get(bb) for bb in range(8)])
        print("  workgroups per CU min/max:", min(len(v) for v in mates.values()), max(len(v) for v in mates.values()), "; CUs seen:", len(np.unique(cu3)))
    order = np.argsort(-en)[:8]
    print("last waves to finish (slot, start, dur, tag):", [(int(slot[i]), round(float(st[i]), 1), round(float(dur[i]), 1), int(tag[i])) for i in order])
    if items:
        wpb = int(os.environ.get("HGS_TL_FWD_WAVES", "4"))
        xcc = (tm[:, 2] >> 32) & 0xf
        blk = slot // wpb
        print("  workgroup -> die: fraction with XCC_ID == workgroup % 8:", float((xcc == blk % 8).mean()), "; XCC_ID of workgroups 0..15:", [int(xcc[blk == b][0]) if (blk == b).any() else -1 for b in range(16)])
        simd_id = (tm[:, 2] >> 4) & 3
        print("  SIMD id of waves 0.. of the first workgroups:", [[int(simd_id[slot == b * wpb + k][0]) if (slot == b * wpb + k).any() else -1 for k in range(wpb)] for b in range(3)])
    if items:       # persistent waves: tag = longest work item (10 ns ticks << 40 | items << 28 | item index << 4 | class + 1)
        imax, icnt, iidx, icls = (tag >> 40) * TICK_US, (tag >> 28) & 0xfff, (tag >> 4) & 0xffffff, (tag & 0xf) - 1
        print("  work items per wave min/p50/max: %d / %d / %d (total %d); longest item of a wave us: p50 %.1f p90 %.1f max %.1f" % (
            icnt.min(), np.percentile(icnt, 50), icnt.max(), icnt.sum(), np.percentile(imax, 50), np.percentile(imax, 90), imax.max()))
        top = np.argsort(-imax)[:12]
        print("  longest items (us, item index in the die's order, long class or -1 = group of short cells, wave slot, wave start, wave end):",
              [(round(float(imax[i]), 1), int(iidx[i]), int(icls[i]), int(slot[i]), round(float(st[i]), 1), round(float(en[i]), 1)) for i in top])
        # SIMD time a wave consumed, assuming its SIMD's resident waves share the issue slots evenly: with end times
        # e1 <= e2 <= ... of the n waves of a SIMD, wave k used sum_{i<=k} (e_i - e_{i-1}) / (n - i + 1)
        key = ((tm[:, 2] >> 32) & 0xf) << 16 | (tm[:, 2] & 0xff30)
        use = np.zeros(len(en))
        for kx in np.unique(key):
            ix = np.nonzero(key == kx)[0]
            ix = ix[np.argsort(en[ix])]
            prev, acc = 0.0, 0.0
            for r, i in enumerate(ix):
                acc += (en[i] - prev) / (len(ix) - r)
                prev = en[i]
                use[i] = acc
        one = icnt == 1
        print("  SIMD time per work item (waves with exactly one item), by class and position in the die's order:")
        for c in sorted(set(icls[one].tolist())):
            m = one & (icls == c)
            qs = np.percentile(iidx[m], [0, 25, 50, 75, 100])
            print(f"    class {c}: {int(m.sum())} items, index {qs[0]:.0f}..{qs[4]:.0f}, SIMD us mean {use[m].mean():.2f} p10 {np.percentile(use[m], 10):.2f} p90 {np.percentile(use[m], 90):.2f} max {use[m].max():.2f}")
        if one.any():
            edges = np.linspace(0, iidx[one].max() + 1, 17)
            print("    by index (16 bins): " + " ".join("%.1f" % (use[one & (iidx >= a) & (iidx < b)].mean() if (one & (iidx >= a) & (iidx < b)).any() else 0) for a, b in zip(edges[:-1], edges[1:])))
        print(f"    per SIMD: total us min/p10/p50/p90/max: " + " / ".join("%.1f" % np.percentile(np.bincount(np.unique(key, return_inverse=True)[1], weights=use), q) for q in (0, 10, 50, 90, 100)))
        return
    for lo, hi in ((0, 1), (1, 64), (64, 256), (256, 512), (512, 1024), (1024, 1 << 31)):
        m = (tag >= lo) & (tag < hi)
        if m.any():
            print(f"  list length [{lo},{hi}): {int(m.sum())} waves, dur mean {dur[m].mean():.2f} max {dur[m].max():.2f}, start p50 {np.percentile(st[m], 50):.1f}, end max {en[m].max():.1f}")


waves(3, "sort_lds")
waves(4, "render_fwd", items=True)
ph = read(5)
gp = read(2)
ok = ph[:, 3] != 0
n = ph[:, 3] & 0xffffffff
t0s = ph[ok, 0].min()
print("\n== rank sort per tile (us; wave 0 of the tile's workgroup): total || ranks | records | tables | allocation | cell lists; "
      "start after the first tile; tiles that took the network fallback")
for lo, hi in ((1, 64), (64, 256), (256, 512), (512, 1024), (1024, 2048), (2048, 4097)):
    m = ok & (n >= lo) & (n < hi)
    if m.any():
        tot = (ph[m, 1] - ph[m, 0]) * TICK_US
        g = gp[m, 0]
        sub = [((g >> (16 * i)) & 0xffff).mean() * TICK_US for i in range(4)] + [(gp[m, 1] & 0xffff).mean() * TICK_US]
        print(f"  n in [{lo},{hi}): {int(m.sum())} tiles: total {tot.mean():.2f} (max {tot.max():.2f}) || " + " | ".join("%.2f" % x for x in sub)
              + f" ; start p50 {np.percentile((ph[m, 0] - t0s) * TICK_US, 50):.1f} max {((ph[m, 0] - t0s) * TICK_US).max():.1f}; end max {((ph[m, 1] - t0s) * TICK_US).max():.1f}; fallback {int(ph[m, 2].sum())}")
        rk = [((gp[m, 2] >> (16 * i)) & 0xffff).mean() * TICK_US for i in range(4)]
        rr = [((gp[m, 3] >> (16 * i)) & 0xffff).mean() * TICK_US for i in range(4)]
        print("      ranking: range %.2f | histogram %.2f | scan %.2f | scatter %.2f | probes %.2f ;  record rounds end at (from the ranks): %s"
              % (rk[0], rk[1], rk[2], rk[3], sub[0] - sum(rk), " ".join("%.2f" % x for x in rr)))

# ---- backward
print('backward ...', flush=True)
g = torch.Generator().manual_seed(1)
gc, gd, ga = ((torch.randn(s, generator=g) * 1e-3).cuda() for s in ((3, 1024, 1024), (1, 1024, 1024), (1, 1024, 1024)))
rc.backward(gc, gd, ga)
torch.cuda.synchronize()
NG = 1 << 15
rc.bin[: NG * 32].zero_()
rc.backward(gc, gd, ga)
torch.cuda.synchronize()
tm = rc.bin[: NG * 32].cpu().numpy().view(np.uint64).reshape(NG, 4)
idx = np.nonzero(tm[:, 3] >> np.uint64(63))[0]
tm = tm[idx].astype(np.int64)
w0 = tm[:, 0].min()
st, en = (tm[:, 0] - w0) * TICK_US, (tm[:, 1] - w0) * TICK_US
dur = en - st
nb = tm[:, 2] & 0xffffffff
chain = (tm[:, 2] >> 32) * TICK_US      # group start -> its first batch's records in registers
print(f"\n== render_bwd: {len(idx)} groups, span {en.max():.1f} us; group duration us: mean {dur.mean():.2f} p10 {np.percentile(dur, 10):.2f} "
      f"p50 {np.percentile(dur, 50):.2f} p90 {np.percentile(dur, 90):.2f} max {dur.max():.2f}; sum {dur.sum() / 1024:.1f} us per SIMD; batches {int(nb.sum())}")
span = en.max()
edges = np.linspace(0, span, 17)
occ = [float(np.clip(np.minimum(en, b) - np.maximum(st, a), 0, None).sum() / (b - a) / 1024) for a, b in zip(edges[:-1], edges[1:])]
print("groups in flight per SIMD over time:", " ".join("%.2f" % o for o in occ))
simd_balance((tm[:, 3] >> 8) & 0xfffff, dur, en, "groups")
live = (tm[:, 3] >> 32) & 0xff             # row-batches with work (a row of a batch whose pixels still contribute)
print(f"  row-batches with work: {int(live.sum())} of {int(4 * nb.sum())} issued ({live.sum() / (4.0 * nb.sum()):.3f}); groups without any: {int((live == 0).sum())}")
for k in range(int(nb.max()), 0, -1):
    m = nb == k
    if m.any():
        print(f"  groups of {k} batches: {int(m.sum())}, dur mean {dur[m].mean():.2f} us (p10 {np.percentile(dur[m], 10):.2f} p90 {np.percentile(dur[m], 90):.2f}), start p10 {np.percentile(st[m], 10):.1f} p50 {np.percentile(st[m], 50):.1f} p90 {np.percentile(st[m], 90):.1f}, end p50 {np.percentile(en[m], 50):.1f} max {en[m].max():.1f}; start chain p50 {np.percentile(chain[m], 50):.2f} p90 {np.percentile(chain[m], 90):.2f}; live row-batches {live[m].mean():.1f} of {4 * k}")
