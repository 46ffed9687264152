"""Per-phase device timestamps of hgs_k_render_bwd.  Needs a -DHGS_BWD_TIMING build of the library:
  bash tools/mkvariant.sh timing "-DHGS_BWD_TIMING"
  HGS_LIB=variants/timing/libhgs_rast.so python tools/bwd_timing.py          (on the GPU box)
Per work item (one wave): t0 start | t1 prologue (item, records, n_contrib) done | accumulated over its batches:
evaluation, stage -> B operand reads, finish (previous batch's sums), MFMA issue | t4 quadrants done | t6 rows written.
Phases in core-clock cycles (readcyclecounter: one counter per XCD, differences only); start / end of the
item also on wall_clock64 (100 MHz, common to the XCDs) for the timeline."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import torch
from abi_runner import RawCall
from humangaussian_amd import synth

cloud = synth.init_cloud(100000, 0, "mid", seed=0)
cam = synth.orbit_camera(10.0, 30.0, 1.75, 55.0, 1024, 1024)
scene = dict(means3D=cloud.means3D, shs=cloud.shs, opacities=cloud.opacities, scales=cloud.scales,
             rotations=cloud.rotations, bg=torch.zeros(3), cam=cam, sh_degree=0)
rc = RawCall(scene, capacity=1 << 19, mapped=0)
assert rc.forward() == 0
groups = rc.status[3]
print("R", rc.status[0], "work items", groups)
g = torch.Generator().manual_seed(1)
gc, gd, ga = ((torch.randn(s, generator=g) * 1e-3).cuda() for s in ((3, 1024, 1024), (1, 1024, 1024), (1, 1024, 1024)))
rc.backward(gc, gd, ga)          # warm
torch.cuda.synchronize()
rc.bin[: groups * 80].zero_()
rc.backward(gc, gd, ga)
torch.cuda.synchronize()
tm = rc.bin[: groups * 80].cpu().numpy().view(np.uint64).reshape(groups, 10).astype(np.int64)
idx = np.nonzero(tm[:, 0])[0]
tm = tm[idx]
print("items with timing", tm.shape[0])
dur = tm[:, 6] - tm[:, 0]                       # core-clock cycles (per-XCD counter: differences only)
print("item duration, cycles: mean %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f" %
      (dur.mean(), *np.percentile(dur, [10, 50, 90]), dur.max()))
w0 = tm[:, 8].min()
start, end = (tm[:, 8] - w0) * 0.01, (tm[:, 9] - w0) * 0.01          # wall clock, us
span = end.max()
wdur = end - start
print("kernel span %.1f us; item duration us: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f; sum %.1f us per SIMD" %
      (span, wdur.mean(), *np.percentile(wdur, [10, 50, 90]), wdur.max(), wdur.sum() / 1024))
edges = np.linspace(0, span, 17)
occ = [float(np.clip(np.minimum(end, b) - np.maximum(start, a), 0, None).sum() / (b - a) / 1024) for a, b in zip(edges[:-1], edges[1:])]
print("resident waves per SIMD over time (16 bins of %.1f us):" % (span / 16), " ".join("%.2f" % o for o in occ))
last = np.argsort(-end)[:10]
print("last items to finish (index in the table, start, duration us):",
      [(int(idx[i]), round(float(start[i]), 1), round(float(wdur[i]), 1)) for i in last])
for lo, hi in ((0, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0)):
    m = (idx >= lo * groups) & (idx < hi * groups)
    print("  table quarter %.2f-%.2f: start p50 %.1f, duration mean %.1f, end max %.1f" %
          (lo, hi, np.percentile(start[m], 50), wdur[m].mean(), end[m].max()))
full = dur > np.percentile(dur, 25)
f = lambda a: float((a[full] / dur[full]).mean())
pro, quad, rows = tm[:, 1] - tm[:, 0], tm[:, 4] - tm[:, 1], tm[:, 6] - tm[:, 4]
ev, stg, fin, mf = tm[:, 2], tm[:, 3], tm[:, 5], tm[:, 7]
print("fractions of an item's duration: prologue %.3f | quadrant sweeps %.3f | rows %.3f" % (f(pro), f(quad), f(rows)))
print("inside the sweeps: evaluation %.3f | stage->B %.3f | finish %.3f | MFMA issue %.3f | per-quadrant setup (rest) %.3f"
      % (f(ev), f(stg), f(fin), f(mf), f(quad - ev - stg - fin - mf)))
print("start-time histogram", np.histogram(start, bins=edges)[0].tolist())
print("end-time histogram", np.histogram(end, bins=edges)[0].tolist())
