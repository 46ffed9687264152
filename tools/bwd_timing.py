"""Per-phase device timestamps of hgs_k_render_bwd.  Needs a -DHGS_BWD_TIMING build of the library:
  bash tools/mkvariant.sh timing "-DHGS_BWD_TIMING"
  HGS_LIB=variants/timing/libhgs_rast.so python tools/bwd_timing.py          (on the GPU box)
Per work item (one wave): t0 start | t1 prologue (item, records, n_contrib) done | accumulated over its batches:
evaluation, stage -> B operand reads, finish (previous batch's sums), MFMA issue | t4 quadrants done | t6 rows written.
s_memtime / readcyclecounter ticks at 100 MHz on this part."""
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
rc.bin[: groups * 64].zero_()
rc.backward(gc, gd, ga)
torch.cuda.synchronize()
tm = rc.bin[: groups * 64].cpu().numpy().view(np.uint64).reshape(groups, 8).astype(np.int64)
tm = tm[tm[:, 0] != 0]
print("items with timing", tm.shape[0])
t0 = tm[:, 0].min()
start, end = tm[:, 0] - t0, tm[:, 6] - t0
dur = end - start
span = end.max()
tick_us = 0.01
print("kernel span %.1f us; item duration us: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" %
      (span * tick_us, dur.mean() * tick_us, *(np.percentile(dur, [10, 50, 90]) * tick_us), dur.max() * tick_us))
print("item start us: p50 %.1f p75 %.1f p90 %.1f max %.1f" % tuple(np.percentile(start, [50, 75, 90, 100]) * tick_us))
full = dur > np.percentile(dur, 25)
f = lambda a: float((a[full] / dur[full]).mean())
pro, quad, rows = tm[:, 1] - tm[:, 0], tm[:, 4] - tm[:, 1], tm[:, 6] - tm[:, 4]
ev, stg, fin, mf = tm[:, 2], tm[:, 3], tm[:, 5], tm[:, 7]
print("fractions of an item's duration: prologue %.3f | quadrant sweeps %.3f | rows %.3f" % (f(pro), f(quad), f(rows)))
print("inside the sweeps: evaluation %.3f | stage->B %.3f | finish %.3f | MFMA issue %.3f | per-quadrant setup (rest) %.3f"
      % (f(ev), f(stg), f(fin), f(mf), f(quad - ev - stg - fin - mf)))
hist, _ = np.histogram(start, bins=10, range=(0, span))
print("start-time histogram", hist.tolist())
hist, _ = np.histogram(end, bins=10, range=(0, span))
print("end-time histogram", hist.tolist())
