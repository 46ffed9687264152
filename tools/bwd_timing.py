"""Per-phase timestamps of hgs_k_render_bwd (needs a -DHGS_BWD_TIMING variant preloaded):
phases 0 start | 1 prologue loads done | 2 compaction done | 3 basis+barrier | 4 loop | 5 barrier | 6 rows."""
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
rc = RawCall(scene, capacity=1 << 19, mapped=0)
rc.forward()
groups = rc.status[3]
print("R", rc.status[0], "groups", groups)
g = torch.Generator().manual_seed(1)
gc = (torch.randn(3, 1024, 1024, generator=g) * 1e-3).cuda(); gd = (torch.randn(1, 1024, 1024, generator=g) * 1e-3).cuda(); ga = (torch.randn(1, 1024, 1024, generator=g) * 1e-3).cuda()
rc.backward(gc, gd, ga)          # warm
torch.cuda.synchronize()
rc.bin[: groups * 4 * 64].zero_()
rc.backward(gc, gd, ga)
torch.cuda.synchronize()
tm = rc.bin[: groups * 64].cpu().numpy().view(np.uint64).reshape(groups, 8)
ok = tm[:, 0] != 0
tm = tm[ok].astype(np.int64)
print("waves with timing", tm.shape[0])
t0 = tm[:, 0].min()
ghz = 0.1    # s_memtime ticks: 100 MHz constant clock on this part? calibrate against kernel span
span = tm[:, 6].max() - t0
print("kernel span ticks", span)
start = tm[:, 0] - t0; end = tm[:, 6] - t0
dur = end - start
print("duration ticks: mean %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f" % (dur.mean(), *np.percentile(dur, [10, 50, 90]), dur.max()))
print("start ticks: p50 %.0f p75 %.0f p90 %.0f max %.0f" % (*np.percentile(start, [50, 75, 90]), start.max()))
late = start > np.percentile(start, 74)
print("first-round waves: n %d mean dur %.0f ; late waves: n %d mean dur %.0f" % ((~late).sum(), dur[~late].mean(), late.sum(), dur[late].mean()))
print("phase fractions: info %.3f quadrants %.3f rows %.3f" % (((tm[:, 1] - tm[:, 0]) / dur).mean(), ((tm[:, 4] - tm[:, 1]) / dur).mean(), ((tm[:, 6] - tm[:, 4]) / dur).mean()))
print("inside batches: eval %.3f stage->B %.3f finish %.3f mfma-issue %.3f of duration" % tuple((tm[:, i] / dur).mean() for i in (2, 3, 5, 7)))
hist, edges = np.histogram(end, bins=10)
print("end-time histogram", hist.tolist())
hist, edges = np.histogram(start, bins=10, range=(0, span))
print("start-time histogram", hist.tolist())
