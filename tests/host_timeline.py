"""Ad-hoc: where does the host time of one fwd+bwd step go?  (run on the GPU box)"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humangaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, synth
from humangaussian_amd import rasterizer as R
dev = torch.device("cuda")
cloud = synth.init_cloud(100000, 0, "mid", 0)
cam = synth.orbit_camera(10, 30, 1.75, 55, 1024, 1024)
leaves = {k: getattr(cloud, k).to(dev).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
rs = GaussianRasterizationSettings(1024, 1024, math.tan(cam.FoVx/2), math.tan(cam.FoVy/2), torch.zeros(3, device=dev), 1.0,
      cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 0, cam.camera_center.to(dev), False, False)
rast = GaussianRasterizer(rs)
g = [torch.randn(s, device=dev) * 1e-3 for s in ((3,1024,1024),(1,1024,1024),(1,1024,1024))]
import ctypes
orig_sync = torch.cuda.Stream.synchronize
T = {"pre": 0.0, "sync": 0.0, "post": 0.0, "bwd": 0.0, "n": 0}
marks = {}
def sync_patch(self):
    marks["s0"] = time.perf_counter(); orig_sync(self); marks["s1"] = time.perf_counter()
torch.cuda.Stream.synchronize = sync_patch
def step(rec):
    for t in leaves.values(): t.grad = None
    m2 = torch.zeros_like(leaves["means3D"], requires_grad=True)
    t0 = time.perf_counter()
    c, r, d, a = rast(means3D=leaves["means3D"], means2D=m2, shs=leaves["shs"], opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"])
    t1 = time.perf_counter()
    torch.autograd.backward([c, d, a], g)
    t2 = time.perf_counter()
    if rec:
        T["pre"] += marks["s0"] - t0; T["sync"] += marks["s1"] - marks["s0"]; T["post"] += t1 - marks["s1"]; T["bwd"] += t2 - t1; T["n"] += 1
for i in range(10): step(False)
orig_sync(torch.cuda.current_stream())
t = time.perf_counter()
for i in range(50): step(True)
orig_sync(torch.cuda.current_stream())
tot = (time.perf_counter() - t) / 50 * 1e6
print("step us", tot, {k: (v / T["n"] * 1e6 if k != "n" else v) for k, v in T.items()})
