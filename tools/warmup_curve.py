"""Step time of the headline step against the steps already run in the process (chunks of 25, then 100 steps):
  python tools/warmup_curve.py          (on the GPU box; run it twice in one call to see a warm box's curve too)
A fresh box runs the step at 0.175-0.19 ms for its first 0.6-1.0 s and at 0.155-0.157 afterwards: the reason
for bench.py's un-timed INIT_SECONDS in front of its first measurement."""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from humangaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, synth
t_start = time.perf_counter()
dev = torch.device("cuda", 0)
RES = 1024
cloud = synth.init_cloud(100_000, 0, "mid", seed=0, source=synth.resolve_cloud_source("auto"))
c = synth.orbit_camera(10.0, 30.0, 1.75, 55.0, RES, RES)
L = {k: getattr(cloud, k).to(dev).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
rs = GaussianRasterizationSettings(RES, RES, math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5), torch.zeros(3, device=dev), 1.0,
                                   c.world_view_transform.to(dev), c.full_proj_transform.to(dev), 0, c.camera_center.to(dev), False, False)
rast = GaussianRasterizer(rs)
g = torch.Generator().manual_seed(1)
gc = (torch.randn((3, RES, RES), generator=g) * 1e-3).to(dev)
gd = (torch.randn((1, RES, RES), generator=g) * 1e-3).to(dev)
ga = (torch.randn((1, RES, RES), generator=g) * 1e-3).to(dev)
def step():
    for t in L.values(): t.grad = None
    m2 = torch.zeros_like(L["means3D"]).requires_grad_(True)
    color, radii, depth, alpha = rast(means3D=L["means3D"], means2D=m2, shs=L["shs"], opacities=L["opacities"], scales=L["scales"], rotations=L["rotations"])
    torch.autograd.backward([color, depth, alpha], [gc, gd, ga])
torch.cuda.synchronize()
print("setup s", round(time.perf_counter() - t_start, 2))
out = []
for chunk in range(60):
    n = 25 if chunk < 20 else 100
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) / n * 1e6, 1))
print(out)
