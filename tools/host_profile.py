"""Where the HOST's time of one step goes (bench.py's headline step, phase by phase, wall clock without device fences):
  python tools/host_profile.py [steps]          (on the GPU box)
Phases: grads cleared | means2D leaf (one fill launch) | forward call (binding: launches + the status wait) | backward call
(autograd engine + binding) | and cProfile's top entries of the same loop."""
import cProfile
import math
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from humangaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, synth
from humangaussian_amd import rasterizer as _rast

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
RES = 1024
cloud = synth.init_cloud(100_000, 0, "mid", seed=0, source=synth.resolve_cloud_source("auto"))
c = synth.orbit_camera(10.0, 30.0, 1.75, 55.0, RES, RES)
L = {k: getattr(cloud, k).to(dev).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
rs = GaussianRasterizationSettings(RES, RES, math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5), torch.zeros(3, device=dev), 1.0,
                                   c.world_view_transform.to(dev), c.full_proj_transform.to(dev), 0, c.camera_center.to(dev),
                                   False, False)
rast = GaussianRasterizer(rs)
g = torch.Generator().manual_seed(1)
gc = (torch.randn((3, RES, RES), generator=g) * 1e-3).to(dev)
gd = (torch.randn((1, RES, RES), generator=g) * 1e-3).to(dev)
ga = (torch.randn((1, RES, RES), generator=g) * 1e-3).to(dev)
acc = [0, 0, 0, 0]


def step(timed):
    t0 = time.perf_counter_ns()
    for t in L.values():
        t.grad = None
    t1 = time.perf_counter_ns()
    means2D = torch.zeros_like(L["means3D"]).requires_grad_(True)
    t2 = time.perf_counter_ns()
    color, radii, depth, alpha = rast(means3D=L["means3D"], means2D=means2D, shs=L["shs"], opacities=L["opacities"],
                                      scales=L["scales"], rotations=L["rotations"])
    t3 = time.perf_counter_ns()
    torch.autograd.backward([color, depth, alpha], [gc, gd, ga])
    t4 = time.perf_counter_ns()
    if timed:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            acc[i] += d


for _ in range(60):
    step(False)
torch.cuda.synchronize()
st0 = _rast._state(dev)
t0 = time.perf_counter()
for _ in range(steps):
    step(True)
torch.cuda.synchronize()
el = time.perf_counter() - t0
st1 = _rast._state(dev)
print("step %.1f us | grads cleared %.1f | means2D leaf %.1f | forward call %.1f | backward call %.1f" %
      ((el / steps * 1e6,) + tuple(a / steps * 1e-3 for a in acc)))
print("binding:", {k: round((st1.host_ns[k] - st0.host_ns[k]) / steps * 1e-3, 1) for k in st1.host_ns},
      "event wait %.1f" % ((st1.wait_ns - st0.wait_ns) / steps * 1e-3))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step(False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

# ---- the drop-in render() on raw parameters with fused activations, the same way
from humangaussian_amd import renderer
raw = {"_xyz": cloud.means3D, "_features_dc": cloud.shs[:, :1], "_features_rest": cloud.shs[:, 1:],
       "_opacity": torch.logit(cloud.opacities.clamp(1e-6, 1 - 1e-6)), "_scaling": torch.log(cloud.scales),
       "_rotation": cloud.rotations}
leaves = {k: v.to(dev).contiguous().requires_grad_(True) for k, v in raw.items()}


class Model:
    active_sh_degree = max_sh_degree = 0
    _opacity, _scaling, _rotation = leaves["_opacity"], leaves["_scaling"], leaves["_rotation"]
    _features_dc, _features_rest = leaves["_features_dc"], leaves["_features_rest"]
    get_xyz = property(lambda m: leaves["_xyz"])
    get_features = property(lambda m: torch.cat((leaves["_features_dc"], leaves["_features_rest"]), dim=1))
    get_opacity = property(lambda m: torch.sigmoid(leaves["_opacity"]))
    get_scaling = property(lambda m: torch.exp(leaves["_scaling"]))
    get_rotation = property(lambda m: torch.nn.functional.normalize(leaves["_rotation"]))


class Pipe:
    convert_SHs_python = compute_cov3D_python = debug = False


model, pipe = Model(), Pipe()
cam = renderer.HostCamera(RES, RES, c.FoVx, c.FoVy, c.world_view_transform.to(dev), c.full_proj_transform.to(dev), c.camera_center.to(dev))
bg = torch.zeros(3, device=dev)
acc2 = [0, 0, 0]


def step2(timed):
    t0 = time.perf_counter_ns()
    for t in leaves.values():
        t.grad = None
    t1 = time.perf_counter_ns()
    pkg = renderer.render(cam, model, pipe, bg, fuse_activations=True)
    t2 = time.perf_counter_ns()
    torch.autograd.backward([pkg["render"], pkg["depth_3dgs"], pkg["alpha_3dgs"]], [gc, gd, ga])
    t3 = time.perf_counter_ns()
    if timed:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2)):
            acc2[i] += d


for _ in range(60):
    step2(False)
torch.cuda.synchronize()
st0 = _rast._state(dev)
t0 = time.perf_counter()
for _ in range(steps):
    step2(True)
torch.cuda.synchronize()
el = time.perf_counter() - t0
st1 = _rast._state(dev)
print("fused drop-in: step %.1f us | grads cleared %.1f | render() %.1f | backward call %.1f" %
      ((el / steps * 1e6,) + tuple(a / steps * 1e-3 for a in acc2)))
print("binding:", {k: round((st1.host_ns[k] - st0.host_ns[k]) / steps * 1e-3, 1) for k in st1.host_ns},
      "event wait %.1f" % ((st1.wait_ns - st0.wait_ns) / steps * 1e-3))
# and the headline step once more (order effects)
for i in range(4):
    acc[i] = 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step(True)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("headline again: step %.1f us | grads cleared %.1f | means2D leaf %.1f | forward call %.1f | backward call %.1f" %
      ((el / steps * 1e6,) + tuple(a / steps * 1e-3 for a in acc)))
