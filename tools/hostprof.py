"""Host-side profile of the training-style step (cProfile) - where do the ~350 us of host time go?"""
import cProfile, pstats, sys, time, math, torch
sys.path.insert(0, ".")
from humangaussian_amd import synth, rasterizer as R
dev = torch.device("cuda:0")
P = 100000
import bench
cloud = synth.init_cloud(P, 0, "mid", seed=0)
leaves = {k: getattr(cloud, k).to(dev).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
cam = bench.camera_for_rank(0)
bg = torch.zeros(3, device=dev)
rs = R.GaussianRasterizationSettings(1024, 1024, math.tan(cam.FoVx * .5), math.tan(cam.FoVy * .5), bg, 1.0,
    cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 0, cam.camera_center.to(dev), False, False)
rast = R.GaussianRasterizer(rs)
gc = torch.randn(3, 1024, 1024, device=dev) * 1e-3
gd = torch.randn(1, 1024, 1024, device=dev) * 1e-3
ga = torch.randn(1, 1024, 1024, device=dev) * 1e-3
def step():
    for t in leaves.values():
        t.grad = None
    m2 = torch.zeros_like(leaves["means3D"], requires_grad=True)
    c, r, d, a = rast(means3D=leaves["means3D"], means2D=m2, shs=leaves["shs"], opacities=leaves["opacities"],
                      scales=leaves["scales"], rotations=leaves["rotations"])
    torch.autograd.backward([c, d, a], [gc, gd, ga])
for _ in range(50): step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize()
print("us/step", (time.perf_counter() - t) / 300 * 1e6)
R._PROF.clear()
ta = {}
import time as _tt
def step2():
    t0 = _tt.perf_counter()
    for t in leaves.values():
        t.grad = None
    m2 = torch.zeros_like(leaves["means3D"], requires_grad=True)
    t1 = _tt.perf_counter()
    c, r, d, a = rast(means3D=leaves["means3D"], means2D=m2, shs=leaves["shs"], opacities=leaves["opacities"],
                      scales=leaves["scales"], rotations=leaves["rotations"])
    t2 = _tt.perf_counter()
    torch.autograd.backward([c, d, a], [gc, gd, ga])
    t3 = _tt.perf_counter()
    ta["s0_zero"] = ta.get("s0_zero", 0) + t1 - t0
    ta["s1_fwd"] = ta.get("s1_fwd", 0) + t2 - t1
    ta["s2_bwd"] = ta.get("s2_bwd", 0) + t3 - t2
for _ in range(300): step2()
torch.cuda.synchronize()
print({k: round(v / 300 * 1e6, 1) for k, v in sorted(ta.items())})
print({k: round(v / 300 * 1e6, 1) for k, v in sorted(R._PROF.items())})
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
