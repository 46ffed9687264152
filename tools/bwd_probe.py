import faulthandler, os, sys, time
faulthandler.dump_traceback_later(25, exit=True)
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from abi_runner import RawCall
from humangaussian_amd import synth
cloud = synth.init_cloud(100000, 0, "mid", seed=0)
cam = synth.orbit_camera(10.0, 30.0, 1.75, 55.0, 1024, 1024)
scene = dict(means3D=cloud.means3D, shs=cloud.shs, opacities=cloud.opacities, scales=cloud.scales,
             rotations=cloud.rotations, bg=torch.zeros(3), cam=cam, sh_degree=0)
rc = RawCall(scene, capacity=int(os.environ.get("CAP", str(1 << 19))), mapped=0)
assert rc.forward() == 0
torch.cuda.synchronize()
print("fwd ok", rc.status, flush=True)
g = torch.Generator().manual_seed(1)
gc, gd, ga = ((torch.randn(s, generator=g) * 1e-3).cuda() for s in ((3, 1024, 1024), (1, 1024, 1024), (1, 1024, 1024)))
t = time.time(); out = rc.backward(gc, gd, ga); print("bwd1 ok", time.time() - t, float(out["means3D"].abs().sum()), flush=True)
t = time.time(); out = rc.backward(gc, gd, ga); print("bwd2 ok", time.time() - t, float(out["means3D"].abs().sum()), flush=True)
