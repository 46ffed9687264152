import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from humangaussian_amd import synth
from abi_runner import RawCall
cl = synth.init_cloud(100000, 0, "mid", 0)
cam = synth.orbit_camera(10, 30, 1.75, 55, 1024, 1024)
sc = dict(means3D=cl.means3D, scales=cl.scales, rotations=cl.rotations, opacities=cl.opacities, shs=cl.shs, cam=cam, sh_degree=0, bg=torch.zeros(3))
rc = RawCall(sc, capacity=1 << 19, mapped=1)
for i in range(3):
    assert rc.forward() == 0
raw = rc.geom[-256:].cpu().numpy().tobytes()
st = np.frombuffer(raw[32:32 + 40], dtype=np.uint64)
print("status", rc.status)
print("stamps delta (us @100MHz):", [(int(st[i + 1]) - int(st[i])) / 100.0 for i in range(4)], "total", (int(st[4]) - int(st[0])) / 100.0)
