"""Generates tests/golden/reference_render_call.npz: the EXACT arguments the reference's own
`render()` (gaussiansplatting/gaussian_renderer/__init__.py:18-104) hands to `GaussianRasterizer.forward` for a
small reference `GaussianModel` and a camera built by the reference's arithmetic.  Run in the build container
(imports /root/reference, which does not exist on the GPU box):

    python tests/golden/make_render_call_fixture.py

tests/test_gpu_api_contract.py::test_reference_render_call_replayed_on_the_hip_rasterizer replays the recorded call
through the HIP rasterizer on the GPU box and checks the result against the oracle."""
import inspect
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REF, ROOT]
ply = types.ModuleType("plyfile")
ply.PlyData = ply.PlyElement = object
sys.modules["plyfile"] = ply
scene = types.ModuleType("gaussiansplatting.scene")
scene.__path__ = [os.path.join(REF, "gaussiansplatting", "scene")]
sys.modules["gaussiansplatting.scene"] = scene
import gaussiansplatting.gaussian_renderer as gr  # noqa: E402
from gaussiansplatting.scene.gaussian_model import GaussianModel  # noqa: E402
from humangaussian_amd import rasterizer as ours  # noqa: E402
from humangaussian_amd import synth  # noqa: E402

_zl = torch.zeros_like
torch.zeros_like = lambda t, **k: _zl(t, **{**k, "device": "cpu"}) if "device" in k else _zl(t, **k)   # render() hard-codes "cuda"

P, deg, H, W = 400, 2, 72, 96
g = torch.Generator().manual_seed(2024)
pc = GaussianModel(deg)
pc._xyz = (torch.rand(P, 3, generator=g) - 0.5) * 0.6
pc._features_dc = torch.randn(P, 1, 3, generator=g) * 0.8
pc._features_rest = torch.randn(P, (deg + 1) ** 2 - 1, 3, generator=g) * 0.3
pc._scaling = torch.log(0.05 * torch.exp(0.5 * torch.randn(P, 3, generator=g)))      # raw: get_scaling = exp
pc._rotation = torch.randn(P, 4, generator=g)                                         # raw: get_rotation = normalize
pc._opacity = torch.logit(0.05 + 0.9 * torch.rand(P, 1, generator=g))                 # raw: get_opacity = sigmoid
pc.active_sh_degree = deg
c = synth.orbit_camera(12.0, 40.0, 2.0, 50.0, H, W)        # pinned to the reference Camera class by reference_helpers.npz
cam = types.SimpleNamespace(FoVx=c.FoVx, FoVy=c.FoVy, image_height=H, image_width=W,
                            world_view_transform=c.world_view_transform, full_proj_transform=c.full_proj_transform,
                            camera_center=c.camera_center)
pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)
seen = {}
sig = inspect.signature(ours.GaussianRasterizer.forward)


def fake_forward(self, *args, **kwargs):
    ba = sig.bind(self, *args, **kwargs)
    ba.apply_defaults()
    seen.update(ba.arguments)
    seen["settings"] = self.raster_settings
    n = ba.arguments["means3D"].shape[0]
    return torch.zeros(3, H, W), torch.ones(n, dtype=torch.int32), torch.zeros(1, H, W), torch.zeros(1, H, W)


ours.GaussianRasterizer.forward = fake_forward
with torch.no_grad():
    gr.render(cam, pc, pipe, torch.tensor([0.2, 0.1, 0.3]), 1.0)
s = seen["settings"]
out = {k: seen[k].detach().numpy() for k in ("means3D", "shs", "opacities", "scales", "rotations")}
assert seen["colors_precomp"] is None and seen["cov3D_precomp"] is None
out.update(bg=s.bg.numpy(), viewmatrix=s.viewmatrix.numpy(), projmatrix=s.projmatrix.numpy(), campos=s.campos.numpy(),
           scalars=np.array([s.image_height, s.image_width, s.tanfovx, s.tanfovy, s.scale_modifier, s.sh_degree], np.float64))
path = os.path.join(ROOT, "tests", "golden", "reference_render_call.npz")
np.savez_compressed(path, **out)
print("wrote", path, {k: v.shape for k, v in out.items()})
