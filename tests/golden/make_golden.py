"""Generates the committed golden fixtures.  Run HERE (the container that has
/root/reference); the GPU box only ever reads the .npz files.

  python tests/golden/make_golden.py

1. reference_helpers.npz - outputs of the REFERENCE's own PyTorch helpers (imported from
   /root/reference, executed on CPU) for the pieces of the rasterizer's math that the
   reference does hold in-tree:
     eval_sh                          gaussiansplatting/utils/sh_utils.py:57-112
     build_rotation / build_scaling_rotation / strip_symmetric
                                      gaussiansplatting/utils/general_utils.py:64-110
     covariance recipe                gaussiansplatting/scene/gaussian_model.py:27-31
     Camera matrices                  gaussiansplatting/scene/cameras.py:17-54
     getProjectionMatrix, fov2focal   gaussiansplatting/utils/graphics_utils.py:73-99
   The reference hard-codes device="cuda"; the generator redirects those allocations to the
   CPU (a patch of torch.zeros / Tensor.cuda that lives only in this script).
2. oracle_scene.npz - a small seeded scene with the fp64 ORACLE's outputs and gradients
   (self-generated regression pin for the HIP path; NOT a reference output - the
   rasterizer's arithmetic is not in the reference tree, see oracle/__init__.py).
"""
import importlib.util
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def reference_helpers():
    sys.path.insert(0, REF)
    _zeros = torch.zeros

    def zeros_cpu(*a, **k):
        if "device" in k:
            k["device"] = "cpu"
        return _zeros(*a, **k)

    torch.zeros = zeros_cpu
    torch.Tensor.cuda = lambda self, *a, **k: self
    from gaussiansplatting.utils import general_utils, graphics_utils, sh_utils
    spec = importlib.util.spec_from_file_location(
        "ref_cameras", os.path.join(REF, "gaussiansplatting/scene/cameras.py"))
    ref_cameras = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_cameras)

    g = torch.Generator().manual_seed(1234)
    out = {}
    # --- SH
    P, M = 64, 16
    sh = torch.randn(P, M, 3, generator=g)
    dirs = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=1)
    out["sh_coeffs"], out["sh_dirs"] = sh.numpy(), dirs.numpy()
    for deg in range(4):
        out[f"sh_eval_deg{deg}"] = sh_utils.eval_sh(deg, sh.transpose(1, 2), dirs).numpy()
    out["rgb2sh_half"] = np.float32(sh_utils.RGB2SH(0.5))
    # --- covariance
    scales = torch.exp(torch.randn(P, 3, generator=g) * 0.5) * 0.05
    rots = torch.nn.functional.normalize(torch.randn(P, 4, generator=g), dim=1)
    for mod in (1.0, 0.7):
        L = general_utils.build_scaling_rotation(mod * scales, rots)
        out[f"cov_mod{mod}"] = general_utils.strip_symmetric(L @ L.transpose(1, 2)).numpy()
    out["cov_scales"], out["cov_rots"] = scales.numpy(), rots.numpy()
    out["rot_mats"] = general_utils.build_rotation(rots).numpy()
    # --- cameras
    from humangaussian_amd import synth
    c2ws, fovys, Vs, Fs, Cs, fovxs, sizes = [], [], [], [], [], [], []
    for i, (el, az, dist, fovy, H, W) in enumerate(
            [(10, 30, 1.75, 55, 1024, 1024), (-25, -140, 1.5, 40, 512, 512),
             (30, 100, 2.0, 70, 800, 600), (0, 0, 2.0, 50, 64, 48)]):
        c2w = torch.from_numpy(synth.c2w_orbit(el, az, dist)).float()
        cam = ref_cameras.Camera(c2w=c2w.clone(), FoVy=math.radians(fovy), height=H, width=W)
        c2ws.append(c2w.numpy()); fovys.append(math.radians(fovy)); sizes.append((H, W))
        Vs.append(cam.world_view_transform.numpy()); Fs.append(cam.full_proj_transform.numpy())
        Cs.append(cam.camera_center.numpy()); fovxs.append(cam.FoVx)
    out["cam_c2w"], out["cam_fovy"] = np.stack(c2ws), np.asarray(fovys, np.float64)
    out["cam_hw"] = np.asarray(sizes, np.int64)
    out["cam_V"], out["cam_full"] = np.stack(Vs), np.stack(Fs)
    out["cam_center"], out["cam_fovx"] = np.stack(Cs), np.asarray(fovxs, np.float64)
    out["proj_matrix"] = graphics_utils.getProjectionMatrix(0.01, 100.0, 0.9, 0.7).numpy()
    torch.zeros = _zeros
    np.savez_compressed(os.path.join(HERE, "reference_helpers.npz"), **out)
    print("wrote reference_helpers.npz", {k: np.shape(v) for k, v in out.items()})


def oracle_scene():
    import oracle
    from helpers import make_scene, oracle_settings
    sc = make_scene(P=160, sh_degree=2, seed=2024, H=48, W=64, spread=0.3)
    st = oracle_settings(sc)
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    ins = {k: sc[k].double().requires_grad_(True) for k in names}
    m2d = torch.zeros(160, 3, dtype=torch.float64, requires_grad=True)
    c, r, d, a = oracle.rasterize(ins["means3D"], m2d, ins["shs"], None, ins["opacities"],
                                  ins["scales"], ins["rotations"], None, st, dtype=torch.float64)
    g = torch.Generator().manual_seed(99)
    wc, wd, wa = (torch.randn(s, generator=g) for s in ((3, 48, 64), (1, 48, 64), (1, 48, 64)))
    ((c * wc).sum() + (d * wd).sum() + (a * wa).sum()).backward()
    r32 = oracle.rasterize(sc["means3D"], None, sc["shs"], None, sc["opacities"], sc["scales"],
                           sc["rotations"], None, st)[1]
    out = {f"in_{k}": sc[k].numpy() for k in names}
    out.update(color=c.detach().numpy(), depth=d.detach().numpy(), alpha=a.detach().numpy(),
               radii=r32.numpy(), w_color=wc.numpy(), w_depth=wd.numpy(), w_alpha=wa.numpy(),
               g_means2D=m2d.grad.numpy(), bg=sc["bg"].numpy(),
               cam=np.asarray([10.0, 30.0, 2.0, 50.0, 48, 64]))
    out.update({f"g_{k}": ins[k].grad.numpy() for k in names})
    np.savez_compressed(os.path.join(HERE, "oracle_scene.npz"), **out)
    print("wrote oracle_scene.npz")


if __name__ == "__main__":
    reference_helpers()
    oracle_scene()
