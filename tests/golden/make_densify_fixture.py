"""Generates tests/golden/reference_densify.npz: the reference's OWN `GaussianModel.densify_and_prune`
(gaussiansplatting/scene/gaussian_model.py:410-423, with `densify_and_clone`, `densify_and_split`, `prune_points` and the
optimizer surgery `cat_tensors_to_optimizer` / `_prune_optimizer` it calls) run on a small model with a live Adam state:
every input (raw parameters, both Adam moments, the densification statistics, the arguments), the normal samples the split
drew, and every output (parameters, moments, statistics after the call).  Run in the build container (imports
/root/reference, which does not exist on the GPU box):

    python tests/golden/make_densify_fixture.py

tests/test_gpu_bookkeeping.py::test_densify_and_prune_matches_the_reference_method replays the state through
humangaussian_amd.densify.densify_and_prune on the GPU (masks / row order exact, tensors <= 1e-6);
tests/test_reference_import_cpu.py::test_densify_fixture_is_what_the_reference_method_produces re-runs this script's
`run()` where the reference is mounted and compares with the committed file."""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GROUPS = (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
          ("scaling", "_scaling"), ("rotation", "_rotation"))
ARGS = dict(max_grad=0.02, min_opacity=0.05, extent=2.0, max_screen_size=20.0)


def _import_model():
    for p in (REF, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    if "plyfile" not in sys.modules:
        ply = types.ModuleType("plyfile")
        ply.PlyData = ply.PlyElement = object
        sys.modules["plyfile"] = ply
    if "gaussiansplatting.scene" not in sys.modules:
        scene = types.ModuleType("gaussiansplatting.scene")
        scene.__path__ = [os.path.join(REF, "gaussiansplatting", "scene")]
        sys.modules["gaussiansplatting.scene"] = scene
    from gaussiansplatting.scene.gaussian_model import GaussianModel
    return GaussianModel


def run():
    """-> dict of numpy arrays (inputs `in_*`, outputs `out_*`, `samples`, `args`)."""
    GaussianModel = _import_model()
    # the reference hard-codes device="cuda": redirect the factory calls it makes to the CPU for the duration of the run
    saved = {n: getattr(torch, n) for n in ("zeros", "normal")}
    drawn = []

    def cpu(fn):
        return lambda *a, **k: fn(*a, **{kk: ("cpu" if kk == "device" else vv) for kk, vv in k.items()})

    def normal(*a, **k):
        out = saved["normal"](*a, **k)
        drawn.append(out.detach().clone())
        return out
    torch.zeros, torch.normal = cpu(saved["zeros"]), normal
    try:
        P, deg = 600, 1
        g = torch.Generator().manual_seed(77)
        pc = GaussianModel(deg)
        pc._xyz = torch.nn.Parameter((torch.rand(P, 3, generator=g) - 0.5) * 1.2)
        pc._features_dc = torch.nn.Parameter(torch.randn(P, 1, 3, generator=g) * 0.8)
        pc._features_rest = torch.nn.Parameter(torch.randn(P, (deg + 1) ** 2 - 1, 3, generator=g) * 0.3)
        pc._scaling = torch.nn.Parameter(torch.log(0.02 * torch.exp(1.2 * torch.randn(P, 3, generator=g))))
        pc._rotation = torch.nn.Parameter(torch.randn(P, 4, generator=g))
        pc._opacity = torch.nn.Parameter(torch.logit(0.01 + 0.98 * torch.rand(P, 1, generator=g)))
        pc.spatial_lr_scale = 1.0
        targs = types.SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                                      position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025,
                                      opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001)
        pc.training_setup(targs)
        for _ in range(3):                                       # a live Adam state: three steps on random gradients
            for _, attr in GROUPS:
                p = getattr(pc, attr)
                p.grad = torch.randn(p.shape, generator=g) * 0.1
            pc.optimizer.step()
        pc.xyz_gradient_accum = torch.rand(P, 1, generator=g) * 0.3
        pc.denom = torch.randint(0, 8, (P, 1), generator=g).float()           # zeros -> NaN -> 0 (gaussian_model.py:412)
        pc.xyz_gradient_accum[pc.denom == 0] = 0.0
        pc.max_radii2D = torch.rand(P, generator=g) * 40.0
        out = {}
        for name, attr in GROUPS:
            p = getattr(pc, attr)
            st = pc.optimizer.state[p]
            out["in" + attr] = p.detach().numpy().copy()
            out["in_exp_avg_" + name] = st["exp_avg"].numpy().copy()
            out["in_exp_avg_sq_" + name] = st["exp_avg_sq"].numpy().copy()
        out["in_xyz_gradient_accum"], out["in_denom"] = pc.xyz_gradient_accum.numpy().copy(), pc.denom.numpy().copy()
        out["in_max_radii2D"] = pc.max_radii2D.numpy().copy()
        torch.manual_seed(5)
        pc.densify_and_prune(ARGS["max_grad"], ARGS["min_opacity"], ARGS["extent"], ARGS["max_screen_size"])
        assert len(drawn) == 1
        out["samples"] = drawn[0].detach().numpy()
        for name, attr in GROUPS:
            p = getattr(pc, attr)
            group = next(gr for gr in pc.optimizer.param_groups if gr["name"] == name)
            assert group["params"][0] is p
            st = pc.optimizer.state[p]
            out["out" + attr] = p.detach().numpy().copy()
            out["out_exp_avg_" + name] = st["exp_avg"].numpy().copy()
            out["out_exp_avg_sq_" + name] = st["exp_avg_sq"].numpy().copy()
        out["out_xyz_gradient_accum"], out["out_denom"] = pc.xyz_gradient_accum.numpy().copy(), pc.denom.numpy().copy()
        out["out_max_radii2D"] = pc.max_radii2D.numpy().copy()
        out["args"] = np.array([ARGS["max_grad"], ARGS["min_opacity"], ARGS["extent"], ARGS["max_screen_size"],
                                targs.percent_dense], np.float64)
        return out
    finally:
        torch.zeros, torch.normal = saved["zeros"], saved["normal"]


if __name__ == "__main__":
    res = run()
    path = os.path.join(ROOT, "tests", "golden", "reference_densify.npz")
    np.savez_compressed(path, **res)
    print("wrote", path, "points", res["in_xyz"].shape[0], "->", res["out_xyz"].shape[0], "split children", res["samples"].shape[0])
