"""The reference's OWN modules, imported unchanged from /root/reference with this repo's
`diff_gaussian_rasterization/` and `simple_knn/` shims on sys.path (no GPU needed):

  gaussiansplatting/gaussian_renderer/__init__.py:14   from diff_gaussian_rasterization import ...
  gaussiansplatting/scene/gaussian_model.py:20         from simple_knn._C import distCUDA2

The real `render()` (gaussian_renderer/__init__.py:18-104) is executed against the real
`GaussianModel` getters; its rasterizer call (`:86-94`) is bound against
`GaussianRasterizer.forward`'s signature and must carry exactly the tensors our forward expects.
/root/reference does not exist on the GPU box: skipped there.
"""
import inspect
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gaussiansplatting")),
                                reason="/root/reference is not mounted here")


@pytest.fixture()
def reference_modules(monkeypatch):
    for p in (REF, ROOT):
        if p not in sys.path:
            monkeypatch.syspath_prepend(p)
    # `plyfile` is not installed: the model only needs the names at import time
    ply = types.ModuleType("plyfile")
    ply.PlyData = ply.PlyElement = object
    monkeypatch.setitem(sys.modules, "plyfile", ply)
    # gaussiansplatting/scene/__init__.py pulls the COLMAP dataset readers (PIL, ...): register
    # the package without running it so that only scene/gaussian_model.py is executed
    scene = types.ModuleType("gaussiansplatting.scene")
    scene.__path__ = [os.path.join(REF, "gaussiansplatting", "scene")]
    monkeypatch.setitem(sys.modules, "gaussiansplatting.scene", scene)
    for name in [m for m in sys.modules if m.startswith("gaussiansplatting.gaussian_renderer")]:
        monkeypatch.delitem(sys.modules, name)
    import gaussiansplatting.gaussian_renderer as gr
    from gaussiansplatting.scene.gaussian_model import GaussianModel
    yield gr, GaussianModel
    for name in [m for m in sys.modules if m.startswith("gaussiansplatting")]:
        sys.modules.pop(name, None)


def test_reference_render_binds_to_our_rasterizer(reference_modules, monkeypatch):
    gr, GaussianModel = reference_modules
    import diff_gaussian_rasterization as dgr
    from humangaussian_amd import rasterizer as ours
    from simple_knn._C import distCUDA2
    from humangaussian_amd.knn import distCUDA2 as our_knn
    # the names the reference imported ARE this repo's objects
    assert gr.GaussianRasterizer is ours.GaussianRasterizer is dgr.GaussianRasterizer
    assert gr.GaussianRasterizationSettings is ours.GaussianRasterizationSettings
    assert distCUDA2 is our_knn

    # the reference hard-codes device="cuda" (gaussian_renderer/__init__.py:26): redirect
    _zl = torch.zeros_like
    monkeypatch.setattr(torch, "zeros_like", lambda t, **k: _zl(t, **{**k, "device": "cpu"}) if "device" in k else _zl(t, **k))

    P, deg = 7, 1
    pc = GaussianModel(deg)
    g = torch.Generator().manual_seed(0)
    pc._xyz = torch.randn(P, 3, generator=g).requires_grad_(True)
    pc._features_dc = torch.randn(P, 1, 3, generator=g).requires_grad_(True)
    pc._features_rest = torch.randn(P, (deg + 1) ** 2 - 1, 3, generator=g).requires_grad_(True)
    pc._scaling = torch.randn(P, 3, generator=g).requires_grad_(True)
    pc._rotation = torch.randn(P, 4, generator=g).requires_grad_(True)
    pc._opacity = torch.randn(P, 1, generator=g).requires_grad_(True)
    pc.active_sh_degree = 1

    cam = types.SimpleNamespace(FoVx=0.9, FoVy=0.8, image_height=32, image_width=48,
                                world_view_transform=torch.eye(4), full_proj_transform=torch.eye(4),
                                camera_center=torch.zeros(3))
    pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)
    seen = {}
    real_sig = inspect.signature(ours.GaussianRasterizer.forward)

    def fake_forward(self, *args, **kwargs):
        # bind exactly as Python would bind the reference's call to OUR forward
        ba = real_sig.bind(self, *args, **kwargs)
        ba.apply_defaults()
        seen.update(ba.arguments)
        seen["settings"] = self.raster_settings
        H, W = self.raster_settings.image_height, self.raster_settings.image_width
        n = ba.arguments["means3D"].shape[0]
        return torch.zeros(3, H, W), torch.ones(n, dtype=torch.int32), torch.zeros(1, H, W), torch.zeros(1, H, W)

    monkeypatch.setattr(ours.GaussianRasterizer, "forward", fake_forward)
    out = gr.render(cam, pc, pipe, torch.zeros(3))
    assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii", "depth_3dgs", "alpha_3dgs"}
    # what arrived at the boundary (gaussian_renderer/__init__.py:86-94)
    assert seen["means3D"].shape == (P, 3) and seen["means2D"].shape == (P, 3) and seen["means2D"].requires_grad
    assert seen["shs"].shape == (P, (deg + 1) ** 2, 3) and seen["colors_precomp"] is None
    assert seen["opacities"].shape == (P, 1) and float(seen["opacities"].min()) > 0       # sigmoid applied
    assert seen["scales"].shape == (P, 3) and float(seen["scales"].min()) > 0             # exp applied
    assert torch.allclose(seen["rotations"].norm(dim=1), torch.ones(P))                   # normalised
    assert seen["cov3D_precomp"] is None
    s = seen["settings"]
    assert isinstance(s, ours.GaussianRasterizationSettings)
    assert (s.image_height, s.image_width, s.sh_degree, s.prefiltered, s.debug) == (32, 48, 1, False, False)
    assert abs(s.tanfovx - 0.4830550656) < 1e-6 and s.scale_modifier == 1.0
    assert out["visibility_filter"].dtype == torch.bool


def test_our_render_mirror_matches_reference_render_arguments(reference_modules, monkeypatch):
    """humangaussian_amd.renderer.render hands the rasterizer the same arguments as the
    reference's render() for the same camera / model / pipe."""
    gr, GaussianModel = reference_modules
    from humangaussian_amd import rasterizer as ours
    from humangaussian_amd import renderer as mirror
    _zl = torch.zeros_like
    monkeypatch.setattr(torch, "zeros_like", lambda t, **k: _zl(t, **{**k, "device": "cpu"}) if "device" in k else _zl(t, **k))
    P, deg = 5, 2
    pc = GaussianModel(deg)
    g = torch.Generator().manual_seed(1)
    pc._xyz = torch.randn(P, 3, generator=g)
    pc._features_dc = torch.randn(P, 1, 3, generator=g)
    pc._features_rest = torch.randn(P, (deg + 1) ** 2 - 1, 3, generator=g)
    pc._scaling = torch.randn(P, 3, generator=g)
    pc._rotation = torch.randn(P, 4, generator=g)
    pc._opacity = torch.randn(P, 1, generator=g)
    pc.active_sh_degree = 2
    cam = types.SimpleNamespace(FoVx=1.1, FoVy=0.7, image_height=16, image_width=16,
                                world_view_transform=torch.eye(4) * 2, full_proj_transform=torch.eye(4) * 3,
                                camera_center=torch.ones(3))
    pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)
    calls = []
    real_sig = inspect.signature(ours.GaussianRasterizer.forward)

    def fake_forward(self, *args, **kwargs):
        ba = real_sig.bind(self, *args, **kwargs)
        ba.apply_defaults()
        calls.append((dict(ba.arguments), self.raster_settings))
        return torch.zeros(3, 16, 16), torch.ones(P, dtype=torch.int32), torch.zeros(1, 16, 16), torch.zeros(1, 16, 16)

    monkeypatch.setattr(ours.GaussianRasterizer, "forward", fake_forward)
    bg = torch.tensor([0.1, 0.2, 0.3])
    a = gr.render(cam, pc, pipe, bg, 0.7)
    b = mirror.render(cam, pc, pipe, bg, 0.7)
    (ka, sa), (kb, sb) = calls
    assert set(a) == set(b)
    for k in ("means3D", "shs", "opacities", "scales", "rotations"):
        assert torch.equal(ka[k], kb[k]), k
    assert ka["colors_precomp"] is None and kb["colors_precomp"] is None
    for f in ("image_height", "image_width", "tanfovx", "tanfovy", "scale_modifier", "sh_degree", "prefiltered", "debug"):
        assert getattr(sa, f) == getattr(sb, f), f
    for f in ("bg", "viewmatrix", "projmatrix", "campos"):
        assert torch.equal(getattr(sa, f), getattr(sb, f)), f


def test_densify_fixture_is_what_the_reference_method_produces():
    """tests/golden/reference_densify.npz (what tests/test_gpu_bookkeeping.py replays on the GPU) IS the reference's own
    `GaussianModel.densify_and_prune` (gaussian_model.py:410-423): re-run here, every array equal."""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("make_densify_fixture", os.path.join(ROOT, "tests", "golden", "make_densify_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        res = mod.run()
    finally:
        for name in [m for m in sys.modules if m.startswith("gaussiansplatting")]:
            sys.modules.pop(name, None)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "reference_densify.npz"))
    assert sorted(fx.files) == sorted(res)
    for k in fx.files:
        assert np.array_equal(fx[k], res[k]), k
    assert res["out_xyz"].shape[0] != res["in_xyz"].shape[0] and res["samples"].shape[0] > 0
