"""SURVEY.md 8(f)-3 / 8(f)-4 on the GPU: the fused bookkeeping kernels against the reference's own
formulas restated in plain torch / numpy (cited line by line below)."""
import numpy as np
import pytest
import torch

from humangaussian_amd import densify
from humangaussian_amd.animation import MeshAnchoredGaussians

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_add_densification_stats_matches_the_reference_loop():
    B, P = 8, 5000
    g = torch.Generator().manual_seed(0)
    grads = [torch.randn(P, 3, generator=g) * 1e-3 for _ in range(B)]           # viewspace_point_list[idx].grad
    radii_v = [torch.where(torch.rand(P, generator=g) < 0.3, torch.zeros(P), torch.rand(P, generator=g) * 40).int()
               for _ in range(B)]
    hand_mask = torch.rand(P, generator=g) < 0.1
    accum0, denom0, maxr0 = torch.rand(P, 1, generator=g), torch.randint(0, 5, (P, 1), generator=g).float(), torch.rand(P, generator=g) * 20
    # --- reference, on the CPU: GaussianDreamer.py:253-256, 289-297, 385-391; gaussian_model.py:434-438
    radii = radii_v[0]
    for b in range(1, B):
        radii = torch.max(radii_v[b], radii)
    visibility_filter = (radii > 0.0) & (~hand_mask)
    vgrad = torch.zeros_like(grads[0])
    for b in range(B):
        vgrad = vgrad + grads[b]
    maxr = maxr0.clone()
    maxr[visibility_filter] = torch.max(maxr[visibility_filter], radii[visibility_filter])
    accum, denom = accum0.clone(), denom0.clone()
    accum[visibility_filter] += torch.norm(vgrad[visibility_filter, :2], dim=-1, keepdim=True)
    denom[visibility_filter] += 1
    # --- one HIP pass
    a_d, d_d, m_d = accum0.to(DEV), denom0.to(DEV), maxr0.to(DEV)
    rmax, vis = densify.add_densification_stats(torch.stack(grads).to(DEV), torch.stack(radii_v).to(DEV), a_d, d_d, m_d,
                                                keep=(~hand_mask).to(DEV))
    assert torch.equal(rmax.cpu(), radii) and torch.equal(vis.cpu(), visibility_filter) and vis.dtype == torch.bool
    assert torch.equal(m_d.cpu(), maxr) and torch.equal(d_d.cpu(), denom)
    assert float((a_d.cpu() - accum).abs().max()) <= 1e-7
    # single view, no mask (the (P,3) / (P,) form)
    a1, d1, m1 = accum0.to(DEV), denom0.to(DEV), maxr0.to(DEV)
    r1, v1 = densify.add_densification_stats(grads[0].to(DEV), radii_v[0].to(DEV), a1, d1, m1)
    assert torch.equal(v1.cpu(), radii_v[0] > 0) and torch.equal(r1.cpu(), radii_v[0])


@pytest.mark.parametrize("raw", [False, True])
def test_densify_masks_match_reference_selections(raw):
    P = 20000
    g = torch.Generator().manual_seed(1)
    accum = torch.rand(P, 1, generator=g) * 1e-3
    denom = torch.randint(0, 4, (P, 1), generator=g).float()            # zeros -> NaN grads -> 0
    accum[denom[:, 0] == 0] = 0.0
    log_scale = torch.randn(P, 3, generator=g) * 0.7 - 4.0
    logit_op = torch.randn(P, 1, generator=g) * 3.0
    max_radii2D = torch.rand(P, generator=g) * 60
    max_grad, percent_dense, extent, min_opacity, size_threshold = 2e-4, 0.01, 1.9, 0.05, 20.0
    # --- reference (gaussian_model.py:405-420, 359-376, 379-386)
    get_scaling, get_opacity = torch.exp(log_scale), torch.sigmoid(logit_op)
    grads = accum / denom
    grads[grads.isnan()] = 0.0
    clone = torch.where(torch.norm(grads, dim=-1) >= max_grad, True, False)
    clone = torch.logical_and(clone, torch.max(get_scaling, dim=1).values <= percent_dense * extent)
    split = torch.where(grads.squeeze() >= max_grad, True, False)
    split = torch.logical_and(split, torch.max(get_scaling, dim=1).values > percent_dense * extent)
    prune = (get_opacity < min_opacity).squeeze()
    big_points_vs = max_radii2D > size_threshold
    big_points_ws = get_scaling.max(dim=1).values > 0.1 * extent
    prune_full = torch.logical_or(torch.logical_or(prune, big_points_vs), big_points_ws)
    sc_in, op_in = (log_scale, logit_op) if raw else (get_scaling, get_opacity)
    c, s, p, counts = densify.densify_masks(accum.to(DEV), denom.to(DEV), sc_in.to(DEV), op_in.to(DEV), max_radii2D.to(DEV),
                                            max_grad, percent_dense, extent, min_opacity, max_screen_size=size_threshold,
                                            raw_params=raw)
    assert c.dtype == torch.bool and torch.equal(c.cpu(), clone) and torch.equal(s.cpu(), split)
    # the fused exp / sigmoid may differ from torch's in the last bit: allow decisions to flip only at the threshold
    diff = p.cpu() != prune_full
    if raw:
        near = ((get_opacity.squeeze() - min_opacity).abs() < 1e-6) | ((get_scaling.max(dim=1).values - 0.1 * extent).abs() < 1e-6)
        assert bool((~diff | near).all())
    else:
        assert not bool(diff.any())
    assert counts.cpu().tolist() == [int(c.sum()), int(s.sum()), int(p.sum())] and int(clone.sum()) > 0 and int(split.sum()) > 0
    # no size threshold (before size_threshold_fix_step): opacity only; prune_only (gaussian_model.py:422-429)
    _, _, p2, _ = densify.densify_masks(accum.to(DEV), denom.to(DEV), get_scaling.to(DEV), get_opacity.to(DEV),
                                        max_radii2D.to(DEV), max_grad, percent_dense, extent, min_opacity)
    assert torch.equal(p2.cpu(), prune)
    _, _, p3, _ = densify.densify_masks(accum.to(DEV), denom.to(DEV), get_scaling.to(DEV), get_opacity.to(DEV),
                                        max_radii2D.to(DEV), max_grad, percent_dense, extent, 0.005, size_thresh=0.01)
    ref3 = torch.logical_or((get_opacity < 0.005).squeeze(), get_scaling.max(dim=1).values > 0.01)
    assert torch.equal(p3.cpu(), ref3)


def test_prune_rows_equals_boolean_indexing_for_all_tensors():
    P = 70001                                   # not a multiple of the 1024-row blocks
    g = torch.Generator().manual_seed(2)
    keep = torch.rand(P, generator=g) < 0.6
    tensors = [torch.randn(P, 3, generator=g), torch.randn(P, 15, 3, generator=g), torch.randn(P, 1, generator=g),
               torch.randn(P, 4, generator=g), torch.randn(P, generator=g)]
    out = densify.prune_rows(keep.to(DEV), [t.to(DEV) for t in tensors])
    for got, t in zip(out, tensors):
        assert got.shape == t[keep].shape and torch.equal(got.cpu(), t[keep])        # _prune_optimizer: x[mask]
    none = densify.prune_rows(torch.zeros(P, dtype=torch.bool, device=DEV), [tensors[0].to(DEV)])
    assert none[0].shape == (0, 3)
    allk = densify.prune_rows(torch.ones(P, dtype=torch.bool, device=DEV), [tensors[3].to(DEV)])
    assert torch.equal(allk[0].cpu(), tensors[3])


def test_reanchor_matches_the_reference_numpy_pass():
    """animation.py:384-403 in numpy (as the reference runs it) vs the device pass."""
    rng = np.random.default_rng(3)
    V, F, P = 3000, 5500, 40000
    vertices = rng.normal(size=(V, 3)).astype(np.float32)
    faces_all = rng.integers(0, V, size=(F, 3)).astype(np.int32)
    faces_all[:, 1] = (faces_all[:, 0] + 1 + rng.integers(0, V - 2, size=F)) % V        # non-degenerate index triples
    mapping_face = rng.integers(0, F, size=P).astype(np.int32)
    uvw = rng.random((P, 3)).astype(np.float32)
    uvw /= uvw.sum(1, keepdims=True)
    dist = (rng.normal(size=P) * 0.01).astype(np.float32)
    anchored = MeshAnchoredGaussians(faces_all, mapping_face, uvw, dist)
    for frame in range(2):
        verts = vertices + 0.05 * np.sin(frame + vertices[:, [1, 2, 0]]).astype(np.float32)
        faces = faces_all[mapping_face]
        v0, v1, v2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
        fnormals = np.cross(v1 - v0, v2 - v0)
        fnormals = fnormals / (np.linalg.norm(fnormals, axis=1, keepdims=True) + 1e-20)
        cpoints = v0 * uvw[:, [0]] + v1 * uvw[:, [1]] + v2 * uvw[:, [2]]
        points = cpoints + dist[:, None] * fnormals
        got = anchored.positions(verts)
        assert got.is_cuda and got.shape == (P, 3) and got.dtype == torch.float32
        assert np.abs(got.cpu().numpy() - points).max() <= 2e-6


def test_densify_and_prune_matches_the_reference_method():
    """SURVEY 8(f)-3 whole: `densify.densify_and_prune` against the reference's OWN `GaussianModel.densify_and_prune`
    (gaussian_model.py:410-423 with clone :393-408, split + sampling :359-391, prune :303-318 and the Adam surgery
    :283-357), recorded in the build container by tests/golden/make_densify_fixture.py: same state in, the reference's
    normal samples replayed; row order and counts exact, every parameter / Adam moment / statistic <= 1e-6, and the
    optimizer object ends up holding the new parameters with their moments."""
    import os
    import types
    import numpy as np
    from humangaussian_amd import densify
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_densify.npz"))
    dev = torch.device("cuda")
    groups = (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
              ("scaling", "_scaling"), ("rotation", "_rotation"))
    pc = types.SimpleNamespace()
    plist = []
    for name, attr in groups:
        p = torch.nn.Parameter(torch.from_numpy(fx["in" + attr]).to(dev))
        setattr(pc, attr, p)
        plist.append({"params": [p], "lr": 1e-3, "name": name})
    pc.optimizer = torch.optim.Adam(plist, lr=0.0, eps=1e-15)
    for name, attr in groups:
        pc.optimizer.state[getattr(pc, attr)] = {"step": torch.tensor(3.0), "exp_avg": torch.from_numpy(fx["in_exp_avg_" + name]).to(dev),
                                                 "exp_avg_sq": torch.from_numpy(fx["in_exp_avg_sq_" + name]).to(dev)}
    pc.xyz_gradient_accum = torch.from_numpy(fx["in_xyz_gradient_accum"]).to(dev)
    pc.denom = torch.from_numpy(fx["in_denom"]).to(dev)
    pc.max_radii2D = torch.from_numpy(fx["in_max_radii2D"]).to(dev)
    max_grad, min_opacity, extent, max_screen_size, pc.percent_dense = (float(x) for x in fx["args"])
    counts = densify.densify_and_prune(pc, max_grad, min_opacity, extent, max_screen_size,
                                       split_samples=torch.from_numpy(fx["samples"]))
    assert counts["children"] == fx["samples"].shape[0] and counts["cloned"] > 20 and counts["split"] > 20 and counts["pruned"] > 20
    assert counts["points"] == fx["out_xyz"].shape[0]
    for name, attr in groups:
        p = getattr(pc, attr)
        group = next(g for g in pc.optimizer.param_groups if g["name"] == name)
        assert group["params"][0] is p and isinstance(p, torch.nn.Parameter) and p.requires_grad and p.is_leaf
        st = pc.optimizer.state[p]
        assert len(pc.optimizer.state) == 6 and float(st["step"]) == 3.0
        for got, key in ((p.detach(), "out" + attr), (st["exp_avg"], "out_exp_avg_" + name), (st["exp_avg_sq"], "out_exp_avg_sq_" + name)):
            want = torch.from_numpy(fx[key])
            assert got.shape == want.shape, (key, got.shape, want.shape)
            err = float((got.cpu() - want).abs().max())
            assert err <= 1e-6 * max(1.0, float(want.abs().max())), (key, err)
    # survivors and clones are COPIES: bit-exact rows (the row ORDER is the reference's)
    n_children = fx["samples"].shape[0]
    assert torch.equal(pc._opacity.detach().cpu(), torch.from_numpy(fx["out_opacity"]))
    assert torch.equal(pc._rotation.detach().cpu(), torch.from_numpy(fx["out_rotation"]))
    assert n_children > 0 and not torch.equal(pc._xyz.detach().cpu()[-1], torch.from_numpy(fx["in_xyz"])[-1])
    for key, got in (("out_xyz_gradient_accum", pc.xyz_gradient_accum), ("out_denom", pc.denom), ("out_max_radii2D", pc.max_radii2D)):
        assert torch.equal(got.cpu(), torch.from_numpy(fx[key])), key
    # the optimizer still works on the new parameters
    for _, attr in groups:
        getattr(pc, attr).grad = torch.ones_like(getattr(pc, attr))
    pc.optimizer.step()
    # a fresh draw (no replayed samples) gives the same counts and different children
    pc2 = types.SimpleNamespace(percent_dense=pc.percent_dense)
    plist = []
    for name, attr in groups:
        p = torch.nn.Parameter(torch.from_numpy(fx["in" + attr]).to(dev))
        setattr(pc2, attr, p)
        plist.append({"params": [p], "lr": 1e-3, "name": name})
    pc2.optimizer = torch.optim.Adam(plist, lr=0.0, eps=1e-15)       # no state yet: moments appear on the first step
    pc2.xyz_gradient_accum = torch.from_numpy(fx["in_xyz_gradient_accum"]).to(dev)
    pc2.denom = torch.from_numpy(fx["in_denom"]).to(dev)
    pc2.max_radii2D = torch.from_numpy(fx["in_max_radii2D"]).to(dev)
    c2 = densify.densify_and_prune(pc2, max_grad, min_opacity, extent, max_screen_size)
    assert c2["cloned"] == counts["cloned"] and c2["split"] == counts["split"] and len(pc2.optimizer.state) == 0
