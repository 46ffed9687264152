"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol
include/hgs_rast.h declares (no compute calls - there is no GPU here), the sizing
functions behave, and the Python API mirrors the reference's names / errors."""
import ctypes
import os
import re

import pytest
import torch

from humangaussian_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "hgs_rast.h")).read()
    declared = set(re.findall(r"^\s*(?:int|size_t)\s+(hgs_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(_lib.EXPORTS), (declared, set(_lib.EXPORTS))
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert lib.hgs_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.HgsStatus) == 32
    s = _lib.HgsSettings
    assert [f[0] for f in s._fields_] == [
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
        "projmatrix", "sh_degree", "campos", "prefiltered", "debug"]
    # natural C layout on LP64: 4+4+4+4 | 8 | 4(+4) | 8 | 8 | 4(+4) | 8 | 4+4
    assert ctypes.sizeof(s) == 72 and s.bg.offset == 16 and s.viewmatrix.offset == 32
    assert s.campos.offset == 56 and s.debug.offset == 68


def test_buffer_sizing(lib):
    g = lib.hgs_geom_bytes(100000, 1024, 1024)
    assert g >= 100000 * 64 + 6 * 4096 * 4 and g % 256 == 0
    assert lib.hgs_geom_bytes(0, 16, 16) > 0 and lib.hgs_geom_bytes(-1, 16, 16) == 0
    assert lib.hgs_img_bytes(1024, 1024) == 1024 * 1024 * 4
    b1, b2 = lib.hgs_bin_bytes(1 << 20), lib.hgs_bin_bytes(1 << 21)
    assert b1 >= (1 << 20) * (8 + 48 + 96) and abs(b2 - 2 * b1) <= 65536
    assert lib.hgs_bin_bytes(0) <= 65536          # only the fixed part of the segment planes
    # backward scratch (include/hgs_rast.h): one 48 B gradient row per entry + 16 (entry, cell) pair rows of 40 B
    assert lib.hgs_bwd_scratch_bytes(1000) == -(-1000 * (48 + 16 * 40) // 256) * 256
    assert lib.hgs_bwd_scratch_bytes(0) == 0 and lib.hgs_bwd_scratch_bytes(-5) == 0
    # sized by a published pair count: 48 B per entry + 40 B per pair; unknown / impossible counts fall back to the worst case
    assert lib.hgs_bwd_scratch_bytes_pairs(1000, 4400) == -(-(1000 * 48 + 4400 * 40) // 256) * 256
    assert lib.hgs_bwd_scratch_bytes_pairs(1000, 0) == lib.hgs_bwd_scratch_bytes(1000)
    assert lib.hgs_bwd_scratch_bytes_pairs(1000, 16001) == lib.hgs_bwd_scratch_bytes(1000)
    assert lib.hgs_bwd_scratch_bytes_pairs(0, 5) == 0


def test_argument_validation_without_gpu(lib):
    s = _lib.HgsSettings()
    assert lib.hgs_forward(ctypes.byref(s), 1, 1, *([None] * 13), 0, None, 0, 0, None, 0, None, None, None) == -1
    assert lib.hgs_mark_visible(None, 1, None, None, None) == -1


def test_python_api_mirrors_reference_names_and_errors():
    import diff_gaussian_rasterization as dgr
    from humangaussian_amd import GaussianRasterizationSettings, GaussianRasterizer
    assert dgr.GaussianRasterizer is GaussianRasterizer
    assert GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
        "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
    rs = GaussianRasterizationSettings(16, 16, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                       torch.zeros(3), False, False)
    r = GaussianRasterizer(raster_settings=rs)
    m = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), shs=torch.zeros(4, 1, 3))
    # the product path has no CPU fallback: CPU tensors fail loudly
    with pytest.raises(RuntimeError, match="HIP device"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), shs=torch.zeros(4, 1, 3),
          scales=torch.ones(4, 3), rotations=torch.ones(4, 4))


def test_simple_knn_module_name_resolves():
    """`from simple_knn._C import distCUDA2` (gaussian_model.py:20, gs_renderer.py:14) resolves to
    the HIP implementation; CPU tensors fail loudly."""
    from simple_knn._C import distCUDA2
    from humangaussian_amd.knn import distCUDA2 as impl
    assert distCUDA2 is impl
    with pytest.raises(RuntimeError, match="HIP device"):
        distCUDA2(torch.zeros(8, 3))
    lib = _lib.load()
    assert lib.hgs_knn_mean_dist2(-1, None, None, None) == -1
    assert lib.hgs_knn_mean_dist2(0, None, None, None) == 0
    assert lib.hgs_reduce_view_packs(0, 1, 1, None, None, None) == -1
    assert lib.hgs_reduce_view_packs(2, 0, 18, None, None, None) == 0
    assert lib.hgs_pack_view_contribution(5, 1, *([None] * 9)) == -1


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "humangaussian_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_backward_group_schedule_visits_every_group_once():
    """Host-side restatement of render_bwd.hip's group_of(), per die (the work tables are per die, Counters::sched; the
    launch's workgroups i with i % 8 == x are the G workgroups of die x, b = i // 8): workgroup b of G draws tickets
    0, 1, 2, ... and ticket t maps to group t * G + (b, or G - 1 - b on odd t).  Every group of the table must be owned by exactly one
    (workgroup, ticket), a workgroup's groups must ascend with the ticket (so it may stop at the first one beyond the
    table), and the snake must give every workgroup the same number of groups +- 1."""
    for G, ngroups in ((1, 7), (4, 4), (256, 3557), (256, 28000), (12, 5), (64, 0)):
        seen, per_wg = set(), []
        for b in range(G):
            mine, t = [], 0
            while True:
                grp = t * G + ((G - 1 - b) if (t & 1) else b)
                if grp >= ngroups:
                    # nothing later may fall inside the table
                    assert all(tt * G + ((G - 1 - b) if (tt & 1) else b) >= ngroups for tt in range(t + 1, t + 4))
                    break
                mine.append(grp)
                t += 1
            assert mine == sorted(mine)
            assert not (seen & set(mine))
            seen |= set(mine)
            per_wg.append(len(mine))
        assert seen == set(range(ngroups))
        assert max(per_wg) - min(per_wg) <= 1


def test_forward_takes_the_per_die_cell_tables_as_one_list():
    """Host-side restatement of render_fwd.hip's fwd_cell_key(): the sort fills one table of non-empty cells per (die,
    length class); the forward indexes a class as the dies' tables one behind the other through a prefix table
    [x] = cells of the class on dies < x.  Every (die, slot) must be reached by exactly one index of the class."""
    import numpy as np
    rng = np.random.default_rng(0)
    for trial in range(20):
        counts = rng.integers(0, 50, size=8)
        if trial == 0:
            counts[:] = 0
        if trial == 1:
            counts[:] = 0
            counts[7] = 5
        pre = np.concatenate([[0], np.cumsum(counts)])
        seen = set()
        for q in range(int(pre[8])):
            x, base = 0, 0
            for d in range(1, 8):
                if q >= pre[d]:
                    x, base = d, int(pre[d])
            assert 0 <= q - base < counts[x], (q, x, counts)
            seen.add((x, q - base))
        assert seen == {(x, i) for x in range(8) for i in range(int(counts[x]))}
