"""Calls libhgs_rast.so directly through the C ABI (ctypes, raw pointers) with buffers the
test owns - the same calls a cgo/JNI/ctypes binding in the reference would make."""
import ctypes
import math

import numpy as np
import torch

from humangaussian_amd import _lib
from humangaussian_amd._lib import HgsSettings, HgsStatus

GEOM_DTYPE = np.dtype([
    ("mx", "<f4"), ("my", "<f4"), ("ca", "<f4"), ("cb", "<f4"), ("cc", "<f4"), ("op", "<f4"),
    ("r", "<f4"), ("g", "<f4"), ("b", "<f4"), ("depth", "<f4"), ("rect_lo", "<u4"),
    ("rect_hi", "<u4"), ("offset", "<u4"), ("radius", "<i4"), ("clamped", "<u4"),
    ("flags", "<u4")])


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class RawCall:
    """One forward (+ optional backward) through the raw ABI on `device`."""

    def __init__(self, scene, device="cuda", scale_modifier=1.0, colors_precomp=None,
                 cov3D_precomp=None, capacity=None, store=True, sh_degree=None, max_tile_hint=0, mapped=0):
        self.lib = _lib.load()
        dev = torch.device(device)
        cam = scene["cam"]
        self.H, self.W = cam.image_height, cam.image_width
        d = lambda t: None if t is None else t.to(dev).float().contiguous()  # noqa: E731
        self.means3D = d(scene["means3D"])
        self.P = int(self.means3D.shape[0])
        self.colors_precomp = d(colors_precomp)
        self.cov3D = d(cov3D_precomp)
        self.shs = None if colors_precomp is not None else d(scene["shs"])
        self.M = 0 if self.shs is None else int(self.shs.shape[1])
        self.opac = d(scene["opacities"])
        self.scales = None if cov3D_precomp is not None else d(scene["scales"])
        self.rots = None if cov3D_precomp is not None else d(scene["rotations"])
        self.bg, self.vm = d(scene["bg"]), d(cam.world_view_transform)
        self.pm, self.cp = d(cam.full_proj_transform), d(cam.camera_center)
        s = HgsSettings()
        s.image_height, s.image_width = self.H, self.W
        s.tanfovx, s.tanfovy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
        s.bg, s.viewmatrix = self.bg.data_ptr(), self.vm.data_ptr()
        s.projmatrix, s.campos = self.pm.data_ptr(), self.cp.data_ptr()
        s.scale_modifier = scale_modifier
        s.sh_degree = scene["sh_degree"] if sh_degree is None else sh_degree
        s.prefiltered = s.debug = 0
        self.settings = s
        self.dev = dev
        self.store = store
        self.max_tile_hint = max_tile_hint
        self.mapped = mapped
        self.capacity = max(1, 8 * self.P) if capacity is None else capacity

    def forward(self):
        lib, dev, P, H, W = self.lib, self.dev, self.P, self.H, self.W
        u8 = lambda n: torch.zeros(int(n), dtype=torch.uint8, device=dev)  # noqa: E731
        self.color = torch.full((3, H, W), float("nan"), device=dev)
        self.depth = torch.full((1, H, W), float("nan"), device=dev)
        self.alpha = torch.full((1, H, W), float("nan"), device=dev)
        self.radii = torch.full((P,), -7, dtype=torch.int32, device=dev)
        self.geom = u8(lib.hgs_geom_bytes(P, H, W))
        self.bin = u8(lib.hgs_bin_bytes(self.capacity))
        self.img = u8(lib.hgs_img_bytes(H, W))
        self.status_host = torch.zeros(8, dtype=torch.int32).pin_memory()
        stream = torch.cuda.current_stream(dev)
        rc = lib.hgs_forward(ctypes.byref(self.settings), P, self.M, _p(self.means3D), _p(self.shs),
                             _p(self.colors_precomp), _p(self.opac), _p(self.scales), _p(self.rots),
                             _p(self.cov3D), _p(self.color), _p(self.depth), _p(self.alpha),
                             _p(self.radii), _p(self.geom), _p(self.bin), self.capacity,
                             _p(self.img), 1 if self.store else 0, self.max_tile_hint,
                             ctypes.c_void_p(self.status_host.data_ptr()), self.mapped, None, None,
                             ctypes.c_void_p(stream.cuda_stream))
        stream.synchronize()
        self.rc = rc
        self.status = [int(x) & 0xFFFFFFFF for x in self.status_host.tolist()]
        return rc

    def geom_records(self):
        raw = self.geom[: self.P * 64].cpu().numpy().tobytes()
        return np.frombuffer(raw, dtype=GEOM_DTYPE)

    def backward(self, g_color, g_depth, g_alpha, use_status=True, pairs_scratch=False):
        lib, dev, P, M = self.lib, self.dev, self.P, self.M
        d = lambda t: None if t is None else t.to(dev).float().contiguous()  # noqa: E731
        gc, gd, ga = d(g_color), d(g_depth), d(g_alpha)
        nan = lambda *s: torch.full(s, float("nan"), device=dev)  # noqa: E731
        out = dict(means3D=nan(P, 3), means2D=nan(P, 3), opacities=nan(P, 1),
                   shs=nan(P, M, 3) if self.shs is not None else None,
                   colors_precomp=nan(P, 3) if self.colors_precomp is not None else None,
                   scales=nan(P, 3) if self.scales is not None else None,
                   rotations=nan(P, 4) if self.rots is not None else None,
                   cov3D_precomp=nan(P, 6) if self.cov3D is not None else None)
        st = HgsStatus()
        (st.num_rendered, st.active_tiles, st.num_pairs, st.bwd_groups, st.overflow) = self.status[:5]
        st.reserved[0], st.reserved[1], st.reserved[2] = self.status[5:8]
        if pairs_scratch:      # sized by the published pair count, with a guard region behind it that must stay untouched
            nbytes = int(lib.hgs_bwd_scratch_bytes_pairs(st.num_rendered, st.num_pairs))
            scratch = torch.full((nbytes + 4096,), 0xAB, dtype=torch.uint8, device=dev)
            self.scratch_bytes = nbytes
        else:
            scratch = torch.zeros(int(lib.hgs_bwd_scratch_bytes(st.num_rendered if use_status else self.capacity)), dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev)
        rc = lib.hgs_backward(ctypes.byref(self.settings), P, M, _p(self.means3D), _p(self.shs),
                              _p(self.colors_precomp), _p(self.opac), _p(self.scales), _p(self.rots),
                              _p(self.cov3D), _p(self.radii), _p(self.color), _p(self.depth),
                              _p(self.alpha), _p(gc), _p(gd), _p(ga), _p(self.geom), _p(self.bin),
                              _p(self.img), (ctypes.byref(st) if use_status else None), self.capacity, _p(scratch), _p(out["means3D"]),
                              _p(out["means2D"]), _p(out["shs"]), _p(out["colors_precomp"]),
                              _p(out["opacities"]), _p(out["scales"]), _p(out["rotations"]),
                              _p(out["cov3D_precomp"]), None, ctypes.c_void_p(stream.cuda_stream))
        stream.synchronize()
        assert rc == 0, rc
        if pairs_scratch:
            assert bool((scratch[self.scratch_bytes:] == 0xAB).all()), "the backward wrote behind a scratch sized by num_pairs"
        return {k: (None if v is None else v.cpu()) for k, v in out.items()}
