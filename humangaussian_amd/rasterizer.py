"""Host-side mirror of the `diff_gaussian_rasterization` Python API (ashawkey fork) on top of
libhgs_rast.so.  Same names, argument meaning, return tuple and error behaviour as the
module the reference imports at
  /root/reference/gaussiansplatting/gaussian_renderer/__init__.py:14  (call :36-51, :86-94)
  /root/reference/gs_renderer.py:10-13                               (call :951-966, :1006-1015)
so those files run unchanged when `diff_gaussian_rasterization` resolves to this package
(the top-level `diff_gaussian_rasterization/` shim re-exports it).

PyTorch is plumbing only: it owns the tensors, the current HIP stream and autograd; all
arithmetic happens in the HIP library reached through ctypes (plain pointers and sizes).
There is no CPU path: non-HIP tensors raise.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _lib
from ._lib import HgsSettings, HgsStatus


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# ---------------------------------------------------------------------------- plumbing

class _Pending:
    """An async forward whose status has not been inspected yet."""
    __slots__ = ("event", "slot", "cap", "hint")


class _DeviceState:
    """Per-device grow-only estimates (entry capacity R, longest tile list) and a small ring
    of pinned status mirrors."""
    RING = 8

    def __init__(self):
        self.capacity = 0
        self.tile_hint = 0          # longest tile list seen (with margin); 0 = unknown
        self.max_R = 0
        self.max_tile = 0
        self.status_ring = torch.zeros(self.RING, 8, dtype=torch.int32).pin_memory()
        self.status_np = self.status_ring.numpy()          # host view of the same pinned words
        self.status_ptr = [self.status_ring[i].data_ptr() for i in range(self.RING)]
        # recorded by the library right behind the status copy (after the scan stage)
        self.status_event = torch.cuda.Event()
        self.status_event.record()
        self.status_event.synchronize()
        self.ring_pos = 0
        self.pending: list = []
        self.synced_calls = 0

    def next_slot(self):
        i = self.ring_pos
        self.ring_pos = (i + 1) % self.RING
        return i

    def observe(self, status):
        self.max_R = max(self.max_R, status[0])
        self.max_tile = max(self.max_tile, status[6])


_device_state: dict = {}
_NULL_CTX = contextlib.nullcontext()
_async_mode = [os.environ.get("HGS_ASYNC", "0") not in ("", "0")]


def set_async(enabled: bool):
    """Opt-in asynchronous mode.  Default (False) mirrors upstream: one host sync per
    forward to read num_rendered, overflow handled transparently by re-running.  With
    async enabled a forward that needs gradients returns WITHOUT synchronising once the
    workload is known (two synchronous calls first): buffers are sized with a 2x margin over
    the largest R seen, the backward needs nothing from the host, and the status words of
    earlier calls are inspected lazily - an overflow (R more than doubled between
    consecutive calls) is then reported as a RuntimeError on a later call, after that
    call's outputs were already handed out.  Use for steady-state training loops."""
    _async_mode[0] = bool(enabled)


def _state(device: torch.device) -> _DeviceState:
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _device_state.get(key)
    if st is None:
        st = _device_state[key] = _DeviceState()
    return st


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _f32c(t: torch.Tensor, device) -> torch.Tensor:
    if t.device != device:
        raise RuntimeError(f"expected a tensor on {device}, got {t.device}")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _opt(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if (t is None or t.numel() == 0) else t


def _make_settings(rs: GaussianRasterizationSettings, device, keep: list) -> HgsSettings:
    bg = _f32c(rs.bg.reshape(-1), device)
    vm = _f32c(rs.viewmatrix, device)
    pm = _f32c(rs.projmatrix, device)
    cp = _f32c(rs.campos.reshape(-1), device)
    if bg.numel() != 3 or vm.numel() != 16 or pm.numel() != 16 or cp.numel() != 3:
        raise RuntimeError("bg/campos must have 3 elements, viewmatrix/projmatrix 16")
    keep.extend([bg, vm, pm, cp])
    s = HgsSettings()
    s.image_height = int(rs.image_height)
    s.image_width = int(rs.image_width)
    s.tanfovx = float(rs.tanfovx)
    s.tanfovy = float(rs.tanfovy)
    s.bg = bg.data_ptr()
    s.scale_modifier = float(rs.scale_modifier)
    s.viewmatrix = vm.data_ptr()
    s.projmatrix = pm.data_ptr()
    s.sh_degree = int(rs.sh_degree)
    s.campos = cp.data_ptr()
    s.prefiltered = int(bool(rs.prefiltered))
    s.debug = int(bool(rs.debug))
    return s


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"libhgs_rast: {what} failed with code {rc}")


# Measurement hook (bench.py only): arrays of hipEvent_t handles recorded after each stage
# of the next forward / backward calls (see HGS_FWD_STAGES / HGS_BWD_STAGES in hgs_rast.h).
_stage_events = {"fwd": None, "bwd": None}


def set_stage_events(fwd=None, bwd=None):
    """fwd / bwd: sequences of raw hipEvent_t handles (ints) or None to disable."""
    def mk(seq):
        if seq is None:
            return None
        arr = (ctypes.c_void_p * len(seq))(*[ctypes.c_void_p(int(h)) for h in seq])
        return arr
    _stage_events["fwd"], _stage_events["bwd"] = mk(fwd), mk(bwd)


def _round_capacity(n: int) -> int:
    return max(1 << 16, (int(n) + 0xFFFF) & ~0xFFFF)


def _read_status(st: _DeviceState, slot: int):
    return [x & 0xFFFFFFFF for x in st.status_np[slot].tolist()]


_size_cache: dict = {}


def _sizes(lib, P: int, H: int, W: int, cap: int):
    """(geom, img, bin, scratch) byte sizes, each rounded to 256 B so the four regions can be
    carved from one allocation.  Cached: four ctypes calls per forward add up."""
    key = (P, H, W, cap)
    r = _size_cache.get(key)
    if r is None:
        if len(_size_cache) > 256:
            _size_cache.clear()
        al = lambda n: (int(n) + 255) & ~255  # noqa: E731
        r = _size_cache[key] = (al(lib.hgs_geom_bytes(P, H, W)), al(lib.hgs_img_bytes(H, W)),
                                al(lib.hgs_bin_bytes(cap)), al(lib.hgs_bwd_scratch_bytes(cap)))
    return r


def _drain_pending(st: _DeviceState, block: bool = False):
    """Inspect the status of earlier async forwards whose copy has landed."""
    while st.pending:
        p = st.pending[0]
        if block:
            p.event.synchronize()
        elif not p.event.query():
            break
        st.pending.pop(0)
        status = _read_status(st, p.slot)
        st.observe(status)
        if status[4]:
            st.capacity = max(st.capacity, _round_capacity(2 * status[0]))
            st.tile_hint = 0
            raise RuntimeError(
                "humangaussian_amd (async mode): an earlier render overflowed its buffers "
                f"(num_rendered={status[0]}, capacity={p.cap}, longest tile list={status[6]}, "
                f"hint={p.hint}); its outputs and gradients were invalid.  Capacity has been "
                "raised; re-run the step (or disable async mode).")


# ------------------------------------------------------------------------ autograd node

class _RasterizeGaussians(torch.autograd.Function):
    """Replaces upstream's `_RasterizeGaussians` (forward -> `_C.rasterize_gaussians`,
    backward -> `_C.rasterize_gaussians_backward`)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                cov3Ds_precomp, raster_settings, want_grad):
        lib = _lib.load()
        device = means3D.device
        if device.type != "cuda":
            raise RuntimeError("humangaussian_amd: tensors must live on a HIP device "
                               "(torch device type 'cuda'); there is no CPU path")
        P = int(means3D.shape[0])
        if P != 0 and (means3D.dim() != 2 or means3D.shape[1] != 3):
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        H, W = int(raster_settings.image_height), int(raster_settings.image_width)

        keep: list = []
        # only switch the current device when it is not already the tensors' device
        switch = torch.cuda.current_device() != (device.index if device.index is not None else 0)
        guard = torch.cuda.device(device) if switch else _NULL_CTX
        with guard:
            settings = _make_settings(raster_settings, device, keep)
            m3 = _f32c(means3D, device)
            sh_ = _opt(sh); cp_ = _opt(colors_precomp)
            sc_ = _opt(scales); ro_ = _opt(rotations); cv_ = _opt(cov3Ds_precomp)
            sh_ = None if sh_ is None else _f32c(sh_, device)
            cp_ = None if cp_ is None else _f32c(cp_, device)
            sc_ = None if sc_ is None else _f32c(sc_, device)
            ro_ = None if ro_ is None else _f32c(ro_, device)
            cv_ = None if cv_ is None else _f32c(cv_, device)
            op_ = _f32c(opacities, device)
            M = int(sh_.shape[1]) if sh_ is not None else 0
            if sh_ is not None and (sh_.dim() != 3 or sh_.shape[0] != P or sh_.shape[2] != 3):
                raise RuntimeError("shs must have dimensions (num_points, M, 3)")

            f32 = torch.float32
            color = torch.empty((3, H, W), dtype=f32, device=device)
            depth = torch.empty((1, H, W), dtype=f32, device=device)
            alpha = torch.empty((1, H, W), dtype=f32, device=device)
            radii = torch.empty((P,), dtype=torch.int32, device=device)

            st = _state(device)
            stream_h = torch.cuda.current_stream(device).cuda_stream
            if st.pending:
                _drain_pending(st)
            go_async = bool(_async_mode[0] and want_grad and P > 0 and st.synced_calls >= 2)
            if go_async:
                cap = max(st.capacity, _round_capacity(2 * st.max_R))
                hint = max(1024, 2 * st.max_tile + 64)
            else:
                cap = max(st.capacity, _round_capacity(4 * P)) if P > 0 else 0
                hint = st.tile_hint
            # one allocation for the four opaque regions [geom | img | bin | backward rows]
            # (the fork keeps three such byte tensors for its backward); a capacity retry
            # re-allocates only the last two.
            g_sz, i_sz, b_sz, s_sz = _sizes(lib, P, H, W, cap)
            work = torch.empty(g_sz + i_sz + b_sz + (s_sz if want_grad else 0), dtype=torch.uint8,
                               device=device)
            base = work.data_ptr()
            geom_p, img_p, bin_p, scr_p = base, base + g_sz, base + g_sz + i_sz, base + g_sz + i_sz + b_sz
            work2 = None
            status = None
            bwd = None
            vp = ctypes.c_void_p
            for _ in range(4):
                slot = st.next_slot()
                rc = lib.hgs_forward(
                    ctypes.byref(settings), P, M, _ptr(m3), _ptr(sh_), _ptr(cp_), _ptr(op_),
                    _ptr(sc_), _ptr(ro_), _ptr(cv_), _ptr(color), _ptr(depth), _ptr(alpha),
                    _ptr(radii), vp(geom_p), vp(bin_p), cap, vp(img_p),
                    1 if want_grad else 0, hint, vp(st.status_ptr[slot]),
                    1,      # torch pinned memory is device-mapped on ROCm: direct kernel store
                    None if go_async else vp(st.status_event.cuda_event),
                    _stage_events["fwd"], vp(stream_h))
                if rc == -2:
                    raise RuntimeError("inconsistent optional inputs (shs/colors_precomp, "
                                       "scales+rotations/cov3D_precomp)")
                _check(rc, "hgs_forward")
                # Host work that does not depend on the result runs HERE, while the GPU is
                # busy with the forward: everything the backward call will need.
                if want_grad and bwd is None:
                    new = lambda *shape: torch.empty(shape, dtype=f32, device=device)  # noqa: E731
                    bwd = dict(
                        d_means3D=new(P, 3), d_means2D=new(P, 3), d_opac=new(*opacities.shape),
                        d_sh=new(P, M, 3) if sh_ is not None else None,
                        d_cp=new(P, 3) if cp_ is not None else None,
                        d_sc=new(P, 3) if sc_ is not None else None,
                        d_ro=new(P, 4) if sc_ is not None else None,
                        d_cv=new(P, 6) if cv_ is not None else None,
                        settings=settings, keep=keep)
                if go_async:
                    p = _Pending()
                    p.event, p.slot, p.cap, p.hint = torch.cuda.Event(), slot, cap, hint
                    p.event.record()
                    st.pending.append(p)
                    if len(st.pending) >= st.RING - 1:     # never let the ring wrap
                        _drain_pending(st, block=True)
                    break
                # One host wait per forward, like upstream's blocking read of num_rendered -
                # but only for the status (published right after the scan stage): fill, sort
                # and blend are already enqueued and keep running while the host goes on.
                st.status_event.synchronize()
                status = _read_status(st, slot)
                if not status[4]:
                    break
                if status[4] & 1:                     # R exceeded the capacity: grow, re-run
                    cap = _round_capacity(int(status[0] * 1.25) + 1)
                    _, _, b_sz, s_sz = _sizes(lib, P, H, W, cap)
                    work2 = torch.empty(b_sz + (s_sz if want_grad else 0), dtype=torch.uint8,
                                        device=device)
                    bin_p = work2.data_ptr()
                    scr_p = bin_p + b_sz
                if status[4] & 2:                     # a tile list outgrew the hint
                    hint = 0
            else:
                raise RuntimeError("libhgs_rast: entry capacity did not converge")
            if status is not None:
                st.observe(status)
                st.synced_calls += 1
                st.capacity = max(st.capacity, cap)
                st.tile_hint = max(1024, int(status[6] * 1.5) + 64)
            if want_grad:
                bwd["cap"] = cap
                bwd["work"] = (work, work2)           # keeps the regions alive until backward
                bwd["ptrs"] = (geom_p, bin_p, img_p, scr_p)
                if status is not None:
                    hs = HgsStatus()
                    (hs.num_rendered, hs.active_tiles, hs.num_buckets, hs.bwd_groups,
                     hs.overflow) = status[:5]
                    hs.reserved[0], hs.reserved[1], hs.reserved[2] = status[5:8]
                    bwd["status"] = hs
                else:
                    bwd["status"] = None

        ctx.P, ctx.M = P, M
        if want_grad:
            ctx.bwd = bwd
            # inputs and outputs go through save_for_backward (version checks, no reference
            # cycle through the outputs); opaque work buffers ride in ctx.bwd
            opt = [t for t in (sh_, cp_, sc_, ro_, cv_) if t is not None]
            ctx.has = (sh_ is not None, cp_ is not None, sc_ is not None, cv_ is not None)
            ctx.save_for_backward(m3, op_, radii, color, depth, alpha, *opt)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        lib = _lib.load()
        saved = ctx.saved_tensors
        m3, op_, radii, color, depth, alpha = saved[:6]
        has_sh, has_cp, has_sr, has_cv = ctx.has
        it = iter(saved[6:])
        sh_ = next(it) if has_sh else None
        cp_ = next(it) if has_cp else None
        sc_ = next(it) if has_sr else None
        ro_ = next(it) if has_sr else None
        cv_ = next(it) if has_cv else None
        device = m3.device
        P, M = ctx.P, ctx.M
        b = ctx.bwd
        gc = None if grad_color is None else _f32c(grad_color, device)
        gd = None if grad_depth is None else _f32c(grad_depth, device)
        ga = None if grad_alpha is None else _f32c(grad_alpha, device)
        hs = b["status"]
        geom_p, bin_p, img_p, scr_p = b["ptrs"]
        vp = ctypes.c_void_p
        rc = lib.hgs_backward(
            ctypes.byref(b["settings"]), P, M, _ptr(m3), _ptr(sh_), _ptr(cp_), _ptr(op_),
            _ptr(sc_), _ptr(ro_), _ptr(cv_), _ptr(radii),
            _ptr(color), _ptr(depth), _ptr(alpha), _ptr(gc), _ptr(gd), _ptr(ga),
            vp(geom_p), vp(bin_p), vp(img_p), None if hs is None else ctypes.byref(hs),
            b["cap"], vp(scr_p),
            _ptr(b["d_means3D"]), _ptr(b["d_means2D"]), _ptr(b["d_sh"]), _ptr(b["d_cp"]),
            _ptr(b["d_opac"]), _ptr(b["d_sc"]), _ptr(b["d_ro"]), _ptr(b["d_cv"]),
            _stage_events["bwd"], vp(torch.cuda.current_stream(device).cuda_stream))
        _check(rc, "hgs_backward")
        ctx.bwd = None
        return (b["d_means3D"], b["d_means2D"], b["d_sh"], b["d_cp"], b["d_opac"], b["d_sc"],
                b["d_ro"], b["d_cv"], None, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                        cov3Ds_precomp, raster_settings):
    want_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad
        for t in (means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                  cov3Ds_precomp))
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales,
                                     rotations, cov3Ds_precomp, raster_settings, want_grad)


# --------------------------------------------------------------------------- the module

class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Frustum test (replaces `_C.mark_visible`)."""
        lib = _lib.load()
        with torch.no_grad():
            device = positions.device
            if device.type != "cuda":
                raise RuntimeError("humangaussian_amd: tensors must live on a HIP device")
            keep: list = []
            with torch.cuda.device(device):
                settings = _make_settings(self.raster_settings, device, keep)
                pos = _f32c(positions, device)
                P = int(pos.shape[0])
                present = torch.zeros((P,), dtype=torch.uint8, device=device)
                rc = lib.hgs_mark_visible(ctypes.byref(settings), P, _ptr(pos), _ptr(present),
                                          ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream))
                _check(rc, "hgs_mark_visible")
            return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or '
                            'precomputed 3D covariance!')
        empty = torch.Tensor([])
        if shs is None:
            shs = empty
        if colors_precomp is None:
            colors_precomp = empty
        if scales is None:
            scales = empty
        if rotations is None:
            rotations = empty
        if cov3D_precomp is None:
            cov3D_precomp = empty
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales,
                                   rotations, cov3D_precomp, raster_settings)
