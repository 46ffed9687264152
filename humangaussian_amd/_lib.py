"""Build + load libhgs_rast.so (the C-ABI HIP library, include/hgs_rast.h) through ctypes.

The library is built IN-TREE (humangaussian_amd/libhgs_rast.so) by one hipcc command for
gfx950; there is no JIT cache and no fallback: if the shared object is missing or does not
export the ABI, importing the rasterizer fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
from ctypes import POINTER, Structure, c_float, c_int32, c_int64, c_size_t, c_uint32, c_void_p

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG_DIR, "csrc")
LIB_PATH = os.environ.get("HGS_LIB") or os.path.join(_PKG_DIR, "libhgs_rast.so")   # HGS_LIB: A/B experiments only
ABI_VERSION = 6

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
               "-shared"]


class HgsSettings(Structure):
    """ctypes mirror of `hgs_settings` (field order = GaussianRasterizationSettings)."""
    _fields_ = [
        ("image_height", c_int32), ("image_width", c_int32),
        ("tanfovx", c_float), ("tanfovy", c_float),
        ("bg", c_void_p), ("scale_modifier", c_float),
        ("viewmatrix", c_void_p), ("projmatrix", c_void_p),
        ("sh_degree", c_int32), ("campos", c_void_p),
        ("prefiltered", c_int32), ("debug", c_int32),
    ]


class HgsStatus(Structure):
    _fields_ = [
        ("num_rendered", c_uint32), ("active_tiles", c_uint32), ("num_buckets", c_uint32),
        ("bwd_groups", c_uint32), ("overflow", c_uint32), ("reserved", c_uint32 * 3),
    ]


EXPORTS = {
    "hgs_abi_version": (ctypes.c_int, []),
    "hgs_geom_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "hgs_bin_bytes": (c_size_t, [c_int64]),
    "hgs_img_bytes": (c_size_t, [c_int32, c_int32]),
    "hgs_bwd_scratch_bytes": (c_size_t, [c_int64]),
    "hgs_forward": (ctypes.c_int, [POINTER(HgsSettings), c_int32, c_int32] + [c_void_p] * 7
                    + [c_void_p] * 4 + [c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int32,
                                        c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "hgs_backward": (ctypes.c_int, [POINTER(HgsSettings), c_int32, c_int32] + [c_void_p] * 8
                     + [c_void_p] * 6 + [c_void_p] * 3 + [POINTER(HgsStatus), c_int64, c_void_p]
                     + [c_void_p] * 8 + [c_void_p, c_void_p]),
    "hgs_mark_visible": (ctypes.c_int, [POINTER(HgsSettings), c_int32, c_void_p, c_void_p,
                                        c_void_p]),
}


def sources():
    return [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC))]


def needs_build() -> bool:
    target = os.path.join(_PKG_DIR, "libhgs_rast.so")
    if not os.path.exists(target):
        return True
    mt = os.path.getmtime(target)
    deps = sources() + [os.path.join(_PKG_DIR, "..", "include", "hgs_rast.h")]
    return any(os.path.getmtime(s) > mt for s in deps)


# translation units: (source, extra flags)
UNITS = [("api.hip", []),
         ("render_bwd.hip", ["-fno-slp-vectorize"])]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the translation units for gfx950 with hipcc and link libhgs_rast.so."""
    target = os.path.join(_PKG_DIR, "libhgs_rast.so")
    if not force and not needs_build():
        return target
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libhgs_rast.so")
    common = [f for f in HIPCC_FLAGS if f != "-shared"]
    objs = []
    for src, extra in UNITS:
        obj = os.path.join(_PKG_DIR, "_build_" + src.replace(".hip", ".o"))
        cmd = [hipcc] + common + extra + ["-c", os.path.join(_CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", target + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(target + ".tmp", target)
    for o in objs:
        os.remove(o)
    return target


_lib = None


def load() -> ctypes.CDLL:
    """dlopen the library and bind every symbol include/hgs_rast.h declares."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP rasterizer extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback for the product path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise ImportError(f"{LIB_PATH} does not export {name}")
        fn.restype = res
        fn.argtypes = args
    if lib.hgs_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.hgs_abi_version()} != {ABI_VERSION}")
    _lib = lib
    return lib
