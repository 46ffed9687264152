"""Build + load the two in-tree shared objects:

  libhgs_rast.so   the C-ABI HIP library (include/hgs_rast.h), hipcc for gfx950; bound here
                   through ctypes (`load()`: sizing functions, raw-ABI tests)
  _hgs_torch.so    the torch binding above it (csrc/torch_binding.cpp, g++, no device code):
                   the C++ autograd node the Python API calls (`load_binding()`)

There is no JIT cache and no fallback: if a shared object is missing or does not export the
ABI, importing the rasterizer fails loudly.
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
ABI_VERSION = 16

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
        ("num_rendered", c_uint32), ("active_tiles", c_uint32), ("num_pairs", c_uint32),
        ("bwd_groups", c_uint32), ("overflow", c_uint32), ("reserved", c_uint32 * 3),
    ]


EXPORTS = {
    "hgs_abi_version": (ctypes.c_int, []),
    "hgs_geom_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "hgs_bin_bytes": (c_size_t, [c_int64]),
    "hgs_img_bytes": (c_size_t, [c_int32, c_int32]),
    "hgs_bwd_scratch_bytes": (c_size_t, [c_int64]),
    "hgs_bwd_scratch_bytes_pairs": (c_size_t, [c_int64, c_int64]),
    "hgs_geom_bytes_batch": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "hgs_img_bytes_batch": (c_size_t, [c_int32, c_int32, c_int32]),
    "hgs_forward_batch": (ctypes.c_int, [POINTER(HgsSettings), c_int32, c_int32, c_int32] + [c_void_p] * 7
                          + [c_void_p] * 4 + [c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int32,
                                              c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "hgs_backward_batch": (ctypes.c_int, [POINTER(HgsSettings), c_int32, c_int32, c_int32] + [c_void_p] * 8
                           + [c_void_p] * 6 + [c_void_p] * 3 + [POINTER(HgsStatus), c_int64, c_void_p]
                           + [c_void_p] * 8 + [c_void_p, c_void_p]),
    "hgs_forward_batch_act": (ctypes.c_int, [POINTER(HgsSettings), c_int32, c_int32, c_int32] + [c_void_p] * 7
                              + [c_void_p] * 4 + [c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int32,
                                                  c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p]),
    "hgs_forward_batch_act_leaf": (ctypes.c_int, [POINTER(HgsSettings), c_int32, c_int32, c_int32] + [c_void_p] * 7
                                   + [c_void_p] * 4 + [c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int32,
                                                       c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "hgs_backward_batch_act": (ctypes.c_int, [POINTER(HgsSettings), c_int32, c_int32, c_int32] + [c_void_p] * 8
                               + [c_void_p] * 6 + [c_void_p] * 3 + [POINTER(HgsStatus), c_int64, c_void_p]
                               + [c_void_p] * 8 + [c_void_p, c_int32, c_void_p]),
    "hgs_forward": (ctypes.c_int, [POINTER(HgsSettings), c_int32, c_int32] + [c_void_p] * 7
                    + [c_void_p] * 4 + [c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int32,
                                        c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "hgs_backward": (ctypes.c_int, [POINTER(HgsSettings), c_int32, c_int32] + [c_void_p] * 8
                     + [c_void_p] * 6 + [c_void_p] * 3 + [POINTER(HgsStatus), c_int64, c_void_p]
                     + [c_void_p] * 8 + [c_void_p, c_void_p]),
    "hgs_mark_visible": (ctypes.c_int, [POINTER(HgsSettings), c_int32, c_void_p, c_void_p,
                                        c_void_p]),
    "hgs_knn_mean_dist2": (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_void_p]),
    "hgs_knn_scratch_bytes": (c_size_t, [c_int32]),
    "hgs_knn_mean_dist2_grid": (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hgs_reduce_view_packs": (ctypes.c_int, [c_int32, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "hgs_reduce_view_packs_acc": (ctypes.c_int, [c_int32, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hgs_pack_view_contribution": (ctypes.c_int, [c_int32, c_int32] + [c_void_p] * 9),
    "hgs_backward_batch_packed": (ctypes.c_int, [POINTER(HgsSettings), c_int32, c_int32, c_int32] + [c_void_p] * 5
                                  + [c_void_p] * 7 + [c_void_p] * 3 + [POINTER(HgsStatus), c_int64, c_void_p]
                                  + [c_void_p] * 2 + [c_void_p, c_int32, c_void_p]),
    "hgs_reduce_view_packs_unpack": (ctypes.c_int, [c_int32, c_int64, c_int32] + [c_void_p] * 9 + [c_void_p]),
    "hgs_densify_stats": (ctypes.c_int, [c_int32, c_int32] + [c_void_p] * 9),
    "hgs_densify_masks": (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p]
                          + [c_float] * 6 + [c_void_p] * 5),
    "hgs_compact_scratch_bytes": (c_size_t, [c_int32]),
    "hgs_compact_index": (ctypes.c_int, [c_int32] + [c_void_p] * 5),
    "hgs_gather_rows": (ctypes.c_int, [c_int64, c_int32] + [c_void_p] * 4),
    "hgs_reanchor": (ctypes.c_int, [c_int32] + [c_void_p] * 7),
}


BINDING_SRC = "torch_binding.cpp"
BINDING_PATH = os.path.join(_PKG_DIR, "_hgs_torch.so")


def sources():
    return [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f != BINDING_SRC]


def needs_build() -> bool:
    target = os.path.join(_PKG_DIR, "libhgs_rast.so")
    if not os.path.exists(target):
        return True
    mt = os.path.getmtime(target)
    deps = sources() + [os.path.join(_PKG_DIR, "..", "include", "hgs_rast.h")]
    return any(os.path.getmtime(s) > mt for s in deps)


# translation units: (source, extra flags)
UNITS = [("api.hip", []),
         ("render_bwd.hip", [])]


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


def binding_needs_build() -> bool:
    if not os.path.exists(BINDING_PATH):
        return True
    mt = os.path.getmtime(BINDING_PATH)
    deps = [os.path.join(_CSRC, BINDING_SRC), os.path.join(_PKG_DIR, "..", "include", "hgs_rast.h")]
    return any(os.path.getmtime(s) > mt for s in deps)


def build_binding(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/torch_binding.cpp with g++ against this interpreter's torch and link it to
    libhgs_rast.so (rpath $ORIGIN).  Host code only: hipcc is not involved."""
    if not force and not binding_needs_build():
        return BINDING_PATH
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("g++ not found: cannot build _hgs_torch.so")
    tlib = ce.library_paths()[0]
    rocm_inc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=_hgs_torch",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += [f"-I{p}" for p in ce.include_paths()] + [f"-I{rocm_inc}", f"-I{sysconfig.get_paths()['include']}"]
    cmd += [os.path.join(_CSRC, BINDING_SRC), "-o", BINDING_PATH + ".tmp", f"-L{tlib}", "-lc10", "-lc10_hip",
            "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python", f"-L{_PKG_DIR}", "-lhgs_rast",
            f"-Wl,-rpath,{tlib}", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("g++ failed for torch_binding.cpp:\n" + res.stdout + res.stderr)
    os.replace(BINDING_PATH + ".tmp", BINDING_PATH)
    return BINDING_PATH


_binding = None


def load_binding():
    """Import humangaussian_amd/_hgs_torch.so (the C++ autograd node).  Fails loudly."""
    global _binding
    if _binding is not None:
        return _binding
    load()                                   # ABI check of libhgs_rast.so first
    if not os.path.exists(BINDING_PATH):
        raise ImportError(
            f"{BINDING_PATH} is missing: the torch binding has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback for the product path.")
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded before the extension)
    spec = importlib.util.spec_from_file_location("humangaussian_amd._hgs_torch", BINDING_PATH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if mod.abi_version() != ABI_VERSION:
        raise ImportError(f"{BINDING_PATH}: linked against ABI {mod.abi_version()} != {ABI_VERSION}")
    _binding = mod
    return mod


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
