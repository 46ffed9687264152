"""Do A/B builds of the library give the same bits?  Forward + backward through the raw ABI on two scenes (a small
cloud with lists > 1024 entries: segments, transmittance products; a wider one), fingerprint of every output and
of the saved state, for the in-tree build and every library given on the command line:
  python tools/cmp_variant.py variants/X/libhgs_rast.so [...]            (on the GPU box)"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
from abi_runner import RawCall
from helpers import make_scene
from humangaussian_amd import _lib


def h(t):
    return hashlib.sha1(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]


def fingerprint():
    out = []
    for name, kw, op in (("long", dict(P=2600, seed=77, H=32, W=32, spread=0.02, scale=0.01, dist=2.0), 0.03),
                         ("wide", dict(P=6000, seed=5, H=128, W=160, spread=0.3, scale=0.05), None)):
        sc = make_scene(**kw)
        if op is not None:
            sc["opacities"] = torch.full_like(sc["opacities"], op)
        rc = RawCall(sc, capacity=1 << 18)
        assert rc.forward() == 0, rc.status
        out.append((name, "status", tuple(rc.status[:4]) + tuple(rc.status[6:8])))
        out.append((name, "outputs", h(rc.color), h(rc.depth), h(rc.alpha), h(rc.radii), h(rc.img)))
        g = torch.Generator().manual_seed(1)
        gc, gd, ga = (torch.randn(s, generator=g) for s in ((3, rc.H, rc.W), (1, rc.H, rc.W), (1, rc.H, rc.W)))
        grads = rc.backward(gc, gd, ga)
        out.append((name, "grads") + tuple(h(v) for v in grads.values() if v is not None))
    return out


ref = fingerprint()
for r in ref:
    print("in-tree", r)
same = True
for path in sys.argv[1:]:
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    got = fingerprint()
    ok = got == ref
    same = same and ok
    print(path, "IDENTICAL" if ok else "DIFFERENT")
    if not ok:
        for a, b in zip(ref, got):
            if a != b:
                print("  ", a, "\n  ", b)
sys.exit(0 if same else 1)
