"""Single-GPU estimate of the per-step view-parallel overhead at world=8 (pack + local reduction;
the all-gather itself needs 8 GPUs)."""
import time, torch, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from humangaussian_amd import view_parallel as vp, _lib
P = 100000
dev = "cuda"
grads = {"means3D": torch.randn(P, 3, device=dev), "means2D": torch.randn(P, 3, device=dev), "shs": torch.randn(P, 1, 3, device=dev),
         "opacities": torch.randn(P, 1, device=dev), "scales": torch.randn(P, 3, device=dev), "rotations": torch.randn(P, 4, device=dev)}
radii = torch.randint(0, 50, (P,), dtype=torch.int32, device=dev)
gathered = torch.randn(8, P, 18, device=dev)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("pack us", t(lambda: vp.pack_contribution(grads, radii)))
print("fused reduce (world 8) us", t(lambda: _lib.load_binding().reduce_view_packs(gathered)))
def loop():
    total = gathered[0].clone()
    for r in range(1, 8):
        total[:, :-1] += gathered[r][:, :-1]
        total[:, -1] = torch.maximum(total[:, -1], gathered[r][:, -1])
    return total
print("torch loop reduce (world 8) us", t(loop))
