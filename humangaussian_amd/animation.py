"""Per-frame re-anchoring of an avatar's Gaussians on the posed SMPL-X mesh, on the device
(SURVEY.md 8(f)-4).  The reference recomputes the positions with numpy on the CPU and uploads them
every frame (/root/reference/animation.py:384-403, `.cuda()` at :403; the D2H of the rendered frame
follows at :484); here the mapping stays resident and one HIP pass writes `xyz`."""
from __future__ import annotations

import torch

from . import _lib


class MeshAnchoredGaussians:
    """Holds the static mapping (face index, barycentric uvw, signed distance along the face normal:
    animation.py:339-345) on the device; `positions(vertices)` -> (P,3) for a posed mesh."""

    def __init__(self, faces, mapping_face, mapping_uvw, mapping_dist, device="cuda"):
        dev = torch.device(device)
        self.faces = torch.as_tensor(faces).to(dev, torch.int32).contiguous()
        self.mapping_face = torch.as_tensor(mapping_face).to(dev, torch.int32).contiguous()
        self.mapping_uvw = torch.as_tensor(mapping_uvw).to(dev, torch.float32).contiguous()
        self.mapping_dist = torch.as_tensor(mapping_dist).to(dev, torch.float32).contiguous()

    def positions(self, vertices) -> torch.Tensor:
        v = torch.as_tensor(vertices).to(self.faces.device, torch.float32).contiguous()
        return _lib.load_binding().reanchor(v, self.faces, self.mapping_face, self.mapping_uvw, self.mapping_dist)
