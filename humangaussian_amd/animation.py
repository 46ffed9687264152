"""The animation path (BASELINE.json configs[4]): per frame re-anchor the avatar's Gaussians
on the posed body mesh, render forward-only, hand the frame to the consumer - frames sharded over the GPUs of a node.

Reference: /root/reference/animation.py
  :384-403   per frame: closest points + signed distance along the face normal -> new xyz, computed with numpy on the
             CPU and uploaded (`.cuda()` at :403)                      -> `MeshAnchoredGaussians.positions` (one HIP pass,
                                                                          the mapping stays resident: hgs_reanchor)
  :477-484   `render_gs`: MiniCam -> `Renderer.render` -> image `.detach().cpu()`   -> `AvatarAnimator.render_frame`
  :966-1004  the frame loop (azimuth i % 360, elevation 0, radius 2, fovy 50; pose i)  -> `render_frames_parallel`:
             frame i belongs to rank i mod G (frames are independent units: no data-path collective is NEEDED; the
             optional image all-gather, one per round of G frames, runs under the render of the next round)

The host logic (`render_frames_parallel`) takes the per-frame render as a callable, so it is testable on CPU with gloo; the
product pieces (`MeshAnchoredGaussians`, `AvatarAnimator`) run on the HIP rasterizer only - there is no CPU fallback.
"""
from __future__ import annotations

import math
import os
from typing import Callable, Iterator, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .view_parallel import _Gather, shard_views


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


# ------------------------------------------------------------------------------ the motion that drives the frames

# toy skeleton on the body mesh (`synth.human_mesh()`: A-pose, z-up, left = +x): SMPL-X joint index,
# centre, and the smooth weight of the part that follows the joint.  Parents before children.
_JOINTS = (
    # name, SMPL-X joint, parent, centre (x, y, z)
    ("l_shoulder", 16, None, (0.20, 0.0, 0.50)), ("l_elbow", 18, "l_shoulder", (0.40, 0.0, 0.30)),
    ("r_shoulder", 17, None, (-0.20, 0.0, 0.50)), ("r_elbow", 19, "r_shoulder", (-0.40, 0.0, 0.30)),
    ("l_hip", 1, None, (0.08, 0.0, -0.05)), ("l_knee", 4, "l_hip", (0.10, -0.03, -0.40)),
    ("r_hip", 2, None, (-0.08, 0.0, -0.05)), ("r_knee", 5, "r_hip", (-0.10, -0.03, -0.40)),
    ("head", 15, None, (0.0, 0.0, 0.58)),
)


def _smooth(t):
    t = t.clamp(0.0, 1.0)
    return t * t * (3.0 - 2.0 * t)


def _part_weights(v: torch.Tensor):
    x, z = v[:, 0], v[:, 2]
    left, right = _smooth(x / 0.03 * 0.5 + 0.5), _smooth(-x / 0.03 * 0.5 + 0.5)
    arm_l, arm_r = _smooth((x - 0.17) / 0.06), _smooth((-x - 0.17) / 0.06)
    return {
        "l_shoulder": arm_l, "l_elbow": _smooth((x - 0.37) / 0.06),
        "r_shoulder": arm_r, "r_elbow": _smooth((-x - 0.37) / 0.06),
        "l_hip": _smooth((-0.02 - z) / 0.08) * left * (1 - arm_l), "l_knee": _smooth((-0.38 - z) / 0.06) * left * (1 - arm_l),
        "r_hip": _smooth((-0.02 - z) / 0.08) * right * (1 - arm_r), "r_knee": _smooth((-0.38 - z) / 0.06) * right * (1 - arm_r),
        "head": _smooth((z - 0.56) / 0.05) * (1 - arm_l) * (1 - arm_r),
    }


def _rodrigues(a: np.ndarray) -> np.ndarray:
    th = float(np.linalg.norm(a))
    if th < 1e-12:
        return np.eye(3)
    k = a / th
    K = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return np.eye(3) + math.sin(th) * K + (1.0 - math.cos(th)) * (K @ K)


class MotionDriver:
    """Posed body-mesh vertices per frame, DRIVEN BY the reference's demo motion `content/amass_test_17.npz` (136 frames
    of 55 SMPL-X joint rotations) where the LOCAL asset humangaussian_amd/data/amass_test_17_poses.npz exists (built from
    the reference tree by data/make_motion.py; AMASS data, not redistributed), by a clip of the caller (`poses_path`: an
    .npz with `poses` (F, 55, 3)), or by a procedural sway of the same nine joints with the same period (`self.source` says
    which).  The SMPL-X model files that turn poses into vertices are not in the reference tree, so the angles of nine
    joints (shoulders, elbows, hips, knees, head) articulate the body mesh through smooth part weights instead - a toy
    skinning that moves the surface the way the real sequence moves its body (amplitude `gain`), which is all the
    rasterize path sees of it: `xyz` changes every frame, nothing else does (animation.py:384-403).  Frame i uses pose
    i mod 136 (animation.py:311-330)."""

    def __init__(self, vertices, device="cuda", gain: float = 0.6, poses_path: Optional[str] = None):
        self.device = torch.device(device)
        self.rest = torch.as_tensor(vertices, dtype=torch.float32).to(self.device)
        self.weights = _part_weights(self.rest)
        from . import data
        path = poses_path or data.MOTION
        if poses_path and not os.path.exists(poses_path):
            raise FileNotFoundError(poses_path)
        self.poses = np.load(path)["poses"].astype(np.float64) if os.path.exists(path) else None
        self.source = ("procedural sway (no motion asset)" if self.poses is None else
                       (os.path.basename(poses_path) if poses_path else "content/amass_test_17.npz poses (local asset)"))
        self.gain = gain
        self.num_poses = 136 if self.poses is None else int(self.poses.shape[0])

    def _angles(self, i: int):
        if self.poses is not None:
            return self.poses[i % self.num_poses]
        ph = 2.0 * math.pi * (i % self.num_poses) / self.num_poses
        a = np.zeros((55, 3))
        for j, (_, sj, _, _) in enumerate(_JOINTS):
            a[sj] = 0.4 * np.array([math.sin(ph + j), math.cos(2 * ph + j), math.sin(3 * ph + 2 * j)])
        return a

    def vertices(self, i: int) -> torch.Tensor:
        ang = self._angles(i)
        rest_c = {n: np.asarray(ce, np.float64) for n, _, _, ce in _JOINTS}
        Rw, cw, pack = {}, {}, np.zeros((len(_JOINTS), 4, 3), np.float32)      # per joint: (R_applied^T - I) rows, centre
        for j, (name, sj, parent, _) in enumerate(_JOINTS):                     # (host: nine 3x3 products, one upload)
            a = ang[sj] * self.gain
            # SMPL-X is y-up, the mesh z-up (poser.py:349-352 swaps y and z: a reflection, so the axis flips sign)
            Rl = _rodrigues(np.array([-a[0], -a[2], -a[1]]))
            if parent is None:
                Rp, c = np.eye(3), rest_c[name]
            else:                                          # the child's centre and frame ride on the parent
                Rp = Rw[parent]
                c = cw[parent] + Rp @ (rest_c[name] - rest_c[parent])
            Ra = Rp @ Rl @ Rp.T                            # the joint's rotation, applied in the posed frame about c
            Rw[name], cw[name] = Rp @ Rl, c
            pack[j, :3], pack[j, 3] = Ra.T - np.eye(3), c
        pk = torch.from_numpy(pack).to(self.device, non_blocking=True)
        v = self.rest.clone()
        for j, (name, _, _, _) in enumerate(_JOINTS):
            v = v + self.weights[name][:, None] * ((v - pk[j, 3]) @ pk[j, :3])
        return v


def human_mesh_anchors(n: int, seed: int = 0, device="cuda", max_dist: float = 0.004):
    """The body mesh (`synth.human_mesh()`: the reference's human.obj where the local asset exists, the procedural capsule
    mesh otherwise) + n Gaussians anchored on it the way animation.py:339-345 anchors a trained avatar: a face, barycentric
    coordinates, a signed distance along the face normal (area-uniform faces, |dist| <= max_dist).
    Returns (vertices (V,3) float32 numpy, MeshAnchoredGaussians)."""
    from . import synth
    mv, mf, _ = synth.human_mesh()
    v, f = mv.astype(np.float64), mf
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    rng = np.random.default_rng(seed)
    tri = np.searchsorted(np.cumsum(area) / area.sum(), rng.uniform(0, 1, n), side="right").clip(0, len(f) - 1)
    r1, r2 = np.sqrt(rng.uniform(0, 1, n)), rng.uniform(0, 1, n)
    uvw = np.stack([1.0 - r1, r1 * (1.0 - r2), r1 * r2], 1).astype(np.float32)
    dist_ = rng.uniform(-max_dist, max_dist, n).astype(np.float32)
    return mv, MeshAnchoredGaussians(f, tri.astype(np.int32), uvw, dist_, device=device)


# ------------------------------------------------------------------------------ one frame, and the sharded loop

class AvatarAnimator:
    """Re-anchor + render of ONE frame on this rank's GPU: `render_frame(vertices, camera)` -> image (3,H,W) in [0,1]
    (`Renderer.render`: gs_renderer.py:923-1028 incl. its clamp), no autograd graph, the frame never leaves the device
    unless the caller moves it.  `gaussians` is any object with the reference GaussianModel getters and a writable
    `_xyz` (animation.py:403 assigns `self.gs.gaussians._xyz`)."""

    def __init__(self, gaussians, anchors: MeshAnchoredGaussians, white_background: bool = True, device="cuda"):
        from .renderer import Renderer
        self.gaussians, self.anchors = gaussians, anchors
        self.renderer = Renderer(gaussians, white_background=white_background, device=device)

    @torch.no_grad()
    def render_frame(self, vertices, camera) -> torch.Tensor:
        self.gaussians._xyz = self.anchors.positions(vertices)               # animation.py:384-403, on the device
        return self.renderer.render(camera)["image"]                         # animation.py:477-482


def orbit_frame_camera(i: int, H: int, W: int, radius: float = 2.0, fovy_deg: float = 50.0, device="cuda"):
    """Camera of frame i as the reference's save loop sets it: elevation 0, azimuth i mod 360, radius 2, fovy 50
    (animation.py:936-945 defaults, :993-1000).  Host-side matrices, one small upload (renderer.cameras_from_c2w)."""
    from . import synth
    from .renderer import cameras_from_c2w
    return cameras_from_c2w(synth.c2w_orbit(0.0, float(i % 360), radius)[None], math.radians(fovy_deg), H, W, device=device)[0]


def render_frames_parallel(frames: Sequence[int], render_frame_fn: Callable[[int], torch.Tensor], group=None,
                           gather: bool = True) -> Iterator[Tuple[int, torch.Tensor]]:
    """Frames of an animation sharded over the ranks of `group`: frame number k of the list belongs to rank k mod G
    (`view_parallel.shard_views`), every rank renders only its own with `render_frame_fn(frame) -> (C,H,W)` tensor.

    gather=True   yields (frame, image) for EVERY frame, in list order, on every rank: round j = the frames j*G .. j*G+G-1,
                  one per rank; the round's images travel in ONE asynchronous all-gather (RCCL's own stream; gloo in the
                  CPU tests) that runs while round j+1 is rendered - only the last round's collective is exposed;
    gather=False  yields only this rank's frames, in order, and uses no collective at all (each rank writes its own frames).
    Works without an initialised process group (one rank)."""
    frames = list(frames)
    on = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if on else 0
    world = dist.get_world_size(group) if on else 1
    if not gather or world == 1:
        for k in shard_views(len(frames), rank, world):
            yield frames[k], render_frame_fn(frames[k])
        return
    if len(frames) < world:
        # checked on EVERY rank before the first collective (the same frame list and world size everywhere): a rank
        # without any frame could not size its slab, and raising only there would leave the others inside the all-gather
        raise ValueError(f"render_frames_parallel(gather=True): {len(frames)} frames for {world} ranks - pass at least "
                         f"one frame per rank, or gather=False")
    rounds = (len(frames) + world - 1) // world
    pending, shape_like = None, None

    def drain(j, g):
        allimg = g.result()                                   # (world, C, H, W)
        for r in range(world):
            k = j * world + r
            if k < len(frames):
                yield frames[k], allimg[r]
    for j in range(rounds):
        k = j * world + rank
        if k < len(frames):
            img = render_frame_fn(frames[k]).contiguous()
            shape_like = img
        else:                                                 # ragged last round: a zero frame nobody yields
            img = torch.zeros_like(shape_like)                #  (len(frames) >= world: this rank rendered round 0)
        started = _Gather(img, group, async_op=True)
        if pending is not None:
            yield from drain(j - 1, pending)
        pending = started
    if pending is not None:
        yield from drain(rounds - 1, pending)
