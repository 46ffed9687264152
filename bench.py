#!/usr/bin/env python
"""bench.py - rasterize fwd+bwd Gaussians/s @1024^2, 100k points (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  N > 1 without a launcher (WORLD_SIZE unset): this script re-executes itself under
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`
  (one rank per GPU over RCCL); under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.

A "step" = one pass of the hot path per rank: forward + backward of V (--views-per-rank, default 1) 1024^2
views of the 100k-Gaussian SMPL-X-like cloud, one after the other through the reference-compatible API
(GaussianRasterizer -> libhgs_rast.so: the reference's loop, GaussianDreamer.py:244-266), inputs resident in
HBM, plus - for N>1 - the all-gather of the per-rank gradient packs, one per ROUND of views (view-parallel, weak
scaling), each overlapped with the render of the next round (view_parallel.py); with V = 1 there is one collective
and nothing to hide it behind.
value = P * V * N * K / t, t = max over ranks of the barrier-bracketed wall time of K steps.
In front of the W warm-up steps the process's first measurement runs un-timed: `--init-steps` first-use steps (capacity
estimates settle) and further steps until `--init-seconds` (2.0) of wall time have passed - a fresh box runs the same step
12-20 % slower for its first 0.6-1.0 s; reported as `init_run`.
The step's `means2D` leaf is built as renderer.render() builds it: storage from torch.empty, the reference's zeros written
by the forward's own kernel (ABI v16); `--torch-zero-means2d` / `--uninitialised-means2d` are the two older forms.
HGS_DIST_BACKEND=gloo + HGS_BENCH_SHARE_DEVICE=1 run the N > 1 branch with all ranks on ONE device (gloo stages the
packs through the host): the configuration of tests/test_gpu_multirank_one_gpu.py, not a measurement.

`--forward-only` is the animation leg (configs[4]): a step = one FRAME per rank (re-anchor on the posed mesh + no-grad render,
frame k on rank k mod N; N > 1: one image all-gather per step, in flight under the next frame's render); value = P * frames / t.

Extra objects on the JSON line:
  roofline      dominant kernel, timed live with HIP events recorded by the library on its launch stream
  cpu_baseline  the PyTorch CPU oracle, full fwd+bwd of the same view on the host cores (rank 0, N=1)
  extra         further measurements of the same path, same rules (rank 0, N=1): the 8-view BATCHED call
                (one launch set for the 8 views of a training step), the `init`-state
                cloud, configs[3] (500k Gaussians, SH degree 3), forward-only (configs[4] shape), the 300-frame
                animation loop, the drop-in render() on raw parameters (un-fused and with fuse_activations)
"""
import argparse
import json
import math
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

P_POINTS = 100_000
RES = 1024
SH_DEGREE = 0
HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy ceiling 6290
INIT_STEPS = 100               # un-timed first-use steps before the W warm-up steps (reported as init_steps): the
                               # estimates settle and the GPU leaves its idle clocks (some boxes need > 20 ms for that)
INIT_SECONDS = 2.0             # ... and the first measurement of a process keeps stepping until this much wall time has
                               # passed: a FRESH box runs the same step 12-20 % slower for its first 0.6-1.0 s (0.175-0.19
                               # ms, then 0.155-0.157 from one step to the next; a second process on the same box starts
                               # at 0.16 and is at 0.155 within 300 steps: host and GPU clocks ramping, tools/warmup_curve.py).
                               # With the driver's --steps 20 --warmup 5 the timed region would sit 20 ms into that ramp.
                               # Reported as init_seconds / init_steps_run; --init-seconds 0 switches it off.

FWD_STAGES = ["preprocess_fwd", "tiles", "fill", "sort", "render_fwd"]
BWD_STAGES = ["render_bwd", "pair_reduce", "preprocess_bwd"]
# upstream's stages: B1 = the blend backward = render_bwd + pair_reduce here
ROOFLINE_STAGES = {"preprocess_fwd": ["preprocess_fwd"], "tiles": ["tiles"], "fill": ["fill"], "sort": ["sort"],
                   "render_fwd": ["render_fwd"], "blend_bwd(render_bwd+pair_reduce)": ["render_bwd", "pair_reduce"],
                   "preprocess_bwd": ["preprocess_bwd"]}


def path_bytes_survey(P, M, R, npix, T, B=1, forward_only=False):
    """SURVEY.md 8(d) / BASELINE.md 3: algorithmic bytes of the whole path per view."""
    if forward_only:
        return B * (P * (120 + 12 * M) + 24 * npix + 8 * T) + 80 * R
    return B * (P * (292 + 36 * M) + 52 * npix + 8 * T) + 164 * R


def algorithmic_bytes(stage, P, M, R, npix, T, B=1):
    """Per-launch algorithmic bytes, SURVEY.md 8(d) terms split by stage;
    B views per launch: per-Gaussian forward terms, entry terms and pixel terms scale with B (R is
    the batch total), the per-Gaussian parameter-gradient write of the backward does not."""
    return {
        "preprocess_fwd": B * P * (44 + 12 * M + 76),
        "tiles": 8 * T * B,
        "fill": B * P * 16 + 12 * R,
        "sort": 24 * R,
        "render_fwd": 44 * R + 24 * npix * B,
        "render_bwd": 44 * R + 28 * npix * B,     # records in, pixel gradients / outputs in (84 R for the
        "pair_reduce": 40 * R,                    # blend backward = 44 R in + 40 R of per-entry gradient out, written by the reduction)
        "preprocess_bwd": P * (60 + 12 * M + 44 + 12 * M) + B * P * (64 + 12) + 40 * R,
    }[stage]


def self_spawn(args):
    """`python bench.py --gpus N` with no launcher: become the launcher the driver would use."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execvpe(cmd[0], cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra measurements (batched, init, config 4)")
    ap.add_argument("--points", type=int, default=P_POINTS)
    ap.add_argument("--sh-degree", type=int, default=SH_DEGREE)
    ap.add_argument("--variant", default="mid", choices=["mid", "init"])
    ap.add_argument("--cloud", default="auto", choices=["auto", "human_obj", "capsule"],
                    help="where the points come from: area-uniform samples of the reference's load/shapes/human.obj (SURVEY.md 8(d); "
                         "a LOCAL asset built from the reference tree, humangaussian_amd/data) or the analytic capsule humanoid "
                         "with the same extents; auto = the mesh where the asset exists (the line says which)")
    ap.add_argument("--uninitialised-means2d", action="store_true",
                    help="hand the rasterizer an uninitialised means2D leaf (its values are never read) instead "
                         "of the zero-filled one the drop-in render() and the reference hand out")
    ap.add_argument("--torch-zero-means2d", action="store_true",
                    help="zero-fill the means2D leaf with torch.zeros (one fill launch per view: what render() did before ABI v16) "
                         "instead of inside the forward's per-Gaussian kernel")
    ap.add_argument("--views", type=int, default=1,
                    help="views per rank per step rendered by ONE batched call (extra measurement when > 1)")
    ap.add_argument("--views-per-rank", type=int, default=1,
                    help="views per rank per step rendered ONE AFTER THE OTHER (the reference's loop); with N > 1 the "
                         "collective of round k runs under the render of round k + 1")
    ap.add_argument("--init-steps", type=int, default=INIT_STEPS, help="un-timed first-use steps before the warm-up")
    ap.add_argument("--init-seconds", type=float, default=INIT_SECONDS,
                    help="the process's first measurement keeps stepping (un-timed) until this much wall time has passed")
    ap.add_argument("--collective", default="auto", choices=["auto", "allgather", "scatter"],
                    help="N > 1: how the per-rank gradient packs are reduced (view_parallel.allgather_reduce); auto times "
                         "both outside the timed region and uses the faster one")
    ap.add_argument("--forward-only", action="store_true",
                    help="the animation leg (configs[4]): a step = one frame per rank - re-anchor on the posed mesh + no-grad render - "
                         "frames sharded over the ranks, + one image all-gather per step for N > 1; reports frames/s")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)

    import torch
    import torch.distributed as dist
    from humangaussian_amd import (GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians_batch,
                                   synth)
    from humangaussian_amd import rasterizer as _rast
    from humangaussian_amd import view_parallel as vp

    args.cloud = synth.resolve_cloud_source(args.cloud)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    backend = os.environ.get("HGS_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if os.environ.get("HGS_BENCH_SHARE_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    VPR = max(1, args.views_per_rank)
    assert not (VPR > 1 and args.views > 1), "--views (batched call) and --views-per-rank (sequential calls) exclude each other"

    def max_over_ranks(values):
        t = torch.tensor(values, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t]

    def camera(i):
        return synth.orbit_camera(10.0, 30.0 + 45.0 * i, 1.75, 55.0, RES, RES)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    PARAMS = ("means3D", "shs", "opacities", "scales", "rotations")
    # the step's means2D leaf (see Workload.step)
    leaf_in_kernel = not (args.uninitialised_means2d or args.torch_zero_means2d)
    leaf_make = torch.zeros if args.torch_zero_means2d else torch.empty
    settled = {}               # filled by the first timed() of the process: the un-timed steps / seconds it ran first

    def init_phase(step_fn, init_steps):
        """The un-timed steps in front of the W warm-up steps: `init_steps` first-use steps and - once per process - further
        steps until INIT_SECONDS of wall time have passed (the same number on every rank: a step may hold a collective)."""
        t_init = time.perf_counter()
        for _ in range(init_steps):
            step_fn()
        fence()
        if init_steps and args.init_seconds > 0 and not settled:
            more = 0
            while True:        # (chunks of 200 steps; every rank takes the same decision: the slowest rank's clock)
                dt = time.perf_counter() - t_init
                if (max_over_ranks([dt])[0] if world > 1 else dt) >= args.init_seconds:
                    break
                for _ in range(200):
                    step_fn()
                fence()
                more += 200
            settled.update(steps=init_steps + more, seconds=round(time.perf_counter() - t_init, 3))

    class Workload:
        """One rank's step: `views` views of a `P`-Gaussian cloud, fwd (+bwd), single or batched call."""

        def __init__(self, P, sh_degree, variant, views, forward_only, first_view, seq_views=1, collectives=True):
            self.P, self.sh_degree, self.views, self.forward_only = P, sh_degree, views, forward_only
            self.seq_views, self.collectives = seq_views, collectives
            cloud = synth.init_cloud(P, sh_degree, variant, seed=0, source=args.cloud)
            self.cloud = cloud
            self.M = cloud.shs.shape[1]
            # sequential mode: round j of the step = global views j * world + rank (view_parallel's round-robin)
            self.cams = [camera(first_view + i) for i in range(views)] if seq_views == 1 else \
                [camera(j * world + rank) for j in range(seq_views)]
            self.leaves = {k: getattr(cloud, k).to(dev).requires_grad_(True)
                           for k in ("means3D", "shs", "opacities", "scales", "rotations")}
            self.shapes = {k: self.leaves[k].shape for k in PARAMS}
            self.shapes["means2D"] = self.leaves["means3D"].shape
            bg = torch.zeros(3, device=dev)
            self.rs = [GaussianRasterizationSettings(
                RES, RES, math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5), bg, 1.0, c.world_view_transform.to(dev),
                c.full_proj_transform.to(dev), sh_degree, c.camera_center.to(dev), False, False) for c in self.cams]
            self.rast = GaussianRasterizer(self.rs[0])
            self.rasts = [GaussianRasterizer(r) for r in self.rs]
            g = torch.Generator().manual_seed(1 + first_view)
            shp = (views,) if views > 1 else ()
            self.gc = (torch.randn(shp + (3, RES, RES), generator=g) * 1e-3).to(dev)
            self.gd = (torch.randn(shp + (1, RES, RES), generator=g) * 1e-3).to(dev)
            self.ga = (torch.randn(shp + (1, RES, RES), generator=g) * 1e-3).to(dev)

        def step(self):
            L = self.leaves
            if self.forward_only:
                with torch.no_grad():
                    if self.views > 1:
                        return rasterize_gaussians_batch(L["means3D"], None, L["shs"], None, L["opacities"],
                                                         L["scales"], L["rotations"], None, self.rs)[0]
                    return self.rast(means3D=L["means3D"], means2D=L["means3D"], shs=L["shs"], opacities=L["opacities"],
                                     scales=L["scales"], rotations=L["rotations"])[0]
            if self.seq_views > 1 or (world > 1 and self.views == 1):
                return self.step_rounds()
            for t in L.values():
                t.grad = None
            # the zero-filled leaf the reference's render() creates per view (gaussian_renderer/__init__.py:26), built the way
            # renderer.render() builds it since ABI v16: storage from torch.empty, the zeros written by the forward's own
            # per-Gaussian kernel (--torch-zero-means2d: one torch fill launch; --uninitialised-means2d: no zeros at all)
            if self.views > 1:
                means2D = leaf_make((self.views,) + tuple(L["means3D"].shape), device=dev).requires_grad_(True)
                color, radii, depth, alpha = rasterize_gaussians_batch(
                    L["means3D"], means2D, L["shs"], None, L["opacities"], L["scales"], L["rotations"], None, self.rs,
                    activation_flags=_rast.ZERO_MEANS2D if leaf_in_kernel else 0)
            else:
                means2D = leaf_make(tuple(L["means3D"].shape), device=dev).requires_grad_(True)
                color, radii, depth, alpha = self.rast(
                    means3D=L["means3D"], means2D=means2D, shs=L["shs"], opacities=L["opacities"], scales=L["scales"],
                    rotations=L["rotations"], zero_means2D=leaf_in_kernel)
            if world > 1:
                # the view-parallel step: the backward's last kernel writes the rank's pack itself (ABI v15: no pack
                # kernel), one collective, the last reduction writes the six gradient tensors + radii (no unpack kernels)
                with _rast.packed_gradients() as pg:
                    torch.autograd.grad([color, depth, alpha], [L[k] for k in PARAMS] + [means2D], [self.gc, self.gd, self.ga])
                    pack = pg.take()
                return vp.allgather_reduce_unpacked(pack, self.shapes, mode=collective_mode[0])
            torch.autograd.backward([color, depth, alpha], [self.gc, self.gd, self.ga])
            return means2D.grad

        def step_rounds(self):
            """The reference's loop over the rank's views; N > 1: one all-gather per round, in flight under the next round."""
            L = self.leaves
            total, pending = None, None
            for j in range(self.seq_views):
                for t in L.values():
                    t.grad = None
                means2D = leaf_make(tuple(L["means3D"].shape), device=dev).requires_grad_(True)
                color, radii, depth, alpha = self.rasts[j](
                    means3D=L["means3D"], means2D=means2D, shs=L["shs"], opacities=L["opacities"], scales=L["scales"],
                    rotations=L["rotations"], zero_means2D=leaf_in_kernel)
                if world > 1 and self.collectives:
                    with _rast.packed_gradients() as pg:
                        torch.autograd.grad([color, depth, alpha], [L[k] for k in PARAMS] + [means2D], [self.gc, self.gd, self.ga])
                        pack = pg.take()
                    if self.seq_views == 1:
                        return vp.allgather_reduce_unpacked(pack, self.shapes, mode=collective_mode[0])
                    started = vp.PackGather(pack, async_op=True)
                    if pending is not None:
                        total = vp.reduce_gathered(pending.result(), total)
                    pending = started
                else:
                    torch.autograd.backward([color, depth, alpha], [self.gc, self.gd, self.ga])
            if pending is not None:
                return vp.reduce_gathered_unpacked(pending.result(), total, self.shapes)
            return means2D.grad

        def timed(self, steps, warmup, init_steps=None):
            init_steps = args.init_steps if init_steps is None else init_steps
            # first-use initialisation, not part of W: the rasterizer's decaying capacity / longest-list
            # estimates settle over the first calls (retries, buffer growth), the GPU leaves its idle clocks
            init_phase(self.step, init_steps)
            for _ in range(warmup):
                self.step()
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            fence()
            elapsed = time.perf_counter() - t0
            if world > 1:
                elapsed = max_over_ranks([elapsed])[0]
            return elapsed

        def stage_times(self, nprof=20):
            """Per-kernel-stage times from events the LIBRARY records on its launch stream."""
            fwd_ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
            bwd_ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            for e in fwd_ev + bwd_ev:
                e.record()
            torch.cuda.synchronize()
            _rast.set_stage_events([e.cuda_event for e in fwd_ev], [e.cuda_event for e in bwd_ev])
            acc = {k: 0.0 for k in FWD_STAGES + BWD_STAGES}
            for _ in range(nprof):
                self.step()
                torch.cuda.synchronize()
                for i, k in enumerate(FWD_STAGES):
                    acc[k] += fwd_ev[i].elapsed_time(fwd_ev[i + 1])
                if not self.forward_only:
                    for i, k in enumerate(BWD_STAGES):
                        acc[k] += bwd_ev[i].elapsed_time(bwd_ev[i + 1])
            _rast.set_stage_events(None, None)
            return {k: v / nprof * 1e3 for k, v in acc.items()}

    class DropInWorkload:
        """The reference-shaped call: renderer.render(camera, GaussianModel-like, pipe, bg) on RAW parameters."""

        def __init__(self, P, fuse=False):
            from humangaussian_amd import renderer
            self.renderer, self.fuse = renderer, fuse
            cloud = synth.init_cloud(P, 0, "mid", seed=0, source=args.cloud)
            raw = {"_xyz": cloud.means3D, "_features_dc": cloud.shs[:, :1], "_features_rest": cloud.shs[:, 1:],
                   "_opacity": torch.logit(cloud.opacities.clamp(1e-6, 1 - 1e-6)), "_scaling": torch.log(cloud.scales),
                   "_rotation": cloud.rotations}
            leaves = {k: v.to(dev).contiguous().requires_grad_(True) for k, v in raw.items()}
            self.leaves = leaves

            class Model:           # gaussian_model.py:95-115
                active_sh_degree = max_sh_degree = 0
                _opacity, _scaling, _rotation = leaves["_opacity"], leaves["_scaling"], leaves["_rotation"]
                _features_dc, _features_rest = leaves["_features_dc"], leaves["_features_rest"]
                get_xyz = property(lambda m: leaves["_xyz"])
                get_features = property(lambda m: torch.cat((leaves["_features_dc"], leaves["_features_rest"]), dim=1))
                get_opacity = property(lambda m: torch.sigmoid(leaves["_opacity"]))
                get_scaling = property(lambda m: torch.exp(leaves["_scaling"]))
                get_rotation = property(lambda m: torch.nn.functional.normalize(leaves["_rotation"]))
            self.model = Model()

            class Pipe:
                convert_SHs_python = compute_cov3D_python = debug = False
            self.pipe = Pipe()
            c = camera(0)
            self.cam = renderer.HostCamera(RES, RES, c.FoVx, c.FoVy, c.world_view_transform.to(dev), c.full_proj_transform.to(dev),
                                           c.camera_center.to(dev))
            self.bg = torch.zeros(3, device=dev)
            g = torch.Generator().manual_seed(1)
            self.gc = (torch.randn((3, RES, RES), generator=g) * 1e-3).to(dev)
            self.gd = (torch.randn((1, RES, RES), generator=g) * 1e-3).to(dev)
            self.ga = (torch.randn((1, RES, RES), generator=g) * 1e-3).to(dev)

        def step(self):
            for t in self.leaves.values():
                t.grad = None
            pkg = self.renderer.render(self.cam, self.model, self.pipe, self.bg, fuse_activations=self.fuse)
            torch.autograd.backward([pkg["render"], pkg["depth_3dgs"], pkg["alpha_3dgs"]], [self.gc, self.gd, self.ga])
            return pkg["viewspace_points"].grad

        def timed(self, steps, warmup, init_steps=None):
            for _ in range((args.init_steps if init_steps is None else init_steps) + warmup):
                self.step()
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            fence()
            return time.perf_counter() - t0

    class AnimationWorkload:
        """configs[4] (animation.py:384-403,477-484,966-1004): per frame re-anchor the Gaussians on the posed body mesh
        (hgs_reanchor; the 136 AMASS poses of content/amass_test_17.npz - a local asset - or a procedural sway drive a toy
        articulation of the body mesh, vertices precomputed), render forward-only through `Renderer.render` (incl. its clamp), frame k on rank k mod N;
        with `gather` one asynchronous image all-gather per round of N frames, in flight under the next round's render."""

        def __init__(self, P, gather):
            import numpy as np
            from humangaussian_amd import animation as an
            from humangaussian_amd.renderer import cameras_from_c2w
            self.P, self.gather, self.an = P, gather, an
            verts, anchors = an.human_mesh_anchors(P, seed=0, device=dev)
            driver = an.MotionDriver(verts, device=dev)
            self.num_poses = driver.num_poses
            self.verts = torch.stack([driver.vertices(i) for i in range(driver.num_poses)]).contiguous()
            self.motion = driver.source
            self.mesh = synth.human_mesh()[2]
            cloud = synth.init_cloud(P, 0, "mid", seed=0, source=args.cloud)

            class Model:
                active_sh_degree = max_sh_degree = 0
                _xyz = None
                get_xyz = property(lambda m: m._xyz)
                get_features = property(lambda m: m._f)
                get_opacity = property(lambda m: m._o)
                get_scaling = property(lambda m: m._s)
                get_rotation = property(lambda m: m._r)
            model = Model()
            model._f, model._o, model._s, model._r = (getattr(cloud, k).to(dev) for k in ("shs", "opacities", "scales", "rotations"))
            self.anim = an.AvatarAnimator(model, anchors, white_background=True, device=dev)
            # the save loop's cameras (animation.py:936-945,993-1000): elevation 0, azimuth i mod 360, radius 2, fovy 50
            self.cams = cameras_from_c2w(np.stack([synth.c2w_orbit(0.0, float(a), 2.0) for a in range(360)]), math.radians(50.0),
                                         RES, RES, device=dev)

        def frame(self, i):
            return self.anim.render_frame(self.verts[i % self.num_poses], self.cams[i % 360])

        def run(self, rounds, first=0):
            n = 0
            for _ in self.an.render_frames_parallel(range(first, first + rounds * world), self.frame, gather=self.gather):
                n += 1
            return n

        def timed(self, steps, warmup, init_steps=None):
            init_phase(lambda: self.run(1), args.init_steps if init_steps is None else init_steps)
            self.run(warmup)
            fence()
            t0 = time.perf_counter()
            self.run(steps, first=7)
            fence()
            elapsed = time.perf_counter() - t0
            if world > 1:
                elapsed = max_over_ranks([elapsed])[0]
            return elapsed

    P, sh_degree = args.points, args.sh_degree
    if args.forward_only:
        # ---------------- the animation leg (configs[4]): frames/s of re-anchor + forward (+ the image gather for N > 1)
        wl = AnimationWorkload(P, gather=world > 1)
        elapsed = wl.timed(args.steps, args.warmup)
        frames = args.steps * world
        line = None
        without = None
        if world > 1:
            wl2 = AnimationWorkload(P, gather=False)
            without = wl2.timed(args.steps, args.warmup, init_steps=min(args.init_steps, 20))
        if rank == 0:
            line = {
                "metric": "rasterize fwd-only Gaussians/sec @1024^2 (animation frames; extra measurement)",
                "value": P * frames / elapsed, "unit": "Gaussians/s", "frames_per_s": frames / elapsed,
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "init_steps": args.init_steps,
                "init_run": dict(settled),
                "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"configs[4]: {P} Gaussians anchored on the {wl.mesh} body mesh, one frame per rank per step: re-anchor "
                                       f"on the posed mesh (hgs_reanchor; motion = {wl.motion}, pose i mod {wl.num_poses}) + no-grad "
                                       "Renderer.render @1024^2 (elevation 0, azimuth i mod 360, radius 2, fovy 50: animation.py:936-1004), "
                                       "frame k on rank k mod N" + ("; ONE asynchronous all-gather of the round's (3,H,W) images per step, "
                                       "in flight under the next frame's render" if world > 1 else ""),
                           "frames_per_step": world, "parallelism": f"frame-parallel x{world}"},
                "animation": {"frames": frames, "image_gather": world > 1, "backend": backend if world > 1 else None,
                              "ms_per_step_without_gather": None if without is None else without / args.steps * 1e3,
                              "exposed_gather_us": None if without is None else (elapsed - without) / args.steps * 1e6},
            }
            print(json.dumps(line))
        if world > 1:
            dist.destroy_process_group()
        return
    collective_mode = [args.collective if args.collective != "auto" else "allgather"]
    main_wl = Workload(P, sh_degree, args.variant, args.views, args.forward_only, first_view=rank * args.views, seq_views=VPR)

    # ---------------- N > 1: what the collective of a step costs, per mode (OUTSIDE the timed region);
    # --collective auto then runs the timed steps with the faster mode (every rank takes the same decision)
    collective = None
    if world > 1:
        # (no pack pass since ABI v15: the backward's last kernel writes the pack; the timed part is the collective + the
        #  reduction that also unpacks)
        pack = torch.zeros((P, 15 + 3 * main_wl.M), device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        nrep = 20
        res = {}
        for mode in vp.COLLECTIVE_MODES:
            acc = 0.0
            for it in range(nrep + 5):
                fence()
                ev[0].record()
                vp.allgather_reduce_unpacked(pack, main_wl.shapes, mode=mode)
                ev[1].record()
                torch.cuda.synchronize()
                if it >= 5:
                    acc += ev[0].elapsed_time(ev[1])
            tt = max_over_ranks([acc / nrep * 1e3])
            res[mode] = {"pack_us": 0.0, "collective_and_reduce_us": tt[0]}
        if args.collective == "auto":
            collective_mode[0] = min(vp.COLLECTIVE_MODES, key=lambda m: res[m]["collective_and_reduce_us"])
            if not all(math.isfinite(res[m]["collective_and_reduce_us"]) and res[m]["collective_and_reduce_us"] > 0 for m in res):
                collective_mode[0] = "scatter" if world >= 8 else "allgather"      # (no usable timing: the fewer-bytes form at 8 ranks)
        # what the collectives ADD to a step, measured: the same step with and without them (outside the timed region),
        # for one view per rank (nothing to overlap: the per-Gaussian backward that produces the gradients is the last
        # kernel of the step) and for two (round 0's all-gather runs under round 1's render)
        exposed = {}
        ksteps = max(10, args.steps // 10)
        for vpr in (1, 2):
            t_with = Workload(P, sh_degree, args.variant, 1, False, 0, seq_views=vpr).timed(ksteps, 3, init_steps=min(args.init_steps, 20))
            t_wo = Workload(P, sh_degree, args.variant, 1, False, 0, seq_views=vpr, collectives=False).timed(ksteps, 3, init_steps=min(args.init_steps, 20))
            exposed[f"views_per_rank_{vpr}"] = {"step_us_with": t_with / ksteps * 1e6, "step_us_without": t_wo / ksteps * 1e6,
                                                "exposed_collective_us": (t_with - t_wo) / ksteps * 1e6}
        collective = {"mode_used": collective_mode[0] if VPR == 1 else "allgather per round, pipelined", "requested": args.collective,
                      "backend": backend, "views_per_rank": VPR, "timings": res, "exposed": exposed,
                      "exposed_collective_us": exposed[f"views_per_rank_{min(VPR, 2)}"]["exposed_collective_us"],
                      "bytes_per_rank_pack": int(pack.numel() * 4),
                      "note": "allgather = ONE all_gather_into_tensor of the per-rank gradient packs + rank-ordered local "
                              "reduction; scatter = all_to_all of row shards + the same local reduction + one all-gather of "
                              "the reduced shards (same bits, (world-1)/world * 2 packs per rank on the links instead of "
                              "world-1); timed outside the step loop, max over ranks"}

    elapsed = main_wl.timed(args.steps, args.warmup)
    ms = elapsed / args.steps * 1e3
    M = main_wl.M

    # ---------------- host/GPU balance (outside the timed region): time the host spends blocked
    # in the one event wait per forward.  wait ~ 0 means the loop is host-bound.
    nhost = 50
    st0 = _rast._state(dev)
    w0 = st0.wait_ns
    th = time.perf_counter()
    for _ in range(nhost):
        main_wl.step()
    torch.cuda.synchronize()
    th = time.perf_counter() - th
    st_now = _rast._state(dev)
    hb = {k: round((st_now.host_ns[k] - st0.host_ns[k]) / nhost * 1e-3, 1) for k in st_now.host_ns}
    host_info = {"step_us": round(th / nhost * 1e6, 1),
                 "event_wait_us": round((st_now.wait_ns - w0) / nhost * 1e-3, 1),
                 # inside the binding (us per step): forward = pre (checks, allocations) + launch (hgs_forward: 5 kernel
                 # launches) + shadow (autograd state, gradient tensors) + wait (status poll) + rest; backward likewise;
                 # step_us - fwd_total - bwd_total = Python, the autograd engine and its thread hand-off
                 "binding_us": hb,
                 "forward_retries_total": int(st_now.retries), "forward_calls_total": int(st_now.calls)}

    # ---------------- host/GPU balance etc. follow below
    # ---------------- per-kernel timing (outside the timed region; library-recorded events)
    stage_us = main_wl.stage_times()
    R = int(_rast._state(dev).max_R)
    npix, T = RES * RES, (RES // 16) ** 2
    B = args.views
    # upstream's stages: the blend backward (B1) is render_bwd + pair_reduce here, priced together
    merged_us = {name: sum(stage_us[k] for k in parts) for name, parts in ROOFLINE_STAGES.items()
                 if not (args.forward_only and any(k in BWD_STAGES for k in parts))}
    merged_bytes = {name: sum(algorithmic_bytes(k, P, M, R, npix, T, B) for k in ROOFLINE_STAGES[name]) for name in merged_us}
    merged_bytes["blend_bwd(render_bwd+pair_reduce)"] = 84 * R + 28 * npix * B       # (B1 = 84 R + 28 N_pix)
    dom = max(merged_us, key=merged_us.get)
    dom_bytes = merged_bytes[dom]
    achieved = dom_bytes / (merged_us[dom] * 1e-6) / 1e9
    # HBM-side bytes of the dominant kernel from the committed PMC passes (tools/gpu_round.sh pmc), with the commit
    # they were taken at: the two must be read together (the counters cannot be collected inside this process)
    traffic, traffic_commit = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    counters = None
    if os.path.exists(tpath) and B == 1:
        try:
            tj = json.load(open(tpath))
            parts = [tj.get(k) for k in ROOFLINE_STAGES[dom]]
            traffic, traffic_commit = (sum(parts) if all(x is not None for x in parts) else None), tj.get("_commit")
        except Exception:
            traffic = None
    cpath = os.path.join(ROOT, "profiles", "sq_counters.json")
    if os.path.exists(cpath) and B == 1:
        try:
            counters = json.load(open(cpath))         # committed SQ-counter passes (tools/sq_pass.sh) -> VALU busy per blend kernel
        except Exception:
            counters = None
    path_bytes = path_bytes_survey(P, M, R, npix, T, B, args.forward_only)
    gpu_us = sum(stage_us.values())
    blend_us = stage_us["render_fwd"] + (0.0 if args.forward_only else stage_us.get("render_bwd", 0.0) + stage_us.get("pair_reduce", 0.0))

    # ---------------- extra measurements of the same path (rank 0, N=1 only; same timing rules)
    extra = None
    if rank == 0 and world == 1 and not args.no_extra and not args.forward_only and args.views == 1 \
            and P == P_POINTS and sh_degree == SH_DEGREE and args.variant == "mid":
        extra = {}

        def pmc_roofline(tag, algorithmic, t_s):
            """algorithmic bytes of the path over the measured step, and - from the committed PMC pass of the
            same workload (profiles/<tag>pmc_traffic.json: bytes through the fabric per step, all kernels) - what the
            memory system actually moved per second"""
            out = {"bound": "hbm", "algorithmic_bytes": algorithmic, "achieved": algorithmic / t_s / 1e9, "peak": HBM_PEAK_GBS,
                   "unit": "GB/s", "frac": algorithmic / t_s / 1e9 / HBM_PEAK_GBS, "traffic": None}
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", tag + "pmc_traffic.json")))
                tr = sum(v for k, v in tj.items() if not k.startswith("_") and isinstance(v, (int, float)))
                out.update({"traffic": tr, "traffic_commit": tj.get("_commit"), "traffic_over_algorithmic": tr / algorithmic,
                            "counter_achieved": tr / t_s / 1e9, "counter_frac": tr / t_s / 1e9 / HBM_PEAK_GBS})
            except Exception:
                pass
            return out

        def measure(name, wl, steps, warmup, units, note, roof=None):
            # three timed repetitions (each K steps behind W warm-up steps); the MEDIAN is reported, all are recorded:
            # these side measurements share the process with everything before them
            ts = [wl.timed(steps, warmup), wl.timed(steps, warmup, init_steps=0), wl.timed(steps, warmup, init_steps=0)]
            t = sorted(ts)[1]
            su = wl.stage_times(10) if hasattr(wl, "stage_times") else None
            extra[name] = {"value": units * steps / t, "unit": "Gaussians/s", "ms_per_step": t / steps * 1e3,
                           "ms_per_step_runs": [x / steps * 1e3 for x in ts],
                           "steps": steps, "warmup": warmup, "init_steps": INIT_STEPS,
                           "workload": note + f" [the MEDIAN of three timed runs of {steps} steps; all in ms_per_step_runs]",
                           "num_rendered_R": int(_rast._state(dev).max_R), "stage_us": su}
            if roof is not None:
                tag, Pn, Mn, Bn = roof
                extra[name]["roofline"] = pmc_roofline(tag, path_bytes_survey(Pn, Mn, extra[name]["num_rendered_R"], npix, T, Bn),
                                                       t / steps)
        k8 = max(20, args.steps // 6)
        measure("batched_8_views", Workload(P, sh_degree, "mid", 8, False, 0), k8, max(5, args.warmup // 5), 8 * P,
                "configs[1] x 8 views batched: the 8 orbit cameras of configs[2] (azim 30+45*i) rendered fwd+bwd by ONE "
                "hgs_forward_batch / hgs_backward_batch call per step; Gaussians/s = 8 * P / step time", roof=("8views_", P, M, 8))
        measure("init_variant", Workload(P, sh_degree, "init", 1, False, 0), max(20, args.steps // 3), max(5, args.warmup // 2),
                P, "configs[1] with the step-0 cloud (opacity 0.1, isotropic scales, identity rotations: no early termination)")
        # the same step with autograd's backward on the CALLING thread (torch.autograd.set_multithreading_enabled(False): no
        # hand-off to the device's engine thread and back): a host-side setting of the caller, the same kernels; it moves
        # the step only where the host is the slower side
        with torch.autograd.set_multithreading_enabled(False):
            measure("single_thread_autograd", Workload(P, sh_degree, "mid", 1, False, 0), max(20, args.steps // 3), max(5, args.warmup // 2),
                    P, "configs[1], the headline step with torch.autograd.set_multithreading_enabled(False) (the backward runs on "
                       "the calling thread); the headline keeps torch's default")
        measure("forward_only", Workload(P, sh_degree, "mid", 1, True, 0), max(20, args.steps // 3), max(5, args.warmup // 2),
                P, "configs[4] shape: no-grad forward of one 1024^2 view (animation path), per GPU")
        measure("drop_in_render", DropInWorkload(P), max(20, args.steps // 3), max(5, args.warmup // 2), P,
                "configs[1] through the drop-in renderer.render() exactly as GaussianDreamer.py:244-266 calls the reference's: a "
                "GaussianModel-shaped object with RAW parameters (get_* = sigmoid / exp / normalize / cat as gaussian_model.py:95-115, "
                "un-fused torch kernels with their autograd), the zero-filled viewspace_points leaf, fwd+bwd of one 1024^2 view")
        measure("drop_in_render_fused", DropInWorkload(P, fuse=True), max(20, args.steps // 3), max(5, args.warmup // 2), P,
                "the same call with renderer.render(..., fuse_activations=True) (or renderer.FUSE_ACTIVATIONS): the RAW "
                "_opacity / _scaling / _rotation go to the rasterizer, sigmoid / exp / normalize run inside its per-Gaussian "
                "kernels forward and backward (values equal to the un-fused path to rounding)")
        anim = AnimationWorkload(P, gather=False)
        measure("animation_frames", anim, 300, max(5, args.warmup // 2), P,
                f"configs[4] at its stated size, 300 frames per timed run: per frame re-anchor on the posed {anim.mesh} body mesh "
                f"(hgs_reanchor; motion = {anim.motion}) + no-grad "
                "Renderer.render @1024^2, camera and pose change every frame (animation.py:384-403,477-484,966-1004); one GPU: "
                "`bench.py --forward-only --gpus N` shards the frames")
        measure("config4_500k_sh3", Workload(500_000, 3, "mid", 1, False, 0), max(20, args.steps // 6), max(5, args.warmup // 5),
                500_000, "configs[3]: 500k Gaussians, SH degree 3, one 1024^2 view, fwd+bwd", roof=("cfg3_", 500_000, 16, 1))

    # ---------------- CPU baseline: the PyTorch oracle on the host cores (rank 0, N=1): the FULL
    # forward + backward of the same view (every tile), 1 warm-up + median of 3.
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.views == 1 and not args.forward_only:
        import oracle
        cloud, cam, rs = main_wl.cloud, main_wl.cams[0], main_wl.rs[0]
        st = oracle.OracleSettings(RES, RES, rs.tanfovx, rs.tanfovy, torch.zeros(3), 1.0, cam.world_view_transform,
                                   cam.full_proj_transform, sh_degree, cam.camera_center, False, False)
        gcc, gdc, gac = main_wl.gc.cpu(), main_wl.gd.cpu(), main_wl.ga.cpu()

        def oracle_fwd_bwd():
            tc = time.perf_counter()
            oracle.forward_backward(cloud.means3D, cloud.shs, None, cloud.opacities, cloud.scales, cloud.rotations, None,
                                    st, gcc, gdc, gac, dtype=torch.float32)
            return time.perf_counter() - tc
        best = None
        for threads in sorted({min(os.cpu_count() or 1, 16), min(os.cpu_count() or 1, 64), os.cpu_count() or 1}):
            torch.set_num_threads(threads)
            oracle_fwd_bwd()                                   # warm-up
            ts = sorted(oracle_fwd_bwd() for _ in range(3))
            if best is None or ts[1] < best[0]:
                best = (ts[1], threads, ts)
            if ts[1] > 8.0:                                    # keep the whole leg bounded
                break
        t_med, threads, ts = best
        cpu = {"value": P / t_med, "unit": "Gaussians/s", "cores": threads, "kind": "port",
               "sample": f"PyTorch CPU oracle (fp32, autograd, tile-streamed backward), the same {P}-Gaussian 1024^2 "
                         f"view, FULL fwd+bwd (all tiles): 1 warm-up + median of 3 = {t_med:.2f} s "
                         f"(runs {', '.join(f'{x:.2f}' for x in ts)}) with torch.set_num_threads({threads}) - the "
                         f"fastest of the thread counts tried - on a host with {os.cpu_count()} cores; "
                         f"torch {torch.__version__}"}

    if rank == 0:
        flops_pair = 25 if args.forward_only else 105
        line = {
            "metric": "rasterize fwd+bwd Gaussians/sec @1024^2, 100k pts" if not args.forward_only
            else "rasterize fwd-only Gaussians/sec @1024^2 (extra measurement)",
            "value": P * args.views * VPR * world * args.steps / elapsed,
            "unit": "Gaussians/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "init_steps": args.init_steps,
            "init_run": dict(settled),      # un-timed steps / seconds the first measurement ran before its W warm-up steps
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[1]: {P} Gaussians on " + ("the reference's load/shapes/human.obj (area-uniform, seed 0, normalised "
                                   "as threestudio/utils/poser.py:337-357: SURVEY.md 8(d)'s cloud)" if args.cloud == "human_obj" else
                                   "a capsule humanoid with SMPL-X extents (the stand-in of rounds 1-4)")
                                   + f", {args.variant}-training state, SH degree {sh_degree}; {args.views * VPR} 1024x1024 orbit view(s) per GPU per step"
                                   + (" in ONE batched call" if args.views > 1 else "")
                                   + (" one after the other (the reference's loop)" if VPR > 1 else "")
                                   + " (elev 10, azim 30+45*view, dist 1.75, fovy 55), "
                                   + ("fwd only" if args.forward_only else "fwd+bwd")
                                   + ("; the step's means2D leaf is uninitialised (--uninitialised-means2d: its values are never read)"
                                      if args.uninitialised_means2d else
                                      "; means2D is the zero-filled leaf of the reference's render(), filled by one torch launch per view (--torch-zero-means2d)"
                                      if args.torch_zero_means2d else
                                      "; means2D is the zero-filled leaf of the reference's render(), built as renderer.render() builds it: "
                                      "fresh storage per view, the zeros written by the forward's per-Gaussian kernel (ABI v16)"),
                       "cloud": args.cloud,
                       "views_per_step": world * args.views * VPR, "views_per_rank_sequential": VPR, "num_rendered_R": int(R),
                       "host_mode": "sync (one host wait per forward for the device-side status, as upstream)",
                       "parallelism": f"view-parallel x{world}" + (f", collective {collective_mode[0]}" if world > 1 else "")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": {"file": "profiles/pmc_traffic.json", "commit": traffic_commit},
                         "algorithmic_bytes": dom_bytes, "avg_us": merged_us[dom],
                         "note": "stage = upstream's stage (B1 = render_bwd + pair_reduce); blend kernels are VALU / LDS / "
                                 "MFMA-issue-bound (the backward's pixel sums run on fp32 MFMA); HBM fraction reported as "
                                 "BASELINE.json asks",
                         # the whole path: SURVEY.md 8(d)'s byte formula over the WALL time of a step (not the event sum)
                         "path": {"algorithmic_bytes": path_bytes, "ms_per_step": ms, "stage_event_us_sum": gpu_us,
                                  "achieved": path_bytes / (ms * 1e-3) / 1e9 / max(1, VPR),
                                  "frac": path_bytes / (ms * 1e-3) / 1e9 / max(1, VPR) / HBM_PEAK_GBS},
                         # how much faster than UPSTREAM'S BRUTE FORCE (256 pixel-Gaussian pairs per entry, ~25 flop forward
                         # + ~80 backward) the blend stages run, expressed against the fp32 vector peak.  NOT a utilisation:
                         # the kernels evaluate ~57 lane slots per entry, not 256 (see valu_busy for what the pipes do)
                         "brute_force_equivalent": {"flops": 256.0 * R * flops_pair,
                                                    "tflops": 256.0 * R * flops_pair / (blend_us * 1e-6) / 1e12,
                                                    "of_fp32_vector_peak_157.3": 256.0 * R * flops_pair / (blend_us * 1e-6) / 1e12 / 157.3,
                                                    "kernels": "render_fwd + render_bwd + pair_reduce"},
                         # counter-derived (committed profiles/sq_counters.json: SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x
                         # 2.4 GHz x kernel time)); live lanes: 34 contributing pixels per 57 evaluated lane slots per entry
                         "valu_busy": (counters or {}).get("valu_busy"), "live_lane_fraction": 34.0 / 57.0},
            "stage_us": stage_us,
            "host": host_info,
            "collective": collective,
            "multi_gpu": (None if world > 1 else
                          "this line is N = 1.  No N > 1 value has ever been measured for this repository: gpurun boxes have one GPU; "
                          "the view-parallel step and the frame-sharded animation leg are covered by gloo world-2/3 tests and by two "
                          "ranks sharing one device (tests/test_gpu_multirank_one_gpu.py); RCCL itself has not executed. "
                          "`python bench.py --gpus N` is the command the first multi-GPU node runs"),
            "cpu_baseline": cpu,
            "extra": extra,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
