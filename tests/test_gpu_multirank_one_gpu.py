"""The N > 1 branch of bench.py and the pipelined view-parallel step, executed on ONE GPU: two ranks share device 0
and talk over gloo (an RCCL communicator needs distinct devices), selected by HGS_DIST_BACKEND=gloo /
HGS_BENCH_SHARE_DEVICE=1.  Everything else is the code the driver's `--gpus N` run executes: the self-spawn under
torch.distributed.run, `pack_view_contribution` / `reduce_view_packs_acc` on the device, one asynchronous all-gather
per round of views, the collective timing block and the `collective` fields of the JSON line."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra):
    env = dict(os.environ, HGS_DIST_BACKEND="gloo", HGS_BENCH_SHARE_DEVICE="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--init-steps", "6", "--points", "20000", "--no-cpu-baseline", "--no-extra"] + extra
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]          # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("vpr", [1, 2])
def test_bench_two_ranks_on_one_device(vpr):
    line = _run_bench(["--views-per-rank", str(vpr)])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 6
    assert line["config"]["views_per_step"] == 2 * vpr
    assert line["value"] > 0 and line["ms_per_step"] > 0
    c = line["collective"]
    assert c["backend"] == "gloo" and c["views_per_rank"] == vpr
    assert set(c["timings"]) == {"allgather", "scatter"}
    for k in ("views_per_rank_1", "views_per_rank_2"):
        assert c["exposed"][k]["step_us_with"] > 0 and c["exposed"][k]["step_us_without"] > 0
    assert c["bytes_per_rank_pack"] == 20000 * 18 * 4          # P x (15 + 3 M) floats, M = 1


@pytest.mark.timeout(900)
def test_pipelined_view_parallel_step_on_one_device_matches_the_serial_loop(tmp_path):
    """render_views_parallel (HIP rasterizer, 5 views on 2 ranks sharing the device, pipelined rounds) against the same
    call in ONE process: bit-identical gradients and radii on both ranks."""
    script = tmp_path / "vp_worker.py"
    script.write_text('''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from helpers import make_scene
from humangaussian_amd import synth, view_parallel as vp
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
sc = make_scene(P=3000, sh_degree=1, seed=3, H=96, W=96, spread=0.3, scale=0.05)
cams = [synth.orbit_camera(8.0 * (v - 2), 70.0 * v, 2.0, 50.0, 96, 96) for v in range(5)]
cams = [c._replace(world_view_transform=c.world_view_transform.cuda(), full_proj_transform=c.full_proj_transform.cuda(),
                   camera_center=c.camera_center.cuda()) for c in cams]
params = {k: sc[k].cuda() for k in ("means3D", "shs", "opacities", "scales", "rotations")}
def loss_grad(v, color, depth, alpha):
    g = torch.Generator().manual_seed(50 + v)
    return (torch.randn(color.shape, generator=g).cuda(), torch.randn(depth.shape, generator=g).cuda(), torch.randn(alpha.shape, generator=g).cuda())
grads, radii, _ = vp.render_views_parallel(cams, params, sc["bg"].cuda(), 1, loss_grad, pipeline=True)
torch.save({"grads": {k: v.cpu() for k, v in grads.items()}, "radii": radii.cpu()}, os.path.join(%r, f"out_{world}_{rank}.pt"))
dist.barrier()
dist.destroy_process_group()
''' % (ROOT, ROOT, str(tmp_path)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    for world in (1, 2):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(29500 + 7 * world + os.getpid() % 200), str(script)]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    ref = torch.load(tmp_path / "out_1_0.pt")
    for r in range(2):
        got = torch.load(tmp_path / f"out_2_{r}.pt")
        assert torch.equal(got["radii"], ref["radii"])
        for k in ref["grads"]:
            assert torch.equal(got["grads"][k], ref["grads"][k]), (r, k)


@pytest.mark.timeout(900)
def test_bench_animation_leg_two_ranks_on_one_device():
    """`bench.py --forward-only --gpus 2` (configs[4]): frames sharded over the ranks, re-anchor + no-grad render per frame,
    one image all-gather per step; the line reports frames/s with and without the gather."""
    line = _run_bench(["--forward-only"])
    assert line["n_gpus"] == 2 and line["config"]["frames_per_step"] == 2 and line["steps"] == 6
    assert line["value"] > 0 and abs(line["value"] - 20000 * line["frames_per_s"]) <= 1e-6 * line["value"]
    a = line["animation"]
    assert a["frames"] == 12 and a["image_gather"] and a["backend"] == "gloo" and a["ms_per_step_without_gather"] > 0
    from humangaussian_amd import data
    assert ("amass_test_17" in line["config"]["workload"]) == data.have_motion()       # the line says which motion drove it


def _run_workers(tmp_path, body, worlds=(1, 2)):
    script = tmp_path / "worker.py"
    script.write_text(('''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
OUT = %r
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
''' % (ROOT, ROOT, str(tmp_path))) + body + '''
dist.barrier()
dist.destroy_process_group()
''')
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    for world in worlds:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(29600 + 11 * world + os.getpid() % 200), str(script)]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])


@pytest.mark.timeout(900)
def test_frame_sharded_animation_on_one_device_equals_the_single_process_loop(tmp_path):
    """configs[4] composed end to end on the HIP path: MotionDriver (AMASS poses) -> hgs_reanchor -> Renderer.render under no_grad
    -> image all-gather, 7 frames on 2 ranks sharing the device: every rank ends up with EVERY frame, in order, bit-identical to
    the one-process loop; consecutive frames differ (the avatar moves and the camera orbits)."""
    _run_workers(tmp_path, '''
import math, numpy as np
from humangaussian_amd import animation as an, synth
P, H, W = 6000, 128, 128
verts, anchors = an.human_mesh_anchors(P, seed=2, device="cuda")
driver = an.MotionDriver(verts, device="cuda")
assert driver.num_poses == 136        # (the AMASS clip where the local asset exists, the procedural sway otherwise)
cloud = synth.init_cloud(P, 0, "mid", seed=2)
class Model:
    active_sh_degree = max_sh_degree = 0
    _xyz = None
    get_xyz = property(lambda m: m._xyz)
    get_features = property(lambda m: cloud.shs.cuda())
    get_opacity = property(lambda m: cloud.opacities.cuda())
    get_scaling = property(lambda m: cloud.scales.cuda() * 3.0)
    get_rotation = property(lambda m: cloud.rotations.cuda())
anim = an.AvatarAnimator(Model(), anchors, white_background=True, device="cuda")
frames = [3 + 20 * k for k in range(7)]
rendered = []
def render(i):
    rendered.append(i)
    return anim.render_frame(driver.vertices(i), an.orbit_frame_camera(i, H, W))
got = list(an.render_frames_parallel(frames, render, gather=True))
assert [i for i, _ in got] == frames and rendered == frames[rank::world]
torch.save({"frames": [i for i, _ in got], "images": torch.stack([img for _, img in got]).cpu()}, os.path.join(OUT, f"anim_{world}_{rank}.pt"))
''')
    ref = torch.load(tmp_path / "anim_1_0.pt")
    imgs = ref["images"]
    assert imgs.shape == (7, 3, 128, 128) and torch.isfinite(imgs).all() and float(imgs.min()) >= 0 and float(imgs.max()) <= 1
    assert float((imgs < 0.999).float().mean()) > 0.02                     # the avatar covers part of the white frame
    for k in range(6):
        assert float((imgs[k + 1] - imgs[k]).abs().mean()) > 1e-4           # pose and camera change from frame to frame
    for r in range(2):
        got = torch.load(tmp_path / f"anim_2_{r}.pt")
        assert got["frames"] == ref["frames"] and torch.equal(got["images"], imgs), r


@pytest.mark.timeout(900)
def test_batched_rank_views_on_one_device(tmp_path):
    """render_views_parallel(batched=True) on the HIP rasterizer: 5 views on 2 ranks sharing the device, each rank's views in ONE
    batched call + one collective: the same bits on both ranks and in both collective modes, the serial loop's values to rounding."""
    _run_workers(tmp_path, '''
from helpers import make_scene
from humangaussian_amd import synth, view_parallel as vp
sc = make_scene(P=3000, sh_degree=1, seed=3, H=96, W=96, spread=0.3, scale=0.05)
cams = [synth.orbit_camera(8.0 * (v - 2), 70.0 * v, 2.0, 50.0, 96, 96) for v in range(5)]
cams = [c._replace(world_view_transform=c.world_view_transform.cuda(), full_proj_transform=c.full_proj_transform.cuda(),
                   camera_center=c.camera_center.cuda()) for c in cams]
params = {k: sc[k].cuda() for k in ("means3D", "shs", "opacities", "scales", "rotations")}
def loss_grad(v, color, depth, alpha):
    g = torch.Generator().manual_seed(50 + v)
    return (torch.randn(color.shape, generator=g).cuda(), torch.randn(depth.shape, generator=g).cuda(), torch.randn(alpha.shape, generator=g).cuda())
res = {}
for mode in ("serial", "allgather", "scatter"):
    kw = dict(pipeline=True) if mode == "serial" else dict(batched=True, collective=mode)
    grads, radii, _ = vp.render_views_parallel(cams, params, sc["bg"].cuda(), 1, loss_grad, **kw)
    res[mode] = {"grads": {k: v.cpu() for k, v in grads.items()}, "radii": radii.cpu()}
torch.save(res, os.path.join(OUT, f"bat_{world}_{rank}.pt"))
''', worlds=(2,))
    a, b = torch.load(tmp_path / "bat_2_0.pt"), torch.load(tmp_path / "bat_2_1.pt")
    ser = a["serial"]
    for mode in ("allgather", "scatter"):
        assert torch.equal(a[mode]["radii"], ser["radii"]) and torch.equal(b[mode]["radii"], ser["radii"])
        for k, ref in ser["grads"].items():
            assert torch.equal(a[mode]["grads"][k], b[mode]["grads"][k]), (mode, k)
            assert torch.equal(a[mode]["grads"][k], a["allgather"]["grads"][k]), (mode, k)
            scale = max(float(ref.abs().max()), 1e-30)
            assert float((a[mode]["grads"][k] - ref).abs().max()) <= 2e-5 * scale, (mode, k)
