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
