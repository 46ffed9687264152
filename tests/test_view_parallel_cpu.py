"""World-size-2 gloo tests of the view-parallel host logic (runs on CPU: the oracle is
injected as the rasterizer; the product default is the HIP rasterizer)."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from helpers import make_scene
from humangaussian_amd import synth
from humangaussian_amd import view_parallel as vp

NUM_VIEWS = 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_render_fn(cam, leaves, means2D, bg, sh_degree):
    st = oracle.OracleSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5),
                               math.tan(cam.FoVy * 0.5), bg, 1.0, cam.world_view_transform,
                               cam.full_proj_transform, sh_degree, cam.camera_center, False, False)
    return oracle.rasterize(leaves["means3D"], means2D, leaves["shs"], None, leaves["opacities"],
                            leaves["scales"], leaves["rotations"], None, st, dtype=torch.float64)


def _scene_and_cams():
    sc = make_scene(P=60, sh_degree=1, seed=5, H=32, W=32, spread=0.3, scale=0.08)
    cams = [synth.orbit_camera(10.0 * (v - 1), 90.0 * v, 2.0, 50.0, 32, 32) for v in range(NUM_VIEWS)]
    params = {k: sc[k].double() for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    return sc, cams, params


def _loss_grad(v, color, depth, alpha):
    g = torch.Generator().manual_seed(100 + v)
    return (torch.randn(color.shape, generator=g, dtype=torch.float64),
            torch.randn(depth.shape, generator=g, dtype=torch.float64), None)


def _worker(rank, world, port, q, pipeline=None, num_views=NUM_VIEWS):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    sc, cams, params = _scene_and_cams()
    cams = (cams * 2)[:num_views]
    grads, radii, outs = vp.render_views_parallel(cams, params, sc["bg"].double(), 1, _loss_grad,
                                                  render_fn=_oracle_render_fn, gather_images=True, pipeline=pipeline)
    # (numpy: pickled by value - a tensor travels as a file descriptor the parent may open after this process has gone)
    q.put((rank, {k: v.numpy().copy() for k, v in grads.items()}, radii.numpy().copy(),
           [(v, c.numpy().copy()) for v, c, _, _ in outs]))
    dist.barrier()
    try:                      # gloo teardown can race with the peer's exit; results are already out
        dist.destroy_process_group()
    except Exception:
        pass


def _from_numpy(r):
    rank, grads, radii, outs = r
    return rank, {k: torch.from_numpy(v) for k, v in grads.items()}, torch.from_numpy(radii), [(v, torch.from_numpy(c)) for v, c in outs]


def test_shard_views_round_robin():
    assert vp.shard_views(8, 0, 8) == [0] and vp.shard_views(8, 7, 8) == [7]
    assert vp.shard_views(300, 3, 8)[:3] == [3, 11, 19]
    assert sorted(sum((vp.shard_views(10, r, 4) for r in range(4)), [])) == list(range(10))


def test_pack_roundtrip():
    P, M = 7, 4
    g = {"means3D": torch.randn(P, 3), "means2D": torch.randn(P, 3), "shs": torch.randn(P, M, 3),
         "opacities": torch.randn(P, 1), "scales": torch.randn(P, 3), "rotations": torch.randn(P, 4)}
    r = torch.randint(0, 500, (P,), dtype=torch.int32)
    pack = vp.pack_contribution(g, r)
    assert pack.shape == (P, 3 + 3 + 3 * M + 1 + 3 + 4 + 1)
    g2, r2 = vp.unpack_contribution(pack, {k: v.shape for k, v in g.items()})
    assert torch.equal(r, r2) and all(torch.equal(g[k], g2[k]) for k in g)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,num_views", [(2, 4), (3, 5)])
def test_pipelined_rounds_equal_the_serial_accumulation_bitwise(world, num_views):
    """Several views per rank: one asynchronous all-gather per round of views, chained in view order
    (`reduce_gathered(gathered, acc_in)`), must give the BITS of the serial loop over the views
    (GaussianDreamer.py:244-266,385-391) on every rank - also when the last round is ragged (5 views on 3 ranks:
    the idle ranks contribute zero packs)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, True, num_views)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
    res = [_from_numpy(r) for r in res]
    sc, cams, params = _scene_and_cams()
    cams = (cams * 2)[:num_views]
    ref, rref, _ = vp.render_views_parallel(cams, params, sc["bg"].double(), 1, _loss_grad,
                                            render_fn=_oracle_render_fn, pipeline=True)       # world 1: the serial chain
    for rank, grads, radii, outs in res:
        assert torch.equal(radii, rref)
        for k in ref:
            assert torch.equal(grads[k], ref[k]), (rank, k)
        assert [v for v, _ in outs] == list(range(num_views))


@pytest.mark.timeout(300)
def test_two_ranks_reproduce_the_serial_accumulation():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, False)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
    res = [_from_numpy(r) for r in res]
    # serial reference = what the single-GPU loop accumulates (GaussianDreamer.py:244-266,385-391)
    sc, cams, params = _scene_and_cams()
    ref, rref, _ = vp.render_views_parallel(cams, params, sc["bg"].double(), 1, _loss_grad,
                                            render_fn=_oracle_render_fn, pipeline=False)
    for rank, grads, radii, outs in res:
        assert torch.equal(radii, rref)
        for k in ref:
            assert torch.allclose(grads[k], ref[k].float().to(grads[k].dtype), rtol=1e-5, atol=1e-7), k
        assert [v for v, _ in outs] == list(range(NUM_VIEWS))      # every rank holds all views
    # both ranks hold bit-identical results (fixed reduction order)
    for k in ref:
        assert torch.equal(res[0][1][k], res[1][1][k])
    for (v0, c0), (v1, c1) in zip(res[0][3], res[1][3]):
        assert v0 == v1 and torch.equal(c0, c1)


def _collective_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(7 + rank)
    P, F = 1001, 18                                    # P not a multiple of the world size
    pack = torch.randn(P, F, generator=g)
    pack[:, -1] = torch.randint(0, 40, (P,), generator=g).float()
    a = vp.allgather_reduce(pack, mode="allgather")
    b = vp.allgather_reduce(pack, mode="scatter")
    q.put((rank, a.numpy().copy(), b.numpy().copy()))
    dist.barrier()
    try:
        dist.destroy_process_group()
    except Exception:
        pass


@pytest.mark.parametrize("world", [2, 3])
def test_scatter_collective_equals_allgather_bitwise(world):
    """mode "scatter" (all-to-all of shards + local rank-ordered reduce + all-gather of the reduced
    shards) returns the bits of the single all-gather + local reduce, on every rank."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_collective_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    res = [(rank, torch.from_numpy(a), torch.from_numpy(b)) for rank, a, b in res]
    ref = res[0][1]
    for rank, a, b in res:
        assert torch.equal(a, ref) and torch.equal(b, ref), rank
    # and it is the rank-ordered sum / max
    packs = []
    for r in range(world):
        g = torch.Generator().manual_seed(7 + r)
        pk = torch.randn(1001, 18, generator=g)
        pk[:, -1] = torch.randint(0, 40, (1001,), generator=g).float()
        packs.append(pk)
    assert torch.equal(ref, vp.reduce_gathered(torch.stack(packs)))


# ------------------------------------------------------------------ a rank's views in ONE batched call (batched=True)

def _oracle_render_batch_fn(cams, leaves, means2D, bg, sh_degree):
    outs = [_oracle_render_fn(c, leaves, means2D[i], bg, sh_degree) for i, c in enumerate(cams)]
    return tuple(torch.stack([o[k] for o in outs]) for k in range(4))


def _batched_worker(rank, world, port, q, num_views):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    sc, cams, params = _scene_and_cams()
    cams = (cams * 2)[:num_views]
    res = {}
    for mode in ("allgather", "scatter"):
        grads, radii, outs = vp.render_views_parallel(cams, params, sc["bg"].double(), 1, _loss_grad, render_fn=_oracle_render_fn,
                                                      render_batch_fn=_oracle_render_batch_fn, batched=True, collective=mode,
                                                      gather_images=True)
        # (numpy: pickled by value - a tensor travels as a file descriptor the parent may open after this process has gone)
        res[mode] = ({k: v.numpy().copy() for k, v in grads.items()}, radii.numpy().copy(), [(v, c.numpy().copy()) for v, c, _, _ in outs])
    q.put((rank, res))
    dist.barrier()
    try:
        dist.destroy_process_group()
    except Exception:
        pass


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,num_views", [(2, 4), (2, 5)])
def test_batched_rank_views_one_collective_matches_the_serial_loop_to_rounding(world, num_views):
    """batched=True: every rank renders ITS views in one batched call and one collective ends the step.  Same values as the
    serial loop to rounding (the sum is associated per rank first), the SAME BITS on every rank and in both collective
    modes, radii max exact, all images on every rank in view order - also with a ragged shard (5 views on 2 ranks)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_batched_worker, args=(r, world, port, q, num_views)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
    sc, cams, params = _scene_and_cams()
    cams = (cams * 2)[:num_views]
    ref, rref, _ = vp.render_views_parallel(cams, params, sc["bg"].double(), 1, _loss_grad, render_fn=_oracle_render_fn, pipeline=True)
    tt = torch.from_numpy
    res = [(rank, {m: ({k: tt(v) for k, v in g.items()}, tt(r), [(v, tt(c)) for v, c in o]) for m, (g, r, o) in modes.items()})
           for rank, modes in res]
    g0, r0, o0 = res[0][1]["allgather"]
    for rank, modes in res:
        for mode, (grads, radii, outs) in modes.items():
            assert torch.equal(radii, rref)
            for k in ref:
                assert torch.equal(grads[k], g0[k]), (rank, mode, k)                       # same bits everywhere
                scale = max(float(ref[k].abs().max()), 1e-30)
                assert float((grads[k].double() - ref[k].double()).abs().max()) <= 1e-5 * scale, (rank, mode, k)
            assert [v for v, _ in outs] == list(range(num_views))
            for (v, c), (v2, c2) in zip(outs, o0):
                assert v == v2 and torch.equal(c, c2)


def test_explicit_pipeline_with_scatter_is_refused_and_an_empty_view_list_gives_zeros():
    sc, cams, params = _scene_and_cams()
    with pytest.raises(ValueError):
        vp.render_views_parallel(cams, params, sc["bg"].double(), 1, _loss_grad, render_fn=_oracle_render_fn, pipeline=True,
                                 collective="scatter")
    with pytest.raises(ValueError):
        vp.render_views_parallel(cams, params, sc["bg"].double(), 1, _loss_grad, render_fn=_oracle_render_fn, pipeline=True, batched=True)
    grads, radii, outs = vp.render_views_parallel([], params, sc["bg"].double(), 1, _loss_grad, render_fn=_oracle_render_fn, pipeline=True)
    assert outs == [] and int(radii.abs().max()) == 0 and all(float(g.abs().max()) == 0 for g in grads.values())


# ------------------------------------------------------------------ animation frames sharded over ranks

def _frame_image(i):
    g = torch.Generator().manual_seed(1000 + i)
    return torch.rand(3, 8, 12, generator=g)


def _frames_worker(rank, world, port, q, frames, gather):
    from humangaussian_amd import animation as an
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rendered = []

    def render(i):
        rendered.append(i)
        return _frame_image(i)
    got = [(i, img.numpy().copy()) for i, img in an.render_frames_parallel(frames, render, gather=gather)]
    q.put((rank, rendered, got))
    dist.barrier()
    try:
        dist.destroy_process_group()
    except Exception:
        pass


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,nframes,gather", [(2, 7, True), (3, 7, True), (2, 6, False)])
def test_animation_frames_are_sharded_round_robin_and_come_back_in_frame_order(world, nframes, gather):
    """configs[4]: frame k of the sequence belongs to rank k mod G (animation.py:966-1004 loops over them on one GPU); with
    the image gather every rank yields EVERY frame, in order, bit-identical to what its owner rendered - also when the
    last round is ragged; without it a rank yields only its own frames and no collective runs."""
    frames = [10 + 3 * k for k in range(nframes)]                    # (frame ids are arbitrary, not 0..n-1)
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_frames_worker, args=(r, world, port, q, frames, gather)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
    for rank, rendered, got in res:
        mine = [frames[k] for k in vp.shard_views(nframes, rank, world)]
        assert rendered == mine                                       # every rank renders only its own frames, in order
        want = frames if gather else mine
        assert [i for i, _ in got] == want
        for i, img in got:
            assert torch.equal(torch.from_numpy(img), _frame_image(i))


def test_animation_frames_without_a_process_group_is_the_plain_loop():
    from humangaussian_amd import animation as an
    got = list(an.render_frames_parallel(range(5), _frame_image))
    assert [i for i, _ in got] == list(range(5)) and all(torch.equal(img, _frame_image(i)) for i, img in got)


def test_auto_collective_rule_and_fused_unpack_fallback():
    """`collective="auto"` without a timing at hand: all-gather up to four ranks, scatter from eight on (7 vs
    1.75 packs per rank on the xGMI links at world 8); on tensors that are not on a HIP device the fused reduce + unpack is
    reduce, then unpack (the same bits)."""
    assert [vp.default_collective(w) for w in (1, 2, 4, 8, 16)] == ["allgather", "allgather", "allgather", "scatter", "scatter"]
    g = torch.Generator().manual_seed(0)
    P, M = 40, 4
    gathered = torch.randn(3, P, 15 + 3 * M, generator=g)
    gathered[:, :, -1] = torch.randint(0, 30, (3, P), generator=g).float()
    shapes = {"means3D": (P, 3), "means2D": (P, 3), "shs": (P, M, 3), "opacities": (P, 1), "scales": (P, 3), "rotations": (P, 4)}
    for acc in (None, torch.randn(P, 15 + 3 * M, generator=g).abs()):
        ref_g, ref_r = vp.unpack_contribution(vp.reduce_gathered(gathered, acc), shapes)
        got_g, got_r = vp.reduce_gathered_unpacked(gathered, acc, shapes)
        assert torch.equal(got_r, ref_r)
        for k in vp.GRAD_KEYS:
            assert torch.equal(got_g[k], ref_g[k]) and got_g[k].shape == torch.Size(shapes[k])
    # one rank, no process group: "auto" resolves and the step runs
    sc, cams, params = _scene_and_cams()
    grads, radii, outs = vp.render_views_parallel(cams[:2], params, sc["bg"].double(), 1, _loss_grad, render_fn=_oracle_render_fn,
                                                  collective="auto", pipeline=False)
    assert len(outs) == 2 and radii.dtype == torch.int32 and grads["means3D"].shape == params["means3D"].shape
