"""CPU, world_size 2 over gloo: the sharding / gather logic of the multi-GPU path, with the oracle standing in for
the GPU renderer.  The gathered frame must be byte-identical to the single-process frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from raytracer_amd import shard  # noqa: E402


def test_tile_strips_cover_the_frame_once():
    for (w, h, world, inter) in [(64, 64, 2, 4), (1920, 1080, 8, 4), (33, 7, 4, 2), (16, 8, 8, 4), (100, 1000, 3, 1)]:
        plan = shard.tile_strips(w, h, world, inter)
        assert len(plan) == world
        cover = np.zeros((h, w), dtype=np.int32)
        for rects in plan:
            for x0, y0, rw, rh in rects:
                assert x0 == 0 and rw == w and (y0 % 8 == 0)
                cover[y0:y0 + rh, x0:x0 + rw] += 1
        assert (cover == 1).all()
        sizes = [shard.rect_pixels(r) for r in plan]
        if h >= 8 * world * inter:
            assert max(sizes) - min(sizes) <= 8 * w * inter
    assert shard.frames_of_rank(1, 4, 10) == [1, 5, 9] and shard.frame_owner(9, 4) == 1


def test_piece_plans_tile_the_receive_buffer_once():
    for (w, h, world, inter, px) in [(64, 64, 2, 4, 16), (1920, 1080, 8, 4, 4), (33, 7, 4, 2, 16), (100, 1000, 3, 1, 1)]:
        plan = shard.tile_strips(w, h, world, inter)
        pieces = shard.strip_pieces(plan, w, px)
        cover = np.zeros(w * h * px, dtype=np.int32)
        sent = [0] * world
        for r, so, ro, nb in pieces:
            assert so == sent[r]                       # a rank's pieces sit back to back in its send buffer
            sent[r] += nb
            cover[ro:ro + nb] += 1
        assert (cover == 1).all()
        assert sent == [shard.rect_pixels(rects) * px for rects in plan]
    assert shard.frame_pieces(3, 10) == [(0, 0, 0, 10), (1, 0, 10, 10), (2, 0, 20, 10)]
    assert shard.frame_pieces(3, 10, live=[True, False, True]) == [(0, 0, 0, 10), (2, 0, 20, 10)]


def test_ray_ranges_cover_the_batch_once():
    for n, world in [(0, 3), (1, 4), (63, 2), (64, 2), (65, 2), (1000, 3), (67108864, 8), (50000, 7), (130, 8)]:
        ranges = shard.ray_ranges(n, world)
        assert len(ranges) == world and ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))          # contiguous, in rank order
        assert all((b % 64 == 0 or b == n) and (e % 64 == 0 or e == n) and b <= e for b, e in ranges)
        sizes = [e - b for b, e in ranges]
        assert max(sizes) - min(sizes) < 128
        cover = np.zeros(n * 16, dtype=np.int32)
        for r, so, ro, nb in shard.ray_pieces(ranges, 16):
            assert so == 0 and nb == sizes[r] * 16 and ro == ranges[r][0] * 16
            cover[ro:ro + nb] += 1
        assert (cover == 1).all()
    with pytest.raises(ValueError):
        shard.ray_ranges(10, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import orc
        import scenes
        from raytracer_amd import rtg
        w, h = 96, 72
        # every rank builds the same (replicated) scene
        tris, xform = scenes.terrain(24, 24, seed=7)
        bvh = rtg.HostBvh(tris)
        objs, lights, view = scenes.mesh_scene_objects(xform, time=10)
        pair = scenes.Pair(None, objs, {2: bvh})

        def render_rect(rect, out):                      # 16-byte hit records as bytes
            hits, _, _ = pair.cpu.render_primary(view, w, h, tile=rect, threads=1)
            out.copy_(torch.from_numpy(np.frombuffer(hits.tobytes(), dtype=np.uint8).copy()))

        frame = shard.render_frame_tiles(render_rect, w, h, 16, torch.uint8, "cpu", interleave=3)
        ok_tiles = True
        if rank == 0:
            full, _, _ = pair.cpu.render_primary(view, w, h, threads=1)
            ok_tiles = frame.numpy().tobytes() == full.tobytes()

        def render_frame(f, out):                        # frame mode: the light orbit changes per frame
            o2, l2, v2 = scenes.mesh_scene_objects(xform, time=f)
            rgb, c8, _ = pair.cpu.render_frame(v2, l2, 40, 30, max_depth=2, threads=1)
            out.copy_(torch.from_numpy(c8.reshape(-1).copy()))

        frames = shard.render_sequence_frames(render_frame, 5, 40 * 30 * 4, torch.uint8, "cpu")
        ok_frames = True
        if rank == 0:
            assert sorted(frames) == [0, 1, 2, 3, 4]
            for f in range(5):
                o2, l2, v2 = scenes.mesh_scene_objects(xform, time=f)
                _, c8, _ = pair.cpu.render_frame(v2, l2, 40, 30, max_depth=2, threads=1)
                ok_frames = ok_frames and frames[f].numpy().tobytes() == c8.tobytes()
            with open(result_path, "w") as fh:
                fh.write(f"{int(ok_tiles)} {int(ok_frames)}")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_gather_equals_single_process(tmp_path):
    result = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, _free_port(), str(result)), nprocs=2, join=True)
    assert result.read_text() == "1 1"


def _worker_unequal(rank, world, port, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sizes = [5, 0, 9][:world]
        local = torch.arange(sizes[rank], dtype=torch.int32) + 100 * rank
        bufs = shard.gather_to_root(local, sizes)
        if rank == 0:
            ok = all(torch.equal(bufs[r], torch.arange(sizes[r], dtype=torch.int32) + 100 * r) for r in range(world))
        # preallocated receive buffers, unequal and equal sizes, reused over two rounds
        out = [torch.full((n,), -1, dtype=torch.int32) for n in sizes] if rank == 0 else None
        eq_out = [torch.full((4,), -1, dtype=torch.int32) for _ in range(world)] if rank == 0 else None
        for rnd in range(2):
            bufs = shard.gather_to_root(local + rnd, sizes, out=out)
            eq = shard.gather_to_root(torch.full((4,), 10 * rank + rnd, dtype=torch.int32), [4] * world, out=eq_out)
            if rank == 0:
                ok = ok and all(bufs[r] is out[r] for r in range(world)) and all(eq[r] is eq_out[r] for r in range(world))
                ok = ok and all(torch.equal(out[r], torch.arange(sizes[r], dtype=torch.int32) + 100 * r + rnd)
                                for r in range(world))
                ok = ok and all(torch.equal(eq_out[r], torch.full((4,), 10 * r + rnd, dtype=torch.int32))
                                for r in range(world))
        # a ray batch cut into contiguous ranges (the bench_mesh distribution): records in ray order on the root;
        # 150 rays over 3 ranks leaves the last rank a ragged range, 100 rays over 3 ranks leaves it none
        import orc
        import scenes
        from raytracer_amd import rtg
        tris, xform = scenes.terrain(12, 12, seed=3)
        objs, _, _ = scenes.mesh_scene_objects(xform, time=0)
        pair = scenes.Pair(None, objs, {2: rtg.HostBvh(tris)})
        for n in (150, 100, 0):
            rays = rtg.bench_mesh_rays((-1.5, -1.5, -1.5, 1.5, 1.5, 1.5), n)

            def trace(begin, end, out, rays=rays):
                hits, _, _ = pair.cpu.trace(rays[begin:end], threads=1)
                out.copy_(torch.from_numpy(np.frombuffer(hits.tobytes(), dtype=np.uint8).copy()))

            got = shard.trace_rays_sharded(trace, n, 16, torch.uint8, "cpu")
            if rank == 0:
                full, _, _ = pair.cpu.trace(rays, threads=1)
                ok = ok and got.shape == (n, 16) and got.numpy().tobytes() == full.tobytes()
            else:
                ok_other = got is None
                assert ok_other
        if rank == 0:
            with open(result_path, "w") as fh:
                fh.write(str(int(ok)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_unequal_and_empty_shards(tmp_path):
    result = tmp_path / "result.txt"
    mp.spawn(_worker_unequal, args=(3, _free_port(), str(result)), nprocs=3, join=True)
    assert result.read_text() == "1"
