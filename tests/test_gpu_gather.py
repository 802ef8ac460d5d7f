"""GPU: rtg_gather_tiles with one rank (the gather degenerates to device copies on the stream; no RCCL needed), and
the piece plans of frame and tile mode.  The multi-rank partition itself is covered on CPUs in test_shard_gloo.py."""
import numpy as np
import pytest

import scenes
from raytracer_amd import rtg, shard

pytestmark = pytest.mark.gpu


def test_one_rank_gather_assembles_the_frame_from_strips(gpu_ctx):
    import torch
    tris, xform = scenes.terrain(50, 50, seed=5)
    mesh = gpu_ctx.mesh(rtg.HostBvh(tris))
    scene = gpu_ctx.scene([rtg.mesh_object(mesh, rtg.material((1, 1, 1)), xform)])
    view = rtg.make_view()
    w, h = 328, 203
    whole = torch.empty(w * h * 16, dtype=torch.uint8, device="cuda")
    scene.render_primary_raw(view, w, h, (0, 0, w, h), whole.data_ptr())
    # tile mode with one rank: the strips are rendered back to back into a send buffer in plan order ...
    plan = shard.tile_strips(w, h, world=1, interleave=5)
    pieces = shard.strip_pieces(plan, w, 16)
    assert sum(p[3] for p in pieces) == w * h * 16 and len(pieces) == 5
    send = torch.empty(w * h * 16, dtype=torch.uint8, device="cuda")
    for (x0, y0, tw, th), (_, so, _, nb) in zip(plan[0], pieces):
        scene.render_primary_raw(view, w, h, (x0, y0, tw, th), send.data_ptr() + so)
        assert nb == tw * th * 16
    # ... and gathered in a shuffled piece order into a poisoned frame
    recv = torch.full((w * h * 16,), 0x5A, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    comm = rtg.Comm(gpu_ctx)
    comm.gather(0, send.data_ptr(), recv.data_ptr(), pieces[::-1])
    gpu_ctx.synchronize()
    assert recv.cpu().numpy().tobytes() == whole.cpu().numpy().tobytes()
    # a caller's stream is honoured
    st = torch.cuda.Stream()
    recv.fill_(0)
    torch.cuda.synchronize()
    comm.gather(0, send.data_ptr(), recv.data_ptr(), pieces, stream_handle=st.cuda_stream)
    st.synchronize()
    assert recv.cpu().numpy().tobytes() == whole.cpu().numpy().tobytes()
    # in-place pieces (send already sits where it lands) are left alone
    comm.gather(0, whole.data_ptr(), whole.data_ptr(), [(0, 0, 0, w * h * 16)])
    with pytest.raises(rtg.RtgError, match="rank 1 of 1"):
        comm.gather(0, send.data_ptr(), recv.data_ptr(), [(1, 0, 0, 16)])
    with pytest.raises(rtg.RtgError, match="device memory"):
        comm.gather(0, np.zeros(16, np.uint8), recv.data_ptr(), [(0, 0, 0, 16)])
    comm.close()
    scene.close()


def test_one_rank_gather_through_rccl(gpu_ctx):
    """The machinery several ranks use -- RCCL bound at run time, ncclGetUniqueId, ncclCommInitRank, one group of
    ncclSend / ncclRecv on the caller's stream, ncclCommDestroy -- on a single GPU: with "comm_loopback" the one-rank
    communicator is a real RCCL communicator and the root sends its own pieces to itself.  Same bytes as the copies,
    also while a traversal runs on another stream."""
    import torch
    tris, xform = scenes.terrain(60, 60, seed=9)
    mesh = gpu_ctx.mesh(rtg.HostBvh(tris))
    scene = gpu_ctx.scene([rtg.mesh_object(mesh, rtg.material((1, 1, 1)), xform)])
    view = rtg.make_view()
    w, h = 640, 360
    n = w * h * 16
    whole = torch.empty(n, dtype=torch.uint8, device="cuda")
    scene.render_primary_raw(view, w, h, (0, 0, w, h), whole.data_ptr())
    plan = shard.tile_strips(w, h, world=1, interleave=9)
    pieces = shard.strip_pieces(plan, w, 16)
    send = torch.empty(n, dtype=torch.uint8, device="cuda")
    for (x0, y0, tw, th), (_, so, _, nb) in zip(plan[0], pieces):
        scene.render_primary_raw(view, w, h, (x0, y0, tw, th), send.data_ptr() + so)
    gpu_ctx.synchronize()
    gpu_ctx.set_option("comm_loopback", 2)
    with pytest.raises(rtg.RtgError, match="id is null"):
        rtg.Comm(gpu_ctx)
    comm = rtg.Comm(gpu_ctx, 1, 0, rtg.comm_unique_id())
    gpu_ctx.set_option("comm_loopback", 1)
    recv = torch.full((n,), 0x5A, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    comm.gather(0, send.data_ptr(), recv.data_ptr(), pieces[::-1])
    gpu_ctx.synchronize()
    assert recv.cpu().numpy().tobytes() == whole.cpu().numpy().tobytes()
    # on a second stream, next to a frame being traced on the context's stream (what bench.py --gpus N does)
    st = torch.cuda.Stream()
    busy = torch.empty(1920 * 1080 * 16, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        recv.fill_(0)
        torch.cuda.synchronize()
        scene.render_primary_raw(view, 1920, 1080, (0, 0, 1920, 1080), busy.data_ptr())
        comm.gather(0, send.data_ptr(), recv.data_ptr(), pieces, stream_handle=st.cuda_stream)
        st.synchronize()
        gpu_ctx.synchronize()
        assert recv.cpu().numpy().tobytes() == whole.cpu().numpy().tobytes()
    # the root of a frame-mode run traces its own frame into slot 0 of the gathered buffer: that piece is in place
    # (nothing is sent for it), the next slot arrives through RCCL
    half = n // 2
    slots = torch.full((n,), 0x5A, dtype=torch.uint8, device="cuda")
    slots[:half] = whole[:half]
    torch.cuda.synchronize()
    comm.gather(0, slots.data_ptr(), slots.data_ptr(), [(0, 0, 0, half), (0, 0, half, half)])
    gpu_ctx.synchronize()
    first = whole[:half].cpu().numpy().tobytes()
    assert slots.cpu().numpy().tobytes() == first + first
    # all pieces in place: an empty group
    comm.gather(0, slots.data_ptr(), slots.data_ptr(), [(0, 0, 0, n)])
    gpu_ctx.synchronize()
    assert slots.cpu().numpy().tobytes() == first + first
    comm.close()
    scene.close()


def test_ray_batch_ranges_assemble_in_ray_order(gpu_ctx):
    """A batch of bench_mesh rays (src/tree/benchmark.rs:15-27) cut as for three ranks: every range traced on its own
    from device memory into the send buffer, gathered by rtg_gather_tiles into a poisoned buffer -- the records of the
    whole batch in ray order, equal to the oracle's."""
    import torch
    tris, xform = scenes.terrain(40, 40, seed=11)
    bvh = rtg.HostBvh(tris)
    objs, _, _ = scenes.mesh_scene_objects(xform, with_default=False)
    pair = scenes.Pair(gpu_ctx, objs, {0: bvh})
    n = 50_000                                     # 782 chunks of 64 rays: ranges of 261, 261 and 260 (ragged) chunks
    rays = rtg.bench_mesh_rays((-1.2, -1.2, -1.2, 1.2, 1.2, 1.2), n)
    d_rays = torch.from_numpy(rays).cuda()
    ranges = shard.ray_ranges(n, 3)
    assert ranges == [(0, 16704), (16704, 33408), (33408, 50000)]
    send = torch.empty(n * 16, dtype=torch.uint8, device="cuda")
    for b, e in ranges:
        pair.gpu.trace_raw(d_rays.data_ptr() + b * 48, e - b, send.data_ptr() + b * 16)
    # as one rank the three ranges sit back to back in the send buffer: the pieces of ray_pieces, re-based
    pieces = [(0, ro, ro, nb) for _, _, ro, nb in shard.ray_pieces(ranges, 16)]
    recv = torch.full((n * 16,), 0x5A, dtype=torch.uint8, device="cuda")
    comm = rtg.Comm(gpu_ctx)
    comm.gather(0, send.data_ptr(), recv.data_ptr(), pieces[::-1])
    gpu_ctx.synchronize()
    want, _, _ = pair.cpu.trace(rays)
    assert recv.cpu().numpy().tobytes() == want.tobytes()
    assert (want["leaf"] != 0xFFFFFFFF).sum() > 100
    comm.close()
