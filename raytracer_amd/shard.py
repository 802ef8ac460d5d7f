"""Sharding of the ray-cast path over the GPUs of one node, one process per GPU.

The path has no exchange step during compute: every pixel / ray is independent (the reference's pixel pass,
src/render/renderer.rs:147-149, maps over `&self`) and frames are independent given `time` (src/bin/local.rs:42-50;
the reference itself fans frames out over HTTP workers, scripts/request_all.sh:2).  So the scene/BVH is replicated on
every GPU and only finished framebuffer pieces move: one gather to the root per frame, over RCCL (backend "nccl")
on GPUs or gloo on CPUs -- the functions here only see `torch.distributed`.

Two partitions are offered:

* frame mode  -- frame f is rendered by rank f mod world (weak scaling: a sequence renders `world` frames at once);
* tile mode   -- one frame is cut into horizontal strips of whole 8-row wave tiles, dealt round-robin
                 (`interleave` strips per rank) so sky-heavy and geometry-heavy rows are spread evenly.

A batch of arbitrary rays (the reference's `bench_mesh`, src/tree/benchmark.rs:15-27, shadow or photon rays) is cut
into contiguous ranges of whole 64-ray wave chunks, one per rank; the hit records land at their rays' positions.
"""
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist

Rect = Tuple[int, int, int, int]   # x0, y0, w, h


def frame_owner(frame: int, world: int) -> int:
    return frame % world


def frames_of_rank(rank: int, world: int, n_frames: int) -> List[int]:
    return list(range(rank, n_frames, world))


def tile_strips(width: int, height: int, world: int, interleave: int = 4, tile_rows: int = 8) -> List[List[Rect]]:
    """Per-rank list of full-width strips.  Strip heights are multiples of `tile_rows` (except the last strip of the
    frame); strip s goes to rank s % world; about `interleave` strips per rank."""
    if world <= 0:
        raise ValueError("world must be positive")
    n_tile_rows = (height + tile_rows - 1) // tile_rows
    n_strips = max(1, min(n_tile_rows, world * max(1, interleave)))
    per = [n_tile_rows // n_strips + (1 if i < n_tile_rows % n_strips else 0) for i in range(n_strips)]
    out: List[List[Rect]] = [[] for _ in range(world)]
    y = 0
    for s, rows in enumerate(per):
        h = min(rows * tile_rows, height - y)
        if h > 0:
            out[s % world].append((0, y, width, h))
        y += h
    assert y == height
    return out


def rect_pixels(rects: Sequence[Rect]) -> int:
    return sum(w * h for _, _, w, h in rects)


def strip_pieces(plan: Sequence[Sequence[Rect]], width: int, pixel_bytes: int):
    """Piece list for rtg_gather_tiles from a tile_strips plan: every rank keeps its strips back to back in its send
    buffer (plan order) and a full-width strip lands at its rows of the frame, so the gather assembles the image.
    Returns [(rank, send_offset, recv_offset, bytes)]."""
    pieces = []
    for rank, rects in enumerate(plan):
        off = 0
        for x0, y0, w, h in rects:
            assert x0 == 0 and w == width, "strips are full-width"
            nb = w * h * pixel_bytes
            pieces.append((rank, off, y0 * width * pixel_bytes, nb))
            off += nb
    return pieces


def frame_pieces(world: int, frame_bytes: int, live=None):
    """Piece list of frame mode: rank r's finished frame lands in slot r of the root's receive buffer."""
    return [(r, 0, r * frame_bytes, frame_bytes) for r in range(world) if live is None or live[r]]


def ray_ranges(n_rays: int, world: int, chunk: int = 64) -> List[Tuple[int, int]]:
    """Per-rank [begin, end) of a ray batch: contiguous, in rank order, every boundary but the last a multiple of
    `chunk` (a wave takes 64 consecutive rays), sizes differing by less than two chunks.  Ranks beyond the batch get an
    empty range."""
    if world <= 0:
        raise ValueError("world must be positive")
    if n_rays < 0:
        raise ValueError("negative ray count")
    n_chunks = (n_rays + chunk - 1) // chunk
    out, begin = [], 0
    for r in range(world):
        end = min(n_rays, begin + (n_chunks // world + (1 if r < n_chunks % world else 0)) * chunk)
        out.append((begin, end))
        begin = end
    assert begin == n_rays
    return out


def ray_pieces(ranges: Sequence[Tuple[int, int]], record_bytes: int):
    """Piece list of a sharded ray batch: rank r's records (from the start of its send buffer) land at its rays'
    positions in the root's receive buffer."""
    return [(r, 0, b * record_bytes, (e - b) * record_bytes) for r, (b, e) in enumerate(ranges) if e > b]


def gather_to_root(local: torch.Tensor, sizes: Sequence[int], dst: int = 0, group=None, out=None):
    """Gathers each rank's flat buffer (`sizes[r]` elements on rank r) to rank `dst`.
    Equal sizes use one `gather` collective; unequal sizes use grouped point-to-point sends, which is what RCCL
    does for gather anyway.  `out` (on `dst` only) is an optional list of preallocated per-rank receive tensors, so
    a per-frame loop allocates nothing.  Returns the list of per-rank tensors on `dst`, None elsewhere.
    The collective is enqueued on the caller's current stream."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert local.numel() == sizes[rank], (local.numel(), sizes[rank])
    if out is not None and rank == dst:
        assert len(out) == world and all(out[r].numel() == sizes[r] for r in range(world))
    if len(set(sizes)) == 1:
        bufs = None
        if rank == dst:
            bufs = out if out is not None else [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, bufs, dst=dst, group=group)
        return bufs
    if rank == dst:
        if out is not None:
            out[dst].copy_(local)
            bufs = list(out)
        else:
            bufs = [local if r == dst else torch.empty(sizes[r], dtype=local.dtype, device=local.device) for r in range(world)]
        ops = [dist.P2POp(dist.irecv, bufs[r], r, group) for r in range(world) if r != dst and sizes[r] > 0]
    else:
        bufs = None
        ops = [dist.P2POp(dist.isend, local, dst, group)] if sizes[rank] > 0 else []
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return bufs


def gather_pieces(send: torch.Tensor, recv, pieces, dst: int = 0, group=None):
    """The protocol of rtg_gather_tiles over `torch.distributed` point-to-point operations (gloo on CPUs): every rank
    passes the same piece list [(rank, send_offset, recv_offset, bytes)] over flat uint8 tensors; a rank sends its
    own pieces in list order, `dst` receives them in list order straight into `recv` and copies its own pieces.
    The stand-in for the native gather where there is no GPU, and the check that a piece plan assembles the frame."""
    rank = dist.get_rank(group)
    ops = []
    for r, so, ro, nb in pieces:
        if nb == 0:
            continue
        if rank == dst:
            if r == dst:
                recv[ro:ro + nb].copy_(send[so:so + nb])
            else:
                ops.append(dist.P2POp(dist.irecv, recv[ro:ro + nb], r, group))
        elif r == rank:
            ops.append(dist.P2POp(dist.isend, send[so:so + nb], dst, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def _gather_finished(device, gather):
    """After a caller-supplied gather: a native gather (rtg.Comm.gather) enqueues its copies and RCCL transfers on a
    stream torch knows nothing about, so nothing torch does afterwards -- reading the assembled result on the root,
    rendering the next piece into the send buffer on any rank -- is ordered behind it.  The GPU is drained here, once
    per gather; a caller that wants the gather to overlap other work drives rtg_gather_tiles itself with events, as
    bench.py does.  The default gather runs through torch.distributed on the current stream and needs none of this."""
    if gather is not None and torch.device(device).type != "cpu":
        torch.cuda.synchronize(torch.device(device))


def render_frame_tiles(render_rect, width: int, height: int, channels: int, dtype, device, interleave: int = 4,
                       dst: int = 0, group=None, gather=None):
    """Tile mode: this rank renders its strips with `render_rect(rect, out_flat_tensor)` back to back into its send
    buffer; one gather lands every strip at its rows of the frame on `dst`, which returns the (height, width,
    channels) frame; other ranks return None.  `gather(send_bytes, recv_bytes_or_None, pieces)` moves the pieces
    (default: gather_pieces over torch.distributed; GPU callers pass rtg.Comm.gather on device addresses)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    plan = tile_strips(width, height, world, interleave)
    item = torch.empty((), dtype=dtype).element_size()
    pieces = strip_pieces(plan, width, channels * item)
    local = torch.empty(rect_pixels(plan[rank]) * channels, dtype=dtype, device=device)
    off = 0
    for rect in plan[rank]:
        n = rect[2] * rect[3] * channels
        render_rect(rect, local[off:off + n])
        off += n
    frame = torch.empty((height, width, channels), dtype=dtype, device=device) if rank == dst else None
    send_b = local.view(torch.uint8)
    recv_b = frame.view(-1).view(torch.uint8) if frame is not None else None
    if gather is None:
        gather_pieces(send_b, recv_b, pieces, dst, group)
    else:
        gather(send_b, recv_b, pieces)
    _gather_finished(device, gather)
    return frame


def render_sequence_frames(render_frame, n_frames: int, frame_elems: int, dtype, device, dst: int = 0, group=None,
                           gather=None):
    """Frame mode: rank r renders frames r, r+world, ...; after each round the `world` frames are gathered on `dst`
    (rank r's frame into slot r of the receive buffer).  `render_frame(frame_index, out_flat_tensor)`; `gather` as in
    render_frame_tiles.  Returns {frame index: tensor} on `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out = {} if rank == dst else None
    local = torch.empty(frame_elems, dtype=dtype, device=device)
    slots = torch.empty((world, frame_elems), dtype=dtype, device=device) if rank == dst else None
    frame_bytes = frame_elems * local.element_size()
    for base in range(0, n_frames, world):
        live = [base + r < n_frames for r in range(world)]
        if live[rank]:
            render_frame(base + rank, local)
        pieces = frame_pieces(world, frame_bytes, live)
        send_b = local.view(torch.uint8)
        recv_b = slots.view(-1).view(torch.uint8) if slots is not None else None
        if gather is None:
            gather_pieces(send_b, recv_b, pieces, dst, group)
        else:
            gather(send_b, recv_b, pieces)
        _gather_finished(device, gather)      # every rank: `local` is rendered into again in the next round
        if rank == dst:
            for r in range(world):
                if live[r]:
                    out[base + r] = slots[r].clone()
    return out


def trace_rays_sharded(trace, n_rays: int, record_elems: int, dtype, device, dst: int = 0, group=None, gather=None):
    """Batch mode: this rank traces its contiguous range with `trace(begin, end, out_flat_tensor)` (`record_elems`
    elements of `dtype` per ray); one gather lands every range at its rays' positions on `dst`, which returns the
    (n_rays, record_elems) records in ray order; other ranks return None.  `gather` as in render_frame_tiles."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ranges = ray_ranges(n_rays, world)
    begin, end = ranges[rank]
    local = torch.empty((end - begin) * record_elems, dtype=dtype, device=device)
    if end > begin:
        trace(begin, end, local)
    pieces = ray_pieces(ranges, record_elems * local.element_size())
    out = torch.empty((n_rays, record_elems), dtype=dtype, device=device) if rank == dst else None
    send_b = local.view(torch.uint8)
    recv_b = out.view(-1).view(torch.uint8) if out is not None else None
    if gather is None:
        gather_pieces(send_b, recv_b, pieces, dst, group)
    else:
        gather(send_b, recv_b, pieces)
    _gather_finished(device, gather)
    return out
