#!/usr/bin/env python3
"""Config 5: a frame sequence at 3840x2160 (1M-triangle terrain, the two orbiting lights of src/lib.rs:90-102 per
frame `time`), rendered to 8-bit PQ colour and sharded over the ranks of one node.

    python tools/config5.py [frames]                                       one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/config5.py [frames]

Frame mode: frame f is rendered by rank f mod N; every round of N frames is gathered to rank 0 (RCCL).  Then one
frame in tile mode: interleaved 8-row strips over the ranks, gathered and assembled on rank 0.  Rank 0 checks that
the tile-mode frame is byte-identical to the same frame rendered whole, and prints frames/s.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from raytracer_amd import rtg, shard
import bench

W, H = 3840, 2160


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    tris = rtg.heightfield(*bench.GRID, bench.SEED, bench.LOCAL_SCALE)
    bvh = rtg.HostBvh(tris)
    ctx = rtg.Context(local_rank)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    if world > 1:
        ctx.set_option("max_blocks", 1248)     # leave wave slots for the RCCL kernels (see bench.py)
    mesh = ctx.mesh(bvh)
    scene = ctx.scene([rtg.mesh_object(mesh, rtg.material((0.8, 0.7, 0.6)), rtg.transform(1.0 / bench.LOCAL_SCALE))])

    def lights_view(frame):
        _, lights, view = rtg.scene_builder(frame)          # SceneBuilder{time: frame}: the light orbit
        return np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE), view

    def render_frame(frame, out):                           # whole frame -> rgba8 bytes
        lights, view = lights_view(frame)
        scene.render_frame_raw(view, lights, len(lights), W, H, (0, 0, W, H), 4, None, out.data_ptr(), None)

    warm = torch.empty(W * H * 4, dtype=torch.uint8, device=dev)
    render_frame(0, warm)
    torch.cuda.synchronize(dev)
    dist.barrier()
    t0 = time.perf_counter()
    frames = shard.render_sequence_frames(render_frame, n_frames, W * H * 4, torch.uint8, dev)
    torch.cuda.synchronize(dev)
    dist.barrier()
    seq_s = time.perf_counter() - t0

    # one frame in tile mode
    probe = min(7, n_frames - 1)
    lights, view = lights_view(probe)

    def render_rect(rect, out):
        scene.render_frame_raw(view, lights, len(lights), W, H, rect, 4, None, out.data_ptr(), None)

    dist.barrier()
    t0 = time.perf_counter()
    tiled = shard.render_frame_tiles(render_rect, W, H, 4, torch.uint8, dev, interleave=4)
    torch.cuda.synchronize(dev)
    dist.barrier()
    tile_s = time.perf_counter() - t0
    if rank == 0:
        same = bool(torch.equal(tiled.reshape(-1), frames[probe]))
        lit = float((frames[probe].view(H, W, 4)[:, :, :3].sum(dim=2) > 0).float().mean())
        print(f"config 5: {n_frames} frames {W}x{H} on {world} GPU(s): {seq_s:.3f} s = {n_frames / seq_s:.1f} frames/s "
              f"({n_frames * W * H / seq_s / 1e6:.0f} Mpixels/s); tile-mode frame {tile_s * 1e3:.1f} ms, "
              f"identical to the whole frame: {same}; lit pixels {lit:.2f}", flush=True)
        assert same
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
