#!/usr/bin/env python3
"""A/B timing of the persistent kernel on the bench frame (config 2) for one build of librtg.so
(RTG_LIBRTG_PATH selects the build).  Prints kernel ms for a launch alone and with two launches in flight, node visits
per ray, and visits per second.  Usage: tools/ab_time.py [label] [key=value ctx options...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from raytracer_amd import rtg

label = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("RTG_LIBRTG_PATH", "tree")
opts = dict(kv.split("=") for kv in sys.argv[2:])
tris = rtg.heightfield(*bench.GRID, bench.SEED, bench.LOCAL_SCALE)
bvh = rtg.HostBvh(tris)
W, H = bench.WIDTH, bench.HEIGHT
n = W * H
ctxs, scenes, streams, bufs = [], [], [], []
mesh = None
for i in range(2):
    c = rtg.Context(0)
    for k, v in opts.items():
        c.set_option(k, int(v))
    st = torch.cuda.Stream()
    c.set_stream(st.cuda_stream)
    if mesh is None:
        mesh = c.mesh(bvh)
    ctxs.append(c)
    streams.append(st)
    scenes.append(c.scene([rtg.mesh_object(mesh, rtg.material((1, 1, 1)), rtg.transform(1.0 / bench.LOCAL_SCALE))]))
    bufs.append(torch.empty(n * 16, dtype=torch.uint8, device="cuda"))
ctr = torch.empty(n * 2, dtype=torch.int32, device="cuda")
tile = (0, 0, W, H)
scenes[0].render_primary_raw(bench.frame_view(rtg, 0), W, H, tile, bufs[0].data_ptr(), None, ctr.data_ptr())
torch.cuda.synchronize()
visits = float(ctr.view(n, 2)[:, 0].to(torch.int64).sum().item())
tests = float(ctr.view(n, 2)[:, 1].to(torch.int64).sum().item())


def run(inflight, frames=40):
    for c in ctxs:
        c.set_option("time_kernels", 2)
        c.kernel_time()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for f in range(frames):
        k = f % inflight
        scenes[k].render_primary_raw(bench.frame_view(rtg, f), W, H, tile, bufs[k].data_ptr())
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / frames * 1e3
    ms = sum(c.kernel_time()[0] for c in ctxs) / frames
    return ms, wall


run(1, 6)
alone_ms, alone_wall = run(1)
two_ms, two_wall = run(2)
print(f"{label:28s} visits/ray {visits / n:7.2f} tris/ray {tests / n:5.2f} | alone: kernel {alone_ms:.4f} ms "
      f"({visits / alone_ms / 1e6:.1f} G visits/s, {n / alone_ms / 1e3:.0f} Mrays/s) | two in flight: {two_wall:.4f} ms per frame "
      f"({n / two_wall / 1e3:.0f} Mrays/s), kernel {two_ms:.4f} ms", flush=True)
