#!/usr/bin/env python3
"""Speed prototype of node layouts (library built with -DRTG_PROTO_LINK, selected by RTG_LIBRTG_PATH): the bench frame
(config 2: alone and two in flight) and the config-4 batch for the layout RTG_PROTO_LAYOUT names, with a checksum of
the hit records so that runs of different libraries and layouts can be compared.
Usage: tools/proto_layout.py c2|c4 [label]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from raytracer_amd import rtg

what = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else "tree"
label += f" layout={os.environ.get('RTG_PROTO_LAYOUT', '-')} top={os.environ.get('RTG_PROTO_TOP', '-')}"


def digest(t):
    return hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:12]


if what == "c2":
    tris = rtg.heightfield(*bench.GRID, bench.SEED, bench.LOCAL_SCALE)
    bvh = rtg.HostBvh(tris)
    W, H = bench.WIDTH, bench.HEIGHT
    n = W * H
    ctxs, scenes, bufs = [], [], []
    mesh = None
    for i in range(2):
        c = rtg.Context(0)
        st = torch.cuda.Stream()
        c.set_stream(st.cuda_stream)
        if mesh is None:
            mesh = c.mesh(bvh)
        ctxs.append((c, st))
        scenes.append(c.scene([rtg.mesh_object(mesh, rtg.material((1, 1, 1)), rtg.transform(1.0 / bench.LOCAL_SCALE))]))
        bufs.append(torch.empty(n * 16, dtype=torch.uint8, device="cuda"))
    tile = (0, 0, W, H)
    ctr = torch.empty(n * 2, dtype=torch.int32, device="cuda")
    scenes[0].render_primary_raw(bench.frame_view(rtg, 0), W, H, tile, bufs[0].data_ptr(), None, ctr.data_ptr())
    torch.cuda.synchronize()
    sums = ctr.view(n, 2).to(torch.int64).sum(dim=0).tolist()
    scenes[0].render_primary_raw(bench.frame_view(rtg, 0), W, H, tile, bufs[0].data_ptr())
    torch.cuda.synchronize()
    d = digest(bufs[0])

    def run(inflight, frames=40):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(frames):
            k = f % inflight
            scenes[k].render_primary_raw(bench.frame_view(rtg, f), W, H, tile, bufs[k].data_ptr())
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / frames * 1e3

    run(1, 6)
    alone = min(run(1) for _ in range(3))
    two = min(run(2) for _ in range(3))
    print(f"{label:40s} c2 hits {d} visits {sums[0]} tris {sums[1]} | alone {alone:.4f} ms ({n / alone / 1e3:.0f} Mrays/s) | "
          f"two in flight {two:.4f} ms per frame ({n / two / 1e3:.0f} Mrays/s)", flush=True)
else:
    ctx = rtg.Context(0)
    tris = rtg.heightfield(bench.C4_GRID[0], bench.C4_GRID[1], bench.SEED, 1.0)
    t0 = time.perf_counter()
    bvh = rtg.HostBvh(tris)
    mesh = ctx.mesh(bvh)
    build = time.perf_counter() - t0
    scene = ctx.scene([rtg.mesh_object(mesh, rtg.material((1, 1, 1)), rtg.transform())])
    rays = rtg.bench_mesh_rays(mesh.bounds(), bench.C4_RAYS, seed=bench.C4_RAY_SEED)
    d_rays = torch.from_numpy(rays).cuda()
    del rays
    hits = torch.empty(bench.C4_RAYS * 16, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ms = []
    for i in range(4):
        t0 = time.perf_counter()
        scene.trace_raw(d_rays.data_ptr(), bench.C4_RAYS, hits.data_ptr())
        ctx.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    print(f"{label:40s} c4 hits {digest(hits)} | host tree + upload {build:.1f} s | calls {' '.join(f'{m:.2f}' for m in ms[1:])} ms "
          f"({bench.C4_RAYS / min(ms[1:]) / 1e3:.0f} Mrays/s)", flush=True)
