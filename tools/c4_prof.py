#!/usr/bin/env python3
"""Configuration 4 as a short program for rocprofv3: builds the 10 001 406-triangle tree on the GPU, uploads the
67 108 864 bench_mesh rays and traces them `launches` times (default 2, after one untimed launch).
Usage: tools/c4_prof.py [launches] [key=value ctx options...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from raytracer_amd import rtg

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx = rtg.Context(0)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
tris = rtg.heightfield(bench.C4_GRID[0], bench.C4_GRID[1], bench.SEED, 1.0)
mesh = ctx.build_mesh(tris)
scene = ctx.scene([rtg.mesh_object(mesh, rtg.material((1, 1, 1)), rtg.transform())])
rays = rtg.bench_mesh_rays(mesh.bounds(), bench.C4_RAYS, seed=bench.C4_RAY_SEED)
d_rays = torch.from_numpy(rays).cuda()
hits = torch.empty(bench.C4_RAYS * 16, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ctx.set_option("time_kernels", 2)
import time
for i in range(launches + 1):
    t0 = time.perf_counter()
    scene.trace_raw(d_rays.data_ptr(), bench.C4_RAYS, hits.data_ptr())
    ctx.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    ms, n = ctx.kernel_time()
    if i:
        print(f"launch {i}: traversal kernel {ms:.3f} ms ({bench.C4_RAYS / ms / 1e3:.1f} Mrays/s), whole call {wall:.3f} ms "
              f"({bench.C4_RAYS / wall / 1e3:.1f} Mrays/s)", flush=True)
