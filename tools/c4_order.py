#!/usr/bin/env python3
"""Configuration 4 traced in one process as given and through the full device sort (rays and answers moved), with the
traversal kernel's time, the whole call's time and a check that the hit records are the same bytes.  (A third way,
coarse bins traced through a sorted index, was measured with profiles/r3_1_kernel/coarse_bins_experiment.patch.)
Usage: tools/c4_order.py [launches per way (default 3)]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from raytracer_amd import rtg

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = rtg.Context(0)
tris = rtg.heightfield(bench.C4_GRID[0], bench.C4_GRID[1], bench.SEED, 1.0)
mesh = ctx.build_mesh(tris)
scene = ctx.scene([rtg.mesh_object(mesh, rtg.material((1, 1, 1)), rtg.transform())])
rays = rtg.bench_mesh_rays(mesh.bounds(), bench.C4_RAYS, seed=bench.C4_RAY_SEED)
d_rays = torch.from_numpy(rays).cuda()
del rays
hits = torch.empty(bench.C4_RAYS * 16, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ctx.set_option("time_kernels", 2)
digests = {}
for name, mode in (("as given", 1), ("full sort", 2), ("as given", 1)):      # ("coarse bins", 3) with the experiment patch
    ctx.set_option("sort_rays", mode)
    walls, kernels = [], []
    for i in range(launches + 1):
        hits.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        scene.trace_raw(d_rays.data_ptr(), bench.C4_RAYS, hits.data_ptr())
        ctx.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        ms, n = ctx.kernel_time()
        if i:
            walls.append(wall)
            kernels.append(ms)
    d = hashlib.sha1(hits.cpu().numpy().tobytes()).hexdigest()[:12]
    digests[name] = d
    print(f"{name:12s} traversal kernel {min(kernels):7.3f} ms, whole call {min(walls):7.3f} ms "
          f"({bench.C4_RAYS / min(walls) / 1e3:.0f} Mrays/s), hits {d}", flush=True)
assert len(set(digests.values())) == 1, digests
