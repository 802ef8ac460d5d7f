#!/usr/bin/env python3
"""Time of the tree build: BvhForest::subdivide + Bvh::new on the host threads against rtg_mesh_build on the GPU,
on the terrain of the bench at a chosen grid; checks that both trees are byte-identical.

    python tools/build_time.py [nx ny] [nocheck]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from raytracer_amd import rtg
import bench


def main():
    args = [a for a in sys.argv[1:] if a != "nocheck"]
    check = "nocheck" not in sys.argv[1:]
    grid = (int(args[0]), int(args[1])) if len(args) >= 2 else bench.GRID
    tris = rtg.heightfield(*grid, bench.SEED, bench.LOCAL_SCALE)
    n = tris.shape[0]
    ctx = rtg.Context(0)
    ctx.build_mesh(tris[:4096]).close()                      # first-use costs (module load, allocator warm-up)
    d_tris = torch.from_numpy(tris).cuda()
    torch.cuda.synchronize()

    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        mesh = ctx.build_mesh((d_tris.data_ptr(), n))
        ctx.synchronize()
        times.append(time.perf_counter() - t0)
        mesh.close()
    t0 = time.perf_counter()
    mesh = ctx.build_mesh(tris)
    ctx.synchronize()
    from_host = time.perf_counter() - t0
    depth = mesh.sizes()[3]
    print(f"{n} triangles, depth {depth}: device build {min(times) * 1e3:.1f} ms (triangles resident), "
          f"{from_host * 1e3:.1f} ms from host memory", flush=True)
    if check:
        t0 = time.perf_counter()
        host = rtg.HostBvh(tris)
        host_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        ref = ctx.mesh(host)
        ctx.synchronize()
        upload_s = time.perf_counter() - t0
        nodes, leaves, source = mesh.read_bvh()
        same = (nodes.tobytes() == host.nodes().tobytes() and leaves.tobytes() == host.leaves().tobytes()
                and np.array_equal(source, host.leaf_source()))
        print(f"host build {host_s:.2f} s on {min(16, os.cpu_count())} threads + {upload_s:.2f} s layout and upload; "
              f"trees identical: {same}", flush=True)
        assert same
        ref.close()


if __name__ == "__main__":
    main()
