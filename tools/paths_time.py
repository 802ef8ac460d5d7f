#!/usr/bin/env python3
"""Photon-pass workload: rays leaving the two lights in n directions each (Fibonacci sphere), traced through the
specular path trees of the glass sphere + checker plane + 1 M-triangle terrain scene with rtg_trace_paths
(directions: Sphere::fibonacci_sphere of the host library, seeded).

    python tools/paths_time.py [directions_per_light] [max_depth]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ctypes as C

import numpy as np
import torch

from raytracer_amd import rtg
import bench


def main():
    per_light = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    max_depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    tris = rtg.heightfield(*bench.GRID, bench.SEED, bench.LOCAL_SCALE)
    ctx = rtg.Context(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    mesh = ctx.build_mesh(tris)
    objs, lights, _ = rtg.scene_builder(0)
    lights = np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE)
    scene = ctx.scene(list(objs) + [rtg.mesh_object(mesh, rtg.material((0.8, 0.7, 0.6)),
                                                     rtg.transform(1.0 / bench.LOCAL_SCALE))])
    dirs = rtg.fibonacci_sphere(per_light, seed=1)
    rays = np.concatenate([np.concatenate([np.broadcast_to(l["position"], dirs.shape), dirs], axis=1) for l in lights])
    n = rays.shape[0]
    d_rays = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
    cap = n * 2
    d_paths = torch.empty(cap * rtg.PATH_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    count = C.c_uint64()

    def run():
        ctx._check(rtg.lib.rtg_trace_paths(scene._h, C.c_void_p(d_rays.data_ptr()), n, max_depth,
                                           C.c_void_p(d_paths.data_ptr()), cap, C.byref(count)))
        ctx.synchronize()

    run()
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
    paths = np.frombuffer(d_paths[:count.value * rtg.PATH_DTYPE.itemsize].cpu().numpy().tobytes(), dtype=rtg.PATH_DTYPE)
    by_depth = np.bincount(paths["depth"], minlength=max_depth)
    ms = min(times) * 1e3
    print(f"{n} photon rays, max_depth {max_depth}: {count.value} paths (by depth {by_depth.tolist()}), "
          f"{(paths['depth'] > 0).sum()} caustic photons, {ms:.2f} ms = {n / ms / 1e3:.0f} M photon rays/s "
          f"({(by_depth.sum()) / ms / 1e3:.0f} M path segments/s)", flush=True)


if __name__ == "__main__":
    main()
