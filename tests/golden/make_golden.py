#!/usr/bin/env python3
"""Generates the golden vectors in this directory from the CPU oracle (oracle/liboracle.so).

The reference renderer cannot be executed in the build environment (Rust toolchain absent, src/mesh/ missing), so
these vectors pin the oracle against regressions and give the GPU tests a fixed target; they are NOT outputs of the
reference itself.  Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import orc  # noqa: E402
import scenes  # noqa: E402
from raytracer_amd import rtg  # noqa: E402


def default_scene():
    objs, lights, view = scenes.default_objects(7)
    pair = scenes.Pair(None, objs, {})
    w, h = 64, 64
    hits, pts, _ = pair.cpu.render_primary(view, w, h, points=True, threads=1)
    rgb, c8, _ = pair.cpu.render_frame(view, lights, w, h, max_depth=4, threads=1)
    np.savez_compressed(os.path.join(HERE, "default_scene_64.npz"), width=w, height=h, time=7, hits=hits, points=pts,
                        rgb_depth4=rgb, rgba8_depth4=c8)


def mesh_scene():
    tris, xform = scenes.terrain(20, 20, seed=7)
    bvh = rtg.HostBvh(tris, threads=1)
    objs, lights, view = scenes.mesh_scene_objects(xform, time=3)
    pair = scenes.Pair(None, objs, {2: bvh})
    w, h = 64, 48
    hits, pts, ctr = pair.cpu.render_primary(view, w, h, points=True, counters=True, threads=1)
    rays = rtg.bench_mesh_rays(bvh.bounds(), 2000, seed=10212233)
    mobjs, _, _ = scenes.mesh_scene_objects(rtg.transform(), with_default=False)
    bare = scenes.Pair(None, mobjs, {0: bvh})
    rhits, _, rctr = bare.cpu.trace(rays, counters=True, threads=1)
    np.savez_compressed(os.path.join(HERE, "terrain_20x20_scene.npz"), width=w, height=h, time=3, grid=20, seed=7,
                        hits=hits, points=pts, counters=ctr, n_nodes=bvh.n_nodes, root=bvh.root,
                        nodes_crc=np.frombuffer(bvh.nodes().tobytes(), dtype=np.uint8).astype(np.uint64).sum(),
                        ray_seed=10212233, ray_hits=rhits, ray_counters=rctr)


def paths_and_builders():
    """Specular path records of seeded rays into the default scene (glass sphere over the checker plane), and the
    leaf orders of the three builders on a small terrain."""
    objs, lights, view = scenes.default_objects(7)
    pair = scenes.Pair(None, objs, {})
    rays = path_rays()
    paths = pair.cpu.trace_paths(rays, max_depth=4)
    tris, _ = scenes.terrain(9, 7, seed=11)
    orders = {s: orc.OracleBvh(tris, strategy=s).leaf_source() for s in ("subdivide", "contract", "hybrid")}
    np.savez_compressed(os.path.join(HERE, "paths_and_builders.npz"), time=7, paths=paths, grid=(9, 7), seed=11,
                        **{"leaf_source_" + k: v for k, v in orders.items()})


def path_rays():
    rays = rtg.bench_mesh_rays([-0.3, -0.3, 0.8, 0.3, 0.3, 1.0], 600, seed=77)
    rays[:, 3:] = np.array([0.0, -0.2, -1.0]) + 0.3 * rays[:, 3:]
    rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    return rays


if __name__ == "__main__":
    if "new" in sys.argv[1:]:                 # only the vectors added last; the older files stay byte-identical
        paths_and_builders()
        sys.exit(0)
    default_scene()
    mesh_scene()
    paths_and_builders()
    print("golden vectors written to", HERE)
