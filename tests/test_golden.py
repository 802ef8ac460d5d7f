"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle): the oracle must keep
reproducing them (CPU), and the GPU path must match them bit for bit (GPU)."""
import os

import numpy as np
import pytest

import scenes
from raytracer_amd import rtg

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    return np.load(os.path.join(HERE, "golden", name), allow_pickle=False)


def _points_equal(a, b):
    for name in rtg.POINT_DTYPE.names:
        x, y = a[name], b[name]
        diff = scenes.float_bits_differ(x, y) if x.dtype.kind == "f" else (x != y)
        if diff.any():
            return False
    return True


def _default(ctx):
    g = _load("default_scene_64.npz")
    objs, lights, view = scenes.default_objects(int(g["time"]))
    return g, scenes.Pair(ctx, objs, {}), lights, view


def _terrain(ctx):
    g = _load("terrain_20x20_scene.npz")
    tris, xform = scenes.terrain(int(g["grid"]), int(g["grid"]), seed=int(g["seed"]))
    bvh = rtg.HostBvh(tris, threads=1)
    objs, lights, view = scenes.mesh_scene_objects(xform, time=int(g["time"]))
    mobjs, _, _ = scenes.mesh_scene_objects(rtg.transform(), with_default=False)
    return g, bvh, scenes.Pair(ctx, objs, {2: bvh}), scenes.Pair(ctx, mobjs, {0: bvh}), view


def test_oracle_reproduces_default_scene_golden():
    g, pair, lights, view = _default(None)
    w, h = int(g["width"]), int(g["height"])
    hits, pts, _ = pair.cpu.render_primary(view, w, h, points=True)
    assert scenes.same_hits(hits, g["hits"]) and _points_equal(pts, g["points"])
    rgb, c8, _ = pair.cpu.render_frame(view, lights, w, h, max_depth=4)
    assert not scenes.float_bits_differ(rgb, g["rgb_depth4"]).any() and np.array_equal(c8, g["rgba8_depth4"])
    # the picture is the expected one: sphere disc in the middle over the checker plane, sky above
    obj = g["hits"]["object"].reshape(h, w)
    assert obj[40, 32] == 0 and obj[60, 5] == 1 and obj[2, 32] == rtg.RTG_NONE


def test_oracle_and_host_builder_reproduce_terrain_golden():
    g, bvh, pair, bare, view = _terrain(None)
    assert bvh.n_nodes == int(g["n_nodes"]) and bvh.root == int(g["root"])
    assert np.frombuffer(bvh.nodes().tobytes(), dtype=np.uint8).astype(np.uint64).sum() == int(g["nodes_crc"])
    w, h = int(g["width"]), int(g["height"])
    hits, pts, ctr = pair.cpu.render_primary(view, w, h, points=True, counters=True)
    assert scenes.same_hits(hits, g["hits"]) and _points_equal(pts, g["points"]) and np.array_equal(ctr, g["counters"])
    rays = rtg.bench_mesh_rays(bvh.bounds(), 2000, seed=int(g["ray_seed"]))
    rhits, _, rctr = bare.cpu.trace(rays, counters=True)
    assert scenes.same_hits(rhits, g["ray_hits"]) and np.array_equal(rctr, g["ray_counters"])


def _path_rays():
    rays = rtg.bench_mesh_rays([-0.3, -0.3, 0.8, 0.3, 0.3, 1.0], 600, seed=77)
    rays[:, 3:] = np.array([0.0, -0.2, -1.0]) + 0.3 * rays[:, 3:]
    rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    return rays


def _paths_equal(a, b):
    if a.shape != b.shape:
        return False
    for f in ("ray", "depth", "modes", "parent"):
        if not np.array_equal(a[f], b[f]):
            return False
    return not scenes.float_bits_differ(a["attenuation"], b["attenuation"]).any() and _points_equal(a["point"], b["point"])


def test_oracle_and_host_builders_reproduce_paths_and_builder_golden():
    import orc
    g = _load("paths_and_builders.npz")
    objs, lights, view = scenes.default_objects(int(g["time"]))
    pair = scenes.Pair(None, objs, {})
    paths = pair.cpu.trace_paths(_path_rays(), max_depth=4)
    assert _paths_equal(paths, g["paths"])
    assert (g["paths"]["depth"] == 3).sum() > 50 and (g["paths"]["modes"] != 0).sum() > 100
    tris, _ = scenes.terrain(int(g["grid"][0]), int(g["grid"][1]), seed=int(g["seed"]))
    for strategy in ("subdivide", "contract", "hybrid"):
        want = g["leaf_source_" + strategy]
        assert np.array_equal(orc.OracleBvh(tris, strategy=strategy).leaf_source(), want)
        assert np.array_equal(rtg.HostBvh(tris, strategy=strategy).leaf_source(), want)
    assert not np.array_equal(g["leaf_source_contract"], g["leaf_source_subdivide"])


@pytest.mark.gpu
def test_gpu_paths_match_golden(gpu_ctx):
    g = _load("paths_and_builders.npz")
    objs, lights, view = scenes.default_objects(int(g["time"]))
    pair = scenes.Pair(gpu_ctx, objs, {})
    assert _paths_equal(pair.gpu.trace_paths(_path_rays(), max_depth=4), g["paths"])


@pytest.mark.gpu
@pytest.mark.parametrize("traversal", [0, 1, 2])
def test_gpu_matches_golden(gpu_ctx, traversal):
    gpu_ctx.set_option("traversal", traversal)
    g, pair, lights, view = _default(gpu_ctx)
    w, h = int(g["width"]), int(g["height"])
    hits, pts, _ = pair.gpu.render_primary(view, w, h, points=True)
    assert scenes.same_hits(hits, g["hits"]) and _points_equal(pts, g["points"])
    rgb, c8, _ = pair.gpu.render_frame(view, lights, w, h, max_depth=4)
    assert not scenes.float_bits_differ(rgb, g["rgb_depth4"]).any()
    assert np.abs(c8.astype(int) - g["rgba8_depth4"].astype(int)).max() <= 1      # 8-bit PQ colour: <= 1 LSB
    g, bvh, pair, bare, view = _terrain(gpu_ctx)
    w, h = int(g["width"]), int(g["height"])
    hits, pts, ctr = pair.gpu.render_primary(view, w, h, points=True, counters=True)
    assert scenes.same_hits(hits, g["hits"]) and _points_equal(pts, g["points"]) and np.array_equal(ctr, g["counters"])
    rays = rtg.bench_mesh_rays(bvh.bounds(), 2000, seed=int(g["ray_seed"]))
    rhits, _, rctr = bare.gpu.trace(rays, counters=True)
    assert scenes.same_hits(rhits, g["ray_hits"]) and np.array_equal(rctr, g["ray_counters"])
    gpu_ctx.set_option("traversal", 2)
