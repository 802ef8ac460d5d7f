"""Trispline patches (src/trispline.rs): the conversion chain and evaluation are pinned by the reference's formulas;
ray/patch intersection does not exist in the reference, so it is this build's definition (see rtg_patch_set in
include/rtg.h) -- "parity unpinned" against the reference, oracle-vs-GPU only."""
import os

import numpy as np
import pytest

import orc
import scenes
from raytracer_amd import rtg

# the patch of the reference's own test (trispline.rs:188-200): an octant of the unit sphere
OCTANT = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 0, -1], [-1, 0, 0], [0, -1, 0],
                   [1, 0, 0], [0, 1, 0], [0, 0, 1], [-0.85, 2.29, 2.29]], dtype=np.float64)


def test_chain_host_equals_oracle_bitwise():
    rng = np.random.default_rng(3)
    for nh in [OCTANT] + [rng.normal(size=(10, 3)) for _ in range(50)]:
        a, b = rtg.trispline_from_normal_hermite(nh), orc.trispline_from_normal_hermite(nh)
        assert a.tobytes() == b.tobytes()
        for t, u in [(0, 0), (1, 0), (0, 1), (0.25, 0.5), (0.3, 0.3)]:
            assert rtg.trispline_eval(a, t, u).tobytes() == orc.trispline_eval(b, t, u).tobytes()


def test_octant_patch_interpolates_corners_and_edges():
    sp = orc.trispline_from_normal_hermite(OCTANT)
    assert np.allclose(orc.trispline_eval(sp, 0, 0)[0], [1, 0, 0], atol=1e-15)      # p0
    assert np.allclose(orc.trispline_eval(sp, 1, 0)[0], [0, 1, 0], atol=1e-14)      # p1 at t = 1
    assert np.allclose(orc.trispline_eval(sp, 0, 1)[0], [0, 0, 1], atol=1e-14)      # p2 at u = 1
    # edge tangents at p0 are perpendicular to n0 and have the edge's length (normalize_speed)
    _, ddt, ddu, n = orc.trispline_eval(sp, 0, 0)
    assert abs(ddt[0]) < 1e-14 and abs(ddu[0]) < 1e-14
    assert np.isclose(np.linalg.norm(ddt), np.sqrt(2)) and np.isclose(np.linalg.norm(ddu), np.sqrt(2))
    assert np.allclose(n, [1, 0, 0], atol=1e-14)
    # the hypotenuse t + u = 1 is the third boundary curve: it stays in the plane x = 0
    for t in np.linspace(0, 1, 11):
        assert abs(orc.trispline_eval(sp, t, 1 - t)[0][0]) < 1e-13
    # and the edge curves bulge outwards like the sphere does (|p| > chord)
    assert np.linalg.norm(orc.trispline_eval(sp, 0.5, 0)[0]) > 0.9


def test_flat_input_gives_a_flat_patch():
    p = np.array([[0, 0, 0], [2, 0, 0.5], [0, 1, 0.25]], dtype=np.float64)
    n = np.cross(p[1] - p[0], p[2] - p[1])
    n /= np.linalg.norm(n)

    def mirror(a, e0, e1):            # a mirrored through the line e0-e1
        e = e1 - e0
        h = (a - e0) - e * np.dot(a - e0, e) / np.dot(e, e)
        return a - 2 * h
    nh = np.array([p[0], p[1], p[2], mirror(p[2], p[0], p[1]), mirror(p[0], p[1], p[2]), mirror(p[1], p[0], p[2]),
                   n, n, n, [0, 0, 0]])
    sp = orc.trispline_from_normal_hermite(nh)
    for t, u in [(0.2, 0.3), (0.5, 0.5), (0.1, 0.0), (0.0, 0.9)]:
        pos, ddt, ddu, nn = orc.trispline_eval(sp, t, u)
        assert np.allclose(pos, p[0] + t * (p[1] - p[0]) + u * (p[2] - p[0]), atol=1e-13)
        assert np.allclose(nn, n, atol=1e-12)


def test_tessellation_host_equals_oracle_and_counts():
    patches = rtg.patch_terrain(3, 2, seed=5)
    assert patches.shape == (12, 10, 3)
    for k in (1, 2, 5):
        ta, ia = rtg.patches_tessellate(patches, k)
        tb, ib = orc.patches_tessellate(patches, k)
        assert ta.shape == (12 * k * k, 3, 3) and ta.tobytes() == tb.tobytes() and ia.tobytes() == ib.tobytes()
        # every facet's corners lie on its patch and inside the parameter triangle
        assert (ia["tu"] >= 0).all() and (ia["tu"].sum(axis=2) <= 1 + 1e-15).all()
        for f in (0, len(ta) // 2, len(ta) - 1):
            for c in range(3):
                pos = orc.trispline_eval(patches[ia[f]["patch"]], *ia[f]["tu"][c])[0]
                assert pos.tobytes() == ta[f, c].tobytes()
    # neighbouring terrain patches share their corner positions
    tris = rtg.heightfield(3, 2, seed=5)
    assert np.allclose(orc.trispline_eval(patches[0], 0, 0)[0], tris[0, 0]) and np.allclose(
        orc.trispline_eval(patches[0], 1, 0)[0], tris[0, 1]) and np.allclose(orc.trispline_eval(patches[0], 0, 1)[0], tris[0, 2])


def _octant_mesh(k=6):
    sp = orc.trispline_from_normal_hermite(OCTANT)
    return rtg.HostPatchMesh(sp.reshape(1, 10, 3), k=k, newton_steps=5)


def test_oracle_patch_hit_lies_on_the_patch_and_on_the_ray():
    pm = _octant_mesh()
    objs, _, _ = scenes.patch_scene_objects(rtg.transform(), with_default=False)
    pair = scenes.Pair(None, objs, {0: pm})
    rng = np.random.default_rng(2)
    target = rng.dirichlet([2, 2, 2], size=400)                     # points near the octant
    target /= np.linalg.norm(target, axis=1, keepdims=True)
    orig = target * 3.0 + rng.normal(scale=0.05, size=target.shape)
    d = target * 0.9 - orig
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([orig, d], axis=1)
    hits, pts, _ = pair.cpu.trace(rays, points=True)
    ok = hits["leaf"] != rtg.RTG_NONE
    assert ok.mean() > 0.7
    sp = pm.patches[0]
    for i in np.flatnonzero(ok)[:100]:
        t, u = pts[i]["manifold_point"]
        assert t >= 0 and u >= 0 and t + u <= 1
        pos, ddt, ddu, n = orc.trispline_eval(sp, t, u)
        assert np.allclose(pos, pts[i]["position"], atol=1e-9)       # on the patch ...
        assert np.allclose(orig[i] + d[i] * hits[i]["t"], pos, atol=1e-9)   # ... and on the ray
        assert np.allclose(pts[i]["geo_normal"], n, atol=1e-12)
        assert 0.8 < np.linalg.norm(pos) < 1.3                       # roughly the sphere octant


@pytest.mark.gpu
@pytest.mark.parametrize("traversal", [0, 1, 2])
def test_gpu_patch_scene_matches_oracle(gpu_ctx, traversal):
    gpu_ctx.set_option("traversal", traversal)
    patches = rtg.patch_terrain(16, 12, seed=7, scale=100.0)
    pm = rtg.HostPatchMesh(patches, k=4, newton_steps=5)
    objs, lights, view = scenes.patch_scene_objects(rtg.transform(0.01), time=20)
    pair = scenes.Pair(gpu_ctx, objs, {2: pm})
    w, h = 240, 160
    g_hits, g_pts, g_ctr = pair.gpu.render_primary(view, w, h, points=True, counters=True)
    c_hits, c_pts, c_ctr = pair.cpu.render_primary(view, w, h, points=True, counters=True)
    assert scenes.hit_mismatches(g_hits, c_hits).size == 0
    for name in rtg.POINT_DTYPE.names:
        a, b = g_pts[name], c_pts[name]
        diff = scenes.float_bits_differ(a, b) if a.dtype.kind == "f" else (a != b)
        assert not diff.any(), name
    assert np.array_equal(g_ctr, c_ctr)
    assert (g_hits["object"] == 2).mean() > 0.3
    g_rgb, g_c8, _ = pair.gpu.render_frame(view, lights, w, h, max_depth=3)
    c_rgb, c_c8, _ = pair.cpu.render_frame(view, lights, w, h, max_depth=3)
    assert not scenes.float_bits_differ(g_rgb, c_rgb).any()
    assert np.abs(g_c8.astype(int) - c_c8.astype(int)).max() <= 1
    gpu_ctx.set_option("traversal", 2)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("RTG_FUZZ_SEEDS", "4"))))
def test_fuzz_patch_scenes(gpu_ctx, seed):
    """Random patch terrains (size, tessellation, Newton steps, placement), with or without the default sphere and
    floor around them, seen from random views: hits, points and visit counters of every traversal, and the shaded
    frame, against the oracle bit for bit."""
    rng = np.random.default_rng(7000 + seed)
    patches = rtg.patch_terrain(int(rng.integers(2, 20)), int(rng.integers(2, 14)), seed=int(rng.integers(1 << 20)), scale=100.0)
    pm = rtg.HostPatchMesh(patches, k=int(rng.choice([1, 2, 3, 4, 6])), newton_steps=int(rng.integers(1, 8)))
    with_default = bool(rng.random() < 0.5)
    xf = rtg.transform(float(rng.uniform(0.005, 0.02)), tuple(rng.normal(scale=0.1, size=3)))
    objs, lights, view = scenes.patch_scene_objects(xf, time=int(rng.integers(0, 100)), with_default=with_default)
    pair = scenes.Pair(gpu_ctx, objs, {(2 if with_default else 0): pm})
    view = rtg.make_view(orig=tuple(rng.normal(scale=0.2, size=3) + [0.0, 0.2, 1.0]), fov=(float(rng.uniform(0.8, 2.2)), float(rng.uniform(0.8, 2.2))))
    w, h = int(rng.integers(20, 200)), int(rng.integers(20, 140))
    c_hits, c_pts, c_ctr = pair.cpu.render_primary(view, w, h, points=True, counters=True)
    c_rgb, c_c8, _ = pair.cpu.render_frame(view, lights, w, h, max_depth=3)
    for traversal in (0, 1, 2):
        gpu_ctx.set_option("traversal", traversal)
        g_hits, g_pts, g_ctr = pair.gpu.render_primary(view, w, h, points=True, counters=True)
        assert scenes.hit_mismatches(g_hits, c_hits).size == 0, (seed, traversal)
        for name in rtg.POINT_DTYPE.names:
            a, b = g_pts[name], c_pts[name]
            diff = scenes.float_bits_differ(a, b) if a.dtype.kind == "f" else (a != b)
            assert not diff.any(), (seed, traversal, name)
        assert np.array_equal(g_ctr, c_ctr), (seed, traversal)
        g_rgb, g_c8, _ = pair.gpu.render_frame(view, lights, w, h, max_depth=3)
        assert not scenes.float_bits_differ(g_rgb, c_rgb).any(), (seed, traversal)
    gpu_ctx.set_option("traversal", 2)


@pytest.mark.gpu
def test_gpu_config3_patches_at_4k(gpu_ctx):
    """The trispline surface configuration, 3840x2160 primary rays (1M facets behind 15 500 patches)."""
    patches = rtg.patch_terrain(125, 62, seed=7, scale=100.0)
    pm = rtg.HostPatchMesh(patches, k=8, newton_steps=5)
    assert pm.bvh.n_leaves == 125 * 62 * 2 * 64
    objs, lights, view = scenes.patch_scene_objects(rtg.transform(0.01), with_default=False)
    pair = scenes.Pair(gpu_ctx, objs, {0: pm})
    w, h = 3840, 2160
    g_hits, _, _ = pair.gpu.render_primary(view, w, h)
    c_hits, _, _ = pair.cpu.render_primary(view, w, h)
    bad = scenes.hit_mismatches(g_hits, c_hits)
    assert bad.size == 0, f"{bad.size} mismatching pixels of {w * h}"
    assert (g_hits["leaf"] != rtg.RTG_NONE).mean() > 0.5


@pytest.mark.gpu
def test_patch_mesh_argument_checks(gpu_ctx):
    pm = _octant_mesh(k=2)
    mesh = gpu_ctx.mesh(pm)
    with pytest.raises(rtg.RtgError) as e:
        gpu_ctx.scene([rtg.mesh_object(mesh, rtg.material())])      # kind MESH with a patch handle
    assert "does not match" in str(e.value)
    bad = rtg.HostPatchMesh(pm.patches, k=2)
    bad.leaf_info["patch"][0] = 7
    with pytest.raises(rtg.RtgError):
        gpu_ctx.mesh(bad)
