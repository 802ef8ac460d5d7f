"""GPU: the HIP path through the C ABI against the CPU oracle on the same inputs.

Bar: hit object / leaf ids and hit times bit-exact; every other field of the RaycastPoint bit-exact as well
(the kernels use the reference's operation order in IEEE binary64 with contraction off); 8-bit PQ colour within
1 LSB (device pow differs from the host libm in the last bits).
"""
import os

import numpy as np
import pytest

import orc
import scenes
from raytracer_amd import rtg

pytestmark = pytest.mark.gpu

TRAVERSALS = [0, 1, 2]   # short stack, threaded, phased threaded (persistent, reciprocal box test)


def _assert_hits_equal(g, c):
    bad = scenes.hit_mismatches(g, c)
    assert bad.size == 0, f"{bad.size} mismatching rays, first {bad[:5]}: gpu {g[bad[:5]]} cpu {c[bad[:5]]}"


def _assert_points_equal(g, c):
    for name in rtg.POINT_DTYPE.names:
        a, b = g[name], c[name]
        diff = scenes.float_bits_differ(a, b) if a.dtype.kind == "f" else (a != b)
        bad = np.flatnonzero(diff.reshape(a.shape[0], -1).any(axis=1))
        assert bad.size == 0, f"field {name}: {bad.size} mismatches, first at {bad[:5]}: gpu {g[name][bad[:3]]} cpu {c[name][bad[:3]]}"


@pytest.mark.parametrize("size", [(64, 64), (300, 300), (1024, 1024), (37, 19)])
def test_default_scene_primary_hits(gpu_ctx, size):
    """Config 1: sphere + checker plane (src/lib.rs:68-108), primary rays."""
    objs, lights, view = scenes.default_objects(0)
    pair = scenes.Pair(gpu_ctx, objs, {})
    w, h = size
    g_hits, g_pts, _ = pair.gpu.render_primary(view, w, h, points=True)
    c_hits, c_pts, _ = pair.cpu.render_primary(view, w, h, points=True)
    _assert_hits_equal(g_hits, c_hits)
    _assert_points_equal(g_pts, c_pts)
    objs_hit = set(np.unique(g_hits["object"]).tolist())
    assert objs_hit == {0, 1, rtg.RTG_NONE}
    assert set(np.unique(g_pts["material_slot"][g_hits["object"] == 1]).tolist()) == {2, 3}   # both checker colours


def test_tiles_reassemble_to_the_full_frame(gpu_ctx):
    objs, lights, view = scenes.default_objects(0)
    pair = scenes.Pair(gpu_ctx, objs, {})
    w, h = 200, 120
    full, _, _ = pair.gpu.render_primary(view, w, h)
    full = full.reshape(h, w)
    for tile in [(0, 0, 100, 60), (100, 0, 100, 60), (0, 60, 200, 60), (33, 17, 71, 29)]:
        part, _, _ = pair.gpu.render_primary(view, w, h, tile=tile)
        x0, y0, tw, th = tile
        assert scenes.same_hits(part.reshape(th, tw), full[y0:y0 + th, x0:x0 + tw])


@pytest.mark.parametrize("traversal", TRAVERSALS)
@pytest.mark.parametrize("with_normals", [False, True])
def test_mesh_scene_primary(gpu_ctx, traversal, with_normals):
    """Sphere + plane + an instanced (scaled) terrain mesh of 20 000 triangles."""
    gpu_ctx.set_option("traversal", traversal)
    tris, xform = scenes.terrain(100, 100, seed=7)
    vn = scenes.vertex_normals_for(tris) if with_normals else None
    bvh = rtg.HostBvh(tris, vertex_normals=vn)
    objs, lights, view = scenes.mesh_scene_objects(xform)
    pair = scenes.Pair(gpu_ctx, objs, {2: bvh})
    w, h = 320, 200
    g_hits, g_pts, g_ctr = pair.gpu.render_primary(view, w, h, points=True, counters=True)
    c_hits, c_pts, c_ctr = pair.cpu.render_primary(view, w, h, points=True, counters=True)
    _assert_hits_equal(g_hits, c_hits)
    _assert_points_equal(g_pts, c_pts)
    assert np.array_equal(g_ctr["nodes_visited"], c_ctr["nodes_visited"])
    assert np.array_equal(g_ctr["tris_tested"], c_ctr["tris_tested"])
    assert (g_hits["object"] == 2).mean() > 0.3
    gpu_ctx.set_option("traversal", 2)


def test_single_mesh_through_the_several_meshes_build(gpu_ctx):
    """A mesh of 2^27 nodes or more is sent to the kernel build that carries 64-bit tree addresses per lane (the one
    scenes with several meshes use); option keep_state forces that dispatch on an ordinary mesh."""
    tris, xform = scenes.terrain(100, 100, seed=5)
    bvh = rtg.HostBvh(tris)
    objs, lights, view = scenes.mesh_scene_objects(xform)
    pair = scenes.Pair(gpu_ctx, objs, {2: bvh})
    bare_objs, _, _ = scenes.mesh_scene_objects(xform, with_default=False)
    bare = scenes.Pair(gpu_ctx, bare_objs, {0: bvh})
    w, h = 256, 160
    rays = rtg.bench_mesh_rays(bvh.bounds(), 30000, seed=3)
    rays[:, :3] /= 100.0                         # into the instance's world space
    want_p, _, want_pc = pair.cpu.render_primary(view, w, h, counters=True)
    want_b, _, want_bc = bare.cpu.trace(rays, counters=True)
    lights = np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE)
    want_rgb, _, _ = pair.cpu.render_frame(view, lights, w, h)
    for keep in (1, 2):
        gpu_ctx.set_option("keep_state", keep)
        got_p, _, got_pc = pair.gpu.render_primary(view, w, h, counters=True)
        _assert_hits_equal(got_p, want_p)
        assert np.array_equal(got_pc, want_pc)
        got_p, _, _ = pair.gpu.render_primary(view, w, h)              # the build without counters
        _assert_hits_equal(got_p, want_p)
        got_b, _, got_bc = bare.gpu.trace(rays, counters=True)
        _assert_hits_equal(got_b, want_b)
        assert np.array_equal(got_bc, want_bc)
        got_b, _, _ = bare.gpu.trace(rays)
        _assert_hits_equal(got_b, want_b)
        got_rgb, _, _ = pair.gpu.render_frame(view, lights, w, h)      # shadow rays
        assert not scenes.float_bits_differ(got_rgb, want_rgb).any()
    gpu_ctx.set_option("keep_state", 1)


@pytest.mark.parametrize("traversal", TRAVERSALS)
def test_incoherent_batch_rays(gpu_ctx, traversal):
    """bench_mesh ray distribution (src/tree/benchmark.rs:15-27) against a bare mesh object."""
    gpu_ctx.set_option("traversal", traversal)
    tris, xform = scenes.terrain(64, 64, seed=11, local_scale=1.0)
    bvh = rtg.HostBvh(tris)
    objs, _, _ = scenes.mesh_scene_objects(rtg.transform(), with_default=False)
    pair = scenes.Pair(gpu_ctx, objs, {0: bvh})
    rays = rtg.bench_mesh_rays(bvh.bounds(), 50000, seed=10212233)
    g_hits, g_pts, g_ctr = pair.gpu.trace(rays, points=True, counters=True)
    c_hits, c_pts, c_ctr = pair.cpu.trace(rays, points=True, counters=True)
    _assert_hits_equal(g_hits, c_hits)
    _assert_points_equal(g_pts, c_pts)
    assert np.array_equal(g_ctr, c_ctr)
    assert 0.05 < (g_hits["leaf"] != rtg.RTG_NONE).mean() < 0.95
    # sorting the batch only permutes the answers
    perm, sorted_rays = rtg.sort_rays(bvh.bounds(), rays)
    s_hits, _, _ = pair.gpu.trace(sorted_rays)
    assert scenes.same_hits(s_hits, g_hits[perm])
    # the device-side sort (key, radix sort, gather, trace, scatter back) returns the caller's order unchanged
    gpu_ctx.set_option("sort_rays", 2)
    d_hits, d_pts, d_ctr = pair.gpu.trace(rays, points=True, counters=True)
    gpu_ctx.set_option("sort_rays", 1)
    _assert_hits_equal(d_hits, c_hits)
    _assert_points_equal(d_pts, c_pts)
    assert np.array_equal(d_ctr, c_ctr)
    gpu_ctx.set_option("traversal", 2)


@pytest.mark.parametrize("traversal", TRAVERSALS)
def test_quirks_nan_inf_ties_and_degenerates(gpu_ctx, traversal):
    """Axis-parallel rays, zero-thickness boxes, degenerate (NaN-time) and coincident triangles."""
    gpu_ctx.set_option("traversal", traversal)
    rng = np.random.default_rng(5)
    base = rng.uniform(-1, 1, size=(200, 1, 3)) + rng.uniform(-0.3, 0.3, size=(200, 3, 3))
    flat = base[:20].copy()
    flat[:, :, 2] = flat[:, :1, 2]                      # axis-aligned flat: invisible through the BVH
    degenerate = base[20:30].copy()
    degenerate[:, 2] = degenerate[:, 1]                 # zero-area: NaN hit time
    dup = base[30:40].copy()                            # coincident copies: lowest leaf index wins
    tris = np.concatenate([base, flat, degenerate, dup])
    bvh = rtg.HostBvh(tris)
    objs, _, _ = scenes.mesh_scene_objects(rtg.transform(), with_default=False)
    pair = scenes.Pair(gpu_ctx, objs, {0: bvh})
    n = 20000
    o = rng.uniform(-1.5, 1.5, size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    axis = rng.integers(0, 3, size=n)
    kind = rng.integers(0, 4, size=n)
    for k in range(3):                                  # a quarter of the rays are exactly axis-parallel
        sel = (kind == 0) & (axis == k)
        d[sel] = 0.0
        d[sel, k] = rng.choice([-1.0, 1.0], size=sel.sum())
    sel = kind == 1                                     # another quarter has one zero direction component
    d[sel, axis[sel]] = 0.0
    rays = np.concatenate([o, d], axis=1)
    g_hits, g_pts, _ = pair.gpu.trace(rays, points=True)
    c_hits, c_pts, _ = pair.cpu.trace(rays, points=True)
    _assert_hits_equal(g_hits, c_hits)
    _assert_points_equal(g_pts, c_pts)
    assert np.isnan(g_hits["t"]).any(), "the degenerate triangles should produce NaN-time hits"
    gpu_ctx.set_option("traversal", 2)


def test_deep_tree_uses_every_stack_tier(gpu_ctx):
    """A skewed mesh whose tree is deeper than the 4 register + 28 LDS levels, so the global overflow slab is used."""
    rng = np.random.default_rng(8)
    n = 3000
    c = np.zeros((n, 1, 3))
    c[:, 0, 0] = np.exp(rng.uniform(0, 40, size=n))     # 1 % clamp on every split -> depth in the hundreds
    c[:, 0, 1:] = rng.uniform(-1, 1, size=(n, 2))
    tris = c + rng.uniform(-0.5, 0.5, size=(n, 3, 3))
    bvh = rtg.HostBvh(tris)
    assert bvh.depth > 40
    objs, _, _ = scenes.mesh_scene_objects(rtg.transform(), with_default=False)
    pair = scenes.Pair(gpu_ctx, objs, {0: bvh})
    m = 4096
    o = np.stack([np.exp(rng.uniform(0, 40, size=m)), rng.uniform(-1, 1, size=m), np.full(m, 5.0)], axis=1)
    d = np.tile([0.0, 0.0, -1.0], (m, 1)) + rng.normal(scale=0.05, size=(m, 3))
    rays = np.concatenate([o, d], axis=1)
    c_hits, _, c_ctr = pair.cpu.trace(rays, counters=True)
    for traversal in TRAVERSALS:
        gpu_ctx.set_option("traversal", traversal)
        g_hits, _, g_ctr = pair.gpu.trace(rays, counters=True)
        _assert_hits_equal(g_hits, c_hits)
        assert np.array_equal(g_ctr, c_ctr)
    gpu_ctx.set_option("traversal", 2)


@pytest.mark.parametrize("exponent", [-45, -39, 0, 20, 39, 45])
@pytest.mark.parametrize("pretest", [2, 1])
def test_single_precision_pretest_borderline_cases(gpu_ctx, exponent, pretest):
    """The f32 box pre-test of the default kernel must never change an answer: lattice geometry whose plane times
    tie exactly (ray origins on box planes, dyadic directions, coincident hit and box times), geometry offset so far
    that f32 cannot resolve the boxes, and scales on either side of the window in which the pre-test is allowed."""
    gpu_ctx.set_option("traversal", 2)
    gpu_ctx.set_option("pretest", pretest)
    rng = np.random.default_rng(100 + exponent)
    scale = 2.0 ** exponent
    # lattice triangles: integer corners, slightly tilted so they are visible through their boxes
    cells = rng.integers(-8, 8, size=(400, 1, 3)).astype(np.float64)
    corners = np.array([[0, 0, 0], [1, 0, 1], [0, 1, 1]], dtype=np.float64)
    lattice = cells + corners[rng.permuted(np.tile(np.arange(3), (400, 1)), axis=1)]
    # far-offset cluster: boxes 1e-3 apart at coordinate 4096 (f32 spacing there is 4.9e-4)
    far = 4096.0 + rng.uniform(0, 0.05, size=(200, 1, 3)) + rng.uniform(-2e-3, 2e-3, size=(200, 3, 3))
    tris = np.concatenate([lattice, far]) * scale
    bvh = rtg.HostBvh(tris)
    objs, _, _ = scenes.mesh_scene_objects(rtg.transform(), with_default=False)
    pair = scenes.Pair(gpu_ctx, objs, {0: bvh})
    n = 30000
    o = rng.integers(-10, 10, size=(n, 3)).astype(np.float64)             # on lattice planes
    o[n // 2:] += rng.choice([0.0, 0.5, 0.25], size=(n - n // 2, 3))
    d = rng.choice([-1.0, -0.5, 0.0, 0.25, 0.5, 1.0, 2.0], size=(n, 3))
    d[(d == 0).all(axis=1)] = [1.0, 0.5, -1.0]
    o2 = 4096.0 + rng.uniform(-0.1, 0.15, size=(n // 3, 3))
    d2 = far.mean(axis=1)[rng.integers(0, 200, size=n // 3)] + rng.normal(scale=5e-4, size=(n // 3, 3)) - o2
    o3 = rng.uniform(-10, 10, size=(n // 3, 3))                            # aimed at the far cluster
    d3 = (4096.0 + rng.uniform(0, 0.05, size=(n // 3, 3))) - o3
    rays = np.concatenate([np.concatenate([o, d], axis=1), np.concatenate([o2, d2], axis=1),
                           np.concatenate([o3, d3], axis=1)])
    rays[:, :3] *= scale
    rays[: n // 2, 3:] *= scale                                             # direction length is free: t scales
    g_hits, g_pts, g_ctr = pair.gpu.trace(rays, points=True, counters=True)
    c_hits, c_pts, c_ctr = pair.cpu.trace(rays, points=True, counters=True)
    gpu_ctx.set_option("pretest", 2)
    _assert_hits_equal(g_hits, c_hits)
    _assert_points_equal(g_pts, c_pts)
    assert np.array_equal(g_ctr, c_ctr)
    assert (g_hits["leaf"] != rtg.RTG_NONE).mean() > 0.05


def test_empty_inputs(gpu_ctx):
    objs, lights, view = scenes.default_objects(0)
    pair = scenes.Pair(gpu_ctx, objs, {})
    hits, _, _ = pair.gpu.trace(np.zeros((0, 6)))
    assert hits.shape == (0,)
    hits, _, _ = pair.gpu.render_primary(view, 16, 16, tile=(3, 3, 0, 5))
    assert hits.shape == (0,)
    empty = scenes.Pair(gpu_ctx, [], {})                 # a scene with no objects: every ray misses
    hits, pts, _ = empty.gpu.render_primary(view, 8, 8, points=True)
    assert (hits["object"] == rtg.RTG_NONE).all() and np.isinf(hits["t"]).all() and (pts["manifold_len"] == 0).all()
    bvh = rtg.HostBvh(np.zeros((0, 3, 3)))               # an empty mesh
    mobjs, _, _ = scenes.mesh_scene_objects(rtg.transform(), with_default=False)
    pm = scenes.Pair(gpu_ctx, mobjs, {0: bvh})
    g, _, _ = pm.gpu.render_primary(view, 8, 8)
    c, _, _ = pm.cpu.render_primary(view, 8, 8)
    _assert_hits_equal(g, c)


def test_invalid_requests_are_reported(gpu_ctx):
    objs, lights, view = scenes.default_objects(0)
    pair = scenes.Pair(gpu_ctx, objs, {})
    with pytest.raises(rtg.RtgError) as e:
        pair.gpu.render_primary(view, 16, 16, tile=(8, 8, 16, 16))
    assert e.value.code == -1 and "outside" in str(e.value)
    nodes = np.zeros(2, dtype=rtg.NODE_DTYPE)
    nodes[1]["child_begin"], nodes[1]["child_end"] = 0, 2     # node 1 lists itself as a child
    with pytest.raises(rtg.RtgError) as e:
        gpu_ctx.mesh(rtg.flat_bvh(nodes, np.zeros((0, 3, 3)), 1))
    assert "reachable twice" in str(e.value)
    nodes[1]["child_end"] = 5
    with pytest.raises(rtg.RtgError):
        gpu_ctx.mesh(rtg.flat_bvh(nodes, np.zeros((0, 3, 3)), 1))


@pytest.mark.parametrize("time", [0, 13])
def test_direct_shading_default_scene(gpu_ctx, time):
    """compute_direct_irrad x diffuse (src/render/renderer.rs:184-205,270-277)."""
    objs, lights, view = scenes.default_objects(time)
    pair = scenes.Pair(gpu_ctx, objs, {})
    w, h = 256, 256
    _, g_pts, _ = pair.gpu.render_primary(view, w, h, points=True)
    g_rgb, g_c8 = pair.gpu.shade_direct(lights, g_pts, rgba8=True)
    c_rgb, c_c8 = pair.cpu.shade_direct(lights, g_pts, rgba8=True)
    assert not scenes.float_bits_differ(g_rgb, c_rgb).any()                     # linear radiance bit-exact
    assert np.abs(g_c8.astype(int) - c_c8.astype(int)).max() <= 1            # 8-bit PQ colour within 1 LSB
    assert g_rgb.max() > 1.0 and (g_rgb[:, 0] == 0).all()                      # blue + green lights only


@pytest.mark.parametrize("max_depth", [1, 4])
def test_fused_frame_default_scene(gpu_ctx, max_depth):
    """raytrace_pixel with the specular path tree (src/render/renderer.rs:266-378), photon term empty."""
    objs, lights, view = scenes.default_objects(5)
    pair = scenes.Pair(gpu_ctx, objs, {})
    w, h = 300, 300
    g_rgb, g_c8, g_hits = pair.gpu.render_frame(view, lights, w, h, max_depth=max_depth)
    c_rgb, c_c8, c_hits = pair.cpu.render_frame(view, lights, w, h, max_depth=max_depth)
    _assert_hits_equal(g_hits, c_hits)
    assert not scenes.float_bits_differ(g_rgb, c_rgb).any()
    assert np.abs(g_c8.astype(int) - c_c8.astype(int)).max() <= 1
    sphere = g_hits["object"] == 0
    if max_depth == 1:
        assert (g_rgb[sphere] == 0).all()                   # diffuse 0 and no specular paths: a black disc
    else:
        assert (g_rgb[sphere].sum(axis=1) > 0).mean() > 0.5  # the glass ball shows the floor through it


def test_fused_frame_mesh_scene_and_cpp_renderer(gpu_ctx):
    tris, xform = scenes.terrain(60, 60, seed=3)
    bvh = rtg.HostBvh(tris)
    objs, lights, view = scenes.mesh_scene_objects(xform, time=40)
    pair = scenes.Pair(gpu_ctx, objs, {2: bvh})
    w, h = 160, 96
    g_rgb, g_c8, g_hits = pair.gpu.render_frame(view, lights, w, h, max_depth=4)
    c_rgb, c_c8, c_hits = pair.cpu.render_frame(view, lights, w, h, max_depth=4)
    _assert_hits_equal(g_hits, c_hits)
    assert not scenes.float_bits_differ(g_rgb, c_rgb).any()
    assert np.abs(g_c8.astype(int) - c_c8.astype(int)).max() <= 1
    # the C++ Renderer (host mirror of Renderer::new/render) drives the same path
    r_rgb, r_hits = rtg.render_scene_builder(gpu_ctx, 5, 120, 80, max_depth=4)
    objs, lights, view = scenes.default_objects(5)
    p2 = scenes.Pair(gpu_ctx, objs, {})
    c_rgb, _, c_hits = p2.cpu.render_frame(view, lights, 120, 80, max_depth=4)
    _assert_hits_equal(r_hits, c_hits)
    assert not scenes.float_bits_differ(r_rgb, c_rgb).any()
    hdr = rtg.to_hdr(r_rgb, 120, 80)
    from test_output_formats import decode_hdr_bytes
    assert hdr.startswith(b"#?RADIANCE\n") and decode_hdr_bytes(hdr).shape == (80, 120, 4)


@pytest.mark.parametrize("traversal", TRAVERSALS)
def test_frame_of_opaque_scene_in_passes(gpu_ctx, traversal):
    """Scenes without a refracting material: the default traversal renders the frame as passes of the persistent
    kernel (primary hits with point records, one shadow-ray batch over all lights, resolve); the other traversals use
    the fused kernel.  All must equal the oracle, on full frames, on a ragged tile and with no light at all."""
    gpu_ctx.set_option("traversal", traversal)
    tris, xform = scenes.terrain(80, 80, seed=9)
    vn = scenes.vertex_normals_for(tris)
    bvh = rtg.HostBvh(tris, vertex_normals=vn)
    _, lights, view = scenes.default_objects(17)
    plane = rtg.plane_object((0, -0.5, 0), (0, 0, -1), (-1, 0, 0), rtg.material((1.0, 1.0, 1.0)), rtg.material((0.1, 0.1, 0.1)))
    ball = rtg.sphere_object((0.3, -0.1, -0.4), 0.15, rtg.material((0.9, 0.2, 0.2)))
    mesh = rtg.mesh_object(None, rtg.material((0.8, 0.7, 0.6)), xform)
    for objs, meshes in (([mesh], {0: bvh}), ([plane, ball, mesh], {2: bvh})):
        pair = scenes.Pair(gpu_ctx, objs, meshes)
        w, h = 200, 120
        for tile, use_lights in ((None, lights), ((13, 7, 150, 90), lights), (None, lights[:0])):
            g_rgb, g_c8, g_hits = pair.gpu.render_frame(view, use_lights, w, h, tile=tile, max_depth=4)
            c_rgb, c_c8, c_hits = pair.cpu.render_frame(view, use_lights, w, h, tile=tile, max_depth=4)
            _assert_hits_equal(g_hits, c_hits)
            assert not scenes.float_bits_differ(g_rgb, c_rgb).any()
            assert np.abs(g_c8.astype(int) - c_c8.astype(int)).max() <= 1
            if len(use_lights):
                assert (g_rgb.sum(axis=1) > 0).mean() > 0.2
            else:
                assert (g_rgb == 0).all()
        # shading of already-built points goes through the same passes
        _, g_pts, _ = pair.gpu.render_primary(view, w, h, points=True)
        g_rgb, g_c8 = pair.gpu.shade_direct(lights, g_pts, rgba8=True)
        c_rgb, c_c8 = pair.cpu.shade_direct(lights, g_pts, rgba8=True)
        assert not scenes.float_bits_differ(g_rgb, c_rgb).any()
        assert np.abs(g_c8.astype(int) - c_c8.astype(int)).max() <= 1
    gpu_ctx.set_option("traversal", 2)


@pytest.mark.parametrize("traversal", TRAVERSALS)
@pytest.mark.parametrize("order", [0, 1])
def test_shadow_ray_meets_nan_hit_in_another_mesh(gpu_ctx, traversal, order):
    """compute_direct_irrad (src/render/renderer.rs:184-205) with two meshes on the shadow path, one of them holding
    zero-area triangles: a NaN-time hit in any mesh wins the scene's `.min()` and un-occludes the light, also when
    another mesh -- before or after it in the object list -- blocks the ray."""
    gpu_ctx.set_option("traversal", traversal)
    try:
        objs, meshes, yi = scenes.shadow_corner_scene(order)
        _, lights, view = scenes.default_objects(17)
        pair = scenes.Pair(gpu_ctx, objs, meshes)
        w, h = 200, 150
        for tile in (None, (9, 60, 180, 85)):
            g_rgb, g_c8, g_hits = pair.gpu.render_frame(view, lights, w, h, tile=tile, max_depth=4)
            c_rgb, c_c8, c_hits = pair.cpu.render_frame(view, lights, w, h, tile=tile, max_depth=4)
            _assert_hits_equal(g_hits, c_hits)
            assert not scenes.float_bits_differ(g_rgb, c_rgb).any()
            assert np.abs(g_c8.astype(int) - c_c8.astype(int)).max() <= 1
        # the same through rtg_shade_direct on already-built points
        _, g_pts, _ = pair.gpu.render_primary(view, w, h, points=True)
        g_rgb, g_c8 = pair.gpu.shade_direct(lights, g_pts, rgba8=True)
        s_rgb, s_c8 = pair.cpu.shade_direct(lights, g_pts, rgba8=True)
        assert not scenes.float_bits_differ(g_rgb, s_rgb).any()
        assert np.abs(g_c8.astype(int) - s_c8.astype(int)).max() <= 1
        # the scene does exercise the corner: without the mesh of zero-area triangles many floor pixels are darker
        keep = [i for i in range(len(objs)) if i != yi]
        without = scenes.Pair(None, [objs[i] for i in keep], {keep.index(i): m for i, m in meshes.items() if i != yi})
        o_rgb, _, o_hits = without.cpu.render_frame(view, lights, w, h, max_depth=4)
        c_rgb, _, c_hits = pair.cpu.render_frame(view, lights, w, h, max_depth=4)
        floor = (c_hits["object"] == 0) & (o_hits["object"] == 0)
        assert (floor & (c_rgb.sum(axis=1) > o_rgb.sum(axis=1))).sum() > 1000
    finally:
        gpu_ctx.set_option("traversal", 2)


@pytest.mark.parametrize("traversal", TRAVERSALS)
def test_many_instances_with_general_affine_transforms(gpu_ctx, traversal):
    """Twenty objects: two meshes instanced under rotated, sheared and mirrored transforms (Transform::from on an
    arbitrary forward matrix, src/geo/transform.rs:51-80), spheres, planes and a glass ball in between, so the object
    list is walked through meshes and primitives in every order and the scene-level minimum sees many candidates."""
    gpu_ctx.set_option("traversal", traversal)
    rng = np.random.default_rng(77)
    tris_a, _ = scenes.terrain(30, 30, seed=3, local_scale=1.0)
    tris_b, _ = scenes.terrain(20, 40, seed=4, local_scale=1.0)
    bvh_a = rtg.HostBvh(tris_a, vertex_normals=scenes.vertex_normals_for(tris_a))
    bvh_b = rtg.HostBvh(tris_b)

    def affine():
        m = np.eye(4)
        a = rng.normal(size=(3, 3))
        q, _ = np.linalg.qr(a)                                    # rotation or reflection
        shear = np.eye(3) + np.triu(rng.uniform(-0.3, 0.3, size=(3, 3)), 1)
        m[:3, :3] = q @ shear @ np.diag(rng.uniform(0.2, 0.6, size=3))
        m[:3, 3] = rng.uniform(-0.6, 0.6, size=3) + [0, 0, -1.0]
        return rtg.transform_from_forward(m)

    objs, meshes = [], {}
    for i in range(20):
        kind = i % 5
        if kind in (0, 3):
            objs.append(rtg.mesh_object(None, rtg.material(tuple(rng.uniform(0.2, 1, size=3))), affine()))
            meshes[i] = bvh_a if kind == 0 else bvh_b
        elif kind == 1:
            objs.append(rtg.sphere_object(tuple(rng.uniform(-0.5, 0.5, size=3) + [0, 0, -1.2]), float(rng.uniform(0.05, 0.2)),
                                          rtg.material((0.5, 0.5, 0.9), dielectric=(1.0, 1.5) if i == 6 else None), affine() if i > 10 else None))
        elif kind == 2:
            objs.append(rtg.plane_object((0, -0.8 - 0.01 * i, 0), (0, 0, -1), (-1, 0, 0), rtg.material((1, 1, 1)), rtg.material((0.1, 0.1, 0.1))))
        else:
            objs.append(rtg.sphere_object((0.0, 0.0, -3.0 - i), 0.5, rtg.material((0.3, 0.8, 0.3))))
    pair = scenes.Pair(gpu_ctx, objs, meshes)
    _, lights, view = scenes.default_objects(9)
    w, h = 192, 144
    g_hits, g_pts, g_ctr = pair.gpu.render_primary(view, w, h, points=True, counters=True)
    c_hits, c_pts, c_ctr = pair.cpu.render_primary(view, w, h, points=True, counters=True)
    _assert_hits_equal(g_hits, c_hits)
    _assert_points_equal(g_pts, c_pts)
    assert np.array_equal(g_ctr, c_ctr)
    assert len(np.unique(g_hits["object"])) >= 8
    g_rgb, g_c8, g_hits = pair.gpu.render_frame(view, lights, w, h, max_depth=4)
    c_rgb, c_c8, c_hits = pair.cpu.render_frame(view, lights, w, h, max_depth=4)
    _assert_hits_equal(g_hits, c_hits)
    assert not scenes.float_bits_differ(g_rgb, c_rgb).any()
    assert np.abs(g_c8.astype(int) - c_c8.astype(int)).max() <= 1
    rays = np.concatenate([rng.uniform(-1, 1, size=(20000, 3)) + [0, 0, -1], rng.normal(size=(20000, 3))], axis=1)
    rays[::1000, 0] = np.nan                                           # origins the sort key cannot place
    rays[500::1000, 1] = np.inf
    g_hits, g_pts, _ = pair.gpu.trace(rays, points=True)
    c_hits, c_pts, _ = pair.cpu.trace(rays, points=True)
    _assert_hits_equal(g_hits, c_hits)
    _assert_points_equal(g_pts, c_pts)
    if traversal == 2:                                                 # the same through the device-side ray sort
        gpu_ctx.set_option("sort_rays", 2)
        s_hits, s_pts, _ = pair.gpu.trace(rays, points=True)
        gpu_ctx.set_option("sort_rays", 1)
        _assert_hits_equal(s_hits, c_hits)
        _assert_points_equal(s_pts, c_pts)
    gpu_ctx.set_option("traversal", 2)


def _glass_scene(rng_seed=5):
    tris, xform = scenes.terrain(40, 40, seed=rng_seed)
    bvh = rtg.HostBvh(tris, vertex_normals=scenes.vertex_normals_for(tris))
    tris2, _ = scenes.terrain(12, 12, seed=rng_seed + 1, local_scale=1.0)
    bvh2 = rtg.HostBvh(tris2)
    _, lights, view = scenes.default_objects(11)
    objs = [
        rtg.sphere_object((0.0, -0.2, 0.0), 0.2, rtg.material((0.0, 0.0, 0.0), dielectric=(1.0, 1.5))),
        rtg.plane_object((0, -0.5, 0), (0, 0, -1), (-1, 0, 0), rtg.material((1, 1, 1)), rtg.material((0.1, 0.1, 0.1))),
        rtg.mesh_object(None, rtg.material((0.8, 0.7, 0.6)), xform),
        rtg.sphere_object((0.35, 0.1, -0.2), 0.15, rtg.material((0.2, 0.1, 0.1), dielectric=(1.0, 1.33))),
        # a small refracting mesh in front of everything
        rtg.mesh_object(None, rtg.material((0.1, 0.2, 0.1), dielectric=(1.0, 1.2)), rtg.transform(0.12, (-3.0, 1.0, 3.0))),
    ]
    return objs, {2: bvh, 4: bvh2}, lights, view


@pytest.mark.parametrize("max_depth", [2, 3, 5, 8])
def test_specular_tree_as_ray_batch_levels(gpu_ctx, max_depth):
    """Frames with refracting spheres and a refracting mesh: the default traversal renders the path tree level by
    level (spawn, trace with the previous level's shadow rays, shade, park per slot, re-sum depth first) and must
    give the recursion's radiance bit for bit -- on a full frame, a ragged tile, and with only 8-bit output."""
    objs, meshes, lights, view = _glass_scene()
    pair = scenes.Pair(gpu_ctx, objs, meshes)
    w, h = 176, 132
    for tile in (None, (21, 9, 130, 101)):
        g_rgb, g_c8, g_hits = pair.gpu.render_frame(view, lights, w, h, tile=tile, max_depth=max_depth)
        c_rgb, c_c8, c_hits = pair.cpu.render_frame(view, lights, w, h, tile=tile, max_depth=max_depth)
        _assert_hits_equal(g_hits, c_hits)
        assert not scenes.float_bits_differ(g_rgb, c_rgb).any()
        assert np.abs(g_c8.astype(int) - c_c8.astype(int)).max() <= 1
    glass = np.isin(c_hits["object"], [0, 3, 4])
    assert glass.mean() > 0.05 and (c_rgb[glass].sum(axis=1) > 0).mean() > 0.5
    only8 = np.zeros((w * h, 4), dtype=np.uint8)
    la = np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE)
    pair.gpu.render_frame_raw(view, la, len(la), w, h, (0, 0, w, h), max_depth, None, only8, None)
    f_rgb, f_c8, _ = pair.gpu.render_frame(view, lights, w, h, max_depth=max_depth)
    assert np.array_equal(only8, f_c8)


def test_specular_slot_table_over_budget_falls_back(gpu_ctx):
    """With max_depth 8 a refracting pixel owns 255 slots of 32 bytes; a frame with more than 263 000 such pixels
    exceeds the 2 GiB slot budget and the fused kernel redoes them.  The result must not change."""
    objs, meshes, lights, view = _glass_scene()
    objs[0] = rtg.sphere_object((0.0, 0.0, 0.0), 0.6, rtg.material((0.0, 0.0, 0.0), dielectric=(1.0, 1.5)))
    pair = scenes.Pair(gpu_ctx, objs, meshes)
    w, h = 1024, 640
    g_rgb, g_c8, g_hits = pair.gpu.render_frame(view, lights, w, h, max_depth=8)
    assert (np.isin(g_hits["object"], [0, 3, 4])).sum() > 270000
    rows = (300, 8)                                           # the oracle checks a strip through the sphere
    c_rgb, c_c8, c_hits = pair.cpu.render_frame(view, lights, w, h, tile=(0, rows[0], w, rows[1]), max_depth=8)
    sl = slice(rows[0] * w, (rows[0] + rows[1]) * w)
    _assert_hits_equal(g_hits[sl], c_hits)
    assert not scenes.float_bits_differ(g_rgb[sl], c_rgb).any()
    assert np.abs(g_c8[sl].astype(int) - c_c8.astype(int)).max() <= 1


@pytest.mark.parametrize("seed", range(int(os.environ.get("RTG_FUZZ_SEEDS", "12"))))
def test_fuzz_scenes_with_special_values(gpu_ctx, seed):
    """Random small scenes (meshes under random affine transforms, spheres, planes, in random order) traced with rays
    that carry special values: zero, denormal, huge and tiny components, infinities and NaNs in origins and
    directions, origins exactly on geometry.  Whatever the reference's arithmetic yields for them -- including NaN
    hit times -- the GPU must yield too."""
    rng = np.random.default_rng(1000 + seed)
    objs, meshes = [], {}
    for i in range(int(rng.integers(1, 6))):
        kind = rng.integers(0, 3)
        m = np.eye(4)
        m[:3, :3] = rng.normal(size=(3, 3)) * rng.choice([1e-3, 0.3, 1.0, 40.0])
        m[:3, 3] = rng.normal(size=3)
        xf = rtg.transform_from_forward(m) if rng.random() < 0.7 else None
        if kind == 0:
            n = int(rng.integers(1, 400))
            tris = rng.normal(size=(n, 1, 3)) * 2 + rng.normal(size=(n, 3, 3)) * rng.choice([0.05, 0.5])
            if rng.random() < 0.3:
                tris[: n // 4, :, int(rng.integers(0, 3))] = np.round(tris[: n // 4, :1, 0], 1)   # axis-aligned flats
            if rng.random() < 0.3:
                tris[n // 2:] *= 2.0 ** int(rng.integers(-60, 60))                                   # outside the f32 window
            objs.append(rtg.mesh_object(None, rtg.material((0.5, 0.5, 0.5)), xf))
            meshes[i] = rtg.HostBvh(np.ascontiguousarray(tris))
        elif kind == 1:
            objs.append(rtg.sphere_object(tuple(rng.normal(size=3)), float(abs(rng.normal()) + 0.01), rtg.material((1, 1, 1)), xf))
        else:
            objs.append(rtg.plane_object(tuple(rng.normal(size=3)), tuple(rng.normal(size=3)), tuple(rng.normal(size=3)),
                                         rtg.material((1, 1, 1)), rtg.material((0.1, 0.1, 0.1)), xf))
    pair = scenes.Pair(gpu_ctx, objs, meshes)
    n = 4000
    rays = np.concatenate([rng.normal(size=(n, 3)) * 3, rng.normal(size=(n, 3))], axis=1)
    special = np.array([0.0, -0.0, 5e-324, 1e-310, 1e-200, 1e200, 1e308, -1e308, np.inf, -np.inf, np.nan, 1.0, -1.0, 2.0 ** -41, 2.0 ** 41])
    sel = rng.random(size=rays.shape) < 0.08
    rays[sel] = rng.choice(special, size=int(sel.sum()))
    rays[: n // 10, 3:] = np.eye(3)[rng.integers(0, 3, size=n // 10)] * rng.choice([-1.0, 1.0], size=(n // 10, 1))
    for traversal in TRAVERSALS:
        gpu_ctx.set_option("traversal", traversal)
        g_hits, g_pts, g_ctr = pair.gpu.trace(rays, points=True, counters=True)
        c_hits, c_pts, c_ctr = pair.cpu.trace(rays, points=True, counters=True)
        _assert_hits_equal(g_hits, c_hits)
        _assert_points_equal(g_pts, c_pts)
        assert np.array_equal(g_ctr, c_ctr)
    gpu_ctx.set_option("traversal", 2)


def test_device_f64_div_and_sqrt_are_correctly_rounded(gpu_ctx):
    """The parity argument rests on IEEE-exact device / and sqrt: sweep them through the ray normalisation."""
    import math
    rng = np.random.default_rng(1)
    n = 200000
    o = rng.uniform(-10, 10, size=(n, 3))
    d = rng.normal(size=(n, 3)) * np.exp(rng.uniform(-20, 20, size=(n, 1)))
    rays = np.concatenate([o, d], axis=1)
    # one huge sphere around everything: t comes from sqrt and three divisions of unnormalised data
    objs = [rtg.sphere_object((0.5, -0.25, 0.125), 100.0, rtg.material((1, 1, 1)))]
    pair = scenes.Pair(gpu_ctx, objs, {})
    g_hits, g_pts, _ = pair.gpu.trace(rays, points=True)
    c_hits, c_pts, _ = pair.cpu.trace(rays, points=True)
    _assert_hits_equal(g_hits, c_hits)
    _assert_points_equal(g_pts, c_pts)
    assert (g_hits["object"] == 0).all()
    del math


@pytest.mark.parametrize("seed", range(int(os.environ.get("RTG_FUZZ_SEEDS", "8"))))
def test_fuzz_shaded_frames(gpu_ctx, seed):
    """Random scenes with refracting and opaque spheres, planes and meshes under random transforms, one to three
    lights anywhere (inside objects too), a random view and a random depth of the path tree: the frame -- primary
    hits, direct light with its shadow rays, the specular tree -- must be the recursion's radiance bit for bit, over
    the whole frame and over a ragged tile, for every traversal."""
    rng = np.random.default_rng(9000 + seed)
    objs, meshes = [], {}
    for i in range(int(rng.integers(1, 7))):
        kind = rng.integers(0, 3)
        m = np.eye(4)
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        m[:3, :3] = q @ np.diag(rng.uniform(0.5, 1.6, size=3)) if rng.random() < 0.5 else q * rng.uniform(0.5, 1.5)
        m[:3, 3] = rng.normal(scale=0.2, size=3)
        xf = rtg.transform_from_forward(m) if rng.random() < 0.6 else None
        glass = rng.random() < 0.4
        mat = rtg.material(tuple(rng.uniform(0.0, 1.0, size=3)),
                           dielectric=(1.0, float(rng.choice([1.1, 1.33, 1.5, 2.4]))) if glass else None)
        centre = rng.uniform(-0.6, 0.6, size=3) + [0.0, 0.0, -0.6]
        if kind == 0:
            if rng.random() < 0.5:
                tris, _ = scenes.terrain(int(rng.integers(3, 14)), int(rng.integers(3, 14)), seed=seed * 7 + i, local_scale=1.0)
                tris = tris * rng.uniform(0.05, 0.2) + centre
            else:
                n = int(rng.integers(1, 200))
                tris = centre + rng.normal(size=(n, 1, 3)) * 0.4 + rng.normal(size=(n, 3, 3)) * rng.choice([0.03, 0.2])
            objs.append(rtg.mesh_object(None, mat, xf))
            meshes[i] = rtg.HostBvh(np.ascontiguousarray(tris))
        elif kind == 1:
            objs.append(rtg.sphere_object(tuple(centre), float(rng.uniform(0.05, 0.4)), mat, xf))
        else:
            t1, t2 = rng.normal(size=3), rng.normal(size=3)
            objs.append(rtg.plane_object(tuple(centre + [0.0, -0.6, 0.0]), tuple(t1), tuple(t2), mat,
                                         rtg.material(tuple(rng.uniform(0.0, 1.0, size=3))), xf))
    pair = scenes.Pair(gpu_ctx, objs, meshes)
    lights = np.zeros(int(rng.integers(1, 4)), dtype=rtg.LIGHT_DTYPE)
    lights["position"] = rng.uniform(-1.5, 1.5, size=(lights.shape[0], 3))
    lights["color"] = rng.uniform(0.0, 200.0, size=(lights.shape[0], 3))
    view = rtg.make_view(orig=tuple(rng.normal(scale=0.3, size=3) + [0.0, 0.0, 1.0]), fov=(float(rng.uniform(0.5, 2.5)), float(rng.uniform(0.5, 2.5))))
    w, h = int(rng.integers(17, 120)), int(rng.integers(17, 90))
    max_depth = int(rng.integers(1, 6))
    tile = (int(rng.integers(0, w // 2)), int(rng.integers(0, h // 2)), int(rng.integers(1, w // 2)), int(rng.integers(1, h // 2)))
    c_rgb, c_c8, c_hits = pair.cpu.render_frame(view, lights, w, h, max_depth=max_depth)
    t_rgb, t_c8, t_hits = pair.cpu.render_frame(view, lights, w, h, tile=tile, max_depth=max_depth)
    if os.environ.get("RTG_FUZZ_VERBOSE"):
        print(f"seed {seed}: {len(objs)} objects {w}x{h} depth {max_depth} hit {float((c_hits['object'] != rtg.RTG_NONE).mean()):.2f} "
              f"lit {float((c_rgb.sum(axis=1) > 0).mean()):.2f} objects seen {np.unique(c_hits['object']).tolist()}")
    for traversal in TRAVERSALS:
        gpu_ctx.set_option("traversal", traversal)
        g_rgb, g_c8, g_hits = pair.gpu.render_frame(view, lights, w, h, max_depth=max_depth)
        _assert_hits_equal(g_hits, c_hits)
        assert not scenes.float_bits_differ(g_rgb, c_rgb).any(), (seed, traversal)
        assert np.abs(g_c8.astype(int) - c_c8.astype(int)).max() <= 1
        g_rgb, g_c8, g_hits = pair.gpu.render_frame(view, lights, w, h, tile=tile, max_depth=max_depth)
        _assert_hits_equal(g_hits, t_hits)
        assert not scenes.float_bits_differ(g_rgb, t_rgb).any(), (seed, traversal, "tile")
    gpu_ctx.set_option("traversal", 2)


@pytest.mark.parametrize("family", [0, 1, 2, 3])
def test_reciprocal_quotient_equals_ieee_division(gpu_ctx, family):
    """The phased kernel's box test replaces x/d by two FMA refinements of x*(1/d); every bit must agree with the
    hardware division over the whole magnitude window the kernel admits (2^28 pairs per family)."""
    import struct
    bad, xb, db = gpu_ctx.selftest_division(1 << 28, family, seed=20261111 + family)
    x, d = struct.unpack("<d", struct.pack("<Q", xb))[0], struct.unpack("<d", struct.pack("<Q", db))[0]
    assert bad == 0, f"family {family}: {bad} quotients differ, e.g. x={x!r} d={d!r}"


def _assert_paths_equal(g, c):
    assert g.shape == c.shape
    for f in ("ray", "depth", "modes", "parent"):
        assert np.array_equal(g[f], c[f]), f
    assert not scenes.float_bits_differ(g["attenuation"], c["attenuation"]).any()
    gp, cp = g["point"], c["point"]
    for f in ("manifold", "manifold_len", "material_slot"):
        assert np.array_equal(gp[f], cp[f]), f
    for f in ("time", "position", "inter_normal", "geo_normal", "manifold_point"):
        assert not scenes.float_bits_differ(gp[f], cp[f]).any(), f


@pytest.mark.parametrize("traversal", [0, 2])
@pytest.mark.parametrize("max_depth", [1, 2, 4, 6])
def test_specular_paths_of_a_ray_batch(gpu_ctx, traversal, max_depth):
    """raytrace_all_specular for a batch (the photon pass's loop body, renderer.rs:119-138): every path record --
    order, modes, parents, attenuation, point -- equals the recursion's, for rays leaving the two lights in all
    directions and for camera rays."""
    objs, meshes, lights, view = _glass_scene()
    pair = scenes.Pair(gpu_ctx, objs, meshes)
    gpu_ctx.set_option("traversal", traversal)
    rng = np.random.default_rng(31)
    d = rng.normal(size=(6000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    la = np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE)
    origins = np.stack([np.asarray(la[i % len(la)]["position"], dtype=np.float64) for i in range(len(d))])
    cam = rtg.bench_mesh_rays([-0.1, -0.1, 0.9, 0.1, 0.1, 1.0], 3000, seed=3)
    cam[:, 3:] = (np.array([0.0, -0.15, -1.0]) + 0.35 * (cam[:, 3:])) / np.linalg.norm(
        np.array([0.0, -0.15, -1.0]) + 0.35 * cam[:, 3:], axis=1, keepdims=True)
    rays = np.concatenate([np.concatenate([origins, d], axis=1), cam])
    try:
        g = pair.gpu.trace_paths(rays, max_depth=max_depth)
    finally:
        gpu_ctx.set_option("traversal", 2)
    c = pair.cpu.trace_paths(rays, max_depth=max_depth)
    _assert_paths_equal(g, c)
    assert (c["depth"] == 0).sum() > 4000
    if max_depth >= 4:
        assert (c["depth"] >= 2).sum() > 200 and (c["modes"] != 0).sum() > 200
        # following parents recovers the surfaces a path bounced off (the reference's `manifolds`)
        deep = np.flatnonzero(c["depth"] == 3)[:50]
        for i in deep:
            chain, at = [], g["parent"][i]
            while at != rtg.RTG_NONE:
                assert g["ray"][at] == g["ray"][i] and g["depth"][at] == g["depth"][i] - 1 - len(chain)
                chain.append(at)
                at = g["parent"][at]
            assert len(chain) == 3


def test_specular_paths_limits_and_plain_scenes(gpu_ctx):
    objs, lights, view = scenes.default_objects(0)            # glass sphere over the checker plane, no mesh
    pair = scenes.Pair(gpu_ctx, objs, {})
    rays = rtg.bench_mesh_rays([-0.3, -0.3, 0.8, 0.3, 0.3, 1.0], 2000, seed=8)
    rays[:, 3:] = (np.array([0.0, -0.2, -1.0]) + 0.3 * rays[:, 3:])
    rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    g = pair.gpu.trace_paths(rays, max_depth=4)
    c = pair.cpu.trace_paths(rays, max_depth=4)
    _assert_paths_equal(g, c)
    assert (c["depth"] > 0).sum() > 100
    with pytest.raises(rtg.RtgError, match="room for"):
        pair.gpu.trace_paths(rays, max_depth=4, capacity=len(c) - 1)
    exact = pair.gpu.trace_paths(rays, max_depth=4, capacity=len(c))
    _assert_paths_equal(exact, c)
    with pytest.raises(rtg.RtgError, match="max_depth"):
        pair.gpu.trace_paths(rays, max_depth=9)
    assert len(pair.gpu.trace_paths(np.zeros((0, 6)), max_depth=4)) == 0
    # rays that hit nothing give no paths
    up = rays.copy()
    up[:, 3:] = [0.0, 1.0, 0.0]
    assert len(pair.gpu.trace_paths(up, max_depth=4)) == 0


def test_contexts_of_one_gpu_share_a_mesh_with_frames_in_flight(gpu_ctx):
    """Two contexts (a stream each) render alternate frames from one uploaded tree; every frame equals the frame a
    single context renders, and the oracle's for the first."""
    import torch
    tris, xform = scenes.terrain(60, 50, seed=12)
    bvh = rtg.HostBvh(tris)
    mesh = gpu_ctx.mesh(bvh)
    obj = rtg.mesh_object(mesh, rtg.material((0.5, 0.5, 0.5)), xform)
    other = rtg.Context(0)
    ctxs = [gpu_ctx, other]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for c, st in zip(ctxs, streams):
        c.set_stream(st.cuda_stream)
    try:
        sc = [c.scene([obj]) for c in ctxs]
        w, h, n_frames = 320, 200, 12
        views = [rtg.make_view(orig=(0.02 * f, 0.01 * f, 1.0)) for f in range(n_frames)]
        out = [torch.empty(w * h * 16, dtype=torch.uint8, device="cuda") for _ in range(n_frames)]
        torch.cuda.synchronize()
        for f in range(n_frames):
            sc[f & 1].render_primary_raw(views[f], w, h, (0, 0, w, h), out[f].data_ptr())
        torch.cuda.synchronize()
        for f in range(n_frames):
            alone, _, _ = sc[0].render_primary(views[f], w, h)
            got = np.frombuffer(out[f].cpu().numpy().tobytes(), dtype=rtg.HIT_DTYPE)
            _assert_hits_equal(got, alone)
        obvh = orc.OracleBvh(flat=bvh.view)
        oscene = orc.OracleScene([orc.oracle_object(obj, obvh)])
        c_hits, _, _ = oscene.render_primary(views[0], w, h)
        _assert_hits_equal(np.frombuffer(out[0].cpu().numpy().tobytes(), dtype=rtg.HIT_DTYPE), c_hits)
        assert (c_hits["leaf"] != rtg.RTG_NONE).mean() > 0.3
        for s_ in sc:
            s_.close()
    finally:
        gpu_ctx.set_stream(None)
        other.close()


def test_kernel_time_reports_every_timed_launch(gpu_ctx):
    tris, xform = scenes.terrain(40, 40, seed=3)
    pair = scenes.Pair(gpu_ctx, scenes.mesh_scene_objects(xform, with_default=False)[0], {0: rtg.HostBvh(tris)})
    view = rtg.make_view()
    assert gpu_ctx.kernel_time() == (0.0, 0)                  # nothing is timed unless asked for
    gpu_ctx.set_option("time_kernels", 2)
    try:
        for _ in range(5):
            pair.gpu.render_primary(view, 256, 192)
        rays = rtg.bench_mesh_rays([-1, -1, -1, 1, 1, 1], 5000, seed=2)
        pair.gpu.trace(rays)
        ms, n = gpu_ctx.kernel_time()
        assert n == 6 and 0.0 < ms < 600.0
        assert gpu_ctx.kernel_time() == (0.0, 0)              # the query starts over
    finally:
        gpu_ctx.set_option("time_kernels", 1)
    pair.gpu.render_primary(view, 64, 64)
    assert gpu_ctx.kernel_time() == (0.0, 0)


def test_frame_in_two_row_bands_equals_one_band(gpu_ctx):
    """An opaque scene's frame is issued as two row bands on two streams; outputs (device and host buffers, full
    frame and ragged tile) equal the one-band frame and the oracle's."""
    import torch
    tris, xform = scenes.terrain(50, 50, seed=21)
    objs, lights, view = scenes.mesh_scene_objects(xform, with_default=True)
    objs[0] = rtg.sphere_object((0.0, -0.2, 0.0), 0.2, rtg.material((0.3, 0.2, 0.6)))      # opaque sphere
    pair = scenes.Pair(gpu_ctx, objs, {2: rtg.HostBvh(tris, vertex_normals=scenes.vertex_normals_for(tris))})
    w, h = 352, 300
    la = np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE)
    try:
        for tile in (None, (5, 3, 301, 277)):
            gpu_ctx.set_option("frame_bands", 1)
            one = pair.gpu.render_frame(view, lights, w, h, tile=tile, max_depth=4)
            gpu_ctx.set_option("frame_bands", 2)
            two = pair.gpu.render_frame(view, lights, w, h, tile=tile, max_depth=4)
            for a, b in zip(one, two):
                assert a.tobytes() == b.tobytes()
            c_rgb, c_c8, c_hits = pair.cpu.render_frame(view, lights, w, h, tile=tile, max_depth=4)
            _assert_hits_equal(two[2], c_hits)
            assert not scenes.float_bits_differ(two[0], c_rgb).any()
            assert np.abs(two[1].astype(int) - c_c8.astype(int)).max() <= 1
        # device buffers, several frames back to back on the context's stream
        n = w * h
        d_rgb = torch.empty(n * 3, dtype=torch.float64, device="cuda")
        d_c8 = torch.empty(n * 4, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for _ in range(3):
            pair.gpu.render_frame_raw(view, la, len(la), w, h, (0, 0, w, h), 4, d_rgb.data_ptr(), d_c8.data_ptr(), None)
        gpu_ctx.synchronize()
        full = pair.gpu.render_frame(view, lights, w, h, max_depth=4)
        assert d_rgb.cpu().numpy().tobytes() == full[0].tobytes()
        assert d_c8.cpu().numpy().tobytes() == full[1].tobytes()
    finally:
        gpu_ctx.set_option("frame_bands", 2)
