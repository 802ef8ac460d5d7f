"""The tree built on the GPU (rtg_mesh_build) is the tree BvhForest::subdivide + Bvh::new make on the host
(tree/bvh_subdivide.rs:8-43, tree/bvh.rs:80-106): node arrays, leaf order and leaf -> triangle map compared byte for
byte with the host builder, on meshes chosen to hit the tie rules (equal centres, equal extents, probes landing on
the split coordinate, the 1 % clamp, signed zeros)."""
import time

import numpy as np
import pytest

import scenes
from raytracer_amd import rtg

pytestmark = pytest.mark.gpu


def _lattice(nx, ny):
    """Axis-aligned flat quads on the integer lattice: every centre and every extent ties with many others."""
    tris = []
    for y in range(ny):
        for x in range(nx):
            a, b, c, d = (x, y, 0.0), (x + 1.0, y, 0.0), (x + 1.0, y + 1.0, 0.0), (x, y + 1.0, 0.0)
            tris.append((a, b, c))
            tris.append((a, c, d))
    return np.array(tris, dtype=np.float64)


def _soup(n, seed, quantum):
    rng = np.random.default_rng(seed)
    base = rng.uniform(-4.0, 4.0, size=(n, 1, 3))
    tris = base + rng.uniform(-0.5, 0.5, size=(n, 3, 3))
    if quantum:
        tris = np.round(tris / quantum) * quantum
    return np.ascontiguousarray(tris)


def _cases():
    yield "terrain", scenes.terrain(60, 40, seed=3)[0]
    yield "lattice", _lattice(37, 23)
    yield "lattice_cube", np.concatenate([_lattice(16, 16), _lattice(16, 16)[:, :, [2, 0, 1]], _lattice(16, 16)[:, :, [1, 2, 0]]])
    yield "duplicates", np.repeat(_soup(1, 1, 0.0), 257, axis=0)
    yield "soup", _soup(5000, 2, 0.0)
    yield "soup_quantised", _soup(5000, 3, 0.25)
    z = _soup(700, 4, 0.5)
    z[z == 0.0] = -0.0                                     # negative zeros next to positive ones
    z[::3][z[::3] == 0.0] = 0.0
    yield "signed_zeros", z
    skew = _soup(3000, 5, 0.0)
    skew[:2900] *= 1e-3                                    # almost everything in one corner: the 1 % clamp decides
    yield "skewed", skew
    for n in (2, 3, 4, 5, 64, 65):
        yield f"n{n}", _soup(n, 10 + n, 0.0)


@pytest.mark.parametrize("name,tris", list(_cases()), ids=[c[0] for c in _cases()])
def test_device_built_tree_equals_host_built_tree(gpu_ctx, name, tris):
    normals = scenes.vertex_normals_for(tris) if name in ("terrain", "soup") else None
    host = rtg.HostBvh(tris, normals)
    mesh = gpu_ctx.build_mesh(tris, normals)
    n_nodes, n_leaves, root, depth = mesh.sizes()
    assert (n_nodes, n_leaves, root, depth) == (host.n_nodes, host.n_leaves, host.root, host.depth)
    nodes, leaves, source = mesh.read_bvh()
    assert np.array_equal(source, host.leaf_source())
    assert leaves.tobytes() == host.leaves().tobytes()
    assert nodes.tobytes() == host.nodes().tobytes()
    assert mesh.bounds().tobytes() == host.bounds().tobytes()

    # the device-written kernel layouts (threaded nodes, single-precision companions, triangles, normals) give the
    # hits and points of the host-flattened mesh, in every traversal
    reference = gpu_ctx.mesh(host)
    lo, hi = host.bounds()[:3], host.bounds()[3:]
    view = rtg.make_view(orig=(float((lo[0] + hi[0]) / 2), float((lo[1] + hi[1]) / 2), float(hi[2] + (hi - lo).max())))
    rays = rtg.bench_mesh_rays(np.concatenate([lo - 0.1, hi + 0.1]), 4096, seed=7)
    for traversal in (0, 1, 2):
        gpu_ctx.set_option("traversal", traversal)
        out = []
        for m in (mesh, reference):
            sc = gpu_ctx.scene([rtg.mesh_object(m, rtg.material((0.5, 0.5, 0.5)))])
            h1, p1, _ = sc.render_primary(view, 96, 64, points=True)
            h2, p2, _ = sc.trace(rays, points=True)
            out.append((h1, p1, h2, p2))
            sc.close()
        for a, b in zip(*out):
            assert a.tobytes() == b.tobytes()
        assert (out[0][0]["leaf"] != 0xFFFFFFFF).any() or name.startswith("lattice") or name.startswith("n")
    gpu_ctx.set_option("traversal", 2)
    mesh.close()
    reference.close()


def test_random_meshes_of_many_sizes(gpu_ctx):
    """Sizes around the sort's internal block boundaries, with and without coordinate ties."""
    rng = np.random.default_rng(99)
    sizes = [6, 7, 8, 9, 31, 33, 127, 129, 255, 256, 257, 511, 513, 1023, 1025, 2047, 2049, 4095, 4097, 8191, 8193,
             16385, 40000, 70001]
    for n in sizes + [int(x) for x in rng.integers(10, 30000, size=16)]:
        for quantum in (0.0, 0.125, 1.0):
            tris = _soup(n, int(rng.integers(1 << 30)), quantum)
            host = rtg.HostBvh(tris)
            mesh = gpu_ctx.build_mesh(tris)
            nodes, leaves, source = mesh.read_bvh()
            assert np.array_equal(source, host.leaf_source()), (n, quantum)
            assert nodes.tobytes() == host.nodes().tobytes(), (n, quantum)
            assert mesh.sizes() == (host.n_nodes, host.n_leaves, host.root, host.depth), (n, quantum)
            # the positions the device builder gives the threaded nodes (pair slots by rank) make the tree the host
            # placement makes: same answers for rays from all around, in the kernels that follow the links
            reference = gpu_ctx.mesh(host)
            lo, hi = host.bounds()[:3], host.bounds()[3:]
            rays = rtg.bench_mesh_rays(np.concatenate([lo - 0.1, hi + 0.1]), 768, seed=n)
            answers = []
            for m in (mesh, reference):
                sc = gpu_ctx.scene([rtg.mesh_object(m, rtg.material((0.5, 0.5, 0.5)))])
                row = []
                for traversal in (2, 1):
                    gpu_ctx.set_option("traversal", traversal)
                    row.append(sc.trace(rays, counters=True))
                answers.append(row)
                sc.close()
            gpu_ctx.set_option("traversal", 2)
            for (h_a, _, c_a), (h_b, _, c_b) in zip(*answers):
                assert h_a.tobytes() == h_b.tobytes() and c_a.tobytes() == c_b.tobytes(), (n, quantum)
            assert answers[0][0][0].tobytes() == answers[0][1][0].tobytes(), (n, quantum)
            reference.close()
            mesh.close()
            host.close()


def test_trivial_and_invalid_inputs(gpu_ctx):
    empty = gpu_ctx.build_mesh(np.zeros((0, 3, 3)))
    host = rtg.HostBvh(np.zeros((0, 3, 3)))
    assert empty.sizes()[:3] == (host.n_nodes, 0, host.root)
    assert empty.read_bvh()[0].tobytes() == host.nodes().tobytes()
    one = _soup(1, 9, 0.0)
    one[0, 1, 2] = -0.0
    single = gpu_ctx.build_mesh(one, scenes.vertex_normals_for(_soup(1, 9, 0.0)))
    host = rtg.HostBvh(one)
    nodes, leaves, source = single.read_bvh()
    assert nodes.tobytes() == host.nodes().tobytes() and leaves.tobytes() == host.leaves().tobytes()
    assert list(source) == [0]
    for bad in (np.nan, np.inf, -np.inf):
        tris = _soup(100, 1, 0.0)
        tris[57, 2, 1] = bad
        with pytest.raises(rtg.RtgError, match="not finite"):
            gpu_ctx.build_mesh(tris)
        with pytest.raises(rtg.RtgError, match="not finite"):
            gpu_ctx.build_mesh(tris[57:58])


def test_triangles_already_on_the_device(gpu_ctx):
    import torch
    tris = scenes.terrain(50, 30, seed=5)[0]
    normals = scenes.vertex_normals_for(tris)
    d_tris = torch.from_numpy(tris).cuda()
    d_normals = torch.from_numpy(normals).cuda()
    torch.cuda.synchronize()
    mesh = gpu_ctx.build_mesh((d_tris.data_ptr(), tris.shape[0]), d_normals.data_ptr())
    host = rtg.HostBvh(tris, normals)
    nodes, leaves, source = mesh.read_bvh()
    assert nodes.tobytes() == host.nodes().tobytes()
    assert leaves.tobytes() == host.leaves().tobytes()
    assert np.array_equal(source, host.leaf_source())


def test_million_triangle_build(gpu_ctx):
    import bench
    tris = rtg.heightfield(*bench.GRID, bench.SEED, bench.LOCAL_SCALE)
    t0 = time.perf_counter()
    host = rtg.HostBvh(tris)
    t1 = time.perf_counter()
    gpu_ctx.build_mesh(tris[:1000]).close()               # first-use costs out of the timing
    gpu_ctx.synchronize()
    t2 = time.perf_counter()
    mesh = gpu_ctx.build_mesh(tris)
    gpu_ctx.synchronize()
    t3 = time.perf_counter()
    print(f"\n1M triangles: host build {t1 - t0:.2f} s, device build (upload included) {t3 - t2:.3f} s")
    nodes, leaves, source = mesh.read_bvh()
    assert np.array_equal(source, host.leaf_source())
    assert nodes.tobytes() == host.nodes().tobytes()
    assert leaves.tobytes() == host.leaves().tobytes()
    assert mesh.sizes() == (host.n_nodes, host.n_leaves, host.root, host.depth)


def test_patch_set_built_on_the_device(gpu_ctx):
    patches = rtg.patch_terrain(12, 9, seed=4, scale=1.0)
    host = rtg.HostPatchMesh(patches, k=3, newton_steps=5)
    built = rtg.Mesh.build_patches(gpu_ctx, patches, k=3, newton_steps=5)
    reference = gpu_ctx.mesh(host)
    assert built.read_bvh()[0].tobytes() == host.bvh.nodes().tobytes()
    lo, hi = host.bvh.bounds()[:3], host.bvh.bounds()[3:]
    view = rtg.make_view(orig=(float((lo[0] + hi[0]) / 2), float((lo[1] + hi[1]) / 2), float(hi[2] + (hi - lo).max())))
    rays = rtg.bench_mesh_rays(np.concatenate([lo - 0.1, hi + 0.1]), 4096, seed=11)
    out = []
    for m in (built, reference):
        sc = gpu_ctx.scene([rtg.patch_object(m, rtg.material((0.5, 0.5, 0.5)))])
        h1, p1, _ = sc.render_primary(view, 128, 96, points=True)
        h2, p2, _ = sc.trace(rays, points=True)
        out.append((h1, p1, h2, p2))
        sc.close()
    for a, b in zip(*out):
        assert a.tobytes() == b.tobytes()
    assert (out[0][0]["leaf"] != 0xFFFFFFFF).mean() > 0.2


def test_build_entry_points_reject_bad_arguments(gpu_ctx):
    import ctypes as C
    lib = rtg.lib
    h = C.c_void_p()
    tris = _soup(10, 1, 0.0)
    assert lib.rtg_mesh_build(gpu_ctx._h, None, 10, None, C.byref(h)) == -1          # RTG_E_INVALID: null triangles
    assert b"null argument" in lib.rtg_last_error(gpu_ctx._h)
    assert lib.rtg_mesh_build(gpu_ctx._h, rtg._ptr(tris), 10, None, None) == -1
    assert lib.rtg_mesh_build(gpu_ctx._h, rtg._ptr(tris), (1 << 30) + 1, None, C.byref(h)) != 0
    assert b"2^30" in lib.rtg_last_error(gpu_ctx._h)
    # patch sets: leaf records must name existing patches
    patches = rtg.patch_terrain(3, 3, seed=1, scale=1.0)
    facets, info = rtg.patches_tessellate(patches, 2)
    info = np.ascontiguousarray(info).copy()
    info["patch"][5] = len(patches)
    ps = rtg.PatchSet()
    ps.patches = patches.ctypes.data
    ps.n_patches = patches.shape[0]
    ps.leaf_info = info.ctypes.data
    ps.newton_steps = 5
    assert lib.rtg_mesh_build_patches(gpu_ctx._h, rtg._ptr(facets), facets.shape[0], C.byref(ps), C.byref(h)) == -1
    assert b"names patch" in lib.rtg_last_error(gpu_ctx._h)
    ps.newton_steps = 65
    info["patch"][5] = 0
    assert lib.rtg_mesh_build_patches(gpu_ctx._h, rtg._ptr(facets), facets.shape[0], C.byref(ps), C.byref(h)) != 0
    # read-back with every output optional
    mesh = gpu_ctx.build_mesh(tris)
    assert lib.rtg_mesh_read_bvh(mesh._h, None, None, None) == 0
    assert lib.rtg_mesh_sizes(mesh._h, None, None, None, None) == 0
    assert lib.rtg_mesh_sizes(None, None, None, None, None) == -1
