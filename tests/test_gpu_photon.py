"""The photon map on the GPU against the oracle, through the C ABI: the photon pass, the k-d tree's nearest
photons and the caustic term (src/render/renderer.rs:116-138, 206-265; src/tree/kd_tree.rs).

Bars: the photon pass and the nearest-photon search are bit-exact (ids, distances, every field of a photon).  The
caustic term goes through sin / cos of the corrected direction, which the GPU's math library and the host's round
differently in the last place, so it is compared within a relative 1e-7 (stated at each check); which points are lit
at all must agree exactly."""
import os

import numpy as np
import pytest

from raytracer_amd import rtg

import orc
import scenes

pytestmark = pytest.mark.gpu


def _lights(lights):
    return np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE)


def uv_sphere(center, radius, n_lat=10, n_lon=16):
    """A triangle mesh of a sphere with every facet twice: facing outwards and facing inwards.  The reference's
    triangles are one-sided (src/geo/triangle.rs:60), so a ray inside a solid only meets the inward copies."""
    tris = []
    c = np.asarray(center, dtype=np.float64)

    def p(i, j):
        th = np.pi * i / n_lat
        ph = 2.0 * np.pi * (j % n_lon) / n_lon
        return c + radius * np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])

    for i in range(n_lat):
        for j in range(n_lon):
            a, b, cc, d = p(i, j), p(i + 1, j), p(i + 1, j + 1), p(i, j + 1)
            if i != 0:
                tris.append([a, d, b])
            if i != n_lat - 1:
                tris.append([b, d, cc])
    tris = np.array(tris)
    # orient every facet outwards
    n = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 1])
    flip = ((tris.mean(axis=1) - c) * n).sum(axis=1) < 0
    tris[flip] = tris[flip][:, [0, 2, 1]]
    return np.ascontiguousarray(np.concatenate([tris, tris[:, [0, 2, 1]]]))


def test_nearest_photons_reference_cases(gpu_ctx):
    """src/tree/kd_tree.rs:221-268."""
    ph = np.zeros(3, dtype=rtg.PHOTON_DTYPE)
    ph["position"] = [[1.0, 0.0, 0.0], [2.0, 0.0, 0.0], [3.0, 0.0, 0.0]]
    pm = rtg.PhotonMap(gpu_ctx, ph)
    for k in (1, 2, 3):
        idx, dist = pm.nearest([[0.0, 0.0, 0.0]], k)
        assert list(idx[0]) == [0, 1, 2][:k] and list(dist[0]) == [1.0, 2.0, 3.0][:k]
    idx, dist = pm.nearest([[0.0, 0.0, 0.0]], 5)
    assert list(idx[0]) == [0, 1, 2, rtg.RTG_NONE, rtg.RTG_NONE] and np.isinf(dist[0, 3:]).all()
    pm.close()
    for seed in range(1, 21):
        rng = np.random.default_rng(seed)
        ph = np.zeros(100, dtype=rtg.PHOTON_DTYPE)
        ph["position"] = np.round(rng.random((100, 3)) * 1000.0)
        pm = rtg.PhotonMap(gpu_ctx, ph)
        kd = orc.OracleKd(ph["position"])
        q = np.round(rng.random((9, 3)) * 1000.0)
        for k in range(1, 10):
            idx, dist = pm.nearest(q, k)
            bidx, bdist = kd.nearest_brute(q, k)
            assert np.array_equal(dist, bdist)
            unique = np.array([np.unique(r).size == k for r in bdist])
            assert np.array_equal(idx[unique], bidx[unique])
        pm.close()


def test_nearest_photons_every_small_size(gpu_ctx):
    """Trees of 1..40 photons (every shape the level-by-level build meets near its last levels), with repeated
    coordinates, signed zeros and an infinite coordinate, against the brute-force sort."""
    rng = np.random.default_rng(5)
    for n in list(range(1, 41)) + [63, 64, 65, 127, 128, 129, 1000]:
        ph = np.zeros(n, dtype=rtg.PHOTON_DTYPE)
        pos = np.round(rng.random((n, 3)) * 8.0) - 4.0        # many equal coordinates
        pos[pos == 0.0] = rng.choice([0.0, -0.0], size=int((pos == 0.0).sum()))
        if n > 4:
            pos[3, 0] = np.inf
        ph["position"] = pos
        pm = rtg.PhotonMap(gpu_ctx, ph)
        kd = orc.OracleKd(pos)
        q = np.concatenate([rng.random((16, 3)) * 10.0 - 5.0, pos[: min(n, 4)]])
        q = q[np.isfinite(q).all(axis=1)]
        for k in (1, 2, 3, 5, 16):
            idx, dist = pm.nearest(q, k)
            bidx, bdist = kd.nearest_brute(q, k)
            assert np.array_equal(dist, bdist), (n, k)
            found = idx != rtg.RTG_NONE
            assert (found.sum(axis=1) == min(n, k)).all()
            for row, drow, qrow in zip(idx, dist, q):          # every id carries its distance and appears once
                ids = row[row != rtg.RTG_NONE]
                assert np.unique(ids).size == ids.size
                with np.errstate(invalid="ignore"):
                    assert np.array_equal(np.sqrt((((qrow - pos[ids]) ** 2)[:, 0] + ((qrow - pos[ids]) ** 2)[:, 1])
                                                  + ((qrow - pos[ids]) ** 2)[:, 2]), drow[: ids.size])
        pm.close()


def test_tree_finished_by_workgroups_is_the_tree_of_the_level_passes(gpu_ctx):
    """The build hands ranges of at most 1024 photons to one workgroup each (option kd_finish, the default); with the
    option off every level is a device-wide pass.  Same entry order (src/tree/seq_tree.rs:36-49), same answers, bit
    for bit -- sizes around the hand-over, coincident photons, photons crowded on a plane."""
    rng = np.random.default_rng(23)
    for n in (1, 2, 3, 17, 1023, 1024, 1025, 2047, 2048, 2049, 3000, 5000, 70_001, 441_112):
        ph = np.zeros(n, dtype=rtg.PHOTON_DTYPE)
        pos = rng.normal(size=(n, 3))
        if n > 100:
            pos[:, 1] = np.where(rng.random(n) < 0.5, -0.5, pos[:, 1])
            pos[40:90] = pos[40]
            pos[::7] = np.round(pos[::7] * 4.0) / 4.0
        ph["position"] = pos
        q = np.concatenate([rng.normal(size=(2000, 3)), pos[:200]])
        got = []
        for mode in (2, 1):
            gpu_ctx.set_option("kd_finish", mode)
            pm = rtg.PhotonMap(gpu_ctx, ph)
            got.append((pm.order(), *pm.nearest(q, 5)))
            pm.close()
        gpu_ctx.set_option("kd_finish", 2)
        assert np.array_equal(np.sort(got[0][0]), np.arange(n)), n
        for a, b in zip(*got):
            assert a.tobytes() == b.tobytes(), n
        if n <= 5000:          # and the order is a k-d order: every range's middle entry splits it along some axis
            order = got[0][0]
            stack = [(0, n)]
            while stack:
                b, e = stack.pop()
                if e - b < 2:
                    continue
                c = b + (e - b) // 2
                seg = pos[order[b:e]]
                ext = seg.max(axis=0) - seg.min(axis=0)
                axis = 2 - int(np.argmax(ext[::-1]))             # the last of equal maxima
                assert (pos[order[b:c], axis] <= pos[order[c], axis]).all() and (pos[order[c + 1:e], axis] >= pos[order[c], axis]).all(), (n, b, e)
                stack += [(b, c), (c + 1, e)]


def test_nearest_photons_large(gpu_ctx):
    rng = np.random.default_rng(11)
    n = 300_000
    ph = np.zeros(n, dtype=rtg.PHOTON_DTYPE)
    pos = rng.normal(size=(n, 3))
    pos[:, 1] = np.where(rng.random(n) < 0.4, -0.5, pos[:, 1])          # photons crowd on a floor
    pos[5000:5100] = pos[5000]                                          # coincident photons
    ph["position"] = pos
    pm = rtg.PhotonMap(gpu_ctx, ph)
    kd = orc.OracleKd(pos)
    q = np.concatenate([rng.normal(size=(40_000, 3)), pos[:2000] + 1e-9, [[1e6, 1e6, 1e6], [np.inf, 0.0, 0.0]]])
    for k in (1, 3, 4, 7, 16):
        idx, dist = pm.nearest(q, k)
        oidx, odist = kd.nearest(q, k)
        assert np.array_equal(dist.view(np.uint64), odist.view(np.uint64)), k
        same = np.array([np.unique(r).size == k for r in odist])
        assert np.array_equal(idx[same], oidx[same])
        # the ids reported carry the distances reported
        d = np.linalg.norm(pos[idx[:1000].reshape(-1)] - np.repeat(q[:1000], k, axis=0), axis=1).reshape(-1, k)
        assert np.allclose(d, dist[:1000], rtol=1e-14)
    bidx, bdist = kd.nearest_brute(q[:300], 16)
    idx, dist = pm.nearest(q[:300], 16)
    assert np.array_equal(dist, bdist)
    # k beyond the limit is refused, k = 0 and no queries are no-ops
    with pytest.raises(rtg.RtgError):
        pm.nearest(q[:4], 17)
    assert pm.nearest(q[:4], 0)[0].shape == (4, 0)
    assert pm.nearest(np.zeros((0, 3)), 3)[0].shape == (0, 3)
    pm.close()
    with pytest.raises(rtg.RtgError):
        bad = ph[:4].copy()
        bad["position"][2, 1] = np.nan
        rtg.PhotonMap(gpu_ctx, bad)


def _glass_mesh_scene(ctx):
    """[glass sphere mesh, checker floor, diffuse terrain mesh]: photons bounce off mesh leaves and land on a plane
    or on terrain triangles."""
    objs, lights, view = scenes.default_objects(0)
    glass = rtg.mesh_object(None, rtg.material((0.0, 0.0, 0.0), dielectric=(1.0, 1.5)))
    tris, xform = scenes.terrain(12, 12, seed=3)
    ground = rtg.mesh_object(None, rtg.material((0.7, 0.6, 0.5)), xform)
    bv = {0: rtg.HostBvh(uv_sphere((0.0, -0.2, 0.0), 0.2)), 2: rtg.HostBvh(tris)}
    return scenes.Pair(ctx, [glass, objs[1], ground], bv), _lights(lights)


def _photon_fields_equal(a, b):
    assert a.shape == b.shape
    for name in rtg.PHOTON_DTYPE.names:
        if name == "_pad":
            continue
        x, y = a[name], b[name]
        if x.dtype.kind == "f":
            assert not scenes.float_bits_differ(x, y).any(), name
        else:
            assert np.array_equal(x, y), name


def test_photon_pass_matches_the_oracle(gpu_ctx):
    objs, lights, view = scenes.default_objects(0)
    lights = _lights(lights)
    pair = scenes.Pair(gpu_ctx, objs, {})
    dirs = np.stack([rtg.fibonacci_zenith_y(50_000, seed=l + 1) for l in range(lights.shape[0])])
    got = pair.gpu.photons_emit(lights, dirs)
    want = pair.cpu.photons_emit(lights, dirs)
    assert got.shape[0] > 1000
    _photon_fields_equal(got, want)
    # more rays than one batch of the pass holds, through the other traversals as well
    big = np.stack([rtg.fibonacci_zenith_y(600_000, seed=7)])
    want = pair.cpu.photons_emit(lights[:1], big)
    for traversal in (2, 1, 0):
        gpu_ctx.set_option("traversal", traversal)
        _photon_fields_equal(pair.gpu.photons_emit(lights[:1], big), want)
    gpu_ctx.set_option("traversal", 2)
    # too little room: nothing is written, the count still comes back
    with pytest.raises(rtg.RtgError) as err:
        pair.gpu.photons_emit(lights, dirs, capacity=10)
    assert err.value.code == -6
    assert pair.gpu.photons_emit(lights, dirs[:, :0]).shape[0] == 0


def test_photon_pass_with_the_cosines_and_sines_taken_on_the_device(gpu_ctx):
    """Option photon_trig = 2: ZenithY::into_normal (src/geo/sphere.rs:29-34) evaluated by the GPU's own sin / cos.
    The last place of those differs between libraries, so the photons are compared with the host-libm pass (which
    equals the oracle bit for bit) within a tolerance: same paths, positions within 1e-9 of the scene's size -- far
    inside the 1e-4 relative the hit times are allowed.  Directions in device memory give the same bytes as
    directions from the host; host mode still refuses them."""
    import torch
    for build in ("default", "meshes"):
        if build == "default":
            objs, lights, _ = scenes.default_objects(0)
            lights = _lights(lights)
            pair = scenes.Pair(gpu_ctx, objs, {})
        else:
            pair, lights = _glass_mesh_scene(gpu_ctx)
        dirs = np.stack([rtg.fibonacci_zenith_y(300_000, seed=l + 11) for l in range(lights.shape[0])])
        want = pair.gpu.photons_emit(lights, dirs)
        gpu_ctx.set_option("photon_trig", 2)
        got = pair.gpu.photons_emit(lights, dirs)
        room = got.shape[0] + 8
        out = torch.zeros(room * rtg.PHOTON_DTYPE.itemsize, dtype=torch.uint8, device="cuda:0")
        d_dirs = torch.from_numpy(dirs).cuda()
        made = pair.gpu.photons_emit_raw(lights, (d_dirs.data_ptr(), dirs.shape[1]), out.data_ptr(), room)
        gpu_ctx.synchronize()
        gpu_ctx.set_option("photon_trig", 1)
        assert want.shape[0] > 1000 and got.shape[0] == want.shape[0] == made
        assert np.frombuffer(out.cpu().numpy().tobytes(), dtype=rtg.PHOTON_DTYPE)[:made].tobytes() == got.tobytes()
        for name in ("n_bounces", "modes", "light_index", "surface_object", "surface_leaf"):
            assert np.array_equal(got[name], want[name]), name
        assert np.array_equal(got["dir"], want["dir"]) and np.array_equal(got["origin"], want["origin"])
        assert np.abs(got["position"] - want["position"]).max() < 1e-9
        assert np.allclose(got["light"], want["light"], rtol=1e-9, atol=0.0)
        assert (got["position"] != want["position"]).any()          # it is another sin / cos
        with pytest.raises(rtg.RtgError, match="host memory"):
            pair.gpu.photons_emit_raw(lights, (d_dirs.data_ptr(), dirs.shape[1]), out.data_ptr(), room)
        # directions made on the device from the seeds the host generator was given: the same map, photon for photon
        seed = 0x1234
        host_dirs = np.stack([rtg.fibonacci_zenith_y(300_000, seed=(seed + 0x9E3779B97F4A7C15 * l) % 2 ** 64)
                              for l in range(lights.shape[0])])
        gpu_ctx.set_option("photon_trig", 2)
        a = pair.gpu.photon_map(lights, host_dirs)
        gpu_ctx.set_option("photon_trig", 1)
        b = pair.gpu.photon_map_fibonacci(lights, 300_000, seed)
        assert a.n == b.n > 1000
        q = got["position"][::5] + 1e-3
        for k in (1, 3):
            ia, da = a.nearest(q, k)
            ib, db = b.nearest(q, k)
            assert np.array_equal(ia, ib) and np.array_equal(da, db)
        a.close()
        b.close()


def test_photons_stay_on_the_device(gpu_ctx):
    """The photon pass into device memory and a photon map made from that memory give what the host round trip
    gives: same records, same nearest photons."""
    import torch
    objs, lights, view = scenes.default_objects(0)
    scene = gpu_ctx.scene(objs)
    dirs = np.stack([rtg.fibonacci_zenith_y(30_000, seed=3 + l) for l in range(2)])
    host = scene.photons_emit(_lights(lights), dirs)
    assert host.shape[0] > 500
    room = host.shape[0] + 10
    dev = torch.zeros(room * rtg.PHOTON_DTYPE.itemsize, dtype=torch.uint8, device="cuda:0")
    made = scene.photons_emit_raw(_lights(lights), dirs, dev.data_ptr(), room)
    gpu_ctx.synchronize()
    assert made == host.shape[0]
    back = np.frombuffer(dev.cpu().numpy().tobytes(), dtype=rtg.PHOTON_DTYPE)[:made]
    assert back.tobytes() == host.tobytes()
    pm_dev = rtg.PhotonMap(gpu_ctx, dev.data_ptr(), n=made)
    pm_host = rtg.PhotonMap(gpu_ctx, host)
    q = host["position"][::7] + 1e-3
    assert np.array_equal(pm_dev.order(), pm_host.order())
    for k in (1, 3):
        a, b = pm_dev.nearest(q, k), pm_host.nearest(q, k)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # too little room: the count comes back with the error
    with pytest.raises(rtg.RtgError) as err:
        scene.photons_emit_raw(_lights(lights), dirs, dev.data_ptr(), 100)
    assert err.value.code == -6
    # the pass and the tree in one call
    pm_one = scene.photon_map(_lights(lights), dirs)
    assert pm_one.n == made
    for k in (1, 3):
        a, b = pm_one.nearest(q, k), pm_host.nearest(q, k)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    pm_one.close()
    empty = scene.photon_map(_lights(lights), np.zeros((2, 0, 2)))
    assert empty.n == 0 and (empty.nearest(q[:3], 2)[0] == rtg.RTG_NONE).all()
    assert empty.order().shape == (0,)
    empty.close()
    pm_dev.close()
    pm_host.close()


def test_photon_pass_through_meshes(gpu_ctx):
    pair, lights = _glass_mesh_scene(gpu_ctx)
    dirs = np.stack([rtg.fibonacci_zenith_y(100_000, seed=l + 21) for l in range(lights.shape[0])])
    got = pair.gpu.photons_emit(lights, dirs)
    want = pair.cpu.photons_emit(lights, dirs)
    assert got.shape[0] > 500 and (got["surface_leaf"][:, 0] != rtg.RTG_NONE).all()
    _photon_fields_equal(got, want)


def _surface_points(pair, positions, lift=0.05):
    """RaycastPoints of whatever lies straight below `positions` (a camera path's hit points)."""
    rays = np.concatenate([positions + [0.0, lift, 0.0], np.broadcast_to([0.0, -1.0, 0.0], positions.shape)], axis=1)
    _, pts, _ = pair.cpu.trace(rays, points=True)
    return pts


# relative tolerance of the caustic term: the corrected direction goes through sin / cos, which the GPU's math
# library and glibc round differently in the last place; five Newton steps and a derivative later the results agree
# to ~1e-12, 1e-7 leaves room for ill-conditioned photons near a caustic's edge
IRRAD_RTOL = 1e-7


def _compare_irradiance(got, want):
    lit_g, lit_w = got.sum(axis=1) > 0, want.sum(axis=1) > 0
    # a photon that lands within a rounding error of newton_epsilon from the point may count on one side only
    assert (lit_g != lit_w).mean() < 0.002
    both = lit_g & lit_w
    close = np.isclose(got[both], want[both], rtol=IRRAD_RTOL, atol=0.0).all(axis=1)
    # the few points where the two sides kept a different photon of a light differ entirely; everything else is tight
    assert close.mean() > 0.995, (close.mean(), np.abs(got[both] / want[both] - 1)[~close][:5])
    return both.sum()


def test_caustic_term_default_scene(gpu_ctx):
    objs, lights, view = scenes.default_objects(0)
    lights = _lights(lights)
    pair = scenes.Pair(gpu_ctx, objs, {})
    dirs = np.stack([rtg.fibonacci_zenith_y(200_000, seed=l + 5) for l in range(lights.shape[0])])
    photons = pair.gpu.photons_emit(lights, dirs)
    pm = rtg.PhotonMap(gpu_ctx, photons)
    kd = orc.OracleKd(photons["position"])
    # what a frame asks for: the hit points of camera rays (every surface, lit or not)
    hits, pts, _ = pair.gpu.render_primary(view, 96, 96, points=True)
    got = pair.gpu.indirect_irrad(pm, pts)
    want = pair.cpu.indirect_irrad(kd, photons, pts)
    assert _compare_irradiance(got, want) > 50
    assert (got[pts["manifold_len"] == 0] == 0).all()
    # and points right under landed photons, where nearly everything is lit; other sample counts and step counts
    rng = np.random.default_rng(2)
    floor = photons[np.abs(photons["position"][:, 1] + 0.5) < 1e-9]["position"]
    pts = _surface_points(pair, floor[rng.choice(floor.shape[0], 4000)] + rng.normal(scale=1e-4, size=(4000, 3)) * [1, 0, 1])
    for samples, steps, eps in ((3, 5, 1e-5), (1, 5, 1e-5), (8, 3, 1e-4), (16, 0, 1e-3)):
        got = pair.gpu.indirect_irrad(pm, pts, samples, steps, eps)
        want = pair.cpu.indirect_irrad(kd, photons, pts, samples, steps, eps)
        assert _compare_irradiance(got, want) > (1000 if steps else 10), (samples, steps)
    # degenerate requests
    assert (pair.gpu.indirect_irrad(pm, pts[:10], photon_samples=0) == 0).all()
    assert pair.gpu.indirect_irrad(pm, pts[:0]).shape == (0, 3)
    empty = rtg.PhotonMap(gpu_ctx, photons[:0])
    assert (pair.gpu.indirect_irrad(empty, pts[:10]) == 0).all()
    two = rtg.PhotonMap(gpu_ctx, photons[:2])                 # fewer photons than samples
    kd2 = orc.OracleKd(photons[:2]["position"])
    near = _surface_points(pair, photons[:2]["position"])
    _ = pair.gpu.indirect_irrad(two, near, 3, 5, 1e-5)
    assert np.isfinite(_).all()
    assert np.allclose(_, pair.cpu.indirect_irrad(kd2, photons[:2], near, 3, 5, 1e-5), rtol=IRRAD_RTOL, atol=0)
    with pytest.raises(rtg.RtgError):
        pair.gpu.indirect_irrad(pm, pts[:4], photon_samples=17)


def test_caustic_term_through_meshes(gpu_ctx):
    pair, lights = _glass_mesh_scene(gpu_ctx)
    dirs = np.stack([rtg.fibonacci_zenith_y(400_000, seed=l + 31) for l in range(lights.shape[0])])
    photons = pair.gpu.photons_emit(lights, dirs)
    pm = rtg.PhotonMap(gpu_ctx, photons)
    kd = orc.OracleKd(photons["position"])
    rng = np.random.default_rng(4)
    landed = photons[photons["position"][:, 1] < -0.3]["position"]      # below the glass: terrain or floor
    pts = _surface_points(pair, landed[rng.choice(landed.shape[0], 3000, replace=False)] + rng.normal(scale=1e-4, size=(3000, 3)) * [1, 0, 1])
    assert (pts["manifold_len"] == 2).sum() > 100 and (pts["manifold_len"] == 1).sum() > 100
    for traversal in (2, 1):
        gpu_ctx.set_option("traversal", traversal)
        got = pair.gpu.indirect_irrad(pm, pts)
        want = pair.cpu.indirect_irrad(kd, photons, pts)
        assert _compare_irradiance(got, want) > 300
    # device-resident points and output
    import torch
    d_pts = torch.from_numpy(pts.view(np.uint8).reshape(-1)).cuda()
    d_rgb = torch.empty(pts.shape[0] * 3, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    pair.gpu.indirect_irrad_raw(pm, d_pts.data_ptr(), pts.shape[0], 3, 5, 1e-5, d_rgb.data_ptr())
    gpu_ctx.synchronize()
    assert np.array_equal(d_rgb.cpu().numpy().reshape(-1, 3), got)


@pytest.mark.parametrize("seed", range(int(os.environ.get("RTG_FUZZ_SEEDS", "6"))))
def test_photon_paths_in_random_scenes(gpu_ctx, seed):
    """Random scenes -- one to three glass objects (spheres and closed meshes, under random similarity or stretched
    transforms, with various refractive indices), a diffuse floor (plane or mesh), two or three lights anywhere above
    -- through the whole photon path: the photon pass bit for bit, nearest photons, and the caustic term at the
    points the photons themselves landed on (so every combination of reflection and refraction that occurs gets its
    Newton correction)."""
    rng = np.random.default_rng(500 + seed)
    objs, bvhs = [], {}
    for i in range(int(rng.integers(1, 4))):
        centre = rng.uniform(-0.5, 0.5, size=3) * [1.0, 0.3, 1.0] + [0.0, 0.1, 0.0]
        radius = float(rng.uniform(0.1, 0.25))
        m = np.eye(4)
        if rng.random() < 0.5:
            m[:3, :3] = np.diag(rng.uniform(0.7, 1.4, size=3))                      # stretched
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        m[:3, :3] = q @ m[:3, :3]
        m[:3, 3] = rng.normal(scale=0.05, size=3)
        xf = rtg.transform_from_forward(m) if rng.random() < 0.7 else None
        glass = rtg.material((0.0, 0.0, 0.0), dielectric=(1.0, float(rng.choice([1.33, 1.5, 2.4]))))
        if rng.random() < 0.5:
            objs.append(rtg.sphere_object(tuple(centre), radius, glass, xf))
        else:
            objs.append(rtg.mesh_object(None, glass, xf))
            bvhs[len(objs) - 1] = rtg.HostBvh(uv_sphere(tuple(centre), radius, n_lat=6, n_lon=9))
    if rng.random() < 0.5:
        objs.append(rtg.plane_object((0.0, -0.5, 0.0), (0.0, 0.0, -1.0), (-1.0, 0.0, 0.0), rtg.material((1, 1, 1)),
                                     rtg.material((0.1, 0.1, 0.1))))
    else:
        tris, xform = scenes.terrain(10, 10, seed=seed)
        objs.append(rtg.mesh_object(None, rtg.material((0.7, 0.6, 0.5)), xform))
        bvhs[len(objs) - 1] = rtg.HostBvh(tris)
    pair = scenes.Pair(gpu_ctx, objs, bvhs)
    n_lights = int(rng.integers(2, 4))
    lights = np.zeros(n_lights, dtype=rtg.LIGHT_DTYPE)
    lights["position"] = rng.uniform(-1.0, 1.0, size=(n_lights, 3)) * [1.0, 0.3, 1.0] + [0.0, 1.2, 0.0]
    lights["color"] = rng.uniform(10.0, 100.0, size=(n_lights, 3))
    dirs = np.stack([rtg.fibonacci_zenith_y(60_000, seed=seed * 10 + l) for l in range(n_lights)])
    for depth in (1, 2, 3):                                   # shallower path trees: fewer bounces per photon, none at depth 1
        shallow = pair.gpu.photons_emit(lights, dirs[:, :5000], max_depth=depth)
        _photon_fields_equal(shallow, pair.cpu.photons_emit(lights, dirs[:, :5000], max_depth=depth))
        assert (shallow["n_bounces"] < depth).all()
    got = pair.gpu.photons_emit(lights, dirs)
    want = pair.cpu.photons_emit(lights, dirs)
    _photon_fields_equal(got, want)
    if got.shape[0] <= 1000:                                  # the glass caught little light: nothing more to compare
        assert seed >= 6                                      # (never so for the seeds that always run)
        return
    pm = pair.gpu.photon_map(lights, dirs)
    assert pm.n == got.shape[0]
    kd = orc.OracleKd(got["position"])
    q = got["position"][:: max(1, got.shape[0] // 500)] + rng.normal(scale=1e-3, size=(got["position"][:: max(1, got.shape[0] // 500)].shape))
    idx, dist = pm.nearest(q, 3)
    oidx, odist = kd.nearest(q, 3)
    assert np.array_equal(dist.view(np.uint64), odist.view(np.uint64))
    # the caustic term where photons landed on the diffuse floor
    floor = len(objs) - 1
    on_floor = got[(got["n_bounces"] >= 1)]
    take = on_floor["position"][rng.choice(on_floor.shape[0], min(800, on_floor.shape[0]), replace=False)]
    pts = _surface_points(pair, take + rng.normal(scale=2e-4, size=take.shape) * [1, 0, 1], lift=0.02)
    pts = pts[(pts["manifold_len"] > 0)]
    last = np.where(pts["manifold_len"] == 2, pts["manifold"][:, 1], pts["manifold"][:, 0])
    pts = np.ascontiguousarray(pts[last == floor])
    if pts.shape[0] <= 100:
        assert seed >= 6
        return
    g = pair.gpu.indirect_irrad(pm, pts)
    w = pair.cpu.indirect_irrad(kd, got, pts)
    lit_g, lit_w = g.sum(axis=1) > 0, w.sum(axis=1) > 0
    assert (lit_g != lit_w).mean() < 0.01
    both = lit_g & lit_w
    if both.sum() <= 100:
        assert seed >= 6
        return
    close = np.isclose(g[both], w[both], rtol=IRRAD_RTOL, atol=0.0).all(axis=1)
    assert close.mean() > 0.98, (seed, close.mean())
    # a small frame from above with the photon term: the plain frame where no photon counts, the oracle's caustic elsewhere
    view = rtg.make_view(orig=(0.0, 0.3, 1.6), fov=(1.2, 1.0))
    fw, fh = 56, 40
    f_got, _ = pair.gpu.render_frame_photons(view, lights, pm, fw, fh, rgba8=False)
    f_want, _ = pair.cpu.render_frame_photons(view, lights, kd, got, fw, fh)
    f_plain, _, _ = pair.gpu.render_frame(view, lights, fw, fh, hits=False)
    caustic = (f_want != f_plain).any(axis=1)
    assert not scenes.float_bits_differ(f_got[~caustic], f_plain[~caustic]).any()
    if caustic.sum() > 20:
        ok = np.isclose(f_got[caustic], f_want[caustic], rtol=IRRAD_RTOL, atol=1e-300).all(axis=1)
        assert ok.mean() > 0.97, (seed, ok.mean(), int(caustic.sum()))
    pm.close()


def test_frame_with_the_photon_term(gpu_ctx):
    """raytrace_pixel with photons (src/render/renderer.rs:266-283): pixels that no photon reaches are bit-identical
    to the photon-less frame and to the oracle; pixels in the caustic agree within IRRAD_RTOL."""
    for build in ("default", "meshes"):
        if build == "default":
            objs, lights, view = scenes.default_objects(0)
            lights = _lights(lights)
            pair = scenes.Pair(gpu_ctx, objs, {})
        else:
            pair, lights = _glass_mesh_scene(gpu_ctx)
            view = scenes.default_objects(0)[2]
        dirs = np.stack([rtg.fibonacci_zenith_y(200_000, seed=l + 41) for l in range(lights.shape[0])])
        photons = pair.gpu.photons_emit(lights, dirs)
        pm = rtg.PhotonMap(gpu_ctx, photons)
        kd = orc.OracleKd(photons["position"])
        w = h = 80
        got, got8 = pair.gpu.render_frame_photons(view, lights, pm, w, h)
        want, want8 = pair.cpu.render_frame_photons(view, lights, kd, photons, w, h)
        plain, _, _ = pair.gpu.render_frame(view, lights, w, h, hits=False)
        caustic = (want != plain).any(axis=1)
        assert caustic.sum() > 100, build
        # outside the caustic: the plain frame, bit for bit
        assert not scenes.float_bits_differ(got[~caustic], plain[~caustic]).any()
        close = np.isclose(got[caustic], want[caustic], rtol=IRRAD_RTOL, atol=1e-300).all(axis=1)
        assert close.mean() > 0.99, (build, close.mean())
        assert (np.abs(got8.astype(int) - want8.astype(int)).max(axis=1) <= 1).mean() > 0.995
        # a tile of the frame is the frame's pixels
        tile = (17, 9, 40, 33)
        t_rgb, _ = pair.gpu.render_frame_photons(view, lights, pm, w, h, tile=tile)
        sub = got.reshape(h, w, 3)[tile[1]:tile[1] + tile[3], tile[0]:tile[0] + tile[2]].reshape(-1, 3)
        assert not scenes.float_bits_differ(t_rgb, sub).any()
        # an empty map gives the plain frame
        empty = rtg.PhotonMap(gpu_ctx, photons[:0])
        e_rgb, _ = pair.gpu.render_frame_photons(view, lights, empty, w, h)
        assert not scenes.float_bits_differ(e_rgb, plain).any()


def test_renderer_with_its_photon_pass(gpu_ctx):
    """The host Renderer with the photon pass switched on (Renderer::render, src/render/renderer.rs:115-154) against
    the same steps taken one by one on the oracle: directions, photon pass, k-d tree, pixel pass."""
    time, w, h, count, seed = 7, 72, 72, 150_000, 5
    got, n_photons, seconds = rtg.render_scene_builder_photons(gpu_ctx, time, w, h, count, seed=seed, photons_on_host=True)
    objs, lights, view = scenes.default_objects(time)
    lights = _lights(lights)
    pair = scenes.Pair(gpu_ctx, objs, {})
    dirs = np.stack([rtg.fibonacci_zenith_y(count, seed=(seed + 0x9E3779B97F4A7C15 * l) % 2 ** 64) for l in range(2)])
    photons = pair.cpu.photons_emit(lights, dirs)
    assert n_photons == photons.shape[0] and n_photons > 1000
    kd = orc.OracleKd(photons["position"])
    want, _ = pair.cpu.render_frame_photons(view, lights, kd, photons, w, h)
    plain, _ = rtg.render_scene_builder(gpu_ctx, time, w, h)
    caustic = (want != plain).any(axis=1)
    assert caustic.sum() > 50
    assert not scenes.float_bits_differ(got[~caustic], plain[~caustic]).any()
    close = np.isclose(got[caustic], want[caustic], rtol=IRRAD_RTOL, atol=1e-300).all(axis=1)
    assert close.mean() > 0.99, close.mean()
    assert all(x >= 0.0 for x in seconds)
    # the whole pass on the GPU (the Renderer's default): same directions, the GPU's sin / cos -- the same photons
    # up to the last places, so the same picture within the caustic's tolerance
    dev, n_dev, _ = rtg.render_scene_builder_photons(gpu_ctx, time, w, h, count, seed=seed)
    assert n_dev == n_photons
    assert not scenes.float_bits_differ(dev[~caustic], plain[~caustic]).any()
    assert np.isclose(dev[caustic], want[caustic], rtol=IRRAD_RTOL, atol=1e-300).all(axis=1).mean() > 0.99


def test_plain_c_caller_renders_the_default_frame(gpu_ctx, tmp_path):
    """examples/default_scene.c -- C11, include/rtg.h and librtg.so only -- renders the reference's default frame with
    its photon pass; given the same directions, the Python harness gets the same doubles."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "default_scene")
    if not os.path.exists(exe):                              # normally built by __graft_entry__.build(); a second of gcc otherwise
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    time, w, h, count = 11, 90, 70, 40_000
    dirs = np.stack([rtg.fibonacci_zenith_y(count, seed=77 + l) for l in range(2)])
    dirs_file, out_file = tmp_path / "dirs.f64", tmp_path / "frame.f64"
    dirs.tofile(dirs_file)
    done = subprocess.run([exe, str(w), str(h), str(count), str(time), str(out_file), str(dirs_file)], capture_output=True, text=True,
                          timeout=120)
    assert done.returncode == 0, done.stderr
    assert "photons from 2 x 40000 directions" in done.stdout
    got = np.fromfile(out_file, dtype=np.float64).reshape(-1, 3)
    objs, lights, view = scenes.default_objects(time)
    scene = gpu_ctx.scene(objs)
    pm = scene.photon_map(_lights(lights), dirs)
    want, _ = scene.render_frame_photons(view, _lights(lights), pm, w, h, rgba8=False)
    plain, _, _ = scene.render_frame(view, _lights(lights), w, h, hits=False)
    assert not scenes.float_bits_differ(got, want).any()
    assert (want != plain).any(axis=1).sum() > 20             # the caustic is in the picture
    # an output name ending in .ppm gets the same frame as an 8-bit picture through the sRGB curve
    ppm_file = tmp_path / "frame.ppm"
    done = subprocess.run([exe, str(w), str(h), str(count), str(time), str(ppm_file), str(dirs_file)], capture_output=True, text=True,
                          timeout=120)
    assert done.returncode == 0, done.stderr
    raw = ppm_file.read_bytes()
    header = b"P6\n%d %d\n255\n" % (w, h)
    assert raw.startswith(header) and len(raw) == len(header) + w * h * 3
    linear = np.clip(want, 0.0, 1.0)
    levels = np.where(linear <= 0.0031308, 12.92 * linear, 1.055 * linear ** (1.0 / 2.4) - 0.055) * 255.0
    picture = np.frombuffer(raw[len(header):], dtype=np.uint8).reshape(-1, 3).astype(np.float64)
    assert np.abs(picture - levels).max() <= 0.5 + 1e-6
    pm.close()
    scene.close()

