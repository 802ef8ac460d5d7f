"""GPU: handles give back what they took.  Contexts, meshes (host-flattened and device-built), scenes, photon maps and
communicators are created, used and destroyed over and over; the device memory in use afterwards is what it was
before, and the process has not grown (the reference leans on Rust's ownership for this: `Arc<Bvh>`, `Renderer`
dropping its `KdTree`)."""
import numpy as np
import pytest

import scenes
from raytracer_amd import rtg

pytestmark = pytest.mark.gpu


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def _one_round(tris, rays, lights, dirs, view):
    ctx = rtg.Context(0)
    host = rtg.HostBvh(tris)
    flattened = ctx.mesh(host)
    built = ctx.build_mesh(tris)
    glass = rtg.sphere_object((0.0, 0.1, 0.0), 0.2, rtg.material((0.0, 0.0, 0.0), dielectric=(1.0, 1.5)))
    scene = ctx.scene([glass, rtg.mesh_object(flattened, rtg.material((0.5, 0.5, 0.5)), rtg.transform(0.01)),
                       rtg.mesh_object(built, rtg.material((0.5, 0.5, 0.5)), rtg.transform(0.01, (0.0, -0.3, 0.0)))])
    scene.render_primary(view, 160, 120, points=True, counters=True)
    scene.trace(rays, points=True)
    scene.render_frame(view, lights, 96, 64)
    scene.trace_paths(rays[:2000])
    pm = scene.photon_map(lights, dirs)
    scene.render_frame_photons(view, lights, pm, 64, 48)
    photons = scene.photons_emit(lights, dirs)
    pm2 = rtg.PhotonMap(ctx, photons)
    pm2.nearest(photons["position"][:100], 3)
    comm = rtg.Comm(ctx)
    with pytest.raises(rtg.RtgError):                      # a failing call in the middle must not strand anything either
        scene.photons_emit(lights, dirs, capacity=1)
    comm.close()
    pm2.close()
    pm.close()
    scene.close()
    built.close()
    flattened.close()
    host.close()
    ctx.close()


def test_create_use_destroy_leaves_no_device_memory_behind():
    tris, _ = scenes.terrain(40, 30, seed=2)
    rng = np.random.default_rng(3)
    rays = np.concatenate([rng.uniform(-1, 1, size=(20000, 3)) + [0, 0.5, 0], rng.normal(size=(20000, 3))], axis=1)
    _, lights, view = scenes.default_objects(0)
    lights = np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE)
    dirs = np.stack([rtg.fibonacci_zenith_y(20_000, seed=l + 1) for l in range(2)])
    import gc
    import psutil
    for _ in range(2):                                     # first uses: the runtime's own one-time allocations
        _one_round(tris, rays, lights, dirs, view)
    gc.collect()
    before, host_before = _free_bytes(), psutil.Process().memory_info().rss
    for _ in range(12):
        _one_round(tris, rays, lights, dirs, view)
    gc.collect()
    after, host_after = _free_bytes(), psutil.Process().memory_info().rss
    assert before - after < 32 * 1024 * 1024, f"{(before - after) / 1e6:.1f} MB of device memory not returned after 12 rounds"
    assert host_after - host_before < 256 * 1024 * 1024, f"host memory grew by {(host_after - host_before) / 1e6:.1f} MB over 12 rounds"


def test_two_contexts_driven_from_two_host_threads():
    """Distinct contexts may be driven from distinct host threads (include/rtg.h): two threads, each with its own
    context and scene over one shared mesh, render different frames at the same time; every frame equals the one a
    single thread renders."""
    import threading
    tris, xform = scenes.terrain(60, 45, seed=4)
    host = rtg.HostBvh(tris)
    main = rtg.Context(0)
    mesh = main.mesh(host)
    obj = rtg.mesh_object(mesh, rtg.material((0.5, 0.5, 0.5)), xform)
    views = [rtg.make_view(orig=(0.1 * i - 0.4, 0.2, 1.0 + 0.05 * i)) for i in range(8)]
    lights = np.ascontiguousarray(scenes.default_objects(3)[1], dtype=rtg.LIGHT_DTYPE)
    w, h = 320, 200
    scene = main.scene([obj])
    want = [(scene.render_primary(v, w, h)[0], scene.render_frame(v, lights, w, h)[0]) for v in views]
    failures = []

    def worker(which):
        try:
            ctx = rtg.Context(0)
            sc = ctx.scene([obj])
            for _ in range(6):
                for i in range(which, len(views), 2):
                    hits = sc.render_primary(views[i], w, h)[0]
                    rgb = sc.render_frame(views[i], lights, w, h)[0]
                    if hits.tobytes() != want[i][0].tobytes() or scenes.float_bits_differ(rgb, want[i][1]).any():
                        failures.append((which, i))
            sc.close()
            ctx.close()
        except Exception as e:                                # noqa: BLE001 -- reported by the assertion below
            failures.append((which, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert failures == []
    scene.close()
    mesh.close()
    main.close()
