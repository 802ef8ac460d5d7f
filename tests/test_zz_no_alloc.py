"""GPU: rtg_render_primary_static -- the primary-ray entry point that allocates nothing -- against
rtg_render_primary, through the C ABI, bit for bit (hit object, leaf and time), on two calls in a row, the second
with a smaller input, so whatever the first call left in the shared per-GPU scratch must not leak into the second."""
import numpy as np
import pytest

import scenes
from raytracer_amd import rtg

pytestmark = pytest.mark.gpu


def _scene(ctx):
    """Sphere + checker plane + mesh: the kernels use both slices of the scratch (work counters and the closest
    sphere / plane hit per pixel)."""
    tris, xform = scenes.terrain(60, 60, seed=7)
    objs, _, view = scenes.mesh_scene_objects(xform)
    mesh = ctx.mesh(rtg.HostBvh(tris))
    objs[2].mesh = mesh.handle
    return ctx.scene(objs), view, mesh


def _static(scene, torch, view, w, h, tile):
    n = tile[2] * tile[3]
    out = torch.full((n * 16,), 0xAB, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    scene.render_primary_static(view, w, h, tile, out.data_ptr())
    scene.ctx.synchronize()
    return np.frombuffer(out.cpu().numpy().tobytes(), dtype=rtg.HIT_DTYPE)


def test_static_entry_point_equals_render_primary_twice_in_a_row(gpu_ctx):
    import torch
    scene, view, mesh = _scene(gpu_ctx)
    first = (640, 400, (0, 0, 640, 400))
    second = (640, 400, (229, 171, 203, 99))            # smaller, ragged: not a multiple of the 8x8 wave tile
    for w, h, tile in (first, second):
        got = _static(scene, torch, view, w, h, tile)
        want, _, _ = scene.render_primary(view, w, h, tile=tile)
        bad = scenes.hit_mismatches(got, want)
        assert bad.size == 0, f"tile {tile}: {bad.size} pixels differ, first {bad[:5]}"
        # the glass sphere wins pixels over the mesh (merge of the sphere / plane slice) and the mesh wins the rest
        assert set(np.unique(want["object"]).tolist()) >= {0, 2}, tile
    # a mesh-only scene (no sphere / plane slice in use) and a scene without any mesh
    tris, xform = scenes.terrain(30, 30, seed=3)
    m2 = gpu_ctx.mesh(rtg.HostBvh(tris))
    only_mesh = gpu_ctx.scene([rtg.mesh_object(m2, rtg.material((1, 1, 1)), xform)])
    objs, _, _ = scenes.default_objects(0)
    no_mesh = gpu_ctx.scene(objs)
    for sc in (only_mesh, no_mesh):
        got = _static(sc, torch, view, 320, 200, (0, 0, 320, 200))
        want, _, _ = sc.render_primary(view, 320, 200)
        assert scenes.hit_mismatches(got, want).size == 0


def test_static_entry_point_leaves_device_memory_as_it_is(gpu_ctx):
    """What the entry point is for: calls on a context that has never rendered through the allocating entry points
    (no scratch buffer of its own yet) leave the GPU's free memory exactly where it was, for frames of several sizes,
    while the same frame through rtg_render_primary makes that context allocate its scratch."""
    import torch
    fresh = rtg.Context(0)
    tris, xform = scenes.terrain(40, 40, seed=5)
    objs, _, view = scenes.mesh_scene_objects(xform)
    mesh = fresh.mesh(rtg.HostBvh(tris))
    objs[2].mesh = mesh.handle
    scene = fresh.scene(objs)
    out = torch.empty(1024 * 768 * 16, dtype=torch.uint8, device="cuda")
    # the first launch on a new stream lets the HIP runtime set up that queue (the kernel's private segment among
    # other things: 40 MiB on this GPU) -- the runtime's doing, once per queue, and no call of the library's
    scene.render_primary_static(view, 64, 64, (0, 0, 64, 64), out.data_ptr())
    fresh.synchronize()
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for w, h in ((1024, 768), (333, 111), (1024, 768)):
        scene.render_primary_static(view, w, h, (0, 0, w, h), out.data_ptr())
    fresh.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    scene.render_primary_raw(view, 1024, 768, (0, 0, 1024, 768), out.data_ptr())      # the allocating path, for contrast
    fresh.synchronize()
    assert torch.cuda.mem_get_info()[0] < free_before
    scene.close()
    mesh.close()
    fresh.close()


def test_static_entry_point_limits(gpu_ctx):
    import torch
    scene, view, mesh = _scene(gpu_ctx)
    out = torch.empty(16 * 64, dtype=torch.uint8, device="cuda")
    with pytest.raises(rtg.RtgError) as e:                              # one pixel row beyond the fixed scratch
        scene.render_primary_static(view, 2048, 1025, (0, 0, 2048, 1025), out.data_ptr())
    assert e.value.code == -6 and "exceed" in str(e.value)
    with pytest.raises(rtg.RtgError) as e:                              # host output would need a staging buffer
        scene.render_primary_static(view, 8, 8, (0, 0, 8, 8), np.empty(64, dtype=rtg.HIT_DTYPE))
    assert e.value.code == -1 and "device memory" in str(e.value)
    with pytest.raises(rtg.RtgError):
        scene.render_primary_static(view, 8, 8, (4, 4, 8, 8), out.data_ptr())
    scene.render_primary_static(view, 8, 8, (3, 3, 0, 2), out.data_ptr())   # empty tile: nothing to do
