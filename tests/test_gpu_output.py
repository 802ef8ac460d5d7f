"""GPU: a frame rendered on the device, taken through the host mirror of Renderer::images (render/renderer.rs:379-385)
and the server's tar framing (bin/server.rs:19-30), decoded again and compared with the frame."""
import io
import math
import tarfile

import numpy as np
import pytest

from raytracer_amd import rtg

import scenes
from test_output_formats import _decode_hdr

pytestmark = pytest.mark.gpu


def _pq(x):
    m1, m2 = 1305.0 / 8192.0, 2523.0 / 32.0
    c1, c2, c3 = 107.0 / 128.0, 2413.0 / 128.0, 2392.0 / 128.0
    return ((c1 + c2 * x ** m1) / (1.0 + c3 * x ** m1)) ** m2


@pytest.mark.parametrize("with_default", [False, True])
def test_rendered_frame_through_images_and_tar(gpu_ctx, with_default):
    """Renderer::render + images() on the GPU, then tar -> decode: image.hdr is the frame's radiance within RGBE
    precision, image.pq.hdr its PQ encoding, perf.hdr the frame's time shared out by the camera rays' visit counters
    (which equal the oracle's).  The radiance itself is bit-identical to the frame entry point's and the oracle's."""
    ctx = gpu_ctx
    w, h = 200, 120
    tris, xform = scenes.terrain(60, 40, seed=5)
    bvh = rtg.HostBvh(tris)
    objs, lights, view = scenes.mesh_scene_objects(xform, with_default=with_default, time=13)
    pair = scenes.Pair(ctx, objs, {len(objs) - 1: bvh})
    rgb, hits, cost, seconds, blob = rtg.render_images(ctx, pair.gpu._objects, lights, (0.0, 0.0, 1.0),
                                                       (math.pi * 0.5, math.pi * 0.5), w, h, max_depth=4)
    g_rgb, _, g_hits = pair.gpu.render_frame(view, lights, w, h, max_depth=4)
    c_rgb, _, c_hits = pair.cpu.render_frame(view, lights, w, h, max_depth=4)
    assert not scenes.float_bits_differ(rgb, g_rgb.reshape(-1, 3)).any()
    assert not scenes.float_bits_differ(rgb, c_rgb.reshape(-1, 3)).any()
    assert scenes.hit_mismatches(hits, c_hits.reshape(-1)).size == 0
    _, _, c_ctr = pair.cpu.render_primary(view, w, h, counters=True)
    c_cost = (c_ctr["nodes_visited"].astype(np.int64) + c_ctr["tris_tested"]).reshape(-1)
    assert np.array_equal(cost.astype(np.int64), c_cost) and cost.max() > 20
    assert seconds > 0.0

    assert len(blob) % 512 == 0 and blob[-1024:] == bytes(1024)
    with tarfile.open(fileobj=io.BytesIO(blob)) as tar:
        members = {m.name: m for m in tar.getmembers()}
        assert sorted(members) == ["image.hdr", "image.pq.hdr", "perf.hdr"]
        image = _decode_hdr(tar.extractfile(members["image.hdr"]).read())
        pq = _decode_hdr(tar.extractfile(members["image.pq.hdr"]).read())
        perf = _decode_hdr(tar.extractfile(members["perf.hdr"]).read())
        assert tar.extractfile(members["image.hdr"]).read() == rtg.to_hdr(rgb, w, h)
    frame = rgb.reshape(h, w, 3)
    assert frame.max() > 0.0
    peak = frame.max(axis=2, keepdims=True)
    assert np.all(np.abs(image - frame) <= peak / 128.0 + 1e-300)
    want_pq = _pq(frame / 80.0)
    assert np.all(np.abs(pq - want_pq) <= want_pq.max(axis=2, keepdims=True) / 128.0 + 1e-12)
    share = (1.0 + c_cost) / (1.0 + c_cost).sum()
    want_perf = (seconds * share * 3000.0).reshape(h, w, 1)
    assert np.all(np.abs(perf - want_perf) <= want_perf / 128.0)
    # pixels whose ray crosses the terrain cost more than sky pixels
    sky = c_hits.reshape(-1)["object"] == rtg.RTG_NONE
    if sky.any() and (~sky).any():
        assert perf.reshape(-1, 3)[~sky, 0].mean() > perf.reshape(-1, 3)[sky, 0].mean()
