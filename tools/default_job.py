#!/usr/bin/env python3
"""The reference's default job (src/bin/local.rs:40-50 over src/lib.rs:109-120): frames of the glass sphere over the
checker floor at 300x300, two lights orbiting with `time`, 10 000 000 photon directions per light, 3 photon samples
and 5 Newton steps per path point; every frame is written as the reference's three images.  Per frame: directions
(the caller's side in the reference: Sphere::fibonacci_sphere), photon pass + k-d tree, pixel pass, image encoding.
The CPU side is the oracle on a reduced photon count (its photon pass is linear in the count).
Usage: tools/default_job.py [frames (default 5)] [photons per light (default 10000000)] [oracle photons per light (default 200000)]
       [host: photon directions and their sin / cos on the host instead of the GPU] [batch=N: option photon_batch]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracer_amd import rtg  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 5
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
oracle_count = int(sys.argv[3]) if len(sys.argv) > 3 else 200_000
W = H = 300

ctx = rtg.Context(0)
for a in sys.argv[4:]:
    if a.startswith("batch="):
        ctx.set_option("photon_batch", int(a[6:]))
ON_HOST = "host" in sys.argv[4:]              # directions and their sin / cos on the host (bit-exact with the oracle)
rtg.render_scene_builder_photons(ctx, 0, W, H, min(count, 1_000_000), photons_on_host="host" in sys.argv[4:])          # warm-up: scratch buffers, first launches
rows = []
for frame in range(frames):
    t0 = time.perf_counter()
    rgb, n_photons, (s_dirs, s_photons, s_pixels) = rtg.render_scene_builder_photons(ctx, frame, W, H, count, seed=frame + 1,
                                                                                     photons_on_host=ON_HOST)
    t1 = time.perf_counter()
    tar = rtg.images_tar(rgb, W, H, seconds=s_pixels)
    t2 = time.perf_counter()
    rows.append((t2 - t0, s_dirs, s_photons, s_pixels, t2 - t1))
    print(f"frame {frame}: {n_photons} photons  total {1e3 * (t2 - t0):8.1f} ms = directions {1e3 * s_dirs:7.1f} + photon pass and tree "
          f"{1e3 * s_photons:7.1f} + pixels {1e3 * s_pixels:6.1f} + images {1e3 * (t2 - t1):5.1f} ({len(tar)} B)", flush=True)
r = np.median(np.array(rows), axis=0)
print(f"median of {frames} frames: {1e3 * r[0]:.1f} ms per frame = {1 / r[0]:.2f} frames/s; without the directions "
      f"{1e3 * (r[0] - r[1]):.1f} ms = {1 / (r[0] - r[1]):.2f} frames/s; {2 * count / r[2] / 1e6:.1f} M photon directions/s in the photon pass")

# The job as a sequence: frame f's images are encoded on a host thread while the GPU renders frame f + 1 (frames are
# independent in the reference too: src/lib.rs renders them from a parallel iterator).
from concurrent.futures import ThreadPoolExecutor  # noqa: E402
with ThreadPoolExecutor(1) as pool:
    t0 = time.perf_counter()
    pending = None
    sizes = []
    for frame in range(frames):
        rgb, _, (_, _, s_pixels) = rtg.render_scene_builder_photons(ctx, frame, W, H, count, seed=frame + 1, photons_on_host=ON_HOST)
        if pending is not None:
            sizes.append(len(pending.result()))
        pending = pool.submit(rtg.images_tar, rgb, W, H, s_pixels)
    sizes.append(len(pending.result()))
    seq = time.perf_counter() - t0
print(f"{frames} frames as a sequence, images encoded while the next frame renders: {1e3 * seq / frames:.1f} ms per frame = "
      f"{frames / seq:.2f} frames/s ({sum(sizes)} B of images)")

if oracle_count:
    import orc  # noqa: E402
    import scenes  # noqa: E402
    objs, lights, view = scenes.default_objects(0)
    lights = np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE)
    cpu = orc.OracleScene(objs)
    dirs = np.stack([rtg.fibonacci_zenith_y(oracle_count, seed=1 + l) for l in range(2)])
    t0 = time.perf_counter()
    photons = cpu.photons_emit(lights, dirs)                 # one thread (the oracle's photon pass is a plain loop)
    t1 = time.perf_counter()
    kd = orc.OracleKd(photons["position"])
    t2 = time.perf_counter()
    cpu.render_frame_photons(view, lights, kd, photons, W, H)
    t3 = time.perf_counter()
    scale = count / oracle_count
    est = (t1 - t0) * scale + (t2 - t1) * scale + (t3 - t2)
    ideal = (t1 - t0) * scale / orc.THREADS + (t2 - t1) * scale + (t3 - t2)
    print(f"oracle, {orc.THREADS} threads for the pixel pass, 1 for the photon pass: {2 * oracle_count} directions {1e3 * (t1 - t0):.0f} ms "
          f"({2 * oracle_count / (t1 - t0) / 1e6:.2f} M/s), tree of {photons.shape[0]} photons {1e3 * (t2 - t1):.0f} ms, pixel pass "
          f"{1e3 * (t3 - t2):.0f} ms; scaled to {2 * count} directions: about {est:.1f} s per frame "
          f"({ideal:.2f} s if the photon pass were spread over {orc.THREADS} threads without loss)")
