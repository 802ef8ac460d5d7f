#!/usr/bin/env python3
"""A frame of the default scene with the photon term, for kernel traces: photon map from 2 x 1 000 000 directions,
then `frames` frames at width x height.  Usage: tools/frame_photons_prof.py [width (1920)] [height (1080)] [frames (5)]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import rtg  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = rtg.Context(0)
objs, lights, view = rtg.scene_builder(0)
lights = np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE)
scene = ctx.scene(objs)
dirs = np.stack([rtg.fibonacci_zenith_y(1_000_000, seed=l + 1) for l in range(2)])
pm = scene.photon_map(lights, dirs)
print(f"{pm.n} photons", flush=True)
scene.render_frame_photons(view, lights, pm, W, H)
ctx.synchronize()
for _ in range(frames):
    t = time.perf_counter()
    scene.render_frame_photons(view, lights, pm, W, H)
    ctx.synchronize()
    print(f"frame {W}x{H} with the photon term: {1e3 * (time.perf_counter() - t):.2f} ms", flush=True)
t = time.perf_counter()
scene.render_frame(view, lights, W, H, hits=False)
ctx.synchronize()
print(f"frame {W}x{H} without: {1e3 * (time.perf_counter() - t):.2f} ms", flush=True)
