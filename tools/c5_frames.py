#!/usr/bin/env python3
"""Renders config 5's frame a few times and nothing else (GPU box): the program to put under
`rocprofv3 --kernel-trace --stats` for the per-kernel split of a shaded 3840x2160 frame.
    python tools/c5_frames.py [frames [option=value ...]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracer_amd import rtg
import bench

W, H = 3840, 2160
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
tris = rtg.heightfield(*bench.GRID, bench.SEED, bench.LOCAL_SCALE)
bvh = rtg.HostBvh(tris)
ctx = rtg.Context(0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
mesh = ctx.mesh(bvh)
scene = ctx.scene([rtg.mesh_object(mesh, rtg.material((0.8, 0.7, 0.6)), rtg.transform(1.0 / bench.LOCAL_SCALE))])
rgba = torch.empty(W * H * 4, dtype=torch.uint8, device="cuda")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for f in range(n_frames + 1):
    if f == 1:
        a.record(stream)
    _, lights, view = rtg.scene_builder(f)
    lights = np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE)
    scene.render_frame_raw(view, lights, len(lights), W, H, (0, 0, W, H), 4, None, rgba.data_ptr(), None)
b.record(stream); torch.cuda.synchronize()
print(f"{n_frames} frames {W}x{H}: {a.elapsed_time(b) / n_frames:.2f} ms per frame")
