#!/usr/bin/env python3
"""The bench frame's camera rays traced as a ray batch (rtg_trace, device memory) against the same frame through
rtg_render_primary: what a caller that keeps its own pixel loop gets.  Usage: tools/batch_vs_primary.py [key=value ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from raytracer_amd import rtg

ctx = rtg.Context(0)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
st = torch.cuda.Stream()
ctx.set_stream(st.cuda_stream)
torch.cuda.set_stream(st)
tris = rtg.heightfield(*bench.GRID, bench.SEED, bench.LOCAL_SCALE)
mesh = ctx.mesh(rtg.HostBvh(tris))
scene = ctx.scene([rtg.mesh_object(mesh, rtg.material((1, 1, 1)), rtg.transform(1.0 / bench.LOCAL_SCALE))])
W, H = bench.WIDTH, bench.HEIGHT
n = W * H
view = bench.frame_view(rtg, 0)
hits_a = torch.empty(n * 16, dtype=torch.uint8, device="cuda")
hits_b = torch.empty(n * 16, dtype=torch.uint8, device="cuda")
scene.render_primary_raw(view, W, H, (0, 0, W, H), hits_a.data_ptr())
# the same rays as a batch: View::get_ray in numpy (division order as in the reference)
x = (np.arange(W) - (W - 1) / 2.0) / W
y = -(np.arange(H) - (H - 1) / 2.0) / H
dx = np.broadcast_to(x[None, :] * view.tan_half_fov[0], (H, W))
dy = np.broadcast_to(y[:, None] * view.tan_half_fov[1], (H, W))
dz = -np.ones((H, W))
length = np.sqrt(((0.0 + dx * dx) + dy * dy) + dz * dz)
rays = np.empty((H, W, 6))
rays[..., 0:3] = list(view.orig)
rays[..., 3], rays[..., 4], rays[..., 5] = dx / length, dy / length, dz / length
d_rays = torch.from_numpy(np.ascontiguousarray(rays.reshape(-1, 6))).cuda()
scene.trace_raw(d_rays.data_ptr(), n, hits_b.data_ptr())
torch.cuda.synchronize()
same = bool((hits_a == hits_b).all().item())


def timed(fn, reps=20):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record(st)
    for _ in range(reps):
        fn()
    b.record(st)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


t_primary = timed(lambda: scene.render_primary_raw(view, W, H, (0, 0, W, H), hits_a.data_ptr()))
t_batch = timed(lambda: scene.trace_raw(d_rays.data_ptr(), n, hits_b.data_ptr()))
print(f"frame {W}x{H}: rtg_render_primary {t_primary:.3f} ms, the same rays through rtg_trace {t_batch:.3f} ms, identical hits: {same}")
