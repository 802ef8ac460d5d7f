#!/usr/bin/env python3
"""Configuration 3 as a short program for rocprofv3: the trispline-patch scene (15 500 patches, 992 000 facets),
3840x2160 primary rays, `launches` timed launches of the default traversal after one untimed one.
Usage: tools/c3_prof.py [launches (3)]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raytracer_amd import rtg

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 3
W, H = 3840, 2160
pm = rtg.HostPatchMesh(rtg.patch_terrain(125, 62, seed=7, scale=100.0), k=8, newton_steps=5)
ctx = rtg.Context(0)
mesh = ctx.mesh(pm)
scene = ctx.scene([rtg.patch_object(mesh, rtg.material((1, 1, 1)), rtg.transform(0.01))])
view = rtg.make_view()
hits = torch.empty(W * H * 16, dtype=torch.uint8, device="cuda")
for i in range(launches + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scene.render_primary_raw(view, W, H, (0, 0, W, H), hits.data_ptr())
    ctx.synchronize()
    print(f"launch {i}: {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
