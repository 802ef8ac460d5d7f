#!/usr/bin/env python3
"""Times rtg_render_frame (primary + direct light + specular tree + PQ) on the bench mesh (GPU box).
    python tools/frame_time.py [W H [traversal [option=value ...]]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from raytracer_amd import rtg
import bench, scenes

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
TRAVERSAL = int(sys.argv[3]) if len(sys.argv) > 3 else 2
tris = rtg.heightfield(1000, 500, bench.SEED, bench.LOCAL_SCALE)
bvh = rtg.HostBvh(tris)
ctx = rtg.Context(0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
ctx.set_option("traversal", TRAVERSAL)
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
mesh = ctx.mesh(bvh)
_, lights, view = scenes.default_objects(0)
mesh_obj = rtg.mesh_object(mesh, rtg.material((0.8, 0.7, 0.6)), rtg.transform(1.0 / bench.LOCAL_SCALE))
default_objs, _, _ = scenes.default_objects(0)      # glass sphere + checker plane
for label, objs in (("mesh only", [mesh_obj]), ("glass sphere + plane + mesh", list(default_objs) + [mesh_obj]),
                    ("glass sphere + plane", list(default_objs))):
    scene = ctx.scene(objs)
    rgba = torch.empty(W * H * 4, dtype=torch.uint8, device="cuda")
    def run():
        scene.render_frame_raw(view, np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE), len(lights), W, H, (0, 0, W, H), 4, None, rgba.data_ptr(), None)
    run(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(5): run()
    b.record(stream); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"traversal {TRAVERSAL}, {label}: {W}x{H} frame {ms:.2f} ms ({W*H/ms/1e3:.1f} Mpixels/s)")
