#!/usr/bin/env python3
"""Config 3: trispline surface scene, 3840x2160 primary rays on one GPU (timing + CPU port on a strip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from raytracer_amd import rtg

W, H = 3840, 2160
patches = rtg.patch_terrain(125, 62, seed=7, scale=100.0)
t0 = time.time(); pm = rtg.HostPatchMesh(patches, k=8, newton_steps=5); print(f"patches {len(patches)} facets {pm.bvh.n_leaves} build {time.time()-t0:.1f}s depth {pm.bvh.depth}", flush=True)
ctx = rtg.Context(0)
for kv in sys.argv[1:]:                       # ctx options, e.g. tri_min=24
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
mesh = ctx.mesh(pm)
scene = ctx.scene([rtg.patch_object(mesh, rtg.material((1, 1, 1)), rtg.transform(0.01))])
view = rtg.make_view()
n = W * H
hits = torch.empty(n * 16, dtype=torch.uint8, device="cuda")
ctr = torch.empty(n * 2, dtype=torch.int32, device="cuda")
scene.render_primary_raw(view, W, H, (0, 0, W, H), hits.data_ptr(), None, ctr.data_ptr()); torch.cuda.synchronize()
c = ctr.view(n, 2).to(torch.int64).sum(0).cpu().numpy() / n
bpr = c[0] * 64 + c[1] * (72 + 240) + 16
print(f"nodes/ray {c[0]:.1f} patch-leaf tests/ray {c[1]:.2f} algorithmic bytes/ray {bpr:.0f} (node 64 B, facet 72 B + patch 240 B, hit 16 B)")
for trav in (1, 2):
    ctx.set_option("traversal", trav)
    scene.render_primary_raw(view, W, H, (0, 0, W, H), hits.data_ptr()); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(5):
        scene.render_primary_raw(view, W, H, (0, 0, W, H), hits.data_ptr())
    b.record(stream); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"traversal {trav}: {ms:.2f} ms  {n / ms / 1e3:.1f} Mrays/s  algorithmic {bpr * n / ms / 1e6:.0f} GB/s", flush=True)
import orc, scenes
objs, _, _ = scenes.patch_scene_objects(rtg.transform(0.01), with_default=False)
pair = scenes.Pair(None, objs, {0: pm})
t0 = time.time(); pair.cpu.render_primary(view, W, H, tile=(0, 1000, W, 160), threads=16); dt = time.time() - t0
print(f"CPU port, 16 threads, rows 1000-1159: {W * 160 / dt / 1e6:.2f} Mrays/s")
