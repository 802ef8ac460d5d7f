#!/usr/bin/env python3
"""Kernel time of the default primary-ray path against the frame size (same camera, growing resolution) and per
horizontal band of the 1920x1080 frame: shows the fixed cost of a persistent launch -- the chain of dependent node
fetches of its last rays -- next to the per-ray cost.

    python tools/frame_scaling.py
"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from raytracer_amd import rtg
import bench
tris = rtg.heightfield(*bench.GRID, bench.SEED, bench.LOCAL_SCALE)
ctx = rtg.Context(0); st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
mesh = ctx.build_mesh(tris)
sc = ctx.scene([rtg.mesh_object(mesh, rtg.material((0.8,0.7,0.6)), rtg.transform(1.0/bench.LOCAL_SCALE))])
view = rtg.make_view()
def t(W,H,tile):
    n = tile[2]*tile[3]
    hits = torch.empty(n*16, dtype=torch.uint8, device="cuda")
    ts=[]
    for i in range(14):
        a,b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); sc.render_primary_raw(view, W, H, tile, hits.data_ptr()); b.record(st); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts[3:]))
# same image, growing resolution: the work per pixel stays comparable
for (W,H) in [(480,270),(960,540),(1920,1080),(2880,1620),(3840,2160),(7680,4320)]:
    ms = t(W,H,(0,0,W,H))
    print(f"{W}x{H}: {ms:.4f} ms, {W*H/ms/1e3:.0f} Mrays/s", flush=True)
# horizontal bands of the 1080p frame
for k in range(8):
    ms = t(1920,1080,(0,k*135,1920,135))
    print(f"band {k}: {ms:.4f} ms", flush=True)
