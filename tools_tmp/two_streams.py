import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from raytracer_amd import rtg
import bench
tris = rtg.heightfield(*bench.GRID, bench.SEED, bench.LOCAL_SCALE)
W,H=1920,1080
ctxs=[]; scenes=[]; streams=[]; hits=[]
for i in range(2):
    ctx = rtg.Context(0); st = torch.cuda.Stream(); ctx.set_stream(st.cuda_stream)
    mesh = ctx.build_mesh(tris)
    sc = ctx.scene([rtg.mesh_object(mesh, rtg.material((0.8,0.7,0.6)), rtg.transform(1.0/bench.LOCAL_SCALE))])
    ctxs.append(ctx); scenes.append((sc,mesh)); streams.append(st)
    hits.append(torch.empty(W*H*16, dtype=torch.uint8, device="cuda"))
view = rtg.make_view()
def run(n, two, blocks=None):
    for c in ctxs:
        c.set_option("max_blocks", blocks or 1024)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for f in range(n):
        k = f & 1 if two else 0
        scenes[k][0].render_primary_raw(view, W, H, (0,0,W,H), hits[k].data_ptr())
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
for rep in range(3):
    print("one stream %.4f ms/frame | two streams %.4f | two streams 512 blocks each %.4f | two streams 768 %.4f" % (run(40,False), run(40,True), run(40,True,512), run(40,True,768)), flush=True)
