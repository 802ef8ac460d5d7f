#!/usr/bin/env python3
"""Config 4 stress: incoherent ray batch (bench_mesh distribution) against a multi-million-triangle BVH.
Usage: tools/config4.py [nx ny n_rays [check]]   (defaults 3163 1581 16777216)
With a 4th argument the host ray sort is skipped and the full-size result is checked instead: the per-lane threaded
kernel and the persistent kernel must agree bit for bit on every ray, and a 262144-ray sample must equal the CPU
oracle (ids, times, visit counters)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracer_amd import rtg

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 3163
ny = int(sys.argv[2]) if len(sys.argv) > 2 else 1581
n_rays = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 24
check = len(sys.argv) > 4
t0 = time.time(); tris = rtg.heightfield(nx, ny, 7, 1.0); t_gen = time.time() - t0
t0 = time.time(); bvh = rtg.HostBvh(tris); t_build = time.time() - t0
print(f"triangles {len(tris)} gen {t_gen:.1f}s build {t_build:.1f}s nodes {bvh.n_nodes} depth {bvh.depth}", flush=True)
ctx = rtg.Context(0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
t0 = time.time(); mesh = ctx.mesh(bvh); print(f"upload {time.time()-t0:.1f}s", flush=True)
scene = ctx.scene([rtg.mesh_object(mesh, rtg.material((1, 1, 1)), rtg.transform())])
t0 = time.time(); rays = rtg.bench_mesh_rays(bvh.bounds(), n_rays, seed=10212233); print(f"rays {time.time()-t0:.1f}s", flush=True)
d_rays = torch.from_numpy(rays).cuda()
hits = torch.empty(n_rays * 16, dtype=torch.uint8, device="cuda")
ctr = torch.empty(n_rays * 2, dtype=torch.int32, device="cuda")

def run(label, rays_t, trav):
    ctx.set_option("traversal", trav)
    scene.trace_raw(rays_t.data_ptr(), n_rays, hits.data_ptr()); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(3):
        scene.trace_raw(rays_t.data_ptr(), n_rays, hits.data_ptr())
    b.record(stream); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    print(f"{label:28s} traversal {trav}: {ms:9.2f} ms  {n_rays / ms / 1e3:8.1f} Mrays/s", flush=True)
    return ms

scene.trace_raw(d_rays.data_ptr(), n_rays, hits.data_ptr(), None, ctr.data_ptr()); torch.cuda.synchronize()
c = ctr.view(n_rays, 2).to(torch.int64).sum(0).cpu().numpy() / n_rays
bpr = c[0] * 64 + c[1] * 72 + 48 + 16
print(f"nodes/ray {c[0]:.1f} tris/ray {c[1]:.2f} algorithmic bytes/ray {bpr:.0f}", flush=True)
for trav in (1, 2):
    ms = run("unsorted", d_rays, trav)
    print(f"    algorithmic {bpr * n_rays / ms / 1e6:.0f} GB/s")
if check:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import orc, scenes
    h2 = hits.clone()                                              # traversal 2 ran last
    ctx.set_option("traversal", 1)
    scene.trace_raw(d_rays.data_ptr(), n_rays, hits.data_ptr()); torch.cuda.synchronize()
    a = h2.view(torch.int64).view(n_rays, 2); b = hits.view(torch.int64).view(n_rays, 2)
    t_a = h2.view(torch.float64).view(n_rays, 2)[:, 1]; t_b = hits.view(torch.float64).view(n_rays, 2)[:, 1]
    differ = ((a != b).any(dim=1) & ~(torch.isnan(t_a) & torch.isnan(t_b) & (a[:, 0] == b[:, 0]))).sum().item()
    n_hit = (h2.view(torch.int32).view(n_rays, 4)[:, 1] != -1).sum().item()
    print(f"full size: {n_rays} rays, {n_hit} hits, threaded vs persistent kernel mismatches: {differ}", flush=True)
    ctx.set_option("traversal", 2)
    m = 1 << 18
    idx = np.sort(np.random.default_rng(1).choice(n_rays, size=m, replace=False))
    sample = np.ascontiguousarray(rays[idx])
    t0 = time.time()
    obvh = orc.OracleBvh(flat=bvh.view)
    oscene = orc.OracleScene([orc.oracle_object(rtg.mesh_object(mesh, rtg.material((1, 1, 1)), rtg.transform()), obvh)])
    c_hits, _, c_ctr = oscene.trace(sample, counters=True, threads=16)
    g = np.frombuffer(h2.cpu().numpy().tobytes(), dtype=rtg.HIT_DTYPE)[idx]
    g_ctr = np.frombuffer(ctr.cpu().numpy().tobytes(), dtype=rtg.COUNTER_DTYPE)[idx]
    bad = scenes.hit_mismatches(g, c_hits)
    print(f"oracle sample: {m} rays in {time.time()-t0:.1f}s, hit mismatches {bad.size}, "
          f"counter mismatches {int((g_ctr != c_ctr).sum())}", flush=True)
    sys.exit(0 if differ == 0 and bad.size == 0 and (g_ctr == c_ctr).all() else 1)
t0 = time.time(); perm, srt = rtg.sort_rays(bvh.bounds(), rays); print(f"host sort {time.time()-t0:.1f}s", flush=True)
d_sorted = torch.from_numpy(srt).cuda()
for trav in (1, 2):
    ms = run("sorted (octant+morton)", d_sorted, trav)
    print(f"    algorithmic {bpr * n_rays / ms / 1e6:.0f} GB/s")
