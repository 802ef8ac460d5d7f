"""GPU, full size: configuration 4 -- 67 108 864 incoherent rays (the `bench_mesh` distribution,
src/tree/benchmark.rs:15-27, seed 10212233) against a 10 001 406-triangle BVH built on the GPU.

The oracle cannot trace 64 M rays in test time, so the bar is split: a 262 144-ray sample must equal the oracle bit
for bit (ids, times, node/triangle visit counters, through the tree read back from the device); every ray must come
out the same from the persistent kernel and from the per-lane threaded kernel (two independent traversals); and the
answers must not depend on where a ray sits in the batch (a rotated batch gives the rotated answers)."""
import numpy as np
import pytest

import orc
import scenes
from raytracer_amd import rtg

pytestmark = pytest.mark.gpu

N_RAYS = 1 << 26
GRID = (3163, 1581)


def _device_hits_differ(torch, a, b, n):
    ia, ib = a.view(torch.int64).view(n, 2), b.view(torch.int64).view(n, 2)
    ta, tb = a.view(torch.float64).view(n, 2)[:, 1], b.view(torch.float64).view(n, 2)[:, 1]
    both_nan = torch.isnan(ta) & torch.isnan(tb) & (ia[:, 0] == ib[:, 0])
    return int(((ia != ib).any(dim=1) & ~both_nan).sum().item())


def test_config4_64m_incoherent_rays_10m_triangles(gpu_ctx):
    import torch
    tris = rtg.heightfield(GRID[0], GRID[1], 7, 1.0)
    assert tris.shape[0] == 10_001_406
    mesh = gpu_ctx.build_mesh(tris)                      # BvhForest::subdivide + Bvh::new on the device
    n_nodes, n_leaves, root, depth = mesh.sizes()
    assert n_nodes == 2 * tris.shape[0] - 1 and n_leaves == tris.shape[0]
    obj = rtg.mesh_object(mesh, rtg.material((1, 1, 1)), rtg.transform())
    scene = gpu_ctx.scene([obj])
    rays = rtg.bench_mesh_rays(mesh.bounds(), N_RAYS, seed=10212233)
    d_rays = torch.from_numpy(rays).cuda()
    hits2 = torch.empty(N_RAYS * 16, dtype=torch.uint8, device="cuda")
    hits1 = torch.empty_like(hits2)
    ctr = torch.empty(N_RAYS * 2, dtype=torch.int32, device="cuda")
    gpu_ctx.set_option("traversal", 2)
    scene.trace_raw(d_rays.data_ptr(), N_RAYS, hits2.data_ptr(), None, ctr.data_ptr())
    gpu_ctx.synchronize()
    # the counting build and the plain build of the persistent kernel agree
    scene.trace_raw(d_rays.data_ptr(), N_RAYS, hits1.data_ptr())
    gpu_ctx.synchronize()
    assert _device_hits_differ(torch, hits2, hits1, N_RAYS) == 0
    # the per-lane threaded kernel agrees on every ray
    gpu_ctx.set_option("traversal", 1)
    scene.trace_raw(d_rays.data_ptr(), N_RAYS, hits1.data_ptr())
    gpu_ctx.synchronize()
    gpu_ctx.set_option("traversal", 2)
    assert _device_hits_differ(torch, hits2, hits1, N_RAYS) == 0
    n_hit = int((hits2.view(torch.int32).view(N_RAYS, 4)[:, 1] != -1).sum().item())
    assert 0.1 < n_hit / N_RAYS < 0.5, n_hit
    # the batch rotated by an odd amount gives the rotated answers (no dependence on the slot, the chunk or the wave)
    shift = 12_345_677
    rolled = torch.roll(d_rays.view(N_RAYS, 6), shifts=shift, dims=0).contiguous()
    torch.cuda.synchronize()                             # torch's stream made the copy, the context's stream reads it
    scene.trace_raw(rolled.data_ptr(), N_RAYS, hits1.data_ptr())
    gpu_ctx.synchronize()
    back = torch.roll(hits1.view(N_RAYS, 16), shifts=-shift, dims=0).contiguous().view(-1)
    assert _device_hits_differ(torch, hits2, back, N_RAYS) == 0
    del rolled, back
    # sorted on the device (octant | Morton key, radix sort, gather, trace, scatter back): the same answers
    gpu_ctx.set_option("sort_rays", 2)
    scene.trace_raw(d_rays.data_ptr(), N_RAYS, hits1.data_ptr())
    gpu_ctx.synchronize()
    gpu_ctx.set_option("sort_rays", 1)
    assert _device_hits_differ(torch, hits2, hits1, N_RAYS) == 0
    del hits1

    # oracle sample: 262 144 rays of the batch through the tree the device built
    m = 1 << 18
    idx = np.sort(np.random.default_rng(1).choice(N_RAYS, size=m, replace=False))
    nodes, leaves, _ = mesh.read_bvh()
    obvh = orc.OracleBvh(flat=rtg.flat_bvh(nodes, leaves, root))
    oscene = orc.OracleScene([orc.oracle_object(obj, obvh)])
    c_hits, _, c_ctr = oscene.trace(np.ascontiguousarray(rays[idx]), counters=True, threads=16)
    d_idx = torch.from_numpy(idx).cuda()
    g_hits = np.frombuffer(hits2.view(N_RAYS, 16)[d_idx].cpu().numpy().tobytes(), dtype=rtg.HIT_DTYPE)
    g_ctr = np.frombuffer(ctr.view(N_RAYS, 2)[d_idx].cpu().numpy().tobytes(), dtype=rtg.COUNTER_DTYPE)
    bad = scenes.hit_mismatches(g_hits, c_hits)
    assert bad.size == 0, f"{bad.size} of {m} sampled rays differ from the oracle"
    assert np.array_equal(g_ctr, c_ctr)
    assert (c_hits["leaf"] != rtg.RTG_NONE).mean() > 0.1
    scene.close()
    mesh.close()
