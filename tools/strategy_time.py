#!/usr/bin/env python3
"""Primary-ray time of the bench scene (1 M-triangle terrain, 1920x1080) for trees made by each of the
reference's builders: BvhForest::subdivide, ::hybrid, ::contract.  Prints nodes visited / triangles tested per ray
(the reference traversal's counts) and the default kernel's time; a row of pixels is compared with the oracle.

    python tools/strategy_time.py [nx ny]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

from raytracer_amd import rtg
import bench


def main():
    grid = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) >= 3 else bench.GRID
    import orc
    import scenes
    tris = rtg.heightfield(*grid, bench.SEED, bench.LOCAL_SCALE)
    ctx = rtg.Context(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    view = rtg.make_view()
    w, h = 1920, 1080
    hits = torch.empty(w * h * 16, dtype=torch.uint8, device="cuda")
    ctr = torch.empty(w * h * 8, dtype=torch.uint8, device="cuda")
    for strategy in ("subdivide", "hybrid", "contract"):
        t0 = time.perf_counter()
        bvh = rtg.HostBvh(tris, strategy=strategy)
        build_s = time.perf_counter() - t0
        mesh = ctx.mesh(bvh)
        xform = rtg.transform(1.0 / bench.LOCAL_SCALE)
        scene = ctx.scene([rtg.mesh_object(mesh, rtg.material((0.8, 0.7, 0.6)), xform)])
        scene.render_primary_raw(view, w, h, (0, 0, w, h), hits.data_ptr(), None, ctr.data_ptr())
        ctx.synchronize()
        c = ctr.cpu().numpy().view(rtg.COUNTER_DTYPE)
        times = []
        for _ in range(12):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ctx.synchronize()
            a.record(torch.cuda.current_stream())
            scene.render_primary_raw(view, w, h, (0, 0, w, h), hits.data_ptr())
            b.record(torch.cuda.current_stream())
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b))
        ms = float(np.median(times[2:]))
        # one row of pixels against the oracle walking the same flattened tree
        row = h // 2
        obvh = orc.OracleBvh(flat=bvh.view)
        mobj = rtg.mesh_object(mesh, rtg.material((0.8, 0.7, 0.6)), xform)
        oscene = orc.OracleScene([orc.oracle_object(mobj, obvh)])
        g = hits.cpu().numpy().view(rtg.HIT_DTYPE)[row * w:(row + 1) * w]
        o, _, _ = oscene.render_primary(view, w, h, tile=(0, row, w, 1))
        bad = scenes.hit_mismatches(g, o).size
        print(f"{strategy:9s}: host build {build_s:6.2f} s, depth {bvh.depth:3d}, nodes/ray {c['nodes_visited'].mean():7.2f}, "
              f"tris/ray {c['tris_tested'].mean():5.2f}, kernel {ms:.3f} ms = {w * h / ms / 1e3:.0f} Mrays/s, "
              f"row mismatches vs oracle {bad}", flush=True)
        assert bad == 0
        scene.close()
        mesh.close()
        bvh.close()


if __name__ == "__main__":
    main()
