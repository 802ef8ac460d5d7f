#!/usr/bin/env python3
"""Prints the phased kernel's per-phase step/lane tallies on the bench workload (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import rtg
import bench

opts = dict(kv.split("=") for kv in sys.argv[1:])
tris = rtg.heightfield(1000, 500, bench.SEED, bench.LOCAL_SCALE)
bvh = rtg.HostBvh(tris)
ctx = rtg.Context(0)
ctx.set_option("traversal", 2)
for k, v in opts.items():
    ctx.set_option(k, int(v))
ctx.set_option("stats", 2)
mesh = ctx.mesh(bvh)
scene = ctx.scene([rtg.mesh_object(mesh, rtg.material((1, 1, 1)), rtg.transform(1.0 / bench.LOCAL_SCALE))])
hits = torch.empty(bench.WIDTH * bench.HEIGHT * 16, dtype=torch.uint8, device="cuda")
scene.render_primary_raw(bench.frame_view(rtg, 0), bench.WIDTH, bench.HEIGHT, (0, 0, bench.WIDTH, bench.HEIGHT), hits.data_ptr())
st = ctx.debug_stats()
w = st["waves"]
print(opts, {k: round(v / w, 1) for k, v in st.items()},
      "node lanes/step %.1f tri lanes/step %.1f obj lanes/step %.1f, different 128-byte lines per node load %.1f" % (
          st["node_lanes"] / max(1, st["node_steps"]), st["tri_lanes"] / max(1, st["tri_steps"]),
          st["obj_lanes"] / max(1, st["obj_steps"]), st["node_lines"] / max(1, st["node_steps"])))
