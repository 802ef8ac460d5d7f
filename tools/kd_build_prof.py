#!/usr/bin/env python3
"""Times rtg_photon_map_create on synthetic photons already in device memory.  Usage: tools/kd_build_prof.py [photons (420000)]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import rtg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 420_000
ctx = rtg.Context(0)
rng = np.random.default_rng(3)
ph = np.zeros(n, dtype=rtg.PHOTON_DTYPE)
ph["position"] = rng.normal(size=(n, 3))
ph["position"][:, 1] = np.where(rng.random(n) < 0.5, -0.5, ph["position"][:, 1])
ph["n_bounces"] = 2
dev = torch.from_numpy(ph.view(np.uint8).reshape(-1)).cuda()
for i in range(4):
    torch.cuda.synchronize()
    t = time.perf_counter()
    pm = rtg.PhotonMap(ctx, dev.data_ptr(), n=n)
    dt = time.perf_counter() - t
    print(f"k-d tree over {n} photons in device memory: {1e3 * dt:.2f} ms", flush=True)
    pm.close()
