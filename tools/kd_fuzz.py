#!/usr/bin/env python3
"""Campaign for the device k-d build and the nearest-photon search: random photon counts, coordinates with many
ties, random k; distances against the oracle's brute-force sort, ids where the distances (including the next one) are unique.
Usage: tools/kd_fuzz.py [cases (500)] [seed (1)]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracer_amd import rtg  # noqa: E402
import orc  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = rtg.Context(0)
checked = 0
for case in range(cases):
    n = int(rng.choice([rng.integers(1, 70), rng.integers(1, 3000), rng.integers(1, 60000)]))
    quantum = float(rng.choice([0.0, 0.0, 1e-3, 0.25, 2.0]))
    pos = rng.normal(size=(n, 3)) * rng.choice([1e-3, 1.0, 1e4], size=3)
    if quantum:
        pos = np.round(pos / quantum) * quantum
    if rng.random() < 0.3:
        pos[:, int(rng.integers(0, 3))] = float(rng.normal())          # a flat cloud
    ph = np.zeros(n, dtype=rtg.PHOTON_DTYPE)
    ph["position"] = pos
    pm = rtg.PhotonMap(ctx, ph)
    kd = orc.OracleKd(pos)
    q = np.concatenate([pos[rng.integers(0, n, size=40)] + rng.normal(scale=1e-3, size=(40, 3)), rng.normal(size=(40, 3)) * 3])
    k = int(rng.integers(1, 17))
    idx, dist = pm.nearest(q, k)
    bidx, bdist = kd.nearest_brute(q, k + 1)                  # one more: a tie for the last place leaves its id open
    assert np.array_equal(dist.view(np.uint64), bdist[:, :k].view(np.uint64)), (case, n, quantum, k)
    unique = np.array([np.unique(r).size == k + 1 for r in bdist])
    assert np.array_equal(idx[unique], bidx[unique][:, :k]), (case, n, quantum, k)
    checked += q.shape[0]
    pm.close()
print(f"{cases} trees, {checked} queries: distances equal the brute-force sort bit for bit")
