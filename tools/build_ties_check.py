#!/usr/bin/env python3
"""Large inputs full of exact ties (coordinates snapped to a coarse lattice, repeated triangles): the tree built on
the GPU must still equal the host builder's byte for byte.  python tools/build_ties_check.py [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from raytracer_amd import rtg


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    ctx = rtg.Context(0)
    rng = np.random.default_rng(17)
    for quantum, repeats in ((1.0 / 8, 1), (1.0, 4), (0.0, 1)):
        base = rng.uniform(-40.0, 40.0, size=(n // repeats, 1, 3))
        tris = base + rng.uniform(-0.5, 0.5, size=(n // repeats, 3, 3))
        if quantum:
            tris = np.round(tris / quantum) * quantum
        tris = np.ascontiguousarray(np.repeat(tris, repeats, axis=0))
        rng.shuffle(tris, axis=0)
        host = rtg.HostBvh(tris)
        mesh = ctx.build_mesh(tris)
        nodes, leaves, source = mesh.read_bvh()
        same = (nodes.tobytes() == host.nodes().tobytes() and np.array_equal(source, host.leaf_source())
                and mesh.sizes() == (host.n_nodes, host.n_leaves, host.root, host.depth))
        centres = (tris.min(axis=1) + tris.max(axis=1)) / 2
        distinct = [len(np.unique(centres[:, a])) for a in range(3)]
        print(f"{len(tris)} triangles, lattice {quantum}, x{repeats}: distinct centres per axis {distinct}, "
              f"depth {host.depth}, identical: {same}", flush=True)
        assert same
        mesh.close()
        host.close()


if __name__ == "__main__":
    main()
