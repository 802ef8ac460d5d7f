#!/usr/bin/env python3
"""What would keeping a passed node's second child in registers save for incoherent rays?  (CPU model, a few thousand
bench_mesh rays against a heightfield tree laid out as the library lays it out.)

The reference's walk (tree/bvh.rs:118-133) visits both children of every node whose box passes: the first at once, the
second after the first one's whole subtree.  With unrelated rays in flight a cache line lives in the L2 for about as
long as a ray takes to fetch a couple of lines of its own, so the second child's record -- it sits beside the first --
is usually fetched from memory again.  The model replays the walk per ray, maps every visit to the 128-byte line of the
node's threaded position, and counts line fetches with (a) the top of the tree always cached and (b) a per-ray LRU of
`C` lines standing in for the ray's share of the L2.  It then repeats the count with the second child's record held
in a per-ray stack of K entries (oldest dropped on overflow) filled when the first child is fetched.
Usage: tools/sibling_cache_estimate.py [quads_x quads_y [rays]]"""
import os
import sys
import math

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import rtg
import bench

nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1581, 790)
n_rays = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
TOP = 12

tris = rtg.heightfield(nx, ny, bench.SEED, 1.0)
bvh = rtg.HostBvh(tris)
nodes = bvh.nodes()
leaves = bvh.leaves()
lo = nodes['lo'].tolist(); hi = nodes['hi'].tolist()
cb = nodes['child_begin'].tolist(); ce = nodes['child_end'].tolist()
lb = nodes['leaf_begin'].tolist(); le = nodes['leaf_end'].tolist()
root = bvh.view.root
n_nodes = len(cb)
print(f"{len(tris)} triangles, {n_nodes} nodes, depth {bvh.depth}", flush=True)

# threaded positions: sibling pairs, the parents of the top levels breadth first, the rest in pre-order of the parents
pos = [0] * n_nodes
depth_of = [0] * n_nodes
g = 0
level = [root]
for d in range(TOP):
    below = []
    for i in level:
        for c in range(cb[i], ce[i]):
            pos[c] = g; g += 1
            depth_of[c] = d + 1
            below.append(c)
    level = below
stack = level[::-1]
while stack:
    i = stack.pop()
    for c in range(cb[i], ce[i]):
        pos[c] = g; g += 1
        depth_of[c] = depth_of[i] + 1
    for c in range(ce[i] - 1, cb[i] - 1, -1):
        stack.append(c)
pos[root] = n_nodes - 1
top_block = sum(1 for i in range(n_nodes) if depth_of[i] <= TOP)   # positions below this are the compact top block
print(f"top block: {top_block} nodes", flush=True)

inf = float('inf')
P = leaves['p'] if leaves.dtype.names else leaves


def box(n, o, d, bound):
    mn, mx = -inf, inf
    l, h = lo[n], hi[n]
    for k in range(3):
        if d[k] != 0.0:
            a = (l[k] - o[k]) / d[k]; b = (h[k] - o[k]) / d[k]
            if a > b: a, b = b, a
            if a > mn: mn = a
            if b < mx: mx = b
            if not (mn < mx): return False
    l0 = mn if mn > 0.0 else 0.0
    return l0 < mx and l0 < bound


def tri(l, o, d):
    v0, v1, v2 = P[l][0], P[l][1], P[l][2]
    nt = np.cross(v1 - v0, v2 - v1)
    nd = nt.dot(d)
    if nd > 0: return None
    t = (v0 - o).dot(nt) / nd if nd != 0 else inf
    if t < 0: return None
    q = o + d * t
    den = nt.dot(nt)
    for a, b in ((v0, v1), (v1, v2), (v2, v0)):
        if np.cross(b - a, q - a).dot(nt) / den < 0: return None
    return t


def walk(o, d):
    """visit list: (node, is_second_child, passed)"""
    best = inf
    out = []
    stack = [(root, False)]
    ol, dl = o.tolist(), d.tolist()
    while stack:
        n, second = stack.pop()
        ok = box(n, ol, dl, best)
        out.append((n, second, ok))
        if not ok: continue
        for l in range(lb[n], le[n]):
            t = tri(l, o, d)
            if t is not None and t < best: best = t
        kids = list(range(cb[n], ce[n]))
        for j in range(len(kids) - 1, -1, -1):
            stack.append((kids[j], j > 0))
    return out


def line_of(p):
    return (64 + 32 * p) >> 7


# siblings: map first child -> second child
second_of = {}
for i in range(n_nodes):
    if ce[i] - cb[i] == 2:
        second_of[cb[i]] = cb[i] + 1


def fetches(visits, C, K):
    """line fetches of one ray below the top block (per-ray LRU of C lines), and the second children served from a
    K-entry stack of records kept since their sibling was fetched (K = 0: none kept)"""
    lru = []
    held = []
    n_fetch = 0
    served = 0
    for n, second, ok in visits:
        p = pos[n]
        if second and K > 0 and held and held[-1] == n:
            held.pop()
            served += 1
            continue
        if second and K > 0:
            while held and depth_of[held[-1]] >= depth_of[n]:
                held.pop()
        if p >= top_block:
            ln = line_of(p)
            if ln in lru:
                lru.remove(ln); lru.append(ln)
            else:
                n_fetch += 1
                lru.append(ln)
                if len(lru) > C: lru.pop(0)
        if K > 0 and n in second_of:
            held.append(second_of[n])
            if len(held) > K: held.pop(0)
    return n_fetch, served


b = bvh.bounds()
rays = rtg.bench_mesh_rays(b, n_rays, seed=bench.C4_RAY_SEED)
walks = []
for r in range(n_rays):
    o = np.array(rays[r, :3]); d = np.array(rays[r, 3:])
    walks.append(walk(o, d))
    if r % 500 == 0: print("ray", r, flush=True)
v = np.array([len(w) for w in walks])
sec = np.array([sum(1 for x in w if x[1]) for w in walks])
deep = np.array([sum(1 for x in w if pos[x[0]] >= top_block) for w in walks])
print(f"visits per ray {v.mean():.1f}, of second children {sec.mean():.1f}, below the top block {deep.mean():.1f}")
for C in (1, 2, 3, 4, 8, 16, 64):
    row = [f"C={C:3d}:"]
    for K in (0, 1, 2, 3, 4, 6, 8):
        f = np.array([fetches(w, C, K) for w in walks])
        row.append(f"K={K} {f[:, 0].mean():5.2f} ({f[:, 1].mean():4.1f})")
    print("  ".join(row), flush=True)
