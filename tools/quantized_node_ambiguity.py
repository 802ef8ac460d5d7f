#!/usr/bin/env python3
"""How often would a 16-byte node (six planes on a B-bit grid over the mesh's extent + one link word) fail to decide
a box test?  Replays the reference traversal (src/tree/bvh.rs:118-133) for random camera rays of the bench frame in
plain Python and, per node visit, evaluates the slab test on the quantised box grown outwards (a miss there proves a
miss) and shrunk inwards by one cell (a hit there proves a hit); a visit neither box decides would need the exact
node.  Usage: tools/quantized_node_ambiguity.py [bits (16)] [rays (1500)]"""
import sys, math, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import rtg
import bench
sys.setrecursionlimit(10000)
BITS=int(sys.argv[1]) if len(sys.argv)>1 else 16
tris = rtg.heightfield(1000,500,bench.SEED,bench.LOCAL_SCALE)
bvh = rtg.HostBvh(tris)
nodes = bvh.nodes(); leaves = bvh.leaves(); root=bvh.root
lo_all = nodes["lo"].min(axis=0); hi_all = nodes["hi"].max(axis=0)
ext = hi_all-lo_all
cells = (1<<BITS)-1
scale = cells/ext
qlo = np.floor((nodes["lo"]-lo_all)*scale); qhi = np.ceil((nodes["hi"]-lo_all)*scale)
olo = lo_all+qlo/scale; ohi = lo_all+qhi/scale      # outer box
ilo = lo_all+(qlo+1)/scale; ihi = lo_all+(qhi-1)/scale   # inner box
# exact where on grid: ignore
cb = nodes["child_begin"]; ce=nodes["child_end"]; lb=nodes["leaf_begin"]; le=nodes["leaf_end"]
NLO=nodes["lo"]; NHI=nodes["hi"]
def slab(lo,hi,o,d,bound):
    mn=-math.inf; mx=math.inf
    for k in range(3):
        if d[k]==0.0: continue
        a=(lo[k]-o[k])/d[k]; b=(hi[k]-o[k])/d[k]
        mn=max(mn,min(a,b)); mx=min(mx,max(a,b))
        if not mn<mx: return False
    l0=max(mn,0.0)
    return l0<mx and l0<bound
def tri(v,o,d):
    e01=v[1]-v[0]; e12=v[2]-v[1]
    nt=np.cross(e01,e12); nd=nt@d
    if nd>0: return None
    t=((v[0]-o)@nt)/nd
    if t<0: return None
    q=o+d*t; den=nt@nt
    for i in range(3):
        c=(np.cross(v[(i+1)%3]-v[i], q-v[i])@nt)/den
        if c<0: return None
    return t
stats={"visits":0,"amb":0,"amb_leafnode":0,"leafnode_visits":0}
def visit(n,o,d,best):
    stats["visits"]+=1
    isleaf = le[n]>lb[n]
    if isleaf: stats["leafnode_visits"]+=1
    exact = slab(NLO[n],NHI[n],o,d,best[0])
    p_in = (ilo[n]<ihi[n]).all() and slab(ilo[n],ihi[n],o,d,best[0])
    f_out = not slab(olo[n],ohi[n],o,d,best[0])
    if not p_in and not f_out:
        stats["amb"]+=1
        if isleaf: stats["amb_leafnode"]+=1
    assert not (p_in and not exact) and not (f_out and exact)
    if not exact: return
    for l in range(lb[n],le[n]):
        t=tri(leaves[l],o,d)
        if t is not None and t<best[0]: best[0]=t
    for c in range(cb[n],ce[n]): visit(c,o,d,best)
rng=np.random.default_rng(1)
W,H=1920,1080
xf = 1.0/bench.LOCAL_SCALE
for i in range(int(sys.argv[2]) if len(sys.argv)>2 else 1500):
    px=rng.integers(0,W); py=rng.integers(0,H)
    sx=(px-(W-1)/2)/W; sy=-(py-(H-1)/2)/H
    dd=np.array([sx*math.tan(math.pi/4), sy*math.tan(math.pi/4), -1.0]); dd/=np.linalg.norm(dd)
    o=np.array([0,0,1.0])*bench.LOCAL_SCALE   # inverse of scale 1/100: object space = world*100
    visit(root,o,dd,[math.inf])
print(BITS, stats, "amb rate %.4f"%(stats["amb"]/stats["visits"]), "visits/ray", stats["visits"]/ (int(sys.argv[2]) if len(sys.argv)>2 else 1500))
