"""Scene fixtures shared by the parity tests: the same description for the oracle and the GPU library."""
import numpy as np

from raytracer_amd import rtg

import orc


def default_objects(time=0):
    """SceneBuilder{time}: [sphere, plane], lights, view (src/lib.rs:42-121)."""
    return rtg.scene_builder(time)


def terrain(nx, ny, seed=7, local_scale=100.0):
    """Terrain triangles in a x`local_scale` local space plus the instance transform that undoes it."""
    tris = rtg.heightfield(nx, ny, seed, local_scale)
    xform = rtg.transform(1.0 / local_scale, (0.0, 0.0, 0.0))
    return tris, xform


def vertex_normals_for(tris):
    """Per-vertex normals: the face normal tilted a little differently at each corner."""
    e1 = tris[:, 1] - tris[:, 0]
    e2 = tris[:, 2] - tris[:, 1]
    n = np.cross(e1, e2)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    out = np.repeat(n[:, None, :], 3, axis=1).copy()
    out[:, 0, 0] += 0.05
    out[:, 1, 1] += 0.05
    out[:, 2, 2] += 0.05
    out /= np.linalg.norm(out, axis=2, keepdims=True)
    return np.ascontiguousarray(out)


class Pair:
    """A GPU scene and the oracle scene built from the same objects."""

    def __init__(self, ctx, objects, bvhs):
        """objects: rtg.Object list (mesh fields unset); bvhs: {object index: HostBvh or rtg.Bvh view}."""
        self.ctx = ctx
        self.meshes = {}
        self.obvhs = {}
        gobjs, oobjs = [], []
        for i, o in enumerate(objects):
            g = rtg.copy_object(o)
            if i in bvhs:
                if isinstance(bvhs[i], rtg.HostPatchMesh):
                    view = bvhs[i].bvh.view
                else:
                    view = bvhs[i].view if isinstance(bvhs[i], rtg.HostBvh) else bvhs[i]
                self.obvhs[i] = orc.OracleBvh(flat=view)
                if isinstance(bvhs[i], rtg.HostPatchMesh):
                    self.obvhs[i].set_patches(bvhs[i].patch_set())
                oobjs.append(orc.oracle_object(o, self.obvhs[i]))
                if ctx is not None:
                    self.meshes[i] = ctx.mesh(bvhs[i])
                    g.mesh = self.meshes[i].handle
            else:
                oobjs.append(orc.oracle_object(o, None))
            gobjs.append(g)
        self._keep = (objects, bvhs)
        self.gpu = ctx.scene(gobjs) if ctx is not None else None
        self.cpu = orc.OracleScene(oobjs)


def mesh_scene_objects(xform, with_default=True, time=0, diffuse=(0.8, 0.7, 0.6)):
    """[sphere, plane, mesh] (or just [mesh]) with the mesh object's handle left unset."""
    mesh = rtg.Object()
    mesh.kind = rtg.MESH
    mesh.transform = xform
    mesh.material = rtg.material(diffuse)
    if with_default:
        objs, lights, view = default_objects(time)
        return objs + [mesh], lights, view
    _, lights, view = default_objects(time)
    return [mesh], lights, view


def float_bits_differ(a, b):
    """Elementwise: True where two float64 arrays differ in any bit, except that any NaN equals any NaN
    (NaN sign and payload are hardware conventions: x86 produces a negative default NaN, gfx950 a positive one)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return (a.view(np.uint64) != b.view(np.uint64)) & ~(np.isnan(a) & np.isnan(b))


def hit_mismatches(a, b):
    bad = (a["object"] != b["object"]) | (a["leaf"] != b["leaf"]) | float_bits_differ(a["t"], b["t"])
    return np.flatnonzero(bad)


def patch_scene_objects(xform, with_default=True, time=0, diffuse=(0.7, 0.8, 0.6)):
    """Like mesh_scene_objects, with a trispline patch object instead of the triangle mesh."""
    objs, lights, view = mesh_scene_objects(xform, with_default, time, diffuse)
    objs[-1].kind = rtg.PATCHES
    return objs, lights, view


def same_hits(a, b):
    """Bit-exact comparison of two HIT_DTYPE arrays."""
    return hit_mismatches(a.reshape(-1), b.reshape(-1)).size == 0


def shadow_corner_scene(order):
    """A checker floor under a canopy of ordinary triangles (mesh X) and, higher up, a mesh Y that holds zero-area
    triangles (v2 == v1: NaN hit time, src/geo/triangle.rs:54-82) next to a few ordinary ones.  A shadow ray from the
    floor that is blocked by X and then meets a zero-area triangle of Y while Y's holder is empty is NOT occluded in
    the reference: the scene's `.min()` orders the NaN time lowest (src/render/scene_object.rs:31-34,
    src/math/scalar.rs:46-60) and `occ.time < dis` is false (src/render/renderer.rs:197-201).
    order 0: [floor, X, Y]; order 1: [floor, Y, X].  Returns (objects, {index: HostBvh}, index of Y)."""
    rng = np.random.default_rng(42)
    c = np.stack([rng.uniform(-0.8, 0.8, 300), rng.uniform(-0.25, 0.0, 300), rng.uniform(-1.2, 0.3, 300)], axis=1)
    canopy = c[:, None, :] + rng.uniform(-0.1, 0.1, size=(300, 3, 3))
    a = np.stack([rng.uniform(-0.7, 0.7, 16), rng.uniform(0.25, 0.6, 16), rng.uniform(-0.6, 0.7, 16)], axis=1)
    b = a + rng.uniform(0.15, 0.45, size=(16, 3)) * rng.choice([-1.0, 1.0], size=(16, 3))
    slivers = np.stack([a, b, b], axis=1)
    few = c[:30, None, :] + [0, 0.45, 0] + rng.uniform(-0.05, 0.05, size=(30, 3, 3))
    floor = rtg.plane_object((0, -0.5, 0), (0, 0, -1), (-1, 0, 0), rtg.material((1.0, 1.0, 1.0)), rtg.material((0.1, 0.1, 0.1)))
    mx = rtg.mesh_object(None, rtg.material((0.8, 0.7, 0.6)))
    my = rtg.mesh_object(None, rtg.material((0.3, 0.7, 0.9)))
    bx = rtg.HostBvh(np.ascontiguousarray(canopy))
    by = rtg.HostBvh(np.ascontiguousarray(np.concatenate([slivers, few])))
    if order == 0:
        return [floor, mx, my], {1: bx, 2: by}, 2
    return [floor, my, mx], {1: by, 2: bx}, 1
