"""ctypes binding of the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import rtg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "liboracle.so")
if not os.path.exists(_SO):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
lib = C.CDLL(_SO)

_vp = C.c_void_p
_dp = C.POINTER(C.c_double)


def _sig(fn, restype, *argtypes):
    fn.restype = restype
    fn.argtypes = list(argtypes)


_sig(lib.orc_minimum, C.c_double, C.c_double, C.c_double)
_sig(lib.orc_maximum, C.c_double, C.c_double, C.c_double)
_sig(lib.orc_real_cmp, C.c_int, C.c_double, C.c_double)
_sig(lib.orc_dot3, C.c_double, _dp, _dp)
_sig(lib.orc_cross, None, _dp, _dp, _dp)
_sig(lib.orc_normalize3, None, _dp, _dp)
for _n in ("orc_mat2_inverse", "orc_mat3_inverse", "orc_mat4_inverse"):
    _sig(getattr(lib, _n), None, _dp, _dp)
for _n in ("orc_mat2_mul", "orc_mat3_mul", "orc_mat4_mul"):
    _sig(getattr(lib, _n), None, _dp, _dp, _dp)
_sig(lib.orc_mat4_from_scale_rotation_translation, None, _dp, _dp, _dp)
_sig(lib.orc_transform_from_forward, None, _dp, C.POINTER(rtg.Transform))
_sig(lib.orc_bounds_union, None, _dp, _dp, _dp)
_sig(lib.orc_bounds_surface_area, C.c_double, _dp)
_sig(lib.orc_smpte2048_encode, C.c_double, C.c_double)
_sig(lib.orc_smpte2048_decode, C.c_double, C.c_double)
_sig(lib.orc_pq8, C.c_uint8, C.c_double)
_sig(lib.orc_find_roots_quadratic, C.c_int, C.c_double, C.c_double, C.c_double, _dp)
_sig(lib.orc_bounds_raycast, C.c_int, _dp, _dp, _dp)
_sig(lib.orc_triangle_raycast, C.c_int, _dp, _dp, C.c_int, _dp)
_sig(lib.orc_sphere_raycast, C.c_int, _dp, C.c_double, _dp, _vp)
_sig(lib.orc_plane_raycast, C.c_int, _dp, _dp, _dp, _dp, _vp, C.POINTER(C.c_int))
_sig(lib.orc_view_get_ray, None, C.POINTER(rtg.View), _dp, _dp)
_sig(lib.orc_pixel_ndc, None, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _dp)
_sig(lib.orc_dielectric, None, _dp, _dp, C.c_double, C.c_double, _dp)
_sig(lib.orc_bvh_build_subdivide, _vp, _vp, C.c_uint64, _vp)
_sig(lib.orc_bvh_build, _vp, _vp, C.c_uint64, _vp, C.c_int)
_sig(lib.orc_bvh_from_flat, _vp, C.POINTER(rtg.Bvh))
_sig(lib.orc_bvh_destroy, None, _vp)
_sig(lib.orc_bvh_n_nodes, C.c_uint64, _vp)
_sig(lib.orc_bvh_n_leaves, C.c_uint64, _vp)
_sig(lib.orc_bvh_root, C.c_uint32, _vp)
_sig(lib.orc_bvh_nodes, _vp, _vp)
_sig(lib.orc_bvh_leaves, _vp, _vp)
_sig(lib.orc_bvh_vertex_normals, _vp, _vp)
_sig(lib.orc_bvh_leaf_source, C.POINTER(C.c_uint32), _vp)
_sig(lib.orc_bvh_depth, C.c_uint32, _vp)
_sig(lib.orc_bvh_raycast, C.c_int, _vp, _dp, _vp, _vp)
_sig(lib.orc_bvh_raycast_brute, C.c_int, _vp, _dp, _vp)
_sig(lib.orc_trispline_from_normal_hermite, None, _vp, _vp)
_sig(lib.orc_trispline_eval, None, _vp, C.c_double, C.c_double, _vp)
_sig(lib.orc_patches_tessellate, None, _vp, C.c_uint64, C.c_uint32, _vp, _vp)
_sig(lib.orc_bvh_set_patches, None, _vp, C.POINTER(rtg.PatchSet))
_sig(lib.orc_scene_create, _vp, C.POINTER(rtg.SceneDesc))
_sig(lib.orc_scene_destroy, None, _vp)
_sig(lib.orc_render_primary, C.c_int, _vp, C.POINTER(rtg.View), C.c_uint32, C.c_uint32, rtg.Rect, _vp, _vp, _vp, C.c_int)
_sig(lib.orc_trace, C.c_int, _vp, _vp, C.c_uint64, _vp, _vp, _vp, C.c_int)
_sig(lib.orc_trace_paths, C.c_int, _vp, _vp, C.c_uint64, C.c_uint32, _vp, C.c_uint64, C.POINTER(C.c_uint64))
_sig(lib.orc_shade_direct, C.c_int, _vp, _vp, C.c_uint32, _vp, C.c_uint64, _vp, _vp, C.c_int)
_sig(lib.orc_render_frame, C.c_int, _vp, C.POINTER(rtg.View), _vp, C.c_uint32, C.c_uint32, C.c_uint32, rtg.Rect,
     C.c_uint32, _vp, _vp, _vp, C.c_int)
_sig(lib.orc_der2_op, None, C.c_int, _dp, _dp, _dp)
_sig(lib.orc_der2_partial_cmp, C.c_int, _dp, _dp)
_sig(lib.orc_der2_known_answer, None, _dp)
_sig(lib.orc_zenith_y_into_normal, None, _dp, _dp)
_sig(lib.orc_fibonacci_sphere, None, C.c_uint64, _vp, _vp)
_sig(lib.orc_kd_create, _vp, _vp, C.c_uint64)
_sig(lib.orc_kd_destroy, None, _vp)
_sig(lib.orc_kd_check, C.c_int, _vp)
_sig(lib.orc_kd_nearest, None, _vp, _vp, C.c_uint64, C.c_uint32, _vp, _vp, C.c_int)
_sig(lib.orc_kd_nearest_brute, None, _vp, C.c_uint64, _vp, C.c_uint64, C.c_uint32, _vp, _vp, C.c_int)
_sig(lib.orc_photons_emit, C.c_int, _vp, _vp, C.c_uint32, _vp, C.c_uint64, C.c_uint32, _vp, C.c_uint64, C.POINTER(C.c_uint64))
_sig(lib.orc_indirect_irrad, C.c_int, _vp, _vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, C.c_uint32, _vp, C.c_int)
_sig(lib.orc_render_frame_photons, C.c_int, _vp, C.POINTER(rtg.View), _vp, C.c_uint32, _vp, _vp, C.c_uint32, C.c_uint32, C.c_double,
     C.c_uint32, C.c_uint32, rtg.Rect, C.c_uint32, _vp, _vp, C.c_int)
_sig(lib.orc_photon_adjust, None, _vp, _vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp)

THREADS = min(8, os.cpu_count() or 1)


def d(*vals):
    vals = [float(v) for v in np.asarray(vals, dtype=np.float64).reshape(-1)]
    return (C.c_double * len(vals))(*vals)


def mat_inverse(m):
    m = np.asarray(m, dtype=np.float64)
    n = m.shape[0]
    out = (C.c_double * (n * n))()
    getattr(lib, f"orc_mat{n}_inverse")(d(m), out)
    return np.array(list(out)).reshape(n, n)


def mat_mul(a, b):
    a = np.asarray(a, dtype=np.float64)
    n = a.shape[0]
    out = (C.c_double * (n * n))()
    getattr(lib, f"orc_mat{n}_mul")(d(a), d(b), out)
    return np.array(list(out)).reshape(n, n)


def bounds_raycast(box, ray):
    out = (C.c_double * 2)()
    ok = lib.orc_bounds_raycast(d(box), d(ray), out)
    return (out[0], out[1]) if ok else None


def triangle_raycast(tri, ray, manifold=False):
    out = (C.c_double * 12)()
    ok = lib.orc_triangle_raycast(d(tri), d(ray), 1 if manifold else 0, out)
    if not ok:
        return None
    o = list(out)
    return {"t": o[0], "bary": o[1:4], "q": o[4:7], "mp": o[7:9], "n": o[9:12]}


def sphere_raycast(center, radius, ray):
    pt = np.zeros(1, dtype=rtg.POINT_DTYPE)
    ok = lib.orc_sphere_raycast(d(center), float(radius), d(ray), pt.ctypes.data)
    return pt[0] if ok else None


def plane_raycast(position, tan1, tan2, ray):
    pt = np.zeros(1, dtype=rtg.POINT_DTYPE)
    m2 = C.c_int(0)
    ok = lib.orc_plane_raycast(d(position), d(tan1), d(tan2), d(ray), pt.ctypes.data, C.byref(m2))
    return (pt[0], bool(m2.value)) if ok else None


def find_roots_quadratic(a2, a1, a0):
    out = (C.c_double * 2)()
    n = lib.orc_find_roots_quadratic(a2, a1, a0, out)
    return [out[i] for i in range(n)]


def dielectric(inc, signed_norm, n1, n2):
    out = (C.c_double * 8)()
    lib.orc_dielectric(d(inc), d(signed_norm), n1, n2, out)
    o = list(out)
    return {"reflectance": o[0], "reflect": o[1:4], "has_refract": o[4] != 0.0, "refract": o[5:8]}


def trispline_from_normal_hermite(nh):
    nh = np.ascontiguousarray(nh, dtype=np.float64).reshape(10, 3)
    out = np.empty((10, 3), dtype=np.float64)
    lib.orc_trispline_from_normal_hermite(nh.ctypes.data, out.ctypes.data)
    return out


def trispline_eval(patch, t, u):
    patch = np.ascontiguousarray(patch, dtype=np.float64).reshape(10, 3)
    out = np.empty((4, 3), dtype=np.float64)
    lib.orc_trispline_eval(patch.ctypes.data, float(t), float(u), out.ctypes.data)
    return out


def patches_tessellate(patches, k):
    patches = np.ascontiguousarray(patches, dtype=np.float64).reshape(-1, 10, 3)
    n = patches.shape[0] * k * k
    tris = np.empty((n, 3, 3), dtype=np.float64)
    info = np.zeros(n, dtype=rtg.PATCH_LEAF_DTYPE)
    lib.orc_patches_tessellate(patches.ctypes.data, patches.shape[0], k, tris.ctypes.data, info.ctypes.data)
    return tris, info


DER2_OPS = {"add": 0, "sub": 1, "mul": 2, "div": 3, "neg": 4, "sqrt": 5, "sin": 6, "cos": 7, "abs": 8, "minimum": 9,
            "maximum": 10}


def der2(op, a, b=(0.0, 0.0, 0.0)):
    """One Der<2> operation on (v, d0, d1) triples."""
    out = (C.c_double * 3)()
    lib.orc_der2_op(DER2_OPS[op], d(a), d(b), out)
    return np.array(list(out))


def der2_partial_cmp(a, b):
    """-1 / 0 / 1, or None when the derived PartialOrd gives no order."""
    c = lib.orc_der2_partial_cmp(d(a), d(b))
    return None if c == 2 else c


def zenith_y_into_normal(dir2):
    out = (C.c_double * 3)()
    lib.orc_zenith_y_into_normal(d(dir2), out)
    return np.array(list(out))


def fibonacci_zenith_y(count, seed=1):
    """Sphere::fibonacci_sphere with numpy's generator standing in for the reference's: (count, 2) ZenithY values."""
    u = np.random.default_rng(seed).random((count, 2))
    out = np.empty((count, 2), dtype=np.float64)
    lib.orc_fibonacci_sphere(count, u.ctypes.data, out.ctypes.data)
    return out


class OracleKd:
    """KdTree over positions; values are the positions' indices."""
    _h = None

    def __init__(self, positions3):
        self.positions = np.ascontiguousarray(positions3, dtype=np.float64).reshape(-1, 3)
        self._h = lib.orc_kd_create(self.positions.ctypes.data, self.positions.shape[0])

    @property
    def handle(self):
        return self._h

    def check(self):
        return bool(lib.orc_kd_check(self._h))

    def nearest(self, queries3, k, threads=THREADS):
        q = np.ascontiguousarray(queries3, dtype=np.float64).reshape(-1, 3)
        idx = np.empty((q.shape[0], k), dtype=np.uint32)
        dist = np.empty((q.shape[0], k), dtype=np.float64)
        if k:
            lib.orc_kd_nearest(self._h, q.ctypes.data, q.shape[0], k, idx.ctypes.data, dist.ctypes.data, threads)
        return idx, dist

    def nearest_brute(self, queries3, k, threads=THREADS):
        q = np.ascontiguousarray(queries3, dtype=np.float64).reshape(-1, 3)
        idx = np.empty((q.shape[0], k), dtype=np.uint32)
        dist = np.empty((q.shape[0], k), dtype=np.float64)
        if k:
            lib.orc_kd_nearest_brute(self.positions.ctypes.data, self.positions.shape[0], q.ctypes.data, q.shape[0], k,
                                     idx.ctypes.data, dist.ctypes.data, threads)
        return idx, dist

    def close(self):
        if self._h:
            lib.orc_kd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown
            pass


class OracleBvh:
    _h = None

    def __init__(self, tris=None, vertex_normals=None, flat=None, strategy="subdivide"):
        if flat is not None:
            self._keep = flat
            self._h = lib.orc_bvh_from_flat(C.byref(flat))
        else:
            tris = np.ascontiguousarray(tris, dtype=np.float64).reshape(-1, 3, 3)
            vn = None if vertex_normals is None else np.ascontiguousarray(vertex_normals, dtype=np.float64)
            self._h = lib.orc_bvh_build(tris.ctypes.data, tris.shape[0], None if vn is None else vn.ctypes.data,
                                        rtg.BUILD_STRATEGIES[strategy])

    @property
    def handle(self):
        return self._h

    @property
    def n_nodes(self):
        return int(lib.orc_bvh_n_nodes(self._h))

    @property
    def n_leaves(self):
        return int(lib.orc_bvh_n_leaves(self._h))

    @property
    def root(self):
        return int(lib.orc_bvh_root(self._h))

    @property
    def depth(self):
        return int(lib.orc_bvh_depth(self._h))

    def nodes(self):
        buf = (C.c_char * (self.n_nodes * 64)).from_address(lib.orc_bvh_nodes(self._h))
        return np.frombuffer(buf, dtype=rtg.NODE_DTYPE).copy()

    def leaves(self):
        if self.n_leaves == 0:
            return np.zeros((0, 3, 3))
        buf = (C.c_char * (self.n_leaves * 72)).from_address(lib.orc_bvh_leaves(self._h))
        return np.frombuffer(buf, dtype=np.float64).reshape(-1, 3, 3).copy()

    def leaf_source(self):
        p = lib.orc_bvh_leaf_source(self._h)
        return np.ctypeslib.as_array(p, shape=(self.n_leaves,)).copy() if self.n_leaves else np.zeros(0, np.uint32)

    def flat(self):
        """rtg_bvh view over copies of the oracle's arrays (to hand to the GPU library)."""
        vn = None
        p = lib.orc_bvh_vertex_normals(self._h)
        if p:
            buf = (C.c_char * (self.n_leaves * 72)).from_address(p)
            vn = np.frombuffer(buf, dtype=np.float64).copy()
        return rtg.flat_bvh(self.nodes(), self.leaves(), self.root, vn)

    def set_patches(self, patch_set):
        self._patch_keep = patch_set
        lib.orc_bvh_set_patches(self._h, C.byref(patch_set))

    def raycast(self, ray, counters=False):
        hit = np.zeros(1, dtype=rtg.HIT_DTYPE)
        ctr = np.zeros(1, dtype=rtg.COUNTER_DTYPE)
        lib.orc_bvh_raycast(self._h, d(ray), hit.ctypes.data, ctr.ctypes.data)
        return (hit[0], ctr[0]) if counters else hit[0]

    def raycast_brute(self, ray):
        hit = np.zeros(1, dtype=rtg.HIT_DTYPE)
        lib.orc_bvh_raycast_brute(self._h, d(ray), hit.ctypes.data)
        return hit[0]

    def close(self):
        if self._h:
            lib.orc_bvh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown
            pass


class OracleScene:
    """objects: rtg.Object list whose `mesh` fields hold OracleBvh handles."""
    _h = None

    def __init__(self, objects):
        self._objects = list(objects)
        arr = (rtg.Object * max(1, len(objects)))()
        for i, o in enumerate(objects):
            C.memmove(C.byref(arr[i]), C.byref(o), C.sizeof(rtg.Object))
        desc = rtg.SceneDesc(arr, len(objects))
        self._h = lib.orc_scene_create(C.byref(desc))

    def render_primary(self, view, width, height, tile=None, points=False, counters=False, threads=THREADS):
        x0, y0, w, h = tile if tile is not None else (0, 0, width, height)
        n = w * h
        hits = np.empty(n, dtype=rtg.HIT_DTYPE)
        pts = np.empty(n, dtype=rtg.POINT_DTYPE) if points else None
        ctr = np.empty(n, dtype=rtg.COUNTER_DTYPE) if counters else None
        lib.orc_render_primary(self._h, C.byref(view), width, height, rtg.Rect(x0, y0, w, h), hits.ctypes.data,
                               None if pts is None else pts.ctypes.data, None if ctr is None else ctr.ctypes.data, threads)
        return hits, pts, ctr

    def trace_paths(self, rays6, max_depth=4):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        n = rays6.shape[0]
        out = np.zeros(n * ((1 << max_depth) - 1), dtype=rtg.PATH_DTYPE)
        count = C.c_uint64()
        lib.orc_trace_paths(self._h, rays6.ctypes.data, n, max_depth, out.ctypes.data, out.shape[0], C.byref(count))
        return out[:count.value]

    def trace(self, rays6, points=False, counters=False, threads=THREADS):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        n = rays6.shape[0]
        hits = np.empty(n, dtype=rtg.HIT_DTYPE)
        pts = np.empty(n, dtype=rtg.POINT_DTYPE) if points else None
        ctr = np.empty(n, dtype=rtg.COUNTER_DTYPE) if counters else None
        lib.orc_trace(self._h, rays6.ctypes.data, n, hits.ctypes.data, None if pts is None else pts.ctypes.data,
                      None if ctr is None else ctr.ctypes.data, threads)
        return hits, pts, ctr

    def shade_direct(self, lights, points, rgba8=False, threads=THREADS):
        lights = np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE)
        points = np.ascontiguousarray(points, dtype=rtg.POINT_DTYPE)
        n = points.shape[0]
        rgb = np.empty((n, 3), dtype=np.float64)
        c8 = np.empty((n, 4), dtype=np.uint8) if rgba8 else None
        lib.orc_shade_direct(self._h, lights.ctypes.data, lights.shape[0], points.ctypes.data, n, rgb.ctypes.data,
                             None if c8 is None else c8.ctypes.data, threads)
        return rgb, c8

    def render_frame(self, view, lights, width, height, tile=None, max_depth=4, rgba8=True, hits=True, threads=THREADS):
        lights = np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE)
        x0, y0, w, h = tile if tile is not None else (0, 0, width, height)
        n = w * h
        rgb = np.empty((n, 3), dtype=np.float64)
        c8 = np.empty((n, 4), dtype=np.uint8) if rgba8 else None
        hh = np.empty(n, dtype=rtg.HIT_DTYPE) if hits else None
        lib.orc_render_frame(self._h, C.byref(view), lights.ctypes.data, lights.shape[0], width, height,
                             rtg.Rect(x0, y0, w, h), max_depth, rgb.ctypes.data, None if c8 is None else c8.ctypes.data,
                             None if hh is None else hh.ctypes.data, threads)
        return rgb, c8, hh

    def photons_emit(self, lights, dirs2, max_depth=4):
        """The photon pass with the directions supplied: dirs2 is (n_lights, count, 2)."""
        lights = np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE)
        dirs2 = np.ascontiguousarray(dirs2, dtype=np.float64).reshape(lights.shape[0], -1, 2)
        cap = dirs2.shape[0] * dirs2.shape[1] * ((1 << max_depth) - 1)
        out = np.zeros(max(cap, 1), dtype=rtg.PHOTON_DTYPE)
        count = C.c_uint64()
        lib.orc_photons_emit(self._h, lights.ctypes.data, lights.shape[0], dirs2.ctypes.data, dirs2.shape[1], max_depth,
                             out.ctypes.data, cap, C.byref(count))
        return out[:count.value].copy()

    def indirect_irrad(self, kd, photons, points, photon_samples=3, newton_steps=5, newton_epsilon=1e-5, max_depth=4,
                       threads=THREADS):
        photons = np.ascontiguousarray(photons, dtype=rtg.PHOTON_DTYPE)
        points = np.ascontiguousarray(points, dtype=rtg.POINT_DTYPE)
        rgb = np.empty((points.shape[0], 3), dtype=np.float64)
        lib.orc_indirect_irrad(self._h, kd.handle, photons.ctypes.data, points.ctypes.data, points.shape[0], photon_samples,
                               newton_steps, float(newton_epsilon), max_depth, rgb.ctypes.data, threads)
        return rgb

    def render_frame_photons(self, view, lights, kd, photons, width, height, tile=None, max_depth=4, photon_samples=3,
                             newton_steps=5, newton_epsilon=1e-5, threads=THREADS):
        lights = np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE)
        photons = np.ascontiguousarray(photons, dtype=rtg.PHOTON_DTYPE)
        x0, y0, w, h = tile if tile is not None else (0, 0, width, height)
        rgb = np.empty((w * h, 3), dtype=np.float64)
        c8 = np.empty((w * h, 4), dtype=np.uint8)
        lib.orc_render_frame_photons(self._h, C.byref(view), lights.ctypes.data, lights.shape[0], kd.handle, photons.ctypes.data,
                                     photon_samples, newton_steps, float(newton_epsilon), width, height, rtg.Rect(x0, y0, w, h),
                                     max_depth, rgb.ctypes.data, c8.ctypes.data, threads)
        return rgb, c8

    def photon_adjust(self, photon, point, newton_steps=5, max_depth=4):
        """Newton iteration + final trace of one photon against one point (see orc_photon_adjust)."""
        ph = np.ascontiguousarray(np.asarray(photon, dtype=rtg.PHOTON_DTYPE).reshape(1))
        pt = np.ascontiguousarray(np.asarray(point, dtype=rtg.POINT_DTYPE).reshape(1))
        out = np.empty(12, dtype=np.float64)
        seps = np.empty(max(2 * newton_steps, 1), dtype=np.float64)
        lib.orc_photon_adjust(self._h, ph.ctypes.data, pt.ctypes.data, newton_steps, max_depth, out.ctypes.data, seps.ctypes.data)
        return {"dir": out[0:2], "landed": out[2] != 0.0, "position": out[3:6], "d0": out[6:9], "d1": out[9:12],
                "seps": seps[:2 * newton_steps].reshape(-1, 2)}

    def close(self):
        if self._h:
            lib.orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown
            pass


def oracle_object(obj, oracle_bvh=None):
    """Copy of an rtg.Object whose mesh handle points at an OracleBvh."""
    c = rtg.copy_object(obj)
    c.mesh = oracle_bvh.handle if oracle_bvh is not None else None
    c._keep = oracle_bvh
    return c
