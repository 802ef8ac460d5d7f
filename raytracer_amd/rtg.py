"""ctypes binding of include/rtg.h (librtg.so) and of the host library (librt_host.so).

Only plain pointers and sizes cross the boundary.  Arrays are numpy arrays
(host memory) or raw device addresses (ints, e.g. ``torch.Tensor.data_ptr()``).
"""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

RTG_NONE = 0xFFFFFFFF
SPHERE, MESH, PLANE, PATCHES = 0, 1, 2, 3


def _load(name):
    # RTG_LIBRTG_PATH: A/B a differently built kernel library without touching the tree (diagnostics only)
    path = os.environ.get("RTG_LIBRTG_PATH") if name == "librtg.so" and os.environ.get("RTG_LIBRTG_PATH") else os.path.join(_HERE, name)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the ray-cast path)")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


lib = _load("librtg.so")
host = _load("librt_host.so")


# ---- PODs of include/rtg.h -------------------------------------------------
class BvhNode(C.Structure):
    _fields_ = [("lo", C.c_double * 3), ("hi", C.c_double * 3), ("child_begin", C.c_uint32),
                ("child_end", C.c_uint32), ("leaf_begin", C.c_uint32), ("leaf_end", C.c_uint32)]


class Bvh(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("n_nodes", C.c_uint64), ("leaves", C.c_void_p), ("n_leaves", C.c_uint64),
                ("vertex_normals", C.c_void_p), ("root", C.c_uint32)]


class PatchSet(C.Structure):
    _fields_ = [("patches", C.c_void_p), ("n_patches", C.c_uint64), ("leaf_info", C.c_void_p),
                ("newton_steps", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("diffuse", C.c_double * 3), ("n1", C.c_double), ("n2", C.c_double), ("has_dielectric", C.c_int32),
                ("_pad", C.c_int32)]


class Transform(C.Structure):
    _fields_ = [("forward", C.c_double * 16), ("reverse", C.c_double * 16), ("forward_norm", C.c_double * 9)]


class Object(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("transform", Transform), ("material", Material),
                ("material2", Material), ("sphere_center", C.c_double * 3), ("sphere_radius", C.c_double),
                ("plane_position", C.c_double * 3), ("plane_tan1", C.c_double * 3), ("plane_tan2", C.c_double * 3),
                ("mesh", C.c_void_p)]


class SceneDesc(C.Structure):
    _fields_ = [("objects", C.POINTER(Object)), ("n_objects", C.c_uint32)]


class View(C.Structure):
    _fields_ = [("orig", C.c_double * 3), ("tan_half_fov", C.c_double * 2)]


class Light(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("color", C.c_double * 3)]


class Rect(C.Structure):
    _fields_ = [("x0", C.c_uint32), ("y0", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32)]


class Piece(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("_pad", C.c_uint32), ("send_offset", C.c_uint64), ("recv_offset", C.c_uint64),
                ("bytes", C.c_uint64)]


COMM_ID_BYTES = 128
STATIC_MAX_PIXELS = 1 << 21


NODE_DTYPE = np.dtype([("lo", "<f8", 3), ("hi", "<f8", 3), ("child_begin", "<u4"), ("child_end", "<u4"),
                       ("leaf_begin", "<u4"), ("leaf_end", "<u4")])
HIT_DTYPE = np.dtype([("object", "<u4"), ("leaf", "<u4"), ("t", "<f8")])
POINT_DTYPE = np.dtype([("time", "<f8"), ("position", "<f8", 3), ("inter_normal", "<f8", 3),
                        ("geo_normal", "<f8", 3), ("manifold_point", "<f8", 2), ("manifold", "<u4", 2),
                        ("manifold_len", "<u4"), ("material_slot", "<u4")])
PATH_DTYPE = np.dtype([("ray", "<u4"), ("depth", "<u4"), ("modes", "<u4"), ("parent", "<u4"), ("attenuation", "<f8"),
                       ("_pad", "<f8"), ("point", POINT_DTYPE)])
COUNTER_DTYPE = np.dtype([("nodes_visited", "<u4"), ("tris_tested", "<u4")])
PATCH_LEAF_DTYPE = np.dtype([("tu", "<f8", (3, 2)), ("patch", "<u4"), ("_pad", "<u4")])
LIGHT_DTYPE = np.dtype([("position", "<f8", 3), ("color", "<f8", 3)])
PHOTON_DTYPE = np.dtype([("position", "<f8", 3), ("origin", "<f8", 3), ("dir", "<f8", 2), ("light", "<f8", 3),
                         ("surface_object", "<u4", 3), ("surface_leaf", "<u4", 3), ("n_bounces", "<u4"),
                         ("modes", "<u4"), ("light_index", "<u4"), ("_pad", "<u4")])
assert PHOTON_DTYPE.itemsize == 128
assert NODE_DTYPE.itemsize == 64 and HIT_DTYPE.itemsize == 16 and POINT_DTYPE.itemsize == 112
assert C.sizeof(BvhNode) == 64 and C.sizeof(Transform) == 328 and C.sizeof(Material) == 48


class RtgError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rtg error {code}: {message}")
        self.code = code


def _sig(fn, restype, *argtypes):
    fn.restype = restype
    fn.argtypes = list(argtypes)


_vp = C.c_void_p
_sig(lib.rtg_version, C.c_char_p)
_sig(lib.rtg_ctx_create, C.c_int, C.c_int, C.POINTER(_vp))
_sig(lib.rtg_ctx_destroy, None, _vp)
_sig(lib.rtg_last_error, C.c_char_p, _vp)
_sig(lib.rtg_ctx_set_stream, C.c_int, _vp, _vp)
_sig(lib.rtg_ctx_get_stream, _vp, _vp)
_sig(lib.rtg_ctx_synchronize, C.c_int, _vp)
_sig(lib.rtg_ctx_set_option, C.c_int, _vp, C.c_char_p, C.c_int64)
_sig(lib.rtg_ctx_kernel_time, C.c_int, _vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64))
_sig(lib.rtg_ctx_debug_stats, C.c_int, _vp, C.POINTER(C.c_uint32))
_sig(lib.rtg_selftest_division, C.c_int, _vp, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64))
_sig(lib.rtg_transform_from_forward, C.c_int, C.POINTER(C.c_double), C.POINTER(Transform))
_sig(lib.rtg_mesh_create, C.c_int, _vp, C.POINTER(Bvh), C.POINTER(_vp))
_sig(lib.rtg_mesh_create_patches, C.c_int, _vp, C.POINTER(Bvh), C.POINTER(PatchSet), C.POINTER(_vp))
_sig(lib.rtg_mesh_destroy, None, _vp)
_sig(lib.rtg_mesh_build, C.c_int, _vp, _vp, C.c_uint64, _vp, C.POINTER(_vp))
_sig(lib.rtg_mesh_build_patches, C.c_int, _vp, _vp, C.c_uint64, C.POINTER(PatchSet), C.POINTER(_vp))
_sig(lib.rtg_mesh_sizes, C.c_int, _vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
     C.POINTER(C.c_uint32))
_sig(lib.rtg_mesh_read_bvh, C.c_int, _vp, _vp, _vp, _vp)
_sig(lib.rtg_mesh_bounds, C.c_int, _vp, C.POINTER(C.c_double))
_sig(lib.rtg_scene_create, C.c_int, _vp, C.POINTER(SceneDesc), C.POINTER(_vp))
_sig(lib.rtg_scene_destroy, None, _vp)
_sig(lib.rtg_render_primary, C.c_int, _vp, C.POINTER(View), C.c_uint32, C.c_uint32, Rect, _vp, _vp, _vp)
_sig(lib.rtg_render_primary_static, C.c_int, _vp, C.POINTER(View), C.c_uint32, C.c_uint32, Rect, _vp)
_sig(lib.rtg_trace, C.c_int, _vp, _vp, C.c_uint64, _vp, _vp, _vp)
_sig(lib.rtg_trace_paths, C.c_int, _vp, _vp, C.c_uint64, C.c_uint32, _vp, C.c_uint64, C.POINTER(C.c_uint64))
_sig(lib.rtg_shade_direct, C.c_int, _vp, _vp, C.c_uint32, _vp, C.c_uint64, _vp, _vp)
_sig(lib.rtg_render_frame, C.c_int, _vp, C.POINTER(View), _vp, C.c_uint32, C.c_uint32, C.c_uint32, Rect,
     C.c_uint32, _vp, _vp, _vp)
_sig(lib.rtg_photons_emit, C.c_int, _vp, _vp, C.c_uint32, _vp, C.c_uint64, C.c_uint32, _vp, C.c_uint64, C.POINTER(C.c_uint64))
_sig(lib.rtg_photon_map_create, C.c_int, _vp, _vp, C.c_uint64, C.POINTER(_vp))
_sig(lib.rtg_photon_map_emit, C.c_int, _vp, _vp, C.c_uint32, _vp, C.c_uint64, C.c_uint32, C.POINTER(_vp), C.POINTER(C.c_uint64))
_sig(lib.rtg_photon_map_emit_fibonacci, C.c_int, _vp, _vp, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(_vp),
     C.POINTER(C.c_uint64))
_sig(lib.rtg_photon_map_destroy, None, _vp)
_sig(lib.rtg_photon_map_nearest, C.c_int, _vp, _vp, C.c_uint64, C.c_uint32, _vp, _vp)
_sig(lib.rtg_photon_map_order, C.c_int, _vp, _vp)
_sig(lib.rtg_render_frame_photons, C.c_int, _vp, C.POINTER(View), _vp, C.c_uint32, _vp, C.c_uint32, C.c_uint32, C.c_double,
     C.c_uint32, C.c_uint32, Rect, C.c_uint32, _vp, _vp)
_sig(lib.rtg_indirect_irrad, C.c_int, _vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, _vp)

_sig(lib.rtg_comm_unique_id, C.c_int, _vp)
_sig(lib.rtg_comm_create, C.c_int, _vp, _vp, C.c_int, C.c_int, C.POINTER(_vp))
_sig(lib.rtg_comm_adopt, C.c_int, _vp, _vp, C.c_int, C.c_int, C.POINTER(_vp))
_sig(lib.rtg_comm_destroy, None, _vp)
_sig(lib.rtg_comm_info, C.c_int, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int))
_sig(lib.rtg_gather_tiles, C.c_int, _vp, C.c_int, _vp, _vp, C.POINTER(Piece), C.c_uint32, _vp)

_sig(host.rth_bvh_build_subdivide, _vp, _vp, C.c_uint64, _vp, C.c_int)
_sig(host.rth_bvh_build, _vp, _vp, C.c_uint64, _vp, C.c_int, C.c_int)
_sig(host.rth_bvh_destroy, None, _vp)
_sig(host.rth_bvh_view, None, _vp, C.POINTER(Bvh))
_sig(host.rth_bvh_leaf_source, C.POINTER(C.c_uint32), _vp)
_sig(host.rth_bvh_depth, C.c_uint32, _vp)
_sig(host.rth_heightfield, None, C.c_uint32, C.c_uint32, C.c_uint64, C.c_double, _vp)
_sig(host.rth_fibonacci_sphere, None, C.c_uint64, C.c_uint64, _vp)
_sig(host.rth_fibonacci_zenith_y, None, C.c_uint64, C.c_uint64, _vp)
_sig(host.rth_bench_mesh_rays, None, C.POINTER(C.c_double), C.c_uint64, C.c_uint64, _vp)
_sig(host.rth_sort_rays, None, C.POINTER(C.c_double), _vp, C.c_uint64, _vp, _vp)
_sig(host.rth_trispline_from_normal_hermite, None, _vp, _vp)
_sig(host.rth_trispline_eval, None, _vp, C.c_double, C.c_double, _vp)
_sig(host.rth_patches_tessellate, None, _vp, C.c_uint64, C.c_uint32, _vp, _vp)
_sig(host.rth_patch_terrain, None, C.c_uint32, C.c_uint32, C.c_uint64, C.c_double, _vp)
_sig(host.rth_transform_build, None, C.c_double, C.POINTER(C.c_double), C.POINTER(Transform))
_sig(host.rth_scene_builder, None, C.c_uint64, C.POINTER(Object), C.POINTER(Light), C.POINTER(View))
_sig(host.rth_render_scene_builder, C.c_int, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp)
_sig(host.rth_render_scene_builder_photons, C.c_int, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
     C.c_int, _vp, C.POINTER(C.c_uint64), _vp)
_sig(host.rth_to_hdr, C.c_uint64, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint64)
_sig(host.rth_images_tar, C.c_uint64, _vp, C.c_uint32, C.c_uint32, C.c_double, _vp, _vp, C.c_uint64)
_sig(host.rth_render_images, C.c_int, _vp, C.POINTER(Object), C.c_uint32, _vp, C.c_uint32, C.POINTER(C.c_double),
     C.POINTER(C.c_double), C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, C.POINTER(C.c_double), _vp, C.c_uint64,
     C.POINTER(C.c_uint64))


def _ptr(a):
    """Device address (int), numpy array, or None -> void*."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return C.c_void_p(a.ctypes.data)
    raise TypeError(f"unsupported buffer {type(a)}")


def version():
    return lib.rtg_version().decode()


# ---- host library helpers -----------------------------------------------------
def heightfield(nx, ny, seed=1, scale=1.0):
    """Synthetic terrain mesh: (2*nx*ny, 3, 3) float64 triangle array."""
    tris = np.empty((2 * nx * ny, 3, 3), dtype=np.float64)
    host.rth_heightfield(nx, ny, seed, float(scale), _ptr(tris))
    return tris


def trispline_from_normal_hermite(nh):
    """NormalHermiteTrispline (10 x 3: p0,p1,p2,p01,p12,p20,n0,n1,n2,control) -> Trispline coefficients (10 x 3)."""
    nh = np.ascontiguousarray(nh, dtype=np.float64).reshape(10, 3)
    out = np.empty((10, 3), dtype=np.float64)
    host.rth_trispline_from_normal_hermite(_ptr(nh), _ptr(out))
    return out


def trispline_eval(patch, t, u):
    """(position, ddt, ddu, normal) of one patch at (t, u)."""
    patch = np.ascontiguousarray(patch, dtype=np.float64).reshape(10, 3)
    out = np.empty((4, 3), dtype=np.float64)
    host.rth_trispline_eval(_ptr(patch), float(t), float(u), _ptr(out))
    return out


def patches_tessellate(patches, k):
    """k*k flat facets per patch: ((n*k*k, 3, 3) triangles, PATCH_LEAF_DTYPE info)."""
    patches = np.ascontiguousarray(patches, dtype=np.float64).reshape(-1, 10, 3)
    n = patches.shape[0] * k * k
    tris = np.empty((n, 3, 3), dtype=np.float64)
    info = np.zeros(n, dtype=PATCH_LEAF_DTYPE)
    host.rth_patches_tessellate(_ptr(patches), patches.shape[0], k, _ptr(tris), _ptr(info))
    return tris, info


def patch_terrain(nx, ny, seed=1, scale=1.0):
    """2*nx*ny trispline patches over the heightfield terrain: (n, 10, 3)."""
    out = np.empty((2 * nx * ny, 10, 3), dtype=np.float64)
    host.rth_patch_terrain(nx, ny, seed, float(scale), _ptr(out))
    return out


class HostPatchMesh:
    """Patches + their facet BVH (host-built): what rtg_mesh_create_patches consumes."""

    def __init__(self, patches, k=4, newton_steps=5, threads=0):
        self.patches = np.ascontiguousarray(patches, dtype=np.float64).reshape(-1, 10, 3)
        tris, info = patches_tessellate(self.patches, k)
        self.bvh = HostBvh(tris, threads=threads)
        self.leaf_info = np.ascontiguousarray(info[self.bvh.leaf_source()])   # facet order -> leaf order
        self.newton_steps = newton_steps
        self.k = k

    def patch_set(self):
        ps = PatchSet()
        ps.patches = self.patches.ctypes.data
        ps.n_patches = self.patches.shape[0]
        ps.leaf_info = self.leaf_info.ctypes.data
        ps.newton_steps = self.newton_steps
        ps._keep = self
        return ps


def fibonacci_sphere(count, seed=1):
    """Sphere::fibonacci_sphere + into_normal: (count, 3) unit directions for the photon pass."""
    out = np.empty((count, 3), dtype=np.float64)
    host.rth_fibonacci_sphere(count, seed, _ptr(out))
    return out


def fibonacci_zenith_y(count, seed=1):
    """Sphere::fibonacci_sphere as the photon pass consumes it: (count, 2) `ZenithY` (zenith, y) pairs."""
    out = np.empty((count, 2), dtype=np.float64)
    host.rth_fibonacci_zenith_y(count, seed, _ptr(out))
    return out


def bench_mesh_rays(lo_hi, n, seed=10212233):
    b = (C.c_double * 6)(*[float(x) for x in lo_hi])
    rays = np.empty((n, 6), dtype=np.float64)
    host.rth_bench_mesh_rays(b, n, seed, _ptr(rays))
    return rays


def sort_rays(lo_hi, rays):
    b = (C.c_double * 6)(*[float(x) for x in lo_hi])
    n = rays.shape[0]
    perm = np.empty(n, dtype=np.uint32)
    out = np.empty_like(rays)
    host.rth_sort_rays(b, _ptr(rays), n, _ptr(perm), _ptr(out))
    return perm, out


def transform(scale=1.0, translate=(0.0, 0.0, 0.0)):
    t = Transform()
    host.rth_transform_build(float(scale), (C.c_double * 3)(*translate), C.byref(t))
    return t


def transform_from_forward(forward16):
    t = Transform()
    f = (C.c_double * 16)(*[float(x) for x in np.asarray(forward16).reshape(16)])
    rc = lib.rtg_transform_from_forward(f, C.byref(t))
    if rc != 0:
        raise RtgError(rc, "rtg_transform_from_forward")
    return t


def material(diffuse=(0.0, 0.0, 0.0), dielectric=None):
    m = Material()
    m.diffuse[:] = diffuse
    if dielectric is not None:
        m.has_dielectric = 1
        m.n1, m.n2 = dielectric
    return m


def scene_builder(time=0):
    """SceneBuilder{time}: ([sphere, plane] objects, [2 lights] as LIGHT_DTYPE array, View)."""
    objs = (Object * 2)()
    lights = (Light * 2)()
    view = View()
    host.rth_scene_builder(time, objs, lights, C.byref(view))
    la = np.zeros(2, dtype=LIGHT_DTYPE)
    for i in range(2):
        la[i]["position"] = list(lights[i].position)
        la[i]["color"] = list(lights[i].color)
    return [objs[0], objs[1]], la, view


def copy_object(o):
    c = Object()
    C.memmove(C.byref(c), C.byref(o), C.sizeof(Object))
    return c


BUILD_STRATEGIES = {"subdivide": 0, "contract": 1, "hybrid": 2}


class HostBvh:
    """BvhForest::new(tris).subdivide() (or .contract() / .hybrid()) + Bvh::new -- flattened on the host."""
    _h = None

    def __init__(self, tris, vertex_normals=None, threads=0, strategy="subdivide"):
        tris = np.ascontiguousarray(tris, dtype=np.float64).reshape(-1, 3, 3)
        self._tris = tris
        self._normals = None if vertex_normals is None else np.ascontiguousarray(vertex_normals, dtype=np.float64)
        self._h = host.rth_bvh_build(_ptr(tris), tris.shape[0], _ptr(self._normals), threads, BUILD_STRATEGIES[strategy])
        self.view = Bvh()
        host.rth_bvh_view(self._h, C.byref(self.view))

    @property
    def n_nodes(self):
        return int(self.view.n_nodes)

    @property
    def n_leaves(self):
        return int(self.view.n_leaves)

    @property
    def root(self):
        return int(self.view.root)

    @property
    def depth(self):
        return int(host.rth_bvh_depth(self._h))

    def nodes(self):
        buf = (C.c_char * (self.n_nodes * 64)).from_address(self.view.nodes)
        return np.frombuffer(buf, dtype=NODE_DTYPE).copy()

    def leaves(self):
        if self.n_leaves == 0:
            return np.zeros((0, 3, 3))
        buf = (C.c_char * (self.n_leaves * 72)).from_address(self.view.leaves)
        return np.frombuffer(buf, dtype=np.float64).reshape(-1, 3, 3).copy()

    def leaf_source(self):
        p = host.rth_bvh_leaf_source(self._h)
        return np.ctypeslib.as_array(p, shape=(self.n_leaves,)).copy() if self.n_leaves else np.zeros(0, np.uint32)

    def bounds(self):
        n = self.nodes()[self.root]
        return np.concatenate([n["lo"], n["hi"]])

    def close(self):
        if self._h:
            host.rth_bvh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: the library globals may already be gone
            pass


def flat_bvh(nodes, leaves, root, vertex_normals=None):
    """Wrap caller-flattened arrays (NODE_DTYPE nodes, (n,3,3) leaves) as an rtg_bvh; keeps references."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    leaves = np.ascontiguousarray(leaves, dtype=np.float64).reshape(-1, 3, 3)
    v = Bvh()
    v.nodes = nodes.ctypes.data if nodes.size else None
    v.n_nodes = nodes.shape[0]
    v.leaves = leaves.ctypes.data if leaves.size else None
    v.n_leaves = leaves.shape[0]
    if vertex_normals is not None:
        vertex_normals = np.ascontiguousarray(vertex_normals, dtype=np.float64)
        v.vertex_normals = vertex_normals.ctypes.data
    v.root = root
    v._keep = (nodes, leaves, vertex_normals)
    return v


# ---- the GPU path ----------------------------------------------------------------
class Context:
    _h = None

    def __init__(self, device=0):
        h = _vp()
        rc = lib.rtg_ctx_create(device, C.byref(h))
        if rc != 0:
            raise RtgError(rc, lib.rtg_last_error(None).decode())
        self._h = h
        self.device = device
        self._scenes = weakref.WeakSet()      # scenes run on their context: they are closed before it
        self._comms = weakref.WeakSet()       # so are communicators

    def _check(self, rc):
        if rc != 0:
            raise RtgError(rc, lib.rtg_last_error(self._h).decode())

    def set_option(self, key, value):
        self._check(lib.rtg_ctx_set_option(self._h, key.encode(), int(value)))

    def set_stream(self, stream_handle):
        self._check(lib.rtg_ctx_set_stream(self._h, C.c_void_p(stream_handle) if stream_handle else None))

    @property
    def stream(self):
        return lib.rtg_ctx_get_stream(self._h) or 0

    def synchronize(self):
        self._check(lib.rtg_ctx_synchronize(self._h))

    def selftest_division(self, n_pairs, family, seed=1):
        """(mismatches, x_bits, d_bits) of the reciprocal-based quotient vs the hardware division."""
        out = (C.c_uint64 * 3)()
        self._check(lib.rtg_selftest_division(self._h, seed, n_pairs, family, out))
        return int(out[0]), int(out[1]), int(out[2])

    def kernel_time(self):
        """(summed execution windows in ms, launches) of the timed persistent-kernel launches since the last call."""
        ms, n = C.c_double(), C.c_uint64()
        self._check(lib.rtg_ctx_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def debug_stats(self):
        out = (C.c_uint32 * 16)()
        self._check(lib.rtg_ctx_debug_stats(self._h, out))
        names = ["iterations", "node_steps", "node_lanes", "tri_steps", "tri_lanes", "obj_steps", "obj_lanes",
                 "refills", "waves", "ambiguous_steps", "ambiguous_lanes", "node_ticks64", "tri_ticks64", "obj_ticks64",
                 "wave_ticks64", "node_lines"]
        return dict(zip(names, list(out)))

    def mesh(self, bvh):
        return Mesh(self, bvh)

    def build_mesh(self, tris, vertex_normals=None):
        """BvhForest::new(tris).subdivide() + Bvh::new on the GPU; `tris` / `vertex_normals` are numpy arrays or
        device addresses (then `tris` is (address, count))."""
        return Mesh.build(self, tris, vertex_normals)

    def scene(self, objects):
        return Scene(self, objects)

    def close(self):
        if self._h:
            for sc in list(self._scenes) + list(self._comms):
                sc.close()
            lib.rtg_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: the library globals may already be gone
            pass


def comm_unique_id():
    """RCCL unique id (bytes) made by one rank and handed to the others."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    rc = lib.rtg_comm_unique_id(buf)
    if rc != 0:
        raise RtgError(rc, lib.rtg_last_error(None).decode())
    return bytes(buf)


class Comm:
    """The ranks that gather finished output (one per GPU); a one-rank communicator needs no id and no RCCL."""
    _h = None

    def __init__(self, ctx, n_ranks=1, rank=0, unique_id=None):
        h = _vp()
        idbuf = None
        if unique_id is not None:
            assert len(unique_id) == COMM_ID_BYTES
            idbuf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        ctx._check(lib.rtg_comm_create(ctx._h, idbuf, n_ranks, rank, C.byref(h)))
        self._h, self.ctx, self.n_ranks, self.rank = h, ctx, n_ranks, rank
        ctx._comms.add(self)

    def gather(self, root, send_ptr, recv_ptr, pieces, stream_handle=None):
        """pieces: [(rank, send_offset, recv_offset, bytes)], the same list on every rank."""
        arr = (Piece * max(1, len(pieces)))()
        for i, (r, so, ro, nb) in enumerate(pieces):
            arr[i].rank, arr[i].send_offset, arr[i].recv_offset, arr[i].bytes = r, so, ro, nb
        self.ctx._check(lib.rtg_gather_tiles(self._h, root, _ptr(send_ptr), _ptr(recv_ptr), arr, len(pieces),
                                             C.c_void_p(stream_handle) if stream_handle else None))

    def info(self):
        """(ranks, this rank, ranks RCCL's communicator reports -- 0 when one rank gathers by device copies)."""
        n, r, c = C.c_int(), C.c_int(), C.c_int()
        self.ctx._check(lib.rtg_comm_info(self._h, C.byref(n), C.byref(r), C.byref(c)))
        return n.value, r.value, c.value

    def close(self):
        if self._h:
            lib.rtg_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown
            pass


class PhotonMap:
    """KdTree<Photon> on the GPU: nearest photons of a batch of points; handed to Scene.indirect_irrad."""
    _h = None

    def __init__(self, ctx, photons, n=None, _handle=None):
        """photons: PHOTON_DTYPE records, or (with n) the address of n records in host or device memory."""
        if _handle is not None:            # made by Scene.photon_map
            self.photons = None
            self._h, self.ctx, self.n = _handle, ctx, n
            ctx._scenes.add(self)
            return
        h = _vp()
        if n is None:
            self.photons = np.ascontiguousarray(photons, dtype=PHOTON_DTYPE)
            n = self.photons.shape[0]
            ptr = _ptr(self.photons) if n else None
        else:
            self.photons = None
            ptr = _ptr(photons) if n else None
        ctx._check(lib.rtg_photon_map_create(ctx._h, ptr, n, C.byref(h)))
        self._h, self.ctx, self.n = h, ctx, n
        ctx._scenes.add(self)              # closed before the context, like a scene

    def nearest(self, queries3, k):
        q = np.ascontiguousarray(queries3, dtype=np.float64).reshape(-1, 3)
        idx = np.empty((q.shape[0], k), dtype=np.uint32)
        dist = np.empty((q.shape[0], k), dtype=np.float64)
        self.ctx._check(lib.rtg_photon_map_nearest(self._h, _ptr(q), q.shape[0], k, _ptr(idx), _ptr(dist)))
        return idx, dist

    def order(self):
        """The photon index at every position of the tree's entry array (KdTree's own order)."""
        out = np.empty(self.n, dtype=np.uint32)
        self.ctx._check(lib.rtg_photon_map_order(self._h, _ptr(out) if self.n else None))
        return out

    def nearest_raw(self, queries_ptr, n, k, index_ptr, distance_ptr):
        self.ctx._check(lib.rtg_photon_map_nearest(self._h, _ptr(queries_ptr), n, k, _ptr(index_ptr), _ptr(distance_ptr)))

    def close(self):
        if self._h:
            lib.rtg_photon_map_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown
            pass


class Mesh:
    _h = None

    def __init__(self, ctx, bvh):
        h = _vp()
        if isinstance(bvh, HostPatchMesh):
            ps = bvh.patch_set()
            ctx._check(lib.rtg_mesh_create_patches(ctx._h, C.byref(bvh.bvh.view), C.byref(ps), C.byref(h)))
        else:
            view = bvh.view if isinstance(bvh, HostBvh) else bvh
            ctx._check(lib.rtg_mesh_create(ctx._h, C.byref(view), C.byref(h)))
        self._keep = bvh
        self._h = h
        self.ctx = ctx

    @classmethod
    def build(cls, ctx, tris, vertex_normals=None):
        self = cls.__new__(cls)
        if isinstance(tris, tuple):
            addr, n = tris
            keep = None
        else:
            tris = np.ascontiguousarray(tris, dtype=np.float64).reshape(-1, 3, 3)
            addr, n, keep = tris, tris.shape[0], tris
        if isinstance(vertex_normals, np.ndarray):
            vertex_normals = np.ascontiguousarray(vertex_normals, dtype=np.float64)
        h = _vp()
        ctx._check(lib.rtg_mesh_build(ctx._h, _ptr(addr), n, _ptr(vertex_normals), C.byref(h)))
        self._keep = keep
        self._h = h
        self.ctx = ctx
        return self

    @classmethod
    def build_patches(cls, ctx, patches, k=4, newton_steps=5):
        """Patches tessellated into k*k facets each, the facet tree built on the GPU."""
        self = cls.__new__(cls)
        patches = np.ascontiguousarray(patches, dtype=np.float64).reshape(-1, 10, 3)
        tris, info = patches_tessellate(patches, k)
        info = np.ascontiguousarray(info)
        ps = PatchSet()
        ps.patches = patches.ctypes.data
        ps.n_patches = patches.shape[0]
        ps.leaf_info = info.ctypes.data
        ps.newton_steps = newton_steps
        h = _vp()
        ctx._check(lib.rtg_mesh_build_patches(ctx._h, _ptr(tris), tris.shape[0], C.byref(ps), C.byref(h)))
        self._keep = None
        self._h = h
        self.ctx = ctx
        return self

    @property
    def handle(self):
        return self._h.value

    def bounds(self):
        out = (C.c_double * 6)()
        self.ctx._check(lib.rtg_mesh_bounds(self._h, out))
        return np.array(list(out))

    def sizes(self):
        """(n_nodes, n_leaves, root, depth) of the flattened tree."""
        nn, nl, root, depth = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint32()
        self.ctx._check(lib.rtg_mesh_sizes(self._h, C.byref(nn), C.byref(nl), C.byref(root), C.byref(depth)))
        return nn.value, nl.value, root.value, depth.value

    def read_bvh(self):
        """(nodes NODE_DTYPE, leaves (n,3,3), leaf_source u32) copied back from the device."""
        nn, nl, _, _ = self.sizes()
        nodes = np.zeros(nn, dtype=NODE_DTYPE)
        leaves = np.zeros((nl, 3, 3))
        source = np.zeros(nl, dtype=np.uint32)
        rc = lib.rtg_mesh_read_bvh(self._h, _ptr(nodes), _ptr(leaves), _ptr(source))
        if rc != 0:
            raise RtgError(rc, lib.rtg_last_error(None).decode())
        return nodes, leaves, source

    def close(self):
        if self._h:
            lib.rtg_mesh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: the library globals may already be gone
            pass


def sphere_object(center, radius, mat, xform=None):
    o = Object()
    o.kind = SPHERE
    o.transform = xform if xform is not None else transform()
    o.material = mat
    o.sphere_center[:] = center
    o.sphere_radius = radius
    return o


def plane_object(position, tan1, tan2, mat1, mat2, xform=None):
    o = Object()
    o.kind = PLANE
    o.transform = xform if xform is not None else transform()
    o.material = mat1
    o.material2 = mat2
    o.plane_position[:] = position
    o.plane_tan1[:] = tan1
    o.plane_tan2[:] = tan2
    return o


def mesh_object(mesh, mat, xform=None):
    o = Object()
    o.kind = MESH
    o.transform = xform if xform is not None else transform()
    o.material = mat
    o.mesh = mesh.handle if isinstance(mesh, Mesh) else mesh
    o._keep = mesh
    return o


def patch_object(mesh, mat, xform=None):
    o = mesh_object(mesh, mat, xform)
    o.kind = PATCHES
    return o


class Scene:
    _h = None

    def __init__(self, ctx, objects):
        self.ctx = ctx
        self._objects = list(objects)
        arr = (Object * max(1, len(objects)))()
        for i, o in enumerate(objects):
            C.memmove(C.byref(arr[i]), C.byref(o), C.sizeof(Object))
        desc = SceneDesc(arr, len(objects))
        h = _vp()
        ctx._check(lib.rtg_scene_create(ctx._h, C.byref(desc), C.byref(h)))
        self._h = h
        ctx._scenes.add(self)

    def render_primary(self, view, width, height, tile=None, points=False, counters=False, out_hits=None):
        """Host-buffer convenience: returns (hits, points|None, counters|None) as numpy arrays over the tile."""
        x0, y0, w, h = tile if tile is not None else (0, 0, width, height)
        n = w * h
        hits = np.empty(n, dtype=HIT_DTYPE) if out_hits is None else out_hits
        pts = np.empty(n, dtype=POINT_DTYPE) if points else None
        ctr = np.empty(n, dtype=COUNTER_DTYPE) if counters else None
        self.ctx._check(lib.rtg_render_primary(self._h, C.byref(view), width, height, Rect(x0, y0, w, h),
                                               _ptr(hits), _ptr(pts), _ptr(ctr)))
        return hits, pts, ctr

    def render_primary_raw(self, view, width, height, tile, hits_ptr, points_ptr=None, counters_ptr=None):
        """Raw pointers (device addresses or numpy arrays); asynchronous for device pointers."""
        self.ctx._check(lib.rtg_render_primary(self._h, C.byref(view), width, height, Rect(*tile),
                                               _ptr(hits_ptr), _ptr(points_ptr), _ptr(counters_ptr)))

    def render_primary_static(self, view, width, height, tile, hits_ptr):
        """rtg_render_primary_static: device output only, no allocation inside the call."""
        self.ctx._check(lib.rtg_render_primary_static(self._h, C.byref(view), width, height, Rect(*tile), _ptr(hits_ptr)))

    def trace(self, rays6, points=False, counters=False):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        n = rays6.shape[0]
        hits = np.empty(n, dtype=HIT_DTYPE)
        pts = np.empty(n, dtype=POINT_DTYPE) if points else None
        ctr = np.empty(n, dtype=COUNTER_DTYPE) if counters else None
        self.ctx._check(lib.rtg_trace(self._h, _ptr(rays6), n, _ptr(hits), _ptr(pts), _ptr(ctr)))
        return hits, pts, ctr

    def trace_paths(self, rays6, max_depth=4, capacity=None):
        """raytrace_all_specular per ray: PATH_DTYPE records in ray order, depth-first within a ray."""
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        n = rays6.shape[0]
        if capacity is None:
            capacity = n * ((1 << max_depth) - 1)
        out = np.zeros(capacity, dtype=PATH_DTYPE)
        count = C.c_uint64()
        self.ctx._check(lib.rtg_trace_paths(self._h, _ptr(rays6), n, max_depth, _ptr(out) if capacity else None, capacity,
                                            C.byref(count)))
        return out[:count.value]

    def trace_raw(self, rays_ptr, n, hits_ptr, points_ptr=None, counters_ptr=None):
        self.ctx._check(lib.rtg_trace(self._h, _ptr(rays_ptr), n, _ptr(hits_ptr), _ptr(points_ptr), _ptr(counters_ptr)))

    def shade_direct(self, lights, points, rgba8=False):
        lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        points = np.ascontiguousarray(points, dtype=POINT_DTYPE)
        n = points.shape[0]
        rgb = np.empty((n, 3), dtype=np.float64)
        c8 = np.empty((n, 4), dtype=np.uint8) if rgba8 else None
        self.ctx._check(lib.rtg_shade_direct(self._h, _ptr(lights), lights.shape[0], _ptr(points), n, _ptr(rgb), _ptr(c8)))
        return rgb, c8

    def render_frame(self, view, lights, width, height, tile=None, max_depth=4, rgba8=True, hits=True):
        lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        x0, y0, w, h = tile if tile is not None else (0, 0, width, height)
        n = w * h
        rgb = np.empty((n, 3), dtype=np.float64)
        c8 = np.empty((n, 4), dtype=np.uint8) if rgba8 else None
        hh = np.empty(n, dtype=HIT_DTYPE) if hits else None
        self.ctx._check(lib.rtg_render_frame(self._h, C.byref(view), _ptr(lights), lights.shape[0], width, height,
                                             Rect(x0, y0, w, h), max_depth, _ptr(rgb), _ptr(c8), _ptr(hh)))
        return rgb, c8, hh

    def photons_emit(self, lights, dirs2, max_depth=4, capacity=None):
        """The photon pass: dirs2 is (n_lights, count, 2) `ZenithY` directions; PHOTON_DTYPE records in path order."""
        lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        dirs2 = np.ascontiguousarray(dirs2, dtype=np.float64).reshape(lights.shape[0], -1, 2)
        if capacity is None:
            capacity = dirs2.shape[0] * dirs2.shape[1] * ((1 << max_depth) - 1)
        out = np.zeros(max(capacity, 1), dtype=PHOTON_DTYPE)
        count = C.c_uint64()
        self.ctx._check(lib.rtg_photons_emit(self._h, _ptr(lights), lights.shape[0], _ptr(dirs2), dirs2.shape[1], max_depth,
                                             _ptr(out) if capacity else None, capacity, C.byref(count)))
        return out[:count.value].copy()

    def photon_map(self, lights, dirs2, max_depth=4):
        """The photon pass and KdTree::new in one call, the photons staying on the GPU: a PhotonMap (its .n = photons)."""
        lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        dirs2 = np.ascontiguousarray(dirs2, dtype=np.float64).reshape(lights.shape[0], -1, 2)
        h, count = _vp(), C.c_uint64()
        self.ctx._check(lib.rtg_photon_map_emit(self._h, _ptr(lights), lights.shape[0], _ptr(dirs2), dirs2.shape[1], max_depth,
                                                C.byref(h), C.byref(count)))
        return PhotonMap(self.ctx, None, n=count.value, _handle=h)

    def photon_map_fibonacci(self, lights, count_per_light, seed, max_depth=4):
        """Sphere::fibonacci_sphere per light, the photon pass and KdTree::new, all on the GPU: a PhotonMap."""
        lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        h, count = _vp(), C.c_uint64()
        self.ctx._check(lib.rtg_photon_map_emit_fibonacci(self._h, _ptr(lights), lights.shape[0], count_per_light, seed, max_depth,
                                                          C.byref(h), C.byref(count)))
        return PhotonMap(self.ctx, None, n=count.value, _handle=h)

    def photons_emit_raw(self, lights, dirs2, out_ptr, capacity, max_depth=4):
        """The photon pass into caller-chosen memory (host or device address of `capacity` records): photons made.
        `dirs2` may be (device address, directions per light) when the context's option photon_trig is 2."""
        lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        if isinstance(dirs2, tuple):
            dirs_ptr, per_light = _ptr(dirs2[0]), int(dirs2[1])
        else:
            dirs2 = np.ascontiguousarray(dirs2, dtype=np.float64).reshape(lights.shape[0], -1, 2)
            dirs_ptr, per_light = _ptr(dirs2), dirs2.shape[1]
        count = C.c_uint64()
        self.ctx._check(lib.rtg_photons_emit(self._h, _ptr(lights), lights.shape[0], dirs_ptr, per_light, max_depth,
                                             _ptr(out_ptr) if capacity else None, capacity, C.byref(count)))
        return count.value

    def indirect_irrad(self, photon_map, points, photon_samples=3, newton_steps=5, newton_epsilon=1e-5):
        """compute_indirect_irrad per point: (n, 3) float64."""
        points = np.ascontiguousarray(points, dtype=POINT_DTYPE)
        rgb = np.empty((points.shape[0], 3), dtype=np.float64)
        self.ctx._check(lib.rtg_indirect_irrad(self._h, photon_map._h, _ptr(points), points.shape[0], photon_samples,
                                               newton_steps, float(newton_epsilon), _ptr(rgb)))
        return rgb

    def render_frame_photons(self, view, lights, photon_map, width, height, tile=None, max_depth=4, photon_samples=3,
                             newton_steps=5, newton_epsilon=1e-5, rgba8=True):
        """raytrace_pixel with the photon term: (rgb (n, 3), rgba8 (n, 4) | None) over the tile."""
        lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        x0, y0, w, h = tile if tile is not None else (0, 0, width, height)
        rgb = np.empty((w * h, 3), dtype=np.float64)
        c8 = np.empty((w * h, 4), dtype=np.uint8) if rgba8 else None
        self.ctx._check(lib.rtg_render_frame_photons(self._h, C.byref(view), _ptr(lights), lights.shape[0], photon_map._h,
                                                     photon_samples, newton_steps, float(newton_epsilon), width, height,
                                                     Rect(x0, y0, w, h), max_depth, _ptr(rgb), _ptr(c8)))
        return rgb, c8

    def indirect_irrad_raw(self, photon_map, points_ptr, n, photon_samples, newton_steps, newton_epsilon, rgb_ptr):
        self.ctx._check(lib.rtg_indirect_irrad(self._h, photon_map._h, _ptr(points_ptr), n, photon_samples, newton_steps,
                                               float(newton_epsilon), _ptr(rgb_ptr)))

    def render_frame_raw(self, view, lights_ptr, n_lights, width, height, tile, max_depth, rgb_ptr, rgba8_ptr, hits_ptr):
        self.ctx._check(lib.rtg_render_frame(self._h, C.byref(view), _ptr(lights_ptr), n_lights, width, height,
                                             Rect(*tile), max_depth, _ptr(rgb_ptr), _ptr(rgba8_ptr), _ptr(hits_ptr)))

    def close(self):
        if self._h:
            lib.rtg_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: the library globals may already be gone
            pass


def make_view(orig=(0.0, 0.0, 1.0), fov=(np.pi * 0.5, np.pi * 0.5)):
    """View::from_fov with tan(fov/2) evaluated on the host."""
    import math
    v = View()
    v.orig[:] = orig
    v.tan_half_fov[0] = math.tan(fov[0] / 2.0)
    v.tan_half_fov[1] = math.tan(fov[1] / 2.0)
    return v


def render_scene_builder(ctx, time, width, height, max_depth=4):
    """Renderer::new(SceneBuilder{time}.scene()).render() at width x height (C++ host path)."""
    rgb = np.empty((height * width, 3), dtype=np.float64)
    hits = np.empty(height * width, dtype=HIT_DTYPE)
    ctx._check(host.rth_render_scene_builder(ctx._h, time, width, height, max_depth, _ptr(rgb), _ptr(hits)))
    return rgb, hits


def render_scene_builder_photons(ctx, time, width, height, photon_count, seed=1, max_depth=4, photons_on_host=False):
    """Renderer::render with its photon pass at width x height: (rgb, photons made, seconds of the three parts).
    photons_on_host: directions and their sin / cos on the host (the oracle's photons bit for bit) instead of the
    whole pass on the GPU."""
    rgb = np.empty((height * width, 3), dtype=np.float64)
    n = C.c_uint64()
    seconds = (C.c_double * 3)()
    ctx._check(host.rth_render_scene_builder_photons(ctx._h, time, width, height, max_depth, photon_count, seed,
                                                     1 if photons_on_host else 0, _ptr(rgb), C.byref(n), seconds))
    return rgb, n.value, tuple(seconds)


def to_hdr(rgb, width, height):
    """ImageBuilder::to_hdr: the Radiance picture of an RGB frame, scanlines run-length coded."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    cap = host.rth_to_hdr(_ptr(rgb), width, height, None, 0)
    out = np.empty(cap, dtype=np.uint8)
    n = host.rth_to_hdr(_ptr(rgb), width, height, _ptr(out), cap)
    return out[:n].tobytes()


def images_tar(rgb, width, height, seconds=0.0, cost=None):
    """Renderer::images() framed as the reference server's tar archive; `cost` (uint32 per pixel) is what perf.hdr
    shares `seconds` out by (None: evenly)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    if cost is not None:
        cost = np.ascontiguousarray(cost, dtype=np.uint32)
        assert cost.size == width * height
    cap = host.rth_images_tar(_ptr(rgb), width, height, seconds, None, None, 0)
    out = np.empty(cap, dtype=np.uint8)
    n = host.rth_images_tar(_ptr(rgb), width, height, seconds, None if cost is None else _ptr(cost), _ptr(out), cap)
    return out[:n].tobytes()


def render_images(ctx, objects, lights, view_orig, fov, width, height, max_depth=4):
    """Renderer::new(scene).render() and .images() on the GPU through the C++ host mirror, the images framed as the
    server's tar archive: (rgb, hits, per-pixel cost, seconds of the pixel pass, tar bytes)."""
    objs = (Object * len(objects))(*objects)
    lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
    orig = (C.c_double * 3)(*view_orig)
    fov2 = (C.c_double * 2)(*fov)
    rgb = np.empty((height * width, 3), dtype=np.float64)
    hits = np.empty(height * width, dtype=HIT_DTYPE)
    cost = np.empty(height * width, dtype=np.uint32)
    seconds = C.c_double()
    need = C.c_uint64()
    cap = host.rth_images_tar(_ptr(rgb), width, height, 0.0, None, None, 0)
    tar = np.empty(cap, dtype=np.uint8)
    ctx._check(host.rth_render_images(ctx._h, objs, len(objects), _ptr(lights), len(lights), orig, fov2, width, height,
                                      max_depth, _ptr(rgb), _ptr(hits), _ptr(cost), C.byref(seconds), _ptr(tar), cap,
                                      C.byref(need)))
    return rgb, hits, cost, seconds.value, tar[:need.value].tobytes()
