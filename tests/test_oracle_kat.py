"""CPU: pin the oracle with the reference's own unit tests that touch the path and with hand-derived known answers.

The reference holds no fixture for any ray routine ("parity unpinned"): the known answers below are derived by hand
from the formulas in src/geo/{bounds,triangle,sphere}.rs and src/render/plane_object.rs.
"""
import math

import numpy as np

import orc
from raytracer_amd import rtg

INF = float("inf")


# ---- the reference's own tests -------------------------------------------------
def test_mat2_inverse_reference():  # src/math/mat.rs:223-227, exact identity
    m = [[1.0, 2.0], [3.0, 4.0]]
    assert np.array_equal(orc.mat_mul(m, orc.mat_inverse(m)), np.eye(2))


def test_mat3_inverse_reference():  # src/math/mat.rs:229-237, exact identity
    m = [[1.0, 2.0, 3.0], [1.5, 2.5, 3.5], [1.25, 2.25, 3.75]]
    assert np.array_equal(orc.mat_mul(m, orc.mat_inverse(m)), np.eye(3))


def test_mat4_inverse_reference():  # src/math/mat.rs:239-253, one-sided 1e-10
    m = [[1.0, 2.0, 3.0, 4.0], [2.0, 4.0, 8.0, 16.0], [4.0, 16.0, 64.0, 256.0], [8.0, 64.0, 8.0, 64.0]]
    diff = orc.mat_mul(m, orc.mat_inverse(m)) - np.eye(4)
    assert (diff < 1.0e-10).all()
    assert (np.abs(diff) < 1.0e-10).all()


def test_bounds_union_surface_area_reference():  # src/tree/bv_kd_tree.rs:80-95
    out = (orc.C.c_double * 6)()
    orc.lib.orc_bounds_union(orc.d([0, 0, 0, 1, 1, 1]), orc.d([1, 1, 1, 2, 2, 2]), out)
    assert list(out) == [0, 0, 0, 2, 2, 2]
    assert orc.lib.orc_bounds_surface_area(out) == 24.0


def test_smpte2048_roundtrip_reference():  # src/geo/color.rs:19-30
    for x in range(100):
        x1 = x / 100.0
        x2 = orc.lib.orc_smpte2048_decode(orc.lib.orc_smpte2048_encode(x1))
        assert abs(x1 - x2) < 1e-10
    assert orc.lib.orc_pq8(0.0) == 0
    assert orc.lib.orc_pq8(80.0 * 1.0) == 255
    assert orc.lib.orc_pq8(float("nan")) == 0


# ---- scalar semantics (src/math/scalar.rs:36-65) ------------------------------------
def test_minimum_maximum_real_cmp():
    nan = float("nan")
    assert math.isnan(orc.lib.orc_minimum(nan, 1.0)) and math.isnan(orc.lib.orc_maximum(1.0, nan))
    assert math.copysign(1.0, orc.lib.orc_minimum(0.0, -0.0)) == -1.0
    assert math.copysign(1.0, orc.lib.orc_maximum(-0.0, 0.0)) == 1.0
    assert orc.lib.orc_real_cmp(nan, 1.0) == -1 and orc.lib.orc_real_cmp(1.0, nan) == 1
    assert orc.lib.orc_real_cmp(nan, nan) == 0 and orc.lib.orc_real_cmp(1.0, 2.0) == -1


def test_dot_fold_order():
    # ((0 + a0*b0) + a1*b1) + a2*b2: 1e16 + 1 - 1e16 loses the 1 in this order
    assert orc.lib.orc_dot3(orc.d([1e16, 1.0, -1e16]), orc.d([1.0, 1.0, 1.0])) == 0.0
    assert orc.lib.orc_dot3(orc.d([1e16, -1e16, 1.0]), orc.d([1.0, 1.0, 1.0])) == 1.0


# ---- box (src/geo/bounds.rs:80-108) ------------------------------------------------
def test_box_known_answers():
    box = [0, 0, 0, 1, 1, 1]
    assert orc.bounds_raycast(box, [-1, .5, .5, 1, 0, 0]) == (1.0, 2.0)
    # quirk: an axis with d == 0 is skipped, not tested for containment
    assert orc.bounds_raycast(box, [-1, 5, .5, 1, 0, 0]) == (1.0, 2.0)
    # quirk: a zero-thickness box is never hit (strict <)
    assert orc.bounds_raycast([0, 0, 0.5, 1, 1, 0.5], [.5, .5, 1, 0, 0, -1]) is None
    # no t >= 0 clamp inside Bounds::raycast
    assert orc.bounds_raycast(box, [2, .5, .5, 1, 0, 0]) == (-2.0, -1.0)
    # all three axes skipped: the full interval survives
    assert orc.bounds_raycast(box, [5, 5, 5, 0, 0, 0]) == (-INF, INF)
    # diagonal miss
    assert orc.bounds_raycast(box, [-1, 3, .5, 1, 1, 0]) is None


# ---- triangle (src/geo/triangle.rs:54-82) -------------------------------------------
TRI = [0, 0, 0, 1, 0, 0, 0, 1, 0]


def test_triangle_known_answers():
    p = orc.triangle_raycast(TRI, [.25, .25, 1, 0, 0, -1])
    assert p["t"] == 1.0 and p["bary"] == [0.5, 0.25, 0.25] and p["q"] == [0.25, 0.25, 0.0]
    assert p["mp"] == [0.5, 0.25] and p["n"][2] == 1.0
    assert orc.triangle_raycast(TRI, [.25, .25, -1, 0, 0, 1]) is None        # back face culled
    assert orc.triangle_raycast(TRI, [.25, .25, -1, 0, 0, 1], manifold=True)["t"] == 1.0
    assert orc.triangle_raycast(TRI, [.25, .25, -1, 0, 0, -1]) is None       # behind the origin
    assert orc.triangle_raycast(TRI, [.75, .75, 1, 0, 0, -1]) is None        # outside an edge
    assert orc.triangle_raycast(TRI, [0, 0, 1, 0, 0, -1])["bary"] == [1.0, 0.0, 0.0]   # on a vertex: c == 0 accepted


def test_triangle_nan_and_inf_hits_are_accepted():
    # degenerate triangle: nt == 0 -> t = 0/0 = NaN; NaN fails every rejection test
    p = orc.triangle_raycast([0, 0, 0, 1, 0, 0, 2, 0, 0], [.5, .5, 1, 0, 0, -1])
    assert p is not None and math.isnan(p["t"])
    # ray parallel to the plane and below it: t = +1/0 = +inf, barycentrics NaN -> accepted
    p = orc.triangle_raycast(TRI, [-1, .25, -1, 1, 0, 0])
    assert p is not None and p["t"] == INF


# ---- sphere (src/geo/sphere.rs:81-108, roots 0.0.7) -------------------------------------
def test_sphere_known_answers():
    p = orc.sphere_raycast([0, -0.2, 0], 0.2, [0, -0.2, 1, 0, 0, -1])
    assert abs(p["time"] - 0.8) < 1e-15 and list(p["geo_normal"]) == [0.0, 0.0, 1.0]
    assert list(p["manifold_point"]) == [0.0, 0.0]
    assert orc.sphere_raycast([0, -0.2, 0], 0.2, [0, 1, 1, 0, 0, -1]) is None            # miss
    inside = orc.sphere_raycast([0, 0, 0], 1.0, [0, 0, 0, 0, 0, -1])                     # origin inside: far root
    assert inside["time"] == 1.0
    assert orc.sphere_raycast([0, 0, 0], 1.0, [0, 0, 5, 0, 0, 1]) is None                # both roots negative


def test_quadratic_roots():
    assert orc.find_roots_quadratic(1.0, -3.0, 2.0) == [1.0, 2.0]
    assert orc.find_roots_quadratic(1.0, 2.0, 1.0) == [-1.0]
    assert orc.find_roots_quadratic(1.0, 0.0, 1.0) == []
    assert orc.find_roots_quadratic(0.0, 2.0, -4.0) == [2.0]
    r = orc.find_roots_quadratic(1.0, -2.0, 0.96)
    assert abs(r[0] - 0.8) < 1e-15 and abs(r[1] - 1.2) < 1e-15


# ---- plane (src/render/plane_object.rs:23-49) ---------------------------------------------
PLANE = ([0, -0.5, 0], [0, 0, -1], [-1, 0, 0])


def test_plane_known_answers():
    pt, second = orc.plane_raycast(*PLANE, [0, 0, 1, 0, -1, 0])
    assert pt["time"] == 0.5 and list(pt["position"]) == [0.0, -0.5, 1.0]
    assert list(pt["manifold_point"]) == [-1.0, 0.0] and second            # -10 + 0 even -> material2
    pt, second = orc.plane_raycast(*PLANE, [0.1, 0, 1, 0, -1, 0])
    assert list(pt["geo_normal"]) == [0.0, 1.0, 0.0]
    assert abs(pt["manifold_point"][1] + 0.1) < 1e-15 and not second       # -10 + -1 = -11, -11 % 2 == -1 -> material1
    assert orc.plane_raycast(*PLANE, [0, 0, 1, 0, 1, 0]) is None           # t < 0
    pt, second = orc.plane_raycast(*PLANE, [0, -0.5, 1, 1, 0, 0])          # in-plane ray: 0/0 = NaN accepted
    assert math.isnan(pt["time"]) and second                                # NaN.round() as i64 == 0 -> even


# ---- camera (src/render/renderer.rs:157-159, src/geo/view.rs:13-19) ---------------------------
def test_pixel_ndc_and_view_ray():
    s = (orc.C.c_double * 2)()
    orc.lib.orc_pixel_ndc(0, 0, 4, 2, s)
    assert list(s) == [(0 - 1.5) / 4, -(0 - 0.5) / 2]        # divides by W, not W-1
    orc.lib.orc_pixel_ndc(3, 1, 4, 2, s)
    assert list(s) == [0.375, -0.25]
    view = rtg.make_view()
    ray = (orc.C.c_double * 6)()
    orc.lib.orc_view_get_ray(orc.C.byref(view), orc.d([0.0, 0.0]), ray)
    assert list(ray) == [0.0, 0.0, 1.0, 0.0, 0.0, -1.0]
    orc.lib.orc_view_get_ray(orc.C.byref(view), orc.d([0.5, -0.5]), ray)
    x = 0.5 * math.tan(math.pi / 4)
    n = math.sqrt(((0.0 + x * x) + x * x) + 1.0)
    assert list(ray)[3:] == [x / n, -x / n, -1.0 / n]


# ---- dielectric (src/render/dielectric.rs:13-52) ------------------------------------------------
def test_dielectric_normal_incidence_and_tir():
    di = orc.dielectric([0, 0, -1], [0, 0, 1], 1.0, 1.5)
    assert di["reflect"] == [0.0, 0.0, 1.0] and di["has_refract"] and di["refract"] == [0.0, 0.0, -1.0]
    assert abs(di["reflectance"] - 0.04) < 1e-15              # ((1.5-1)/(1.5+1))^2
    # leaving glass at a grazing angle: total internal reflection
    s = math.sqrt(0.5)
    di = orc.dielectric([s, 0, s], [0, 0, 1], 1.0, 1.5)
    assert not di["has_refract"] and di["reflectance"] == 1.0


# ---- transform (src/geo/transform.rs:51-58, src/math/mat.rs:172-177) -------------------------------
def test_transform_translation_applied_before_scale():
    out = (orc.C.c_double * 16)()
    orc.lib.orc_mat4_from_scale_rotation_translation(orc.d([2, 2, 2]), orc.d([1, 2, 3]), out)
    m = np.array(list(out)).reshape(4, 4)
    assert np.array_equal(m, [[2, 0, 0, 2], [0, 2, 0, 4], [0, 0, 2, 6], [0, 0, 0, 1]])
    t = rtg.Transform()
    orc.lib.orc_transform_from_forward(out, orc.C.byref(t))
    assert np.array_equal(np.array(list(t.reverse)).reshape(4, 4), [[.5, 0, 0, -1], [0, .5, 0, -2], [0, 0, .5, -3], [0, 0, 0, 1]])
    assert np.array_equal(np.array(list(t.forward_norm)).reshape(3, 3), np.eye(3) * 0.5)
    # the product library's Transform::from is the same function
    t2 = rtg.transform_from_forward(m)
    assert bytes(t) == bytes(t2)
    assert bytes(rtg.transform(2.0, (1, 2, 3))) == bytes(t)
