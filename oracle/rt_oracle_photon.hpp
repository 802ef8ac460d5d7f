// rt_oracle_photon.hpp -- CPU oracle, second part: the photon map.  Included by rt_oracle.cpp.
//
// TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED except where noted: the reference's photon pass draws its directions
// from an entropy-seeded generator (render/renderer.rs:112,117) and sums per-light contributions in HashMap order
// (:259-263), so no two of its own runs agree in the last bits; what is pinned by the reference's own tests is the
// k-nearest-neighbour search (tree/kd_tree.rs:221-268: equal to a brute-force sort) and Der addition
// (math/scalar.rs:360-367).
//
// Everything below restates a reference routine generically over the scalar T, as the reference's
// `Object::raycast<T: Scalar>` does: T = double, or T = Der2, the forward-mode dual number with two derivative
// slots (`Der<2>`, math/scalar.rs:72-128).  Paths are relative to the reference's src/ directory.
#pragma once

namespace {
namespace gen {

// ---- math/scalar.rs:72-128, 222-342 -------------------------------------------------
struct Der2 {
    double v;
    double d[2];
};
inline Der2 der_from(double v) { return Der2{v, {0.0, 0.0}}; }                    // scalar.rs:222-224
inline Der2 der_var(double v, int n) {                                            // scalar.rs:79-83
    Der2 r{v, {0.0, 0.0}};
    r.d[n] = 1.0;
    return r;
}
inline Der2 operator+(Der2 x, Der2 y) { return Der2{x.v + y.v, {x.d[0] + y.d[0], x.d[1] + y.d[1]}}; }
inline Der2 operator-(Der2 x, Der2 y) { return Der2{x.v - y.v, {x.d[0] - y.d[0], x.d[1] - y.d[1]}}; }
inline Der2 operator*(Der2 x, Der2 y) {                                           // x.v * y.d + x.d * y.v
    return Der2{x.v * y.v, {x.v * y.d[0] + x.d[0] * y.v, x.v * y.d[1] + x.d[1] * y.v}};
}
inline Der2 operator/(Der2 x, Der2 y) {                                           // (y.v * x.d - x.v * y.d) / (y.v * y.v)
    return Der2{x.v / y.v,
                {(y.v * x.d[0] - x.v * y.d[0]) / (y.v * y.v), (y.v * x.d[1] - x.v * y.d[1]) / (y.v * y.v)}};
}
inline Der2 operator-(Der2 x) { return Der2{-x.v, {x.d[0] * -1.0, x.d[1] * -1.0}}; }   // oper1(|x| -x, |x| -1.0)
// #[derive(PartialOrd)] on Der { v, d: Vector<2, f64> }: lexicographic, undefined (all comparisons false) as soon
// as a NaN decides
inline int der_partial_cmp(const Der2 &a, const Der2 &b) {     // -1, 0, 1, or 2 for "no order"
    const double x[3] = {a.v, a.d[0], a.d[1]}, y[3] = {b.v, b.d[0], b.d[1]};
    for (int i = 0; i < 3; i++) {
        if (x[i] < y[i]) return -1;
        if (x[i] > y[i]) return 1;
        if (!(x[i] == y[i])) return 2;
    }
    return 0;
}
inline bool operator<(const Der2 &a, const Der2 &b) { return der_partial_cmp(a, b) == -1; }
inline bool operator>(const Der2 &a, const Der2 &b) { return der_partial_cmp(a, b) == 1; }
inline bool operator>=(const Der2 &a, const Der2 &b) {
    int c = der_partial_cmp(a, b);
    return c == 0 || c == 1;
}

// the operations of `trait Scalar` / `roots::FloatType` the path uses, for both scalars
inline double s_from(double x, double) { return x; }
inline Der2 s_from(double x, Der2) { return der_from(x); }
template <class T> T from(double x) { return s_from(x, T{}); }
inline double val(double x) { return x; }
inline double val(const Der2 &x) { return x.v; }                                  // into_const
inline double s_sqrt(double x) { return std::sqrt(x); }
inline Der2 s_sqrt(Der2 x) {                                                      // oper1(sqrt, 1 / (2 sqrt x))
    double k = 1.0 / (2.0 * std::sqrt(x.v));
    return Der2{std::sqrt(x.v), {x.d[0] * k, x.d[1] * k}};
}
// sin and cos as two separate libm calls, which is what rustc (LLVM) emits for `x.sin()` and `x.cos()`.  gcc would
// merge the pair into one sincos(x), and glibc's sincos rounds differently from its sin / cos in the last place for
// about one argument in a thousand -- enough to move a photon.
static double (*volatile libm_sin)(double) = ::sin;
static double (*volatile libm_cos)(double) = ::cos;
inline double s_sin(double x) { return libm_sin(x); }
inline double s_cos(double x) { return libm_cos(x); }
inline Der2 s_sin(Der2 x) { return Der2{libm_sin(x.v), {x.d[0] * libm_cos(x.v), x.d[1] * libm_cos(x.v)}}; }
inline Der2 s_cos(Der2 x) { return Der2{libm_cos(x.v), {x.d[0] * -libm_sin(x.v), x.d[1] * -libm_sin(x.v)}}; }
inline double s_abs(double x) { return std::fabs(x); }
inline Der2 s_abs(Der2 x) { return x.v < 0.0 ? -x : x; }                          // scalar.rs:267-273
inline double s_maximum(double a, double b) { return f_maximum(a, b); }
inline Der2 s_maximum(Der2 a, Der2 b) {                                           // scalar.rs:312-322
    if (a.v != a.v) return a;
    if (b.v != b.v) return b;
    return a.v <= b.v ? b : a;
}
inline double s_minimum(double a, double b) { return f_minimum(a, b); }
inline Der2 s_minimum(Der2 a, Der2 b) {                                           // scalar.rs:301-311
    if (a.v != a.v) return a;
    if (b.v != b.v) return b;
    return a.v <= b.v ? a : b;
}
inline bool s_is_zero(double x) { return x == 0.0; }
inline bool s_is_zero(const Der2 &x) { return x.v == 0.0 && x.d[0] == 0.0 && x.d[1] == 0.0; }   // derived PartialEq
template <class T> bool s_is_finite(const T &x) { return std::isfinite(val(x)); }
template <class T> int s_real_cmp(const T &a, const T &b) { return real_cmp(val(a), val(b)); }

// ---- math/vec.rs ---------------------------------------------------------------------
template <class T> struct Vec3T {
    T c[3];
    T &operator[](int i) { return c[i]; }
    const T &operator[](int i) const { return c[i]; }
};
template <class T> Vec3T<T> vmk(T x, T y, T z) { return Vec3T<T>{{x, y, z}}; }
template <class T> Vec3T<T> vcast(V3 a) { return vmk(from<T>(a[0]), from<T>(a[1]), from<T>(a[2])); }
template <class T> V3 vval(const Vec3T<T> &a) { return mk(val(a[0]), val(a[1]), val(a[2])); }
template <class T> Vec3T<T> operator+(Vec3T<T> a, Vec3T<T> b) { return vmk(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
template <class T> Vec3T<T> operator-(Vec3T<T> a, Vec3T<T> b) { return vmk(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
template <class T> Vec3T<T> operator*(Vec3T<T> a, T s) { return vmk(a[0] * s, a[1] * s, a[2] * s); }
template <class T> Vec3T<T> operator/(Vec3T<T> a, T s) { return vmk(a[0] / s, a[1] / s, a[2] / s); }
template <class T> Vec3T<T> operator-(Vec3T<T> a) { return vmk(-a[0], -a[1], -a[2]); }
template <class T> T vdot(Vec3T<T> a, Vec3T<T> b) {                               // Sum folds from T::from(0.0)
    return ((from<T>(0.0) + a[0] * b[0]) + a[1] * b[1]) + a[2] * b[2];
}
template <class T> T vlength(Vec3T<T> a) { return s_sqrt(vdot(a, a)); }
template <class T> Vec3T<T> vnormalize(Vec3T<T> a) { return a / vlength(a); }
template <class T> Vec3T<T> vcross(Vec3T<T> a, Vec3T<T> b) {                      // vec.rs:199-203
    return vmk(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}

// ---- geo/ray.rs ----------------------------------------------------------------------
template <class T> struct RayT {
    Vec3T<T> orig, dir;
    Vec3T<T> pos(T t) const { return orig + dir * t; }
};
template <class T> RayT<T> bounce(Vec3T<T> orig, Vec3T<T> dir) { return RayT<T>{orig + dir * from<T>(1e-5), dir}; }

// ---- math/mat.rs:155-160,180-183 with the matrix cast to T (geo/transform.rs:60-80) ----
template <class T> Vec3T<T> mat4_apply(const Mat<4> &m, Vec3T<T> p, double w) {
    const T v[4] = {p[0], p[1], p[2], from<T>(w)};
    Vec3T<T> out;
    for (int i = 0; i < 3; i++) {
        T acc = from<T>(0.0);
        for (int j = 0; j < 4; j++) acc = acc + from<T>(m.m[i][j]) * v[j];
        out[i] = acc;
    }
    return out;
}
template <class T> Vec3T<T> forward_norm_t(const Transform &t, Vec3T<T> n) {
    Vec3T<T> out;
    for (int i = 0; i < 3; i++) {
        T acc = from<T>(0.0);
        for (int j = 0; j < 3; j++) acc = acc + from<T>(t.forward_norm.m[i][j]) * n[j];
        out[i] = acc;
    }
    return vnormalize(out);
}

// ---- render/object.rs ----------------------------------------------------------------
template <class T> struct PointT {
    T time;
    Vec3T<T> position, inter_normal, geo_normal;
    Manifold manifold;
    T mp[2];
    Material material;
};
inline bool manifold_pop(Manifold &m, uint32_t &end) {                            // object.rs:50-57
    if (m.len == 0) return false;
    m.len -= 1;
    end = m.array[m.len];
    m.array[m.len] = 0;
    return true;
}

// ---- roots 0.0.7 find_roots_quadratic, generic (see rt_oracle.cpp for the provenance) ----
template <class T> int roots_quadratic(T a2, T a1, T a0, T out[2]) {
    const T zero = from<T>(0.0);
    if (s_is_zero(a2)) {
        if (s_is_zero(a1)) {
            if (s_is_zero(a0)) {
                out[0] = zero;
                return 1;
            }
            return 0;
        }
        out[0] = -a0 / a1;
        return 1;
    }
    T discriminant = a1 * a1 - from<T>(4.0) * a2 * a0;
    if (discriminant < zero) return 0;
    T a2x2 = from<T>(2.0) * a2;
    if (s_is_zero(discriminant)) {
        out[0] = -a1 / a2x2;
        return 1;
    }
    T sq = s_sqrt(discriminant);
    T same_sign, diff_sign;
    if (a1 < zero) {
        same_sign = -a1 + sq;
        diff_sign = -a1 - sq;
    } else {
        same_sign = -a1 - sq;
        diff_sign = -a1 + sq;
    }
    T x1, x2;
    if (s_abs(same_sign) > s_abs(a2x2)) {
        T a0x2 = from<T>(2.0) * a0;
        if (s_abs(diff_sign) > s_abs(a2x2)) {
            x1 = a0x2 / same_sign;
            x2 = a0x2 / diff_sign;
        } else {
            x1 = a0x2 / same_sign;
            x2 = same_sign / a2x2;
        }
    } else {
        x1 = diff_sign / a2x2;
        x2 = same_sign / a2x2;
    }
    if (x1 < x2) {
        out[0] = x1;
        out[1] = x2;
    } else {
        out[0] = x2;
        out[1] = x1;
    }
    return 2;
}

// ---- geo/sphere.rs:81-108 --------------------------------------------------------------
template <class T> bool sphere_raycast_t(V3 centre, double rad, const RayT<T> &ray, PointT<T> &out) {
    Vec3T<T> e = ray.orig, d = ray.dir, o = vcast<T>(centre);
    T a = vdot(d, d);
    T b = from<T>(2.0) * vdot(e - o, d);
    T c = vdot(e - o, e - o) - from<T>(rad * rad);
    T roots[2];
    int n = roots_quadratic(a, b, c, roots);
    bool have = false;
    T t = from<T>(0.0);
    for (int i = 0; i < n; i++) {
        if (!(roots[i] >= from<T>(0.0))) continue;
        if (!have || s_real_cmp(roots[i], t) < 0) {
            have = true;
            t = roots[i];
        }
    }
    if (!have) return false;
    Vec3T<T> position = ray.pos(t);
    Vec3T<T> norm = vnormalize(position - o);
    out.time = t;
    out.position = position;
    out.inter_normal = norm;
    out.geo_normal = norm;
    out.manifold = Manifold{{0, 0}, 0};
    out.mp[0] = norm[0];
    out.mp[1] = norm[1];
    return true;
}

// ---- render/plane_object.rs:23-49 ------------------------------------------------------
template <class T> bool plane_raycast_t(const Plane &pl, const RayT<T> &ray, PointT<T> &out, bool &second) {
    Vec3T<T> p = vcast<T>(pl.position);
    Vec3T<T> n = vcast<T>(cross(pl.tan1, pl.tan2));
    Vec3T<T> e = ray.orig, d = ray.dir;
    T t = vdot(p - e, n) / vdot(d, n);
    if (t < from<T>(0.0)) return false;
    Vec3T<T> x = ray.pos(t);
    T m1 = vdot(x - p, vcast<T>(pl.tan1));
    T m2 = vdot(x - p, vcast<T>(pl.tan2));
    uint64_t sum = (uint64_t)round_to_i64(val(m1) * 10.0) + (uint64_t)round_to_i64(val(m2) * 10.0);
    second = ((int64_t)sum % 2) == 0;
    out.time = t;
    out.position = x;
    out.inter_normal = n;
    out.geo_normal = n;
    out.manifold = Manifold{{0, 0}, 0};
    out.mp[0] = m1;
    out.mp[1] = m2;
    return true;
}

// ---- geo/triangle.rs:54-82 and the inferred TriVerts (see rt_oracle.cpp) -------------------
template <class T> bool triverts_raycast_t(const rtg_tri &tri, const double *normals9, const RayT<T> &ray, bool manifold,
                                           PointT<T> &out) {
    Vec3T<T> vert[3];
    for (int i = 0; i < 3; i++) vert[i] = vcast<T>(mk(tri.p[i][0], tri.p[i][1], tri.p[i][2]));
    const T zero = from<T>(0.0);
    Vec3T<T> nt = vcross(vert[1] - vert[0], vert[2] - vert[1]);
    T denom = vdot(nt, nt);
    if (vdot(nt, ray.dir) > zero && !manifold) return false;
    T t = vdot(vert[0] - ray.orig, nt) / vdot(ray.dir, nt);
    if (t < zero && !manifold) return false;
    Vec3T<T> q = ray.pos(t);
    T bary[3] = {zero, zero, zero};
    for (int i = 0; i < 3; i++) {
        Vec3T<T> va = vert[i], vb = vert[(i + 1) % 3];
        T c = vdot(vcross(vb - va, q - va), nt) / denom;
        if (c < zero && !manifold) return false;
        bary[(i + 2) % 3] = c;
    }
    out.time = t;
    out.position = q;
    out.geo_normal = vnormalize(nt);
    if (normals9) {
        Vec3T<T> n0 = vcast<T>(mk(normals9[0], normals9[1], normals9[2])), n1 = vcast<T>(mk(normals9[3], normals9[4], normals9[5])),
                 n2 = vcast<T>(mk(normals9[6], normals9[7], normals9[8]));
        out.inter_normal = vnormalize((n0 * bary[0] + n1 * bary[1]) + n2 * bary[2]);
    } else {
        out.inter_normal = out.geo_normal;
    }
    out.manifold = Manifold{{0, 0}, 0};
    out.mp[0] = bary[0];
    out.mp[1] = bary[1];
    return true;
}

// ---- geo/bounds.rs:80-108, tree/bvh.rs:107-133 -------------------------------------------
template <class T> struct IntervalT {
    T lo, hi;
};
template <class T> bool interval_intersect_t(const IntervalT<T> &a, const IntervalT<T> &b, IntervalT<T> &out) {
    T lo = s_maximum(a.lo, b.lo), hi = s_minimum(a.hi, b.hi);
    if (lo < hi) {
        out = IntervalT<T>{lo, hi};
        return true;
    }
    return false;
}
template <class T> bool bounds_raycast_t(const rtg_bvh_node &n, const RayT<T> &ray, IntervalT<T> &out) {
    IntervalT<T> iv{from<T>(-INF), from<T>(INF)};
    for (int i = 0; i < 3; i++) {
        T a = (from<T>(n.lo[i]) - ray.orig[i]) / ray.dir[i];
        T b = (from<T>(n.hi[i]) - ray.orig[i]) / ray.dir[i];
        if (s_is_finite(a) && s_is_finite(b)) {
            IntervalT<T> next;
            if (!interval_intersect_t(iv, IntervalT<T>{s_minimum(a, b), s_maximum(a, b)}, next)) return false;
            iv = next;
        }
    }
    out = iv;
    return true;
}
template <class T> struct HolderT {
    bool has = false;
    PointT<T> point;
    void add(bool some, const PointT<T> &p) {
        if (!some) return;
        if (has) {
            if (p.time < point.time) point = p;
        } else {
            has = true;
            point = p;
        }
    }
};
template <class T> void raycast_rec_t(const orc_bvh &b, const RayT<T> &ray, uint32_t node, HolderT<T> &output) {
    const rtg_bvh_node &n = b.nodes[node];
    IntervalT<T> times, both;
    if (!bounds_raycast_t(n, ray, times)) return;
    IntervalT<T> held{from<T>(0.0), output.has ? output.point.time : from<T>(INF)};
    if (!interval_intersect_t(times, held, both)) return;
    for (uint32_t child = n.leaf_begin; child < n.leaf_end; child++) {
        PointT<T> p;
        bool some = triverts_raycast_t(b.leaves[child], b.normals.empty() ? nullptr : &b.normals[(size_t)child * 9], ray, false, p);
        if (some) p.manifold = p.manifold.push(child);
        output.add(some, p);
    }
    for (uint32_t child = n.child_begin; child < n.child_end; child++) raycast_rec_t(b, ray, child, output);
}
template <class T> bool bvh_raycast_t(const orc_bvh &b, const RayT<T> &ray, const Manifold *manifold, PointT<T> &out) {
    if (!b.patches.empty()) return false;                        // patch sets have no generic intersection
    if (manifold) {                                                               // bvh.rs:108-111
        Manifold inner = *manifold;
        uint32_t index = 0;
        if (!manifold_pop(inner, index) || index >= b.leaves.size()) return false;   // the reference unwraps / panics
        if (!triverts_raycast_t(b.leaves[index], b.normals.empty() ? nullptr : &b.normals[(size_t)index * 9], ray, true, out))
            return false;
        out.manifold = out.manifold.push(index);
        return true;
    }
    if (b.nodes.empty()) return false;
    HolderT<T> holder;
    raycast_rec_t(b, ray, b.root, holder);
    if (!holder.has) return false;
    out = holder.point;
    return true;
}

// ---- render/{any_object,transform_object,scene_object}.rs ---------------------------------
template <class T> bool object_raycast_t(const orc_scene::Obj &o, const RayT<T> &outer, const Manifold *manifold, PointT<T> &out) {
    RayT<T> inner{mat4_apply(o.transform.reverse, outer.orig, 1.0), mat4_apply(o.transform.reverse, outer.dir, 0.0)};
    PointT<T> p;
    switch (o.kind) {
    case RTG_SPHERE:
        if (!sphere_raycast_t(o.sphere_center, o.sphere_radius, inner, p)) return false;
        p.material = o.material;
        break;
    case RTG_MESH:
        if (!o.mesh || !bvh_raycast_t(*o.mesh, inner, manifold, p)) return false;
        p.material = o.material;
        break;
    case RTG_PLANE: {
        bool second = false;
        if (!plane_raycast_t(o.plane, inner, p, second)) return false;
        p.material = second ? o.material2 : o.material;
        break;
    }
    default:
        return false;
    }
    out = p;
    out.position = outer.pos(p.time);
    out.inter_normal = forward_norm_t(o.transform, p.inter_normal);
    out.geo_normal = forward_norm_t(o.transform, p.geo_normal);
    return true;
}
// SceneObject::raycast (scene_object.rs:19-36): with a manifold only the named object is asked, with the manifold's
// remainder; otherwise every object in order and `.min()` under real_cmp.
template <class T> bool scene_raycast_t(const orc_scene &s, const RayT<T> &ray, const Manifold *manifold, PointT<T> &out) {
    if (manifold) {
        Manifold inner = *manifold;
        uint32_t index = 0;
        if (!manifold_pop(inner, index) || index >= s.objects.size()) return false;
        PointT<T> p;
        if (!object_raycast_t(s.objects[index], ray, &inner, p)) return false;
        p.manifold = p.manifold.push(index);
        out = p;
        return true;
    }
    bool have = false;
    for (uint32_t i = 0; i < s.objects.size(); i++) {
        PointT<T> p;
        if (!object_raycast_t(s.objects[i], ray, (const Manifold *)nullptr, p)) continue;
        p.manifold = p.manifold.push(i);
        if (!have || s_real_cmp(out.time, p.time) > 0) {
            out = p;
            have = true;
        }
    }
    return have;
}

// ---- render/dielectric.rs:13-52 ------------------------------------------------------------
template <class T> struct DielectricT {
    T reflectance;
    Vec3T<T> reflect, refract;
    bool has_refract;
};
template <class T> DielectricT<T> dielectric_t(Vec3T<T> inc, Vec3T<T> signed_norm, T n1, T n2) {
    const T zero = from<T>(0.0), one = from<T>(1.0);
    inc = vnormalize(inc);
    T dot = vdot(inc, signed_norm);
    Vec3T<T> norm;
    T cos_theta_i, n_i, n_t;
    if (dot < zero) {
        norm = signed_norm;
        cos_theta_i = -dot;
        n_i = n1;
        n_t = n2;
    } else {
        norm = -signed_norm;
        cos_theta_i = dot;
        n_i = n2;
        n_t = n1;
    }
    T eta = n_i / n_t;
    T sin2_theta_i = s_maximum(one - cos_theta_i * cos_theta_i, zero);
    T sin2_theta_t = eta * eta * sin2_theta_i;
    DielectricT<T> r;
    r.refract = vmk(zero, zero, zero);
    if (sin2_theta_t >= one) {
        r.has_refract = false;
        r.reflectance = one;
    } else {
        T cos_theta_t = s_sqrt(one - sin2_theta_t);
        r.has_refract = true;
        r.refract = inc * eta + norm * (eta * cos_theta_i - cos_theta_t);
        T r_par = ((n_t * cos_theta_i) - (n_i * cos_theta_t)) / ((n_t * cos_theta_i) + (n_i * cos_theta_t));
        T r_perp = ((n_i * cos_theta_i) - (n_t * cos_theta_t)) / ((n_i * cos_theta_i) + (n_t * cos_theta_t));
        r.reflectance = (r_par * r_par + r_perp * r_perp) / from<T>(2.0);
    }
    r.reflect = inc - norm * from<T>(2.0) * vdot(inc, norm);
    return r;
}

// ---- render/renderer.rs:284-378 with filters ----------------------------------------------
template <class T> struct PathT {
    PointT<T> point;
    std::vector<Manifold> manifolds;
    std::vector<int> modes;          // 0 Reflect, 1 Refract
    T attenuation;
};
// filter_manifolds: entries of Option<Manifold> (has == false: None); filter_modes == nullptr: None
struct OptManifold {
    bool has;
    Manifold m;
};
template <class T>
void all_specular_filtered(const orc_scene &s, const RayT<T> &ray, T attenuation, const OptManifold *filter_manifolds,
                           size_t n_filter_manifolds, const int *filter_modes, size_t n_filter_modes, bool have_modes,
                           uint32_t max_depth, std::vector<Manifold> &out_manifolds, std::vector<int> &out_modes,
                           std::vector<PathT<T>> &output) {
    if (out_manifolds.size() >= max_depth) return;
    const OptManifold *fm = n_filter_manifolds >= 1 ? filter_manifolds : nullptr;        // slice_pop
    const OptManifold *rest_m = n_filter_manifolds >= 1 ? filter_manifolds + 1 : nullptr;
    const size_t n_rest_m = n_filter_manifolds >= 1 ? n_filter_manifolds - 1 : 0;
    PointT<T> first;
    if (!scene_raycast_t(s, ray, (fm && fm->has) ? &fm->m : nullptr, first)) return;
    if (!have_modes || n_filter_modes == 0) output.push_back(PathT<T>{first, out_manifolds, out_modes, attenuation});
    if (!first.material.has_dielectric) return;
    DielectricT<T> di = dielectric_t(ray.dir, first.geo_normal, from<T>(first.material.n1), from<T>(first.material.n2));
    int filter_mode = -1;                 // None
    const int *rest_modes = nullptr;
    size_t n_rest_modes = 0;
    if (have_modes) {
        if (n_filter_modes == 0) return;
        filter_mode = filter_modes[0];
        rest_modes = filter_modes + 1;
        n_rest_modes = n_filter_modes - 1;
    }
    if (filter_mode == -1 || filter_mode == 0) {
        out_manifolds.push_back(first.manifold);
        out_modes.push_back(0);
        all_specular_filtered(s, bounce(first.position, di.reflect), attenuation * di.reflectance, rest_m, n_rest_m, rest_modes,
                              n_rest_modes, have_modes, max_depth, out_manifolds, out_modes, output);
        out_manifolds.pop_back();
        out_modes.pop_back();
    }
    if (di.has_refract && (filter_mode == -1 || filter_mode == 1)) {
        out_manifolds.push_back(first.manifold);
        out_modes.push_back(1);
        all_specular_filtered(s, bounce(first.position, di.refract), attenuation * (from<T>(1.0) - di.reflectance), rest_m,
                              n_rest_m, rest_modes, n_rest_modes, have_modes, max_depth, out_manifolds, out_modes, output);
        out_manifolds.pop_back();
        out_modes.pop_back();
    }
}

// ---- geo/sphere.rs:19-33 --------------------------------------------------------------------
template <class T> Vec3T<T> zenith_y_into_normal(T zenith, T y) {
    T radius = s_sqrt(from<T>(1.0) - y * y);
    T x = s_cos(zenith) * radius;
    T z = s_sin(zenith) * radius;
    return vmk(x, y, z);
}

// ---- tree/kd_tree.rs, tree/seq_tree.rs, util/capped_heap.rs ------------------------------------
// The tree lives in a slice: the centre of a range is its element at (len / 2) (seq_tree.rs:41; the reference
// computes the centre relative to the sub-slice), left and right of it the two subtrees.
struct KdEntry {
    int axis = 0;
    Bounds bounds{mk(QNAN, QNAN, QNAN), mk(QNAN, QNAN, QNAN)};
    V3 position;
    uint32_t value;
};
inline Bounds kd_build_rec(KdEntry *tree, size_t len) {                            // kd_tree.rs:79-95
    Bounds bounds = bounds_empty();
    for (size_t i = 0; i < len; i++) bounds = bounds_union(bounds, Bounds{tree[i].position, tree[i].position});
    int axis = 0;                          // max_by_key: the last of equal maxima wins
    for (int i = 1; i < 3; i++)
        if (!(bounds_dim(bounds, i) < bounds_dim(bounds, axis))) axis = i;
    if (len == 0) return bounds_empty();
    const size_t center = len / 2;
    // select_nth_unstable_by_key: which of several equal keys lands in the centre is the library's choice
    std::nth_element(tree, tree + center, tree + len,
                     [axis](const KdEntry &a, const KdEntry &b) { return a.position[axis] < b.position[axis]; });
    tree[center].axis = axis;
    Bounds b = bounds_empty();
    b = bounds_union(b, kd_build_rec(tree, center));
    b = bounds_union(b, kd_build_rec(tree + center + 1, len - center - 1));
    b = bounds_union(b, Bounds{tree[center].position, tree[center].position});
    tree[center].bounds = b;
    return b;
}
inline double bounds_distance(const Bounds &b, V3 p) {                             // bounds.rs:41-55
    double total = 0.0;
    for (int i = 0; i < 3; i++) {
        double d;
        if (p[i] < b.min[i])
            d = b.min[i] - p[i];
        else if (p[i] > b.max[i])
            d = p[i] - b.max[i];
        else
            d = 0.0;
        total += d * d;
    }
    return std::sqrt(total);
}
struct KdNeighbor {
    double distance;
    uint32_t value;
};
// CappedHeap over a max-heap ordered by real_cmp(distance) (capped_heap.rs, binary_heap.rs).
struct CappedHeap {
    std::vector<KdNeighbor> heap;
    size_t cap;
    static bool less(const KdNeighbor &a, const KdNeighbor &b) { return real_cmp(a.distance, b.distance) < 0; }
    void push(KdNeighbor x) {
        if (heap.size() == cap) {          // push_pop (binary_heap.rs:81-93): a newcomer above the largest is dropped,
            if (cap == 0 || real_cmp(x.distance, heap.front().distance) > 0) return;   // one equal to it replaces it
            std::pop_heap(heap.begin(), heap.end(), less);
            heap.back() = x;
            std::push_heap(heap.begin(), heap.end(), less);
        } else {
            heap.push_back(x);
            std::push_heap(heap.begin(), heap.end(), less);
        }
    }
};
inline void kd_nearest_rec(const KdEntry *tree, size_t len, V3 query, CappedHeap &heap) {   // kd_tree.rs:122-154
    if (len == 0) return;
    const size_t c = len / 2;
    const KdEntry &center = tree[c];
    if (heap.heap.size() == heap.cap) {
        double max_so_far = heap.heap.empty() ? -INF : heap.heap.front().distance;
        double min_possible = bounds_distance(center.bounds, query);
        if (min_possible > max_so_far) return;
    }
    const int axis = center.axis;
    const bool left_first = query[axis] < center.position[axis];
    const KdEntry *first = left_first ? tree : tree + c + 1, *second = left_first ? tree + c + 1 : tree;
    const size_t n_first = left_first ? c : len - c - 1, n_second = left_first ? len - c - 1 : c;
    kd_nearest_rec(first, n_first, query, heap);
    heap.push(KdNeighbor{length(query - center.position), center.value});
    bool more = heap.heap.size() < heap.cap;
    if (!more) {
        double max_so_far = heap.heap.empty() ? -INF : heap.heap.front().distance;
        double min_possible = std::fabs(query[axis] - center.position[axis]);
        more = min_possible < max_so_far;
    }
    if (more) kd_nearest_rec(second, n_second, query, heap);
}
inline std::vector<KdNeighbor> kd_nearest(const std::vector<KdEntry> &tree, V3 query, size_t count) {
    CappedHeap heap;
    heap.cap = count;
    kd_nearest_rec(tree.data(), tree.size(), query, heap);
    std::sort_heap(heap.heap.begin(), heap.heap.end(), CappedHeap::less);         // into_sorted_vec: ascending
    return heap.heap;
}

// ---- render/renderer.rs:206-265 ------------------------------------------------------------------
inline Manifold manifold_of(uint32_t object, uint32_t leaf) {
    Manifold m{{0, 0}, 0};
    if (leaf != RTG_NONE) m = m.push(leaf);
    return m.push(object);
}
inline V3 compute_indirect_irrad(const orc_scene &s, const std::vector<KdEntry> &tree, const rtg_photon *photons,
                                 const rtg_point &p, uint32_t photon_samples, uint32_t newton_steps, double newton_epsilon,
                                 uint32_t max_depth) {
    V3 total = mk(0.0, 0.0, 0.0);
    if (p.manifold_len == 0) return total;
    const V3 p_position = mk(p.position[0], p.position[1], p.position[2]);
    Manifold p_manifold{{p.manifold[0], p.manifold[1]}, p.manifold_len};
    struct Adjusted {
        bool has = false;
        V3 light, d0, d1;
    };
    std::vector<Adjusted> by_light;       // HashMap<light_index, AdjustedPhoton>: a later photon of a light replaces it
    for (const KdNeighbor &nb : kd_nearest(tree, p_position, photon_samples)) {
        const rtg_photon &ph = photons[nb.value];
        double dir[2] = {ph.dir[0], ph.dir[1]};
        std::vector<OptManifold> filter_manifolds;
        std::vector<int> modes;
        for (uint32_t i = 0; i < ph.n_bounces; i++) {
            filter_manifolds.push_back(OptManifold{true, manifold_of(ph.surface_object[i], ph.surface_leaf[i])});
            modes.push_back((ph.modes >> i) & 1u);
        }
        filter_manifolds.push_back(OptManifold{true, p_manifold});
        const Vec3T<Der2> origin = vcast<Der2>(mk(ph.origin[0], ph.origin[1], ph.origin[2]));
        for (uint32_t step = 0; step < newton_steps; step++) {
            RayT<Der2> ray{origin, zenith_y_into_normal(der_var(dir[0], 0), der_var(dir[1], 1))};
            std::vector<PathT<Der2>> hit;
            std::vector<Manifold> om;
            std::vector<int> omodes;
            all_specular_filtered(s, ray, der_from(1.0), filter_manifolds.data(), filter_manifolds.size(), modes.data(),
                                  modes.size(), true, max_depth, om, omodes, hit);
            if (hit.empty()) break;
            const PointT<Der2> &h = hit[0].point;
            Der2 sep[2] = {h.mp[0] - der_from(p.manifold_point[0]), h.mp[1] - der_from(p.manifold_point[1])};
            Mat<2> jac;                    // from_row_vector(sep.map(|x| x.d))
            jac.m[0][0] = sep[0].d[0];
            jac.m[0][1] = sep[0].d[1];
            jac.m[1][0] = sep[1].d[0];
            jac.m[1][1] = sep[1].d[1];
            Mat<2> inv = inverse2(jac);
            const double sv[2] = {sep[0].v, sep[1].v};
            double step_v[2];
            matvec(inv, sv, step_v);
            dir[0] = dir[0] - step_v[0];
            dir[1] = dir[1] - step_v[1];
        }
        RayT<Der2> ray{origin, zenith_y_into_normal(der_var(dir[0], 0), der_var(dir[1], 1))};
        std::vector<PathT<Der2>> real_photon;
        std::vector<Manifold> om;
        std::vector<int> omodes;
        all_specular_filtered(s, ray, der_from(1.0), (const OptManifold *)nullptr, 0, modes.data(), modes.size(), true,
                              max_depth, om, omodes, real_photon);
        if (real_photon.empty()) continue;
        const PointT<Der2> &rp = real_photon[0].point;
        if (length(vval(rp.position) - p_position) < newton_epsilon) {
            if (by_light.size() <= ph.light_index) by_light.resize((size_t)ph.light_index + 1);
            Adjusted &a = by_light[ph.light_index];
            a.has = true;
            a.light = mk(ph.light[0], ph.light[1], ph.light[2]);
            a.d0 = mk(rp.position[0].d[0], rp.position[1].d[0], rp.position[2].d[0]);
            a.d1 = mk(rp.position[0].d[1], rp.position[1].d[1], rp.position[2].d[1]);
        }
    }
    // the reference walks a HashMap (any order); here ascending light index
    for (const Adjusted &a : by_light) {
        if (!a.has) continue;
        V3 area_vector = cross(a.d0, a.d1);
        double area = 4.0 * PI * length(area_vector);
        total = total + a.light / area;
    }
    return total;
}

} // namespace gen
} // namespace
