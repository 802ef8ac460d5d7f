// rt_oracle.cpp -- CPU oracle for the ray-cast / BVH-traversal path.
//
// TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  PARITY UNPINNED for the ray
// routines: no reference fixture covers them and the Rust crate cannot be
// built in this environment.
//
// Every function restates one reference routine in plain IEEE binary64, with
// the reference's operation order (Rust never contracts a*b+c; build this file
// with -ffp-contract=off and without -ffast-math).  Paths are relative to the
// reference's src/ directory.
#include "rt_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

namespace {

const double INF = std::numeric_limits<double>::infinity();
const double QNAN = std::numeric_limits<double>::quiet_NaN();
const double PI = 3.14159265358979323846264338327950288; // std::f64::consts::PI

// ---- math/scalar.rs:36-65 ------------------------------------------------
// f64::minimum / f64::maximum: NaN-propagating, -0 < +0.
double f_minimum(double a, double b) {
    if (a != a) return a;
    if (b != b) return b;
    if (a < b) return a;
    if (b < a) return b;
    return std::signbit(a) ? a : b;
}
double f_maximum(double a, double b) {
    if (a != a) return a;
    if (b != b) return b;
    if (a > b) return a;
    if (b > a) return b;
    return std::signbit(a) ? b : a;
}
// real_cmp (scalar.rs:46-60): NaN sorts below every number, NaN == NaN.
int real_cmp(double a, double b) {
    bool an = a != a, bn = b != b;
    if (an) return bn ? 0 : -1;
    if (bn) return 1;
    return a < b ? -1 : (a > b ? 1 : 0);
}

// ---- math/vec.rs ---------------------------------------------------------
struct V2 {
    double v[2];
};
struct V3 {
    double v[3];
    double operator[](int i) const { return v[i]; }
    double &operator[](int i) { return v[i]; }
};
V3 mk(double x, double y, double z) { return V3{{x, y, z}}; }
V3 operator+(V3 a, V3 b) { return mk(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }  // vec.rs:107-110
V3 operator-(V3 a, V3 b) { return mk(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }  // vec.rs:116-119
V3 operator*(V3 a, double s) { return mk(a[0] * s, a[1] * s, a[2] * s); }      // vec.rs:125-128
V3 operator/(V3 a, double s) { return mk(a[0] / s, a[1] / s, a[2] / s); }      // vec.rs:146-149
V3 operator-(V3 a) { return mk(-a[0], -a[1], -a[2]); }                         // vec.rs:151-154
// dot (vec.rs:36-38): products summed left to right by `Sum`, which folds from 0.0.
double dot(V3 a, V3 b) { return ((0.0 + a[0] * b[0]) + a[1] * b[1]) + a[2] * b[2]; }
double length(V3 a) { return std::sqrt(dot(a, a)); }                           // vec.rs:39-42
V3 normalize(V3 a) { return a / length(a); }                                   // vec.rs:43-45
V3 cross(V3 a, V3 b) {                                                         // vec.rs:199-203
    return mk(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
V3 map_mul(V3 a, V3 b) { return mk(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }   // vec.rs:61-63

// ---- math/mat.rs ---------------------------------------------------------
template <int N> struct Mat {
    double m[N][N];
};
template <int N> double rowdot(const double *row, const double *v) {
    double s = 0.0;
    for (int k = 0; k < N; k++) s = s + row[k] * v[k];
    return s;
}
// Matrix * Vector (mat.rs:180-183): per-row dot.
template <int N> void matvec(const Mat<N> &a, const double *v, double *out) {
    double tmp[N];
    for (int i = 0; i < N; i++) tmp[i] = rowdot<N>(a.m[i], v);
    for (int i = 0; i < N; i++) out[i] = tmp[i];
}
// Matrix * Matrix (mat.rs:185-190): self applied to every column of rhs.
template <int N> Mat<N> matmul(const Mat<N> &a, const Mat<N> &b) {
    Mat<N> r;
    for (int j = 0; j < N; j++) {
        double col[N], o[N];
        for (int k = 0; k < N; k++) col[k] = b.m[k][j];
        matvec<N>(a, col, o);
        for (int i = 0; i < N; i++) r.m[i][j] = o[i];
    }
    return r;
}
Mat<2> inverse2(const Mat<2> &x) {                                              // mat.rs:67-70
    double a = x.m[0][0], b = x.m[0][1], c = x.m[1][0], d = x.m[1][1];
    double s = 1.0 / (a * d - b * c);
    Mat<2> r = {{{d * s, (-b) * s}, {(-c) * s, a * s}}};
    return r;
}
Mat<3> inverse3(const Mat<3> &x) {                                              // mat.rs:80-87
    double a = x.m[0][0], b = x.m[0][1], c = x.m[0][2];
    double d = x.m[1][0], e = x.m[1][1], f = x.m[1][2];
    double g = x.m[2][0], h = x.m[2][1], i = x.m[2][2];
    double s = 1.0 / (a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g));
    Mat<3> r = {{{(e * i - f * h) * s, (c * h - b * i) * s, (b * f - c * e) * s},
                 {(f * g - d * i) * s, (a * i - c * g) * s, (c * d - a * f) * s},
                 {(d * h - e * g) * s, (b * g - a * h) * s, (a * e - b * d) * s}}};
    return r;
}
Mat<4> inverse4(const Mat<4> &x) {                                              // mat.rs:100-154
    double m00 = x.m[0][0], m01 = x.m[0][1], m02 = x.m[0][2], m03 = x.m[0][3];
    double m10 = x.m[1][0], m11 = x.m[1][1], m12 = x.m[1][2], m13 = x.m[1][3];
    double m20 = x.m[2][0], m21 = x.m[2][1], m22 = x.m[2][2], m23 = x.m[2][3];
    double m30 = x.m[3][0], m31 = x.m[3][1], m32 = x.m[3][2], m33 = x.m[3][3];
    double a2323 = m22 * m33 - m23 * m32;
    double a1323 = m21 * m33 - m23 * m31;
    double a1223 = m21 * m32 - m22 * m31;
    double a0323 = m20 * m33 - m23 * m30;
    double a0223 = m20 * m32 - m22 * m30;
    double a0123 = m20 * m31 - m21 * m30;
    double a2313 = m12 * m33 - m13 * m32;
    double a1313 = m11 * m33 - m13 * m31;
    double a1213 = m11 * m32 - m12 * m31;
    double a2312 = m12 * m23 - m13 * m22;
    double a1312 = m11 * m23 - m13 * m21;
    double a1212 = m11 * m22 - m12 * m21;
    double a0313 = m10 * m33 - m13 * m30;
    double a0213 = m10 * m32 - m12 * m30;
    double a0312 = m10 * m23 - m13 * m20;
    double a0212 = m10 * m22 - m12 * m20;
    double a0113 = m10 * m31 - m11 * m30;
    double a0112 = m10 * m21 - m11 * m20;
    double det = 1.0 / (m00 * (m11 * a2323 - m12 * a1323 + m13 * a1223) -
                        m01 * (m10 * a2323 - m12 * a0323 + m13 * a0223) +
                        m02 * (m10 * a1323 - m11 * a0323 + m13 * a0123) -
                        m03 * (m10 * a1223 - m11 * a0223 + m12 * a0123));
    Mat<4> r;
    r.m[0][0] = det * (m11 * a2323 - m12 * a1323 + m13 * a1223);
    r.m[0][1] = det * -(m01 * a2323 - m02 * a1323 + m03 * a1223);
    r.m[0][2] = det * (m01 * a2313 - m02 * a1313 + m03 * a1213);
    r.m[0][3] = det * -(m01 * a2312 - m02 * a1312 + m03 * a1212);
    r.m[1][0] = det * -(m10 * a2323 - m12 * a0323 + m13 * a0223);
    r.m[1][1] = det * (m00 * a2323 - m02 * a0323 + m03 * a0223);
    r.m[1][2] = det * -(m00 * a2313 - m02 * a0313 + m03 * a0213);
    r.m[1][3] = det * (m00 * a2312 - m02 * a0312 + m03 * a0212);
    r.m[2][0] = det * (m10 * a1323 - m11 * a0323 + m13 * a0123);
    r.m[2][1] = det * -(m00 * a1323 - m01 * a0323 + m03 * a0123);
    r.m[2][2] = det * (m00 * a1313 - m01 * a0313 + m03 * a0113);
    r.m[2][3] = det * -(m00 * a1312 - m01 * a0312 + m03 * a0112);
    r.m[3][0] = det * -(m10 * a1223 - m11 * a0223 + m12 * a0123);
    r.m[3][1] = det * (m00 * a1223 - m01 * a0223 + m02 * a0123);
    r.m[3][2] = det * -(m00 * a1213 - m01 * a0213 + m02 * a0113);
    r.m[3][3] = det * (m00 * a1212 - m01 * a0212 + m02 * a0112);
    return r;
}
template <int N> Mat<N> identity() {
    Mat<N> r;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) r.m[i][j] = i == j ? 1.0 : 0.0;
    return r;
}

// ---- geo/transform.rs:16-21,51-80 -----------------------------------------
struct Transform {
    Mat<3> forward_norm;
    Mat<4> forward;
    Mat<3> reverse_norm;
    Mat<4> reverse;
};
Transform transform_from(const Mat<4> &forward) {                               // transform.rs:51-58
    Transform t;
    t.forward = forward;
    t.reverse = inverse4(forward);
    for (int i = 0; i < 3; i++)                                                 // Mat3::from_mat4(..).transpose()
        for (int j = 0; j < 3; j++) t.reverse_norm.m[i][j] = forward.m[j][i];
    t.forward_norm = inverse3(t.reverse_norm);
    return t;
}
V3 transform_position(const Mat<4> &m, V3 p) {                                  // mat.rs:155-157
    double v[4] = {p[0], p[1], p[2], 1.0}, o[4];
    matvec<4>(m, v, o);
    return mk(o[0], o[1], o[2]);
}
V3 transform_tangent(const Mat<4> &m, V3 p) {                                   // mat.rs:158-160
    double v[4] = {p[0], p[1], p[2], 0.0}, o[4];
    matvec<4>(m, v, o);
    return mk(o[0], o[1], o[2]);
}
V3 forward_norm(const Transform &t, V3 n) {                                     // transform.rs:78-80
    double o[3];
    matvec<3>(t.forward_norm, n.v, o);
    return normalize(mk(o[0], o[1], o[2]));
}

// ---- geo/ray.rs ----------------------------------------------------------
struct Ray {
    V3 orig, dir;
    V3 pos(double t) const { return orig + dir * t; }                           // ray.rs:21-23
};
Ray new_bounce(V3 orig, V3 dir) { return Ray{orig + dir * 1e-5, dir}; }         // ray.rs:10,16-18
Ray ray_from6(const double *r) { return Ray{mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])}; }

// ---- geo/bounds.rs -------------------------------------------------------
struct Bounds {
    V3 min, max;
};
Bounds bounds_empty() { return Bounds{mk(INF, INF, INF), mk(-INF, -INF, -INF)}; }   // bounds.rs:25-27
Bounds bounds_union(const Bounds &a, const Bounds &b) {                         // bounds.rs:28-33
    Bounds r;
    for (int i = 0; i < 3; i++) {
        r.min[i] = f_minimum(a.min[i], b.min[i]);
        r.max[i] = f_maximum(a.max[i], b.max[i]);
    }
    return r;
}
V3 bounds_center(const Bounds &b) { return (b.min + b.max) / 2.0; }             // bounds.rs:38-40
double bounds_dim(const Bounds &b, int i) { return b.max[i] - b.min[i]; }       // bounds.rs:70-72
double bounds_surface_area(const Bounds &b) {                                   // bounds.rs:73-76
    double dx = bounds_dim(b, 0), dy = bounds_dim(b, 1), dz = bounds_dim(b, 2);
    return (dx * dy + dx * dz + dy * dz) * 2.0;
}
struct Interval {
    double min, max;
};
bool interval_intersect(const Interval &a, const Interval &b, Interval &out) {  // bounds.rs:100-108
    double mn = f_maximum(a.min, b.min);
    double mx = f_minimum(a.max, b.max);
    if (mn < mx) {
        out = Interval{mn, mx};
        return true;
    }
    return false;
}
bool bounds_raycast(const Bounds &b, const Ray &ray, Interval &out) {           // bounds.rs:80-90
    Interval iv{-INF, INF};
    for (int i = 0; i < 3; i++) {
        double a = (b.min[i] - ray.orig[i]) / ray.dir[i];
        double c = (b.max[i] - ray.orig[i]) / ray.dir[i];
        if (std::isfinite(a) && std::isfinite(c)) {
            Interval n{f_minimum(a, c), f_maximum(a, c)};
            if (!interval_intersect(iv, n, iv)) return false;
        }
    }
    out = iv;
    return true;
}

// ---- render/material.rs, render/object.rs ---------------------------------
struct Material {
    V3 diffuse;
    bool has_dielectric;
    double n1, n2;
};
Material material_from(const rtg_material &m) {
    return Material{mk(m.diffuse[0], m.diffuse[1], m.diffuse[2]), m.has_dielectric != 0, m.n1, m.n2};
}
struct Manifold {                                                               // object.rs:12-16
    uint32_t array[2];
    uint32_t len;
    Manifold push(uint32_t x) const {                                           // object.rs:44-49
        Manifold m = *this;
        m.array[m.len] = x;
        m.len += 1;
        return m;
    }
};
struct RaycastPoint {                                                           // object.rs:18-27
    double time;
    V3 position, inter_normal, geo_normal;
    Manifold manifold;
    double mp[2];
    Material material;
    uint32_t material_slot;
};
struct Holder {                                                                 // object.rs:29-31,60-79
    bool has = false;
    RaycastPoint point;
    void add(bool some, const RaycastPoint &p) {
        if (!some) return;
        if (has) {
            if (p.time < point.time) point = p;
        } else {
            has = true;
            point = p;
        }
    }
    Interval interval() const { return Interval{0.0, has ? point.time : INF}; }
};

// ---- geo/triangle.rs -----------------------------------------------------
struct TrianglePoint {
    double time;
    V3 bary, position;
    double mp[2];
    V3 geo_normal;
};
Bounds triangle_bounds(const V3 v[3]) {                                         // triangle.rs:43-45
    Bounds b = bounds_empty();
    for (int i = 0; i < 3; i++) b = bounds_union(b, Bounds{v[i], v[i]});
    return b;
}
bool triangle_raycast(const V3 vert[3], const Ray &ray, bool manifold, TrianglePoint &out) {  // triangle.rs:54-82
    V3 v0 = vert[0], v1 = vert[1], v2 = vert[2];
    V3 nt = cross(v1 - v0, v2 - v1);
    double denom = dot(nt, nt);
    if (dot(nt, ray.dir) > 0.0 && !manifold) return false;
    V3 e = ray.orig, d = ray.dir;
    double t = dot(v0 - e, nt) / dot(d, nt);
    if (t < 0.0 && !manifold) return false;
    V3 q = ray.pos(t);
    V3 bary = mk(0.0, 0.0, 0.0);
    for (int i = 0; i < 3; i++) {
        V3 va = vert[i], vb = vert[(i + 1) % 3];
        double c = dot(cross(vb - va, q - va), nt) / denom;
        if (c < 0.0 && !manifold) return false;
        bary[(i + 2) % 3] = c;
    }
    out.time = t;
    out.bary = bary;
    out.position = q;
    out.mp[0] = bary[0];
    out.mp[1] = bary[1];
    out.geo_normal = normalize(nt);
    return true;
}

// TriVerts is missing from the reference snapshot (src/mesh/).  Contract
// inferred from its call sites (tree/bvh.rs:41,69,110,124-125): wraps
// Triangle::raycast and converts to a RaycastPoint with an empty manifold and
// Material::nan().  inter_normal = normalize(sum bary_i * n_i) when vertex
// normals exist, else the geometric normal (this build's decision).
struct TriVerts {
    V3 p[3];
};
bool triverts_raycast(const TriVerts &tv, const V3 *normals, const Ray &ray, bool manifold,
                      RaycastPoint &out) {
    TrianglePoint tp;
    if (!triangle_raycast(tv.p, ray, manifold, tp)) return false;
    out.time = tp.time;
    out.position = tp.position;
    out.geo_normal = tp.geo_normal;
    if (normals)
        out.inter_normal =
            normalize((normals[0] * tp.bary[0] + normals[1] * tp.bary[1]) + normals[2] * tp.bary[2]);
    else
        out.inter_normal = tp.geo_normal;
    out.manifold = Manifold{{0, 0}, 0};
    out.mp[0] = tp.mp[0];
    out.mp[1] = tp.mp[1];
    out.material = Material{mk(QNAN, QNAN, QNAN), false, 0.0, 0.0};
    out.material_slot = 0;
    return true;
}

// ---- roots 0.0.7 find_roots_quadratic (third-party; Cargo.lock:883-884) ----
// Not vendored in the reference.  Restated from the crate's published
// algorithm: discriminant a1^2 - 4*a2*a0; none if negative; one root
// -a1/(2*a2) if zero; otherwise avoid the smallest divisor among
// {2*a2, -a1-+sqrt(D)} and return the roots ascending.  a2 == 0 falls back to
// the linear equation.
int find_roots_quadratic(double a2, double a1, double a0, double out[2]) {
    if (a2 == 0.0) {
        if (a1 == 0.0) {
            if (a0 == 0.0) {
                out[0] = 0.0;
                return 1;
            }
            return 0;
        }
        out[0] = -a0 / a1;
        return 1;
    }
    double discriminant = a1 * a1 - 4.0 * a2 * a0;
    if (discriminant < 0.0) return 0;
    double a2x2 = 2.0 * a2;
    if (discriminant == 0.0) {
        out[0] = -a1 / a2x2;
        return 1;
    }
    double sq = std::sqrt(discriminant);
    double same_sign, diff_sign;
    if (a1 < 0.0) {
        same_sign = -a1 + sq;
        diff_sign = -a1 - sq;
    } else {
        same_sign = -a1 - sq;
        diff_sign = -a1 + sq;
    }
    double x1, x2;
    if (std::fabs(same_sign) > std::fabs(a2x2)) {
        double a0x2 = 2.0 * a0;
        if (std::fabs(diff_sign) > std::fabs(a2x2)) {
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

// ---- geo/sphere.rs:81-108 --------------------------------------------------
bool sphere_raycast(V3 o, double rad, const Ray &ray, RaycastPoint &out) {
    V3 e = ray.orig, d = ray.dir;
    double a = dot(d, d);
    double b = 2.0 * dot(e - o, d);
    double c = dot(e - o, e - o) - rad * rad;
    double roots[2];
    int n = find_roots_quadratic(a, b, c, roots);
    bool have = false;
    double t = 0.0;
    for (int i = 0; i < n; i++) {            // filter(>= 0).min_by(real_cmp): first minimum wins
        if (!(roots[i] >= 0.0)) continue;
        if (!have || real_cmp(roots[i], t) < 0) {
            have = true;
            t = roots[i];
        }
    }
    if (!have) return false;
    V3 position = ray.pos(t);
    V3 norm = normalize(position - o);
    out.time = t;
    out.position = position;
    out.inter_normal = norm;
    out.geo_normal = norm;
    out.manifold = Manifold{{0, 0}, 0};
    out.mp[0] = norm[0];
    out.mp[1] = norm[1];
    out.material = Material{mk(QNAN, QNAN, QNAN), false, 0.0, 0.0};
    out.material_slot = 0;
    return true;
}

// `x.round() as i64`: round half away from zero, saturating cast, NaN -> 0.
int64_t round_to_i64(double x) {
    double r = std::round(x);
    if (r != r) return 0;
    if (r >= 9223372036854775808.0) return std::numeric_limits<int64_t>::max();
    if (r <= -9223372036854775808.0) return std::numeric_limits<int64_t>::min();
    return (int64_t)r;
}

// ---- render/plane_object.rs:23-49 -------------------------------------------
struct Plane {
    V3 position, tan1, tan2;
};
bool plane_raycast(const Plane &pl, const Ray &ray, RaycastPoint &out, bool &second) {
    V3 p = pl.position;
    V3 n = cross(pl.tan1, pl.tan2);
    V3 e = ray.orig, d = ray.dir;
    double t = dot(p - e, n) / dot(d, n);
    if (t < 0.0) return false;
    V3 x = ray.pos(t);
    double m1 = dot(x - p, pl.tan1);
    double m2 = dot(x - p, pl.tan2);
    double scale = 10.0;
    // release-mode i64 addition wraps; `% 2 == 0` is the low bit of the sum
    uint64_t sum = (uint64_t)round_to_i64(m1 * scale) + (uint64_t)round_to_i64(m2 * scale);
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

// ---- tree/bvh.rs -----------------------------------------------------------
struct BvhTree {                                                                // bvh.rs:26-31
    std::vector<uint32_t> leaves;   // indices into the source triangle array
    std::vector<BvhTree> nodes;
    Bounds bounds = bounds_empty();
};
struct Counters {
    uint64_t nodes = 0, tris = 0;
};
} // namespace

struct orc_bvh {
    std::vector<rtg_bvh_node> nodes;
    std::vector<rtg_tri> leaves;
    std::vector<double> normals;     // empty or 9 per leaf
    std::vector<uint32_t> leaf_source;
    uint32_t root = 0;
    // trispline patch set (empty for plain triangle meshes)
    std::vector<rtg_trispline> patches;
    std::vector<rtg_patch_leaf> leaf_info;
    uint32_t newton_steps = 0;
};

namespace {
// ---- trispline.rs (orphan file, written against glam::DVec3) ----------------------
// glam semantics: dot = (x*x' + y*y') + z*z' (no leading 0.0), normalize = v * (1/length).
double g_dot(V3 a, V3 b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
double g_length(V3 a) { return std::sqrt(g_dot(a, a)); }
V3 g_normalize(V3 a) { return a * (1.0 / g_length(a)); }
V3 smul(double s, V3 a) { return mk(s * a[0], s * a[1], s * a[2]); }

struct Trispline {                      // trispline.rs:55-66, coefficient order of rtg_trispline
    V3 m00, m01, m02, m03, m10, m11, m12, m20, m21, m30;
};
Trispline trispline_from_c(const rtg_trispline &c) {
    auto v = [&](int i) { return mk(c.m[i][0], c.m[i][1], c.m[i][2]); };
    return Trispline{v(0), v(1), v(2), v(3), v(4), v(5), v(6), v(7), v(8), v(9)};
}
void trispline_to_c(const Trispline &t, rtg_trispline &c) {
    const V3 *src[10] = {&t.m00, &t.m01, &t.m02, &t.m03, &t.m10, &t.m11, &t.m12, &t.m20, &t.m21, &t.m30};
    for (int i = 0; i < 10; i++)
        for (int k = 0; k < 3; k++) c.m[i][k] = (*src[i])[k];
}
V3 trispline_position(const Trispline &s, double t, double u) {                 // trispline.rs:162-167
    return s.m30 * (t * t * t) + s.m20 * (t * t) + s.m10 * t + s.m00 + s.m21 * (t * t * u) + s.m11 * (t * u) +
           s.m01 * u + s.m12 * (t * u * u) + s.m02 * (u * u) + s.m03 * (u * u * u);
}
V3 trispline_ddt(const Trispline &s, double t, double u) {                      // trispline.rs:168-172
    return s.m30 * (3.0 * t * t) + s.m20 * (2.0 * t) + s.m10 + s.m21 * (2.0 * t * u) + s.m11 * u + s.m12 * (u * u);
}
V3 trispline_ddu(const Trispline &s, double t, double u) {                      // trispline.rs:173-177
    return s.m21 * (t * t) + s.m11 * (t) + s.m01 + s.m12 * (t * 2.0 * u) + s.m02 * (2.0 * u) + s.m03 * (3.0 * u * u);
}
V3 trispline_normal(const Trispline &s, double t, double u) {                   // trispline.rs:178-180
    return g_normalize(cross(trispline_ddt(s, t, u), trispline_ddu(s, t, u)));
}
struct HermiteSpline {
    V3 p0, p1, d0, d1;
};
struct Spline {
    V3 a, b, c, d;
};
void hermite_normalize_speed(HermiteSpline &h) {                                // trispline.rs:144-150
    double distance = g_length(h.p0 - h.p1);
    h.d0 = g_normalize(h.d0) * distance;
    h.d1 = g_normalize(h.d1) * distance;
}
Spline spline_from_hermite(const HermiteSpline &h) {                            // trispline.rs:126-134
    Spline s;
    s.a = smul(2.0, h.p0) - smul(2.0, h.p1) + smul(1.0, h.d0) + smul(1.0, h.d1);
    s.b = smul(-3.0, h.p0) + smul(3.0, h.p1) - smul(2.0, h.d0) - smul(1.0, h.d1);
    s.c = h.d0;
    s.d = h.p0;
    return s;
}
// NormalHermiteTrispline -> HermiteTrispline -> SplineTrispline -> Trispline (trispline.rs:68-124).
// in: p0,p1,p2,p01,p12,p20,n0,n1,n2,control
Trispline trispline_from_normal_hermite(const V3 in[10]) {
    V3 p0 = in[0], p1 = in[1], p2 = in[2], p01 = in[3], p12 = in[4], p20 = in[5];
    V3 n0 = in[6], n1 = in[7], n2 = in[8], control = in[9];
    V3 d01 = cross(p2 - p01, n0);
    V3 d10 = cross(p2 - p01, n1);
    V3 d02 = -cross(p1 - p20, n0);
    V3 d20 = -cross(p1 - p20, n2);
    V3 d12 = -cross(p0 - p12, n1);
    V3 d21 = -cross(p0 - p12, n2);
    HermiteSpline h01{p0, p1, d01, d10}, h02{p0, p2, d02, d20}, h21{p2, p1, d21, d12};
    hermite_normalize_speed(h01);
    hermite_normalize_speed(h02);
    hermite_normalize_speed(h21);
    Spline f = spline_from_hermite(h01), g = spline_from_hermite(h02), h = spline_from_hermite(h21);
    V3 s1 = f.c - g.c - smul(2.0, g.b) - smul(3.0, g.a) - h.c;                   // trispline.rs:105
    V3 s3 = f.a - g.a - h.a;                                                     // trispline.rs:107
    Trispline t;
    t.m11 = control;
    t.m12 = -t.m11 - s1;
    t.m21 = s3 + t.m12;
    t.m00 = f.d;
    t.m01 = g.c;
    t.m02 = g.b;
    t.m03 = g.a;
    t.m10 = f.c;
    t.m20 = f.b;
    t.m30 = f.a;
    return t;
}

Bounds node_bounds(const rtg_bvh_node &n) {
    return Bounds{mk(n.lo[0], n.lo[1], n.lo[2]), mk(n.hi[0], n.hi[1], n.hi[2])};
}
TriVerts leaf_tri(const rtg_tri &t) {
    TriVerts tv;
    for (int i = 0; i < 3; i++) tv.p[i] = mk(t.p[i][0], t.p[i][1], t.p[i][2]);
    return tv;
}
// Ray / patch intersection: this build's definition (the reference has none), see rtg_patch_set in rtg.h.
bool patch_leaf_raycast(const orc_bvh &b, uint32_t leaf, const Ray &ray, RaycastPoint &out) {
    TriVerts tv = leaf_tri(b.leaves[leaf]);
    TrianglePoint tp;
    if (!triangle_raycast(tv.p, ray, false, tp)) return false;
    const rtg_patch_leaf &li = b.leaf_info[leaf];
    Trispline sp = trispline_from_c(b.patches[li.patch]);
    double t = ((0.0 + tp.bary[0] * li.tu[0][0]) + tp.bary[1] * li.tu[1][0]) + tp.bary[2] * li.tu[2][0];
    double u = ((0.0 + tp.bary[0] * li.tu[0][1]) + tp.bary[1] * li.tu[1][1]) + tp.bary[2] * li.tu[2][1];
    double tau = tp.time;
    V3 c = -ray.dir;
    for (uint32_t it = 0; it < b.newton_steps; it++) {
        V3 a = trispline_ddt(sp, t, u), bb = trispline_ddu(sp, t, u);
        V3 r = ray.pos(tau) - trispline_position(sp, t, u);
        V3 bc = cross(bb, c);
        double det = dot(a, bc);
        double dt = dot(r, bc) / det;
        double du = dot(a, cross(r, c)) / det;
        double dtau = dot(a, cross(bb, r)) / det;
        t = t + dt;
        u = u + du;
        tau = tau + dtau;
    }
    V3 f = trispline_position(sp, t, u) - ray.pos(tau);
    double tol2 = 1e-20 * (dot(sp.m10, sp.m10) + dot(sp.m01, sp.m01));
    if (!(t >= 0.0 && u >= 0.0 && t + u <= 1.0 && tau >= 0.0 && dot(f, f) <= tol2)) return false;
    V3 n = trispline_normal(sp, t, u);
    out.time = tau;
    out.position = ray.pos(tau);
    out.inter_normal = n;
    out.geo_normal = n;
    out.manifold = Manifold{{0, 0}, 0};
    out.mp[0] = t;
    out.mp[1] = u;
    out.material = Material{mk(QNAN, QNAN, QNAN), false, 0.0, 0.0};
    out.material_slot = 0;
    return true;
}
bool leaf_raycast(const orc_bvh &b, uint32_t leaf, const Ray &ray, bool manifold, RaycastPoint &out) {
    if (!b.patches.empty()) return patch_leaf_raycast(b, leaf, ray, out);
    TriVerts tv = leaf_tri(b.leaves[leaf]);
    if (b.normals.empty()) return triverts_raycast(tv, nullptr, ray, manifold, out);
    V3 n[3];
    const double *src = &b.normals[(size_t)leaf * 9];
    for (int i = 0; i < 3; i++) n[i] = mk(src[i * 3], src[i * 3 + 1], src[i * 3 + 2]);
    return triverts_raycast(tv, n, ray, manifold, out);
}

// Bvh::raycast_rec (bvh.rs:118-133): box test, then every leaf of the node in
// index order, then every child node in index order.
void raycast_rec(const orc_bvh &b, const Ray &ray, uint32_t node, Holder &output, Counters &ctr) {
    const rtg_bvh_node &n = b.nodes[node];
    ctr.nodes++;
    Interval bounds_times;
    if (bounds_raycast(node_bounds(n), ray, bounds_times)) {
        Interval iv = output.interval(), tmp;
        if (interval_intersect(bounds_times, iv, tmp)) {
            for (uint32_t child = n.leaf_begin; child < n.leaf_end; child++) {
                ctr.tris++;
                RaycastPoint p;
                bool some = leaf_raycast(b, child, ray, false, p);
                if (some) p.manifold = p.manifold.push(child);
                output.add(some, p);
            }
            for (uint32_t child = n.child_begin; child < n.child_end; child++)
                raycast_rec(b, ray, child, output, ctr);
        }
    }
}
bool bvh_raycast(const orc_bvh &b, const Ray &ray, RaycastPoint &out, Counters &ctr) {  // bvh.rs:107-117
    if (b.nodes.empty()) return false;
    Holder h;
    raycast_rec(b, ray, b.root, h, ctr);
    if (h.has) out = h.point;
    return h.has;
}

// Bvh::new / add_bvh_tree (bvh.rs:80-106).
rtg_bvh_node add_bvh_tree(orc_bvh &out, const BvhTree &tree, const rtg_tri *tris, const double *normals) {
    uint32_t leaf_start = (uint32_t)out.leaves.size();
    for (uint32_t src : tree.leaves) {
        out.leaves.push_back(tris[src]);
        out.leaf_source.push_back(src);
        if (normals) out.normals.insert(out.normals.end(), normals + (size_t)src * 9, normals + (size_t)src * 9 + 9);
    }
    uint32_t leaf_end = (uint32_t)out.leaves.size();
    std::vector<rtg_bvh_node> kids;
    kids.reserve(tree.nodes.size());
    for (const BvhTree &child : tree.nodes) kids.push_back(add_bvh_tree(out, child, tris, normals));
    uint32_t nodes_start = (uint32_t)out.nodes.size();
    out.nodes.insert(out.nodes.end(), kids.begin(), kids.end());
    uint32_t nodes_end = (uint32_t)out.nodes.size();
    rtg_bvh_node e;
    for (int i = 0; i < 3; i++) {
        e.lo[i] = tree.bounds.min[i];
        e.hi[i] = tree.bounds.max[i];
    }
    e.child_begin = nodes_start;
    e.child_end = nodes_end;
    e.leaf_begin = leaf_start;
    e.leaf_end = leaf_end;
    return e;
}

// ---- tree/bvh_subdivide.rs --------------------------------------------------
// slice::binary_search_by as implemented by the standard library the reference
// was written against (Rust 1.52 - 1.81): probe mid = left + size/2, return the
// first probed match.  Ok(i) and Err(i) both map to i (bvh_subdivide.rs:29-31).
size_t binary_search_by_key(const std::vector<BvhTree> &trees, int axis, double key) {
    size_t size = trees.size(), left = 0, right = size;
    while (left < right) {
        size_t mid = left + size / 2;
        double v = bounds_center(trees[mid].bounds)[axis];
        if (v < key)
            left = mid + 1;
        else if (v > key)
            right = mid;
        else
            return mid;
        size = right - left;
    }
    return left;
}
BvhTree subdivide(std::vector<BvhTree> trees) {                                 // bvh_subdivide.rs:9-20
    if (trees.empty()) return BvhTree();
    if (trees.size() == 1) return std::move(trees[0]);
    // subdivide_once (bvh_subdivide.rs:21-43)
    Bounds bounds = bounds_empty();
    for (const BvhTree &t : trees) bounds = bounds_union(bounds, t.bounds);
    int axis = 0;                                   // max_by_key: the last maximum wins
    for (int i = 1; i < 3; i++)
        if (bounds_dim(bounds, i) >= bounds_dim(bounds, axis)) axis = i;
    double split_coordinate = bounds.min[axis] + bounds_dim(bounds, axis) / 2.0;
    std::stable_sort(trees.begin(), trees.end(), [axis](const BvhTree &a, const BvhTree &b) {
        return bounds_center(a.bounds)[axis] < bounds_center(b.bounds)[axis];
    });
    size_t split = binary_search_by_key(trees, axis, split_coordinate);
    size_t min_split = std::max<size_t>((size_t)((double)trees.size() * 0.01), 1);
    size_t max_split = trees.size() - min_split;
    if (split < min_split) split = min_split;
    if (split > max_split) split = max_split;
    std::vector<BvhTree> right(std::make_move_iterator(trees.begin() + split),
                               std::make_move_iterator(trees.end()));
    trees.resize(split);
    BvhTree node;
    BvhTree l = subdivide(std::move(trees));
    node.bounds = bounds_union(node.bounds, l.bounds);                          // add_node (bvh.rs:72-75)
    node.nodes.push_back(std::move(l));
    BvhTree r = subdivide(std::move(right));
    node.bounds = bounds_union(node.bounds, r.bounds);
    node.nodes.push_back(std::move(r));
    return node;
}

// ---- tree/bvh_cluster.rs:42-116, tree/bvh_joint.rs:3-8 ------------------------
// contract_once by its definition, without the k-d tree and the priority queue: while two or more trees remain,
// the tree whose nearest remaining neighbour (smallest union surface area, bv_kd_tree.rs:34-36) is farthest is
// taken (the reference pops the maximum, bvh_cluster.rs:67) and joined with that neighbour, popped tree first
// (bvh_cluster.rs:103-109); the forest becomes the pairs in the order made, then the odd tree out
// (bvh_cluster.rs:111-115).  Equal distances: the lower index wins both choices (the reference's own order there
// depends on the internals of the priority-queue crate and of select_nth_unstable: parity unpinned for ties).
void contract_once(std::vector<BvhTree> &trees) {
    const size_t n = trees.size();
    if (n < 2) return;
    std::vector<char> done(n, 0);
    std::vector<size_t> head(n, n);
    std::vector<double> head_distance(n, 0.0);
    auto find_head = [&](size_t i) {
        head[i] = n;
        for (size_t j = 0; j < n; j++) {
            if (j == i || done[j]) continue;
            double d = bounds_surface_area(bounds_union(trees[i].bounds, trees[j].bounds));
            if (head[i] == n || d < head_distance[i]) {
                head[i] = j;
                head_distance[i] = d;
            }
        }
    };
    for (size_t i = 0; i < n; i++) find_head(i);
    std::vector<std::pair<size_t, size_t>> contractions;
    size_t remaining = n;
    while (remaining >= 2) {
        size_t n1 = n;
        for (size_t i = 0; i < n; i++)
            if (!done[i] && (n1 == n || head_distance[i] > head_distance[n1])) n1 = i;
        size_t n2 = head[n1];
        done[n1] = done[n2] = 1;
        remaining -= 2;
        contractions.emplace_back(n1, n2);
        if (remaining >= 2)
            for (size_t i = 0; i < n; i++)
                if (!done[i] && (head[i] == n1 || head[i] == n2)) find_head(i);
    }
    std::vector<BvhTree> next;
    for (auto &c : contractions) {
        BvhTree node;
        node.bounds = bounds_union(node.bounds, trees[c.first].bounds);             // add_node (bvh.rs:72-75)
        node.nodes.push_back(std::move(trees[c.first]));
        node.bounds = bounds_union(node.bounds, trees[c.second].bounds);
        node.nodes.push_back(std::move(trees[c.second]));
        next.push_back(std::move(node));
    }
    for (size_t i = 0; i < n; i++)
        if (!done[i]) next.push_back(std::move(trees[i]));
    trees = std::move(next);
}

uint32_t depth_of(const orc_bvh &b, uint32_t node) {
    uint32_t d = 0;
    const rtg_bvh_node &n = b.nodes[node];
    for (uint32_t c = n.child_begin; c < n.child_end; c++) d = std::max(d, depth_of(b, c));
    return d + 1;
}

// ---- render/dielectric.rs:13-52 ---------------------------------------------
struct Dielectric {
    double reflectance;
    V3 reflect;
    bool has_refract;
    V3 refract;
};
Dielectric dielectric_new(V3 inc, V3 signed_norm, double n1, double n2) {
    inc = normalize(inc);
    double d = dot(inc, signed_norm);
    V3 norm;
    double cos_theta_i, n_i, n_t;
    if (d < 0.0) {
        norm = signed_norm;
        cos_theta_i = -d;
        n_i = n1;
        n_t = n2;
    } else {
        norm = -signed_norm;
        cos_theta_i = d;
        n_i = n2;
        n_t = n1;
    }
    double eta = n_i / n_t;
    double sin2_theta_i = f_maximum(1.0 - cos_theta_i * cos_theta_i, 0.0);
    double sin2_theta_t = eta * eta * sin2_theta_i;
    Dielectric r;
    r.refract = mk(0, 0, 0);
    if (sin2_theta_t >= 1.0) {
        r.has_refract = false;
        r.reflectance = 1.0;
    } else {
        double cos_theta_t = std::sqrt(1.0 - sin2_theta_t);
        r.has_refract = true;
        r.refract = inc * eta + norm * (eta * cos_theta_i - cos_theta_t);
        double r_par = ((n_t * cos_theta_i) - (n_i * cos_theta_t)) / ((n_t * cos_theta_i) + (n_i * cos_theta_t));
        double r_perp = ((n_i * cos_theta_i) - (n_t * cos_theta_t)) / ((n_i * cos_theta_i) + (n_t * cos_theta_t));
        r.reflectance = (r_par * r_par + r_perp * r_perp) / 2.0;
    }
    r.reflect = inc - norm * 2.0 * dot(inc, norm);
    return r;
}

// ---- geo/color.rs:5-17 --------------------------------------------------------
const double PQ_M1 = 1305.0 / 8192.0, PQ_M2 = 2523.0 / 32.0;
const double PQ_C1 = 107.0 / 128.0, PQ_C2 = 2413.0 / 128.0, PQ_C3 = 2392.0 / 128.0;
double smpte2048_decode(double x) {
    return std::pow(f_maximum(0.0, std::pow(x, 1.0 / PQ_M2) - PQ_C1) / (PQ_C2 - PQ_C3 * std::pow(x, 1.0 / PQ_M2)),
                    1.0 / PQ_M1);
}
double smpte2048_encode(double x) {
    return std::pow((PQ_C1 + PQ_C2 * std::pow(x, PQ_M1)) / (1.0 + PQ_C3 * std::pow(x, PQ_M1)), PQ_M2);
}
// 8-bit quantisation is this build's definition (the reference only writes
// float .hdr files): round(255 * clamp01(encode(c / 80))), image.rs:39.
uint8_t pq8(double c) {
    double e = smpte2048_encode(c / 80.0);
    if (!(e > 0.0)) e = 0.0;   // also maps NaN to 0
    if (e > 1.0) e = 1.0;
    return (uint8_t)std::floor(255.0 * e + 0.5);
}
} // namespace

// ---- render/{scene_object,transform_object,any_object,...}.rs ------------------
struct orc_scene {
    struct Obj {
        int kind;
        Transform transform;
        Material material, material2;
        V3 sphere_center;
        double sphere_radius;
        Plane plane;
        const orc_bvh *mesh;
    };
    std::vector<Obj> objects;
};

namespace {
bool any_object_raycast(const orc_scene::Obj &o, uint32_t index, const Ray &ray, RaycastPoint &out,
                        Counters &ctr) {                                        // any_object.rs:15-23
    switch (o.kind) {
    case RTG_SPHERE:                                                            // sphere_object.rs:20-25
        if (!sphere_raycast(o.sphere_center, o.sphere_radius, ray, out)) return false;
        out.material = o.material;
        out.material_slot = index * 2;
        return true;
    case RTG_PATCHES:
    case RTG_MESH:                                                              // mesh_object.rs:22-27
        if (!o.mesh || !bvh_raycast(*o.mesh, ray, out, ctr)) return false;
        out.material = o.material;
        out.material_slot = index * 2;
        return true;
    case RTG_PLANE: {
        bool second = false;
        if (!plane_raycast(o.plane, ray, out, second)) return false;
        out.material = second ? o.material2 : o.material;
        out.material_slot = index * 2 + (second ? 1 : 0);
        return true;
    }
    default:
        return false;
    }
}
bool transform_object_raycast(const orc_scene::Obj &o, uint32_t index, const Ray &outer,
                              RaycastPoint &out, Counters &ctr) {               // transform_object.rs:17-28
    Ray inner{transform_position(o.transform.reverse, outer.orig),
              transform_tangent(o.transform.reverse, outer.dir)};               // transform.rs:69-77
    RaycastPoint p;
    if (!any_object_raycast(o, index, inner, p, ctr)) return false;
    out = p;
    out.position = outer.pos(p.time);
    out.inter_normal = forward_norm(o.transform, p.inter_normal);
    out.geo_normal = forward_norm(o.transform, p.geo_normal);
    return true;
}
// SceneObject::raycast(ray, None) (scene_object.rs:19-36): every object in
// order, `.min()` keeps the first minimum under real_cmp (NaN lowest).
bool scene_raycast(const orc_scene &s, const Ray &ray, RaycastPoint &out, Counters &ctr) {
    bool have = false;
    for (uint32_t i = 0; i < s.objects.size(); i++) {
        RaycastPoint p;
        if (!transform_object_raycast(s.objects[i], i, ray, p, ctr)) continue;
        p.manifold = p.manifold.push(i);
        if (!have || real_cmp(out.time, p.time) > 0) {
            out = p;
            have = true;
        }
    }
    return have;
}

void point_to_c(bool some, const RaycastPoint &p, rtg_hit *hit, rtg_point *pt) {
    if (hit) {
        if (!some) {
            hit->object = RTG_NONE;
            hit->leaf = RTG_NONE;
            hit->t = INF;
        } else if (p.manifold.len == 2) {
            hit->leaf = p.manifold.array[0];
            hit->object = p.manifold.array[1];
            hit->t = p.time;
        } else {
            hit->leaf = RTG_NONE;
            hit->object = p.manifold.array[0];
            hit->t = p.time;
        }
    }
    if (pt) {
        std::memset(pt, 0, sizeof(*pt));
        if (!some) {
            pt->time = INF;
            return;
        }
        pt->time = p.time;
        for (int i = 0; i < 3; i++) {
            pt->position[i] = p.position[i];
            pt->inter_normal[i] = p.inter_normal[i];
            pt->geo_normal[i] = p.geo_normal[i];
        }
        pt->manifold_point[0] = p.mp[0];
        pt->manifold_point[1] = p.mp[1];
        pt->manifold[0] = p.manifold.array[0];
        pt->manifold[1] = p.manifold.array[1];
        pt->manifold_len = p.manifold.len;
        pt->material_slot = p.material_slot;
    }
}

// ---- render/renderer.rs ----------------------------------------------------------
void pixel_ndc(uint32_t x, uint32_t y, uint32_t w, uint32_t h, double s[2]) {   // renderer.rs:157-159
    s[0] = ((double)x - ((double)w - 1.0) / 2.0) / (double)w;
    s[1] = -((double)y - ((double)h - 1.0) / 2.0) / (double)h;
}
Ray view_get_ray(const rtg_view &v, const double s[2]) {                        // view.rs:13-19
    double x = s[0] * v.tan_half_fov[0];
    double y = s[1] * v.tan_half_fov[1];
    double z = -1.0;
    V3 d = normalize(mk(x, y, z));
    return Ray{mk(v.orig[0], v.orig[1], v.orig[2]), d};
}
V3 compute_direct_irrad(const orc_scene &s, const rtg_light *lights, uint32_t nl, V3 position,
                        V3 inter_normal, Counters &ctr) {                       // renderer.rs:184-205
    V3 lighting = mk(1.0, 1.0, 1.0) * 0.0;
    for (uint32_t i = 0; i < nl; i++) {
        V3 lp = mk(lights[i].position[0], lights[i].position[1], lights[i].position[2]);
        V3 color = mk(lights[i].color[0], lights[i].color[1], lights[i].color[2]);
        V3 disp = lp - position;
        double dis2 = dot(disp, disp);
        double dis = std::sqrt(dis2);
        V3 dir = disp / dis;
        double d = dot(dir, inter_normal);
        if (d < 0.0) continue;
        Ray ray = new_bounce(position, dir);
        RaycastPoint occlude;
        if (scene_raycast(s, ray, occlude, ctr)) {
            if (occlude.time < dis) continue;
        }
        lighting = lighting + color * d / (4.0 * PI * dis2);
    }
    return lighting;
}
struct PathPoint {
    RaycastPoint point;
    double attenuation;
    uint32_t depth, modes, parent;   // output_modes as bits (1 = Refract), index of the path bounced off last
};
// raytrace_all_specular_rec (renderer.rs:296-378) with no filters; `depth` is
// output_manifolds.len(), `max_depth` the reference's constant 4 (renderer.rs:312).
void all_specular_rec(const orc_scene &s, const Ray &ray, double attenuation, uint32_t depth,
                      uint32_t max_depth, std::vector<PathPoint> &output, Counters &ctr, uint32_t modes = 0,
                      uint32_t parent = RTG_NONE) {
    if (depth >= max_depth) return;
    RaycastPoint first;
    if (!scene_raycast(s, ray, first, ctr)) return;
    const uint32_t self = (uint32_t)output.size();
    output.push_back(PathPoint{first, attenuation, depth, modes, parent});
    if (first.material.has_dielectric) {
        Dielectric di = dielectric_new(ray.dir, first.geo_normal, first.material.n1, first.material.n2);
        Ray reflect = new_bounce(first.position, di.reflect);
        all_specular_rec(s, reflect, attenuation * di.reflectance, depth + 1, max_depth, output, ctr, modes, self);
        if (di.has_refract) {
            Ray refract = new_bounce(first.position, di.refract);
            all_specular_rec(s, refract, attenuation * (1.0 - di.reflectance), depth + 1, max_depth, output, ctr,
                             modes | (1u << depth), self);
        }
    }
}
// raytrace_pixel (renderer.rs:266-283) with the photon map empty: the indirect
// term is Color::default() and the ambient term 0.0 (renderer.rs:181-183).
V3 raytrace_pixel(const orc_scene &s, const rtg_light *lights, uint32_t nl, const Ray &ray,
                  uint32_t max_depth, bool &primary_some, RaycastPoint &primary, Counters &ctr) {
    std::vector<PathPoint> paths;
    all_specular_rec(s, ray, 1.0, 0, max_depth, paths, ctr);
    primary_some = !paths.empty();
    if (primary_some) primary = paths[0].point;
    V3 total = mk(0.0, 0.0, 0.0);
    for (const PathPoint &pp : paths) {
        V3 irrad = (mk(0.0, 0.0, 0.0) + compute_direct_irrad(s, lights, nl, pp.point.position, pp.point.inter_normal, ctr)) +
                   mk(0.0, 0.0, 0.0);
        total = total + map_mul(irrad, pp.point.material.diffuse) * pp.attenuation;
    }
    return total;
}

template <class F> void parallel_for(uint64_t n, int threads, F f) {
    if (threads <= 1 || n < 2) {
        for (uint64_t i = 0; i < n; i++) f(i);
        return;
    }
    std::atomic<uint64_t> next(0);
    const uint64_t chunk = 256;
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&]() {
            for (;;) {
                uint64_t b = next.fetch_add(chunk);
                if (b >= n) break;
                uint64_t e = std::min(n, b + chunk);
                for (uint64_t i = b; i < e; i++) f(i);
            }
        });
    for (auto &th : pool) th.join();
}
Material material_of_slot(const orc_scene &s, uint32_t slot) {
    const orc_scene::Obj &o = s.objects[slot / 2];
    return (slot & 1) ? o.material2 : o.material;
}
} // namespace

#include "rt_oracle_photon.hpp"

// =============================== C API ========================================
extern "C" {

double orc_minimum(double a, double b) { return f_minimum(a, b); }
double orc_maximum(double a, double b) { return f_maximum(a, b); }
int orc_real_cmp(double a, double b) { return real_cmp(a, b); }
double orc_dot3(const double a[3], const double b[3]) { return dot(mk(a[0], a[1], a[2]), mk(b[0], b[1], b[2])); }
void orc_cross(const double a[3], const double b[3], double out[3]) {
    V3 r = cross(mk(a[0], a[1], a[2]), mk(b[0], b[1], b[2]));
    for (int i = 0; i < 3; i++) out[i] = r[i];
}
void orc_normalize3(const double a[3], double out[3]) {
    V3 r = normalize(mk(a[0], a[1], a[2]));
    for (int i = 0; i < 3; i++) out[i] = r[i];
}
void orc_mat2_inverse(const double m[4], double out[4]) {
    Mat<2> a;
    std::memcpy(a.m, m, sizeof(a.m));
    Mat<2> r = inverse2(a);
    std::memcpy(out, r.m, sizeof(r.m));
}
void orc_mat3_inverse(const double m[9], double out[9]) {
    Mat<3> a;
    std::memcpy(a.m, m, sizeof(a.m));
    Mat<3> r = inverse3(a);
    std::memcpy(out, r.m, sizeof(r.m));
}
void orc_mat4_inverse(const double m[16], double out[16]) {
    Mat<4> a;
    std::memcpy(a.m, m, sizeof(a.m));
    Mat<4> r = inverse4(a);
    std::memcpy(out, r.m, sizeof(r.m));
}
void orc_mat4_mul(const double a[16], const double b[16], double out[16]) {
    Mat<4> x, y;
    std::memcpy(x.m, a, sizeof(x.m));
    std::memcpy(y.m, b, sizeof(y.m));
    Mat<4> r = matmul<4>(x, y);
    std::memcpy(out, r.m, sizeof(r.m));
}
void orc_mat3_mul(const double a[9], const double b[9], double out[9]) {
    Mat<3> x, y;
    std::memcpy(x.m, a, sizeof(x.m));
    std::memcpy(y.m, b, sizeof(y.m));
    Mat<3> r = matmul<3>(x, y);
    std::memcpy(out, r.m, sizeof(r.m));
}
void orc_mat2_mul(const double a[4], const double b[4], double out[4]) {
    Mat<2> x, y;
    std::memcpy(x.m, a, sizeof(x.m));
    std::memcpy(y.m, b, sizeof(y.m));
    Mat<2> r = matmul<2>(x, y);
    std::memcpy(out, r.m, sizeof(r.m));
}
// Mat4::from_scale_rotation_translation (mat.rs:161-177): S * R * T with R = I.
void orc_mat4_from_scale_rotation_translation(const double scale[3], const double translate[3], double out[16]) {
    Mat<4> s = identity<4>(), r = identity<4>(), t = identity<4>();
    for (int i = 0; i < 3; i++) {
        s.m[i][i] = scale[i];
        t.m[i][3] = translate[i];
    }
    s.m[3][3] = 1.0;
    Mat<4> m = matmul<4>(matmul<4>(s, r), t);
    std::memcpy(out, m.m, sizeof(m.m));
}
void orc_transform_from_forward(const double forward[16], rtg_transform *out) {
    Mat<4> f;
    std::memcpy(f.m, forward, sizeof(f.m));
    Transform t = transform_from(f);
    std::memcpy(out->forward, t.forward.m, sizeof(t.forward.m));
    std::memcpy(out->reverse, t.reverse.m, sizeof(t.reverse.m));
    std::memcpy(out->forward_norm, t.forward_norm.m, sizeof(t.forward_norm.m));
}
void orc_bounds_union(const double a[6], const double b[6], double out[6]) {
    Bounds r = bounds_union(Bounds{mk(a[0], a[1], a[2]), mk(a[3], a[4], a[5])},
                            Bounds{mk(b[0], b[1], b[2]), mk(b[3], b[4], b[5])});
    for (int i = 0; i < 3; i++) {
        out[i] = r.min[i];
        out[3 + i] = r.max[i];
    }
}
double orc_bounds_surface_area(const double b[6]) {
    return bounds_surface_area(Bounds{mk(b[0], b[1], b[2]), mk(b[3], b[4], b[5])});
}
double orc_smpte2048_encode(double x) { return smpte2048_encode(x); }
double orc_smpte2048_decode(double x) { return smpte2048_decode(x); }
uint8_t orc_pq8(double c) { return pq8(c); }
int orc_find_roots_quadratic(double a2, double a1, double a0, double out[2]) {
    return find_roots_quadratic(a2, a1, a0, out);
}

int orc_bounds_raycast(const double b[6], const double ray6[6], double out[2]) {
    Interval iv;
    if (!bounds_raycast(Bounds{mk(b[0], b[1], b[2]), mk(b[3], b[4], b[5])}, ray_from6(ray6), iv)) return 0;
    out[0] = iv.min;
    out[1] = iv.max;
    return 1;
}
int orc_triangle_raycast(const double tri9[9], const double ray6[6], int manifold, double out[12]) {
    V3 v[3];
    for (int i = 0; i < 3; i++) v[i] = mk(tri9[i * 3], tri9[i * 3 + 1], tri9[i * 3 + 2]);
    TrianglePoint tp;
    if (!triangle_raycast(v, ray_from6(ray6), manifold != 0, tp)) return 0;
    out[0] = tp.time;
    for (int i = 0; i < 3; i++) {
        out[1 + i] = tp.bary[i];
        out[4 + i] = tp.position[i];
        out[9 + i] = tp.geo_normal[i];
    }
    out[7] = tp.mp[0];
    out[8] = tp.mp[1];
    return 1;
}
int orc_sphere_raycast(const double c[3], double radius, const double ray6[6], rtg_point *out) {
    RaycastPoint p;
    bool some = sphere_raycast(mk(c[0], c[1], c[2]), radius, ray_from6(ray6), p);
    point_to_c(some, p, nullptr, out);
    return some ? 1 : 0;
}
int orc_plane_raycast(const double position[3], const double tan1[3], const double tan2[3],
                      const double ray6[6], rtg_point *out, int *out_material2) {
    Plane pl{mk(position[0], position[1], position[2]), mk(tan1[0], tan1[1], tan1[2]), mk(tan2[0], tan2[1], tan2[2])};
    RaycastPoint p;
    bool second = false;
    bool some = plane_raycast(pl, ray_from6(ray6), p, second);
    p.material_slot = second ? 1 : 0;
    point_to_c(some, p, nullptr, out);
    if (out_material2) *out_material2 = second ? 1 : 0;
    return some ? 1 : 0;
}
void orc_view_get_ray(const rtg_view *view, const double s[2], double out[6]) {
    Ray r = view_get_ray(*view, s);
    for (int i = 0; i < 3; i++) {
        out[i] = r.orig[i];
        out[3 + i] = r.dir[i];
    }
}
void orc_pixel_ndc(uint32_t x, uint32_t y, uint32_t w, uint32_t h, double out_s[2]) { pixel_ndc(x, y, w, h, out_s); }
void orc_dielectric(const double inc[3], const double sn[3], double n1, double n2, double out[8]) {
    Dielectric d = dielectric_new(mk(inc[0], inc[1], inc[2]), mk(sn[0], sn[1], sn[2]), n1, n2);
    out[0] = d.reflectance;
    for (int i = 0; i < 3; i++) {
        out[1 + i] = d.reflect[i];
        out[5 + i] = d.refract[i];
    }
    out[4] = d.has_refract ? 1.0 : 0.0;
}

orc_bvh *orc_bvh_build_subdivide(const rtg_tri *tris, uint64_t n, const double *vertex_normals) {
    return orc_bvh_build(tris, n, vertex_normals, 0);
}
orc_bvh *orc_bvh_build(const rtg_tri *tris, uint64_t n, const double *vertex_normals, int strategy) {
    std::vector<BvhTree> forest((size_t)n);                                     // BvhForest::new (bvh.rs:48-58)
    for (uint64_t i = 0; i < n; i++) {
        TriVerts tv = leaf_tri(tris[i]);
        forest[i].bounds = bounds_union(forest[i].bounds, triangle_bounds(tv.p)); // add_leaf (bvh.rs:68-71)
        forest[i].leaves.push_back((uint32_t)i);
    }
    BvhTree tree;
    if (strategy == 1) {                                                        // contract (bvh_cluster.rs:43-48)
        while (forest.size() > 1) contract_once(forest);
        if (!forest.empty()) tree = std::move(forest[0]);
    } else {
        if (strategy == 2) contract_once(forest);                               // hybrid (bvh_joint.rs:4-7)
        tree = subdivide(std::move(forest));
    }
    orc_bvh *b = new orc_bvh();                                                 // Bvh::new (bvh.rs:80-90)
    rtg_bvh_node root = add_bvh_tree(*b, tree, tris, vertex_normals);
    b->root = (uint32_t)b->nodes.size();
    b->nodes.push_back(root);
    return b;
}
orc_bvh *orc_bvh_from_flat(const rtg_bvh *flat) {
    orc_bvh *b = new orc_bvh();
    b->nodes.assign(flat->nodes, flat->nodes + flat->n_nodes);
    b->leaves.assign(flat->leaves, flat->leaves + flat->n_leaves);
    if (flat->vertex_normals) b->normals.assign(flat->vertex_normals, flat->vertex_normals + flat->n_leaves * 9);
    b->leaf_source.resize(flat->n_leaves);
    for (uint64_t i = 0; i < flat->n_leaves; i++) b->leaf_source[i] = (uint32_t)i;
    b->root = flat->root;
    return b;
}
void orc_bvh_destroy(orc_bvh *b) { delete b; }
uint64_t orc_bvh_n_nodes(const orc_bvh *b) { return b->nodes.size(); }
uint64_t orc_bvh_n_leaves(const orc_bvh *b) { return b->leaves.size(); }
uint32_t orc_bvh_root(const orc_bvh *b) { return b->root; }
const rtg_bvh_node *orc_bvh_nodes(const orc_bvh *b) { return b->nodes.data(); }
const rtg_tri *orc_bvh_leaves(const orc_bvh *b) { return b->leaves.data(); }
const double *orc_bvh_vertex_normals(const orc_bvh *b) { return b->normals.empty() ? nullptr : b->normals.data(); }
const uint32_t *orc_bvh_leaf_source(const orc_bvh *b) { return b->leaf_source.data(); }
uint32_t orc_bvh_depth(const orc_bvh *b) { return b->nodes.empty() ? 0 : depth_of(*b, b->root); }

int orc_bvh_raycast(const orc_bvh *b, const double ray6[6], rtg_hit *out, rtg_ray_counters *ctr) {
    Counters c;
    RaycastPoint p;
    bool some = bvh_raycast(*b, ray_from6(ray6), p, c);
    out->object = RTG_NONE;
    out->leaf = some ? p.manifold.array[0] : RTG_NONE;
    out->t = some ? p.time : INF;
    if (ctr) {
        ctr->nodes_visited = (uint32_t)c.nodes;
        ctr->tris_tested = (uint32_t)c.tris;
    }
    return some ? 1 : 0;
}
int orc_bvh_raycast_brute(const orc_bvh *b, const double ray6[6], rtg_hit *out) {
    Holder h;
    Ray ray = ray_from6(ray6);
    for (uint32_t i = 0; i < b->leaves.size(); i++) {
        RaycastPoint p;
        bool some = leaf_raycast(*b, i, ray, false, p);
        if (some) p.manifold = p.manifold.push(i);
        h.add(some, p);
    }
    out->object = RTG_NONE;
    out->leaf = h.has ? h.point.manifold.array[0] : RTG_NONE;
    out->t = h.has ? h.point.time : INF;
    return h.has ? 1 : 0;
}

void orc_trispline_from_normal_hermite(const double in30[30], rtg_trispline *out) {
    V3 in[10];
    for (int i = 0; i < 10; i++) in[i] = mk(in30[i * 3], in30[i * 3 + 1], in30[i * 3 + 2]);
    trispline_to_c(trispline_from_normal_hermite(in), *out);
}
/* out12 = position, ddt, ddu, normal */
void orc_trispline_eval(const rtg_trispline *sp, double t, double u, double out12[12]) {
    Trispline s = trispline_from_c(*sp);
    V3 v[4] = {trispline_position(s, t, u), trispline_ddt(s, t, u), trispline_ddu(s, t, u), trispline_normal(s, t, u)};
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 3; k++) out12[i * 3 + k] = v[i][k];
}
/* k*k facets per patch, in the order: for i, for j: "up" facet, then (when it exists) "down" facet */
void orc_patches_tessellate(const rtg_trispline *patches, uint64_t n, uint32_t k, rtg_tri *out_tris,
                            rtg_patch_leaf *out_info) {
    uint64_t w = 0;
    for (uint64_t p = 0; p < n; p++) {
        Trispline s = trispline_from_c(patches[p]);
        auto emit = [&](const double tu[3][2]) {
            for (int c = 0; c < 3; c++) {
                V3 q = trispline_position(s, tu[c][0], tu[c][1]);
                for (int a = 0; a < 3; a++) out_tris[w].p[c][a] = q[a];
                out_info[w].tu[c][0] = tu[c][0];
                out_info[w].tu[c][1] = tu[c][1];
            }
            out_info[w].patch = (uint32_t)p;
            out_info[w]._pad = 0;
            w++;
        };
        for (uint32_t i = 0; i < k; i++)
            for (uint32_t j = 0; j + i < k; j++) {
                double t0 = (double)i / (double)k, t1 = (double)(i + 1) / (double)k;
                double u0 = (double)j / (double)k, u1 = (double)(j + 1) / (double)k;
                const double up[3][2] = {{t0, u0}, {t1, u0}, {t0, u1}};
                emit(up);
                if (j + i + 1 < k) {
                    const double down[3][2] = {{t1, u0}, {t1, u1}, {t0, u1}};
                    emit(down);
                }
            }
    }
}
/* leaf_info in leaf order (as rtg_patch_set) */
void orc_bvh_set_patches(orc_bvh *b, const rtg_patch_set *ps) {
    b->patches.assign(ps->patches, ps->patches + ps->n_patches);
    b->leaf_info.assign(ps->leaf_info, ps->leaf_info + b->leaves.size());
    b->newton_steps = ps->newton_steps;
}

orc_scene *orc_scene_create(const rtg_scene_desc *desc) {
    orc_scene *s = new orc_scene();
    for (uint32_t i = 0; i < desc->n_objects; i++) {
        const rtg_object &o = desc->objects[i];
        orc_scene::Obj x;
        x.kind = o.kind;
        std::memcpy(x.transform.forward.m, o.transform.forward, sizeof(x.transform.forward.m));
        std::memcpy(x.transform.reverse.m, o.transform.reverse, sizeof(x.transform.reverse.m));
        std::memcpy(x.transform.forward_norm.m, o.transform.forward_norm, sizeof(x.transform.forward_norm.m));
        x.transform.reverse_norm = identity<3>();
        x.material = material_from(o.material);
        x.material2 = material_from(o.material2);
        x.sphere_center = mk(o.sphere_center[0], o.sphere_center[1], o.sphere_center[2]);
        x.sphere_radius = o.sphere_radius;
        x.plane = Plane{mk(o.plane_position[0], o.plane_position[1], o.plane_position[2]),
                        mk(o.plane_tan1[0], o.plane_tan1[1], o.plane_tan1[2]),
                        mk(o.plane_tan2[0], o.plane_tan2[1], o.plane_tan2[2])};
        x.mesh = reinterpret_cast<const orc_bvh *>(o.mesh);
        s->objects.push_back(x);
    }
    return s;
}
void orc_scene_destroy(orc_scene *s) { delete s; }

int orc_render_primary(const orc_scene *s, const rtg_view *view, uint32_t width, uint32_t height,
                       rtg_rect tile, rtg_hit *out_hits, rtg_point *out_points,
                       rtg_ray_counters *out_counters, int threads) {
    uint64_t n = (uint64_t)tile.w * tile.h;
    parallel_for(n, threads, [&](uint64_t i) {
        uint32_t x = tile.x0 + (uint32_t)(i % tile.w), y = tile.y0 + (uint32_t)(i / tile.w);
        double sxy[2];
        pixel_ndc(x, y, width, height, sxy);
        Ray ray = view_get_ray(*view, sxy);
        Counters c;
        RaycastPoint p;
        bool some = scene_raycast(*s, ray, p, c);
        point_to_c(some, p, out_hits ? out_hits + i : nullptr, out_points ? out_points + i : nullptr);
        if (out_counters) {
            out_counters[i].nodes_visited = (uint32_t)c.nodes;
            out_counters[i].tris_tested = (uint32_t)c.tris;
        }
    });
    return 0;
}
int orc_trace(const orc_scene *s, const double *rays6, uint64_t n, rtg_hit *out_hits,
              rtg_point *out_points, rtg_ray_counters *out_counters, int threads) {
    parallel_for(n, threads, [&](uint64_t i) {
        Ray ray = ray_from6(rays6 + i * 6);
        Counters c;
        RaycastPoint p;
        bool some = scene_raycast(*s, ray, p, c);
        point_to_c(some, p, out_hits ? out_hits + i : nullptr, out_points ? out_points + i : nullptr);
        if (out_counters) {
            out_counters[i].nodes_visited = (uint32_t)c.nodes;
            out_counters[i].tris_tested = (uint32_t)c.tris;
        }
    });
    return 0;
}
int orc_trace_paths(const orc_scene *s, const double *rays6, uint64_t n, uint32_t max_depth, rtg_path *out,
                    uint64_t capacity, uint64_t *n_paths) {
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; i++) {
        std::vector<PathPoint> paths;
        Counters c;
        all_specular_rec(*s, ray_from6(rays6 + i * 6), 1.0, 0, max_depth, paths, c);
        for (const PathPoint &pp : paths) {
            if (total < capacity) {
                rtg_path &o = out[total];
                std::memset(&o, 0, sizeof(o));
                o.ray = (uint32_t)i;
                o.depth = pp.depth;
                o.modes = pp.modes;
                o.parent = pp.parent == RTG_NONE ? RTG_NONE : (uint32_t)(total - (&pp - paths.data()) + pp.parent);
                o.attenuation = pp.attenuation;
                point_to_c(true, pp.point, nullptr, &o.point);
            }
            total++;
        }
    }
    *n_paths = total;
    return total <= capacity ? 0 : -1;
}
int orc_shade_direct(const orc_scene *s, const rtg_light *lights, uint32_t n_lights,
                     const rtg_point *points, uint64_t n, double *rgb_out, uint8_t *rgba8_out, int threads) {
    parallel_for(n, threads, [&](uint64_t i) {
        const rtg_point &pt = points[i];
        V3 total = mk(0.0, 0.0, 0.0);
        if (pt.manifold_len != 0) {
            Counters c;
            V3 pos = mk(pt.position[0], pt.position[1], pt.position[2]);
            V3 nrm = mk(pt.inter_normal[0], pt.inter_normal[1], pt.inter_normal[2]);
            V3 irrad = (mk(0.0, 0.0, 0.0) + compute_direct_irrad(*s, lights, n_lights, pos, nrm, c)) + mk(0.0, 0.0, 0.0);
            total = total + map_mul(irrad, material_of_slot(*s, pt.material_slot).diffuse) * 1.0;
        }
        if (rgb_out)
            for (int k = 0; k < 3; k++) rgb_out[i * 3 + k] = total[k];
        if (rgba8_out) {
            for (int k = 0; k < 3; k++) rgba8_out[i * 4 + k] = pq8(total[k]);
            rgba8_out[i * 4 + 3] = 255;
        }
    });
    return 0;
}
int orc_render_frame(const orc_scene *s, const rtg_view *view, const rtg_light *lights,
                     uint32_t n_lights, uint32_t width, uint32_t height, rtg_rect tile,
                     uint32_t max_depth, double *out_rgb, uint8_t *out_rgba8, rtg_hit *out_hits, int threads) {
    uint64_t n = (uint64_t)tile.w * tile.h;
    parallel_for(n, threads, [&](uint64_t i) {
        uint32_t x = tile.x0 + (uint32_t)(i % tile.w), y = tile.y0 + (uint32_t)(i / tile.w);
        double sxy[2];
        pixel_ndc(x, y, width, height, sxy);
        Ray ray = view_get_ray(*view, sxy);
        Counters c;
        bool some = false;
        RaycastPoint primary;
        V3 total = raytrace_pixel(*s, lights, n_lights, ray, max_depth, some, primary, c);
        if (out_hits) point_to_c(some, primary, out_hits + i, nullptr);
        if (out_rgb)
            for (int k = 0; k < 3; k++) out_rgb[i * 3 + k] = total[k];
        if (out_rgba8) {
            for (int k = 0; k < 3; k++) out_rgba8[i * 4 + k] = pq8(total[k]);
            out_rgba8[i * 4 + 3] = 255;
        }
    });
    return 0;
}

} // extern "C"

// ---- photon map -----------------------------------------------------------------------
struct orc_kd {
    std::vector<gen::KdEntry> tree;
};

extern "C" {

// op: 0 add, 1 sub, 2 mul, 3 div, 4 neg, 5 sqrt, 6 sin, 7 cos, 8 abs, 9 minimum, 10 maximum (a, b, out = v, d0, d1)
void orc_der2_op(int op, const double a[3], const double b[3], double out[3]) {
    gen::Der2 x{a[0], {a[1], a[2]}}, y{b[0], {b[1], b[2]}}, r{QNAN, {QNAN, QNAN}};
    switch (op) {
    case 0: r = x + y; break;
    case 1: r = x - y; break;
    case 2: r = x * y; break;
    case 3: r = x / y; break;
    case 4: r = -x; break;
    case 5: r = gen::s_sqrt(x); break;
    case 6: r = gen::s_sin(x); break;
    case 7: r = gen::s_cos(x); break;
    case 8: r = gen::s_abs(x); break;
    case 9: r = gen::s_minimum(x, y); break;
    case 10: r = gen::s_maximum(x, y); break;
    default: break;
    }
    out[0] = r.v;
    out[1] = r.d[0];
    out[2] = r.d[1];
}
// derived PartialOrd of Der<2>: -1, 0, 1, or 2 when the values have no order
int orc_der2_partial_cmp(const double a[3], const double b[3]) {
    return gen::der_partial_cmp(gen::Der2{a[0], {a[1], a[2]}}, gen::Der2{b[0], {b[1], b[2]}});
}
// the reference's own Der test (math/scalar.rs:360-367): var(1,0) + var(2,1) + from(3)
void orc_der2_known_answer(double out[3]) {
    gen::Der2 f = gen::der_var(1.0, 0) + gen::der_var(2.0, 1) + gen::der_from(3.0);
    out[0] = f.v;
    out[1] = f.d[0];
    out[2] = f.d[1];
}
void orc_zenith_y_into_normal(const double dir2[2], double out[3]) {
    gen::Vec3T<double> n = gen::zenith_y_into_normal<double>(dir2[0], dir2[1]);
    for (int i = 0; i < 3; i++) out[i] = n[i];
}
// Sphere::fibonacci_sphere (geo/sphere.rs:52-80) with the two uniform draws per direction supplied by the caller
void orc_fibonacci_sphere(uint64_t count, const double *uniform2, double *out_dirs2) {
    const double c = (double)count;
    const double phi = PI * (3.0 - std::sqrt(5.0));
    const double d = 2.0 * PI / phi;
    const double dydi = 2.0 / (c + d);
    const double dydj = dydi * d;
    for (uint64_t i = 0; i < count; i++) {
        double x = (double)i + uniform2[i * 2], y = uniform2[i * 2 + 1];
        out_dirs2[i * 2] = std::fmod(phi * x, 2.0 * PI);
        out_dirs2[i * 2 + 1] = -1.0 + dydi * x + dydj * y;
    }
}

orc_kd *orc_kd_create(const double *positions3, uint64_t n) {
    orc_kd *kd = new orc_kd();
    kd->tree.resize(n);
    for (uint64_t i = 0; i < n; i++) {
        kd->tree[i].position = mk(positions3[i * 3], positions3[i * 3 + 1], positions3[i * 3 + 2]);
        kd->tree[i].value = (uint32_t)i;
    }
    gen::kd_build_rec(kd->tree.data(), kd->tree.size());
    return kd;
}
void orc_kd_destroy(orc_kd *kd) { delete kd; }
// KdTree::assert_kd_tree over every subtree (tree/kd_tree.rs:98-114 checks the root's split): 1 when it holds
static bool kd_check_rec(const gen::KdEntry *t, size_t len) {
    if (len == 0) return true;
    size_t c = len / 2;
    int axis = t[c].axis;
    for (size_t i = 0; i < c; i++)
        if (!(t[i].position[axis] <= t[c].position[axis])) return false;
    for (size_t i = c + 1; i < len; i++)
        if (!(t[c].position[axis] <= t[i].position[axis])) return false;
    return kd_check_rec(t, c) && kd_check_rec(t + c + 1, len - c - 1);
}
int orc_kd_check(const orc_kd *kd) { return kd_check_rec(kd->tree.data(), kd->tree.size()) ? 1 : 0; }
// KdTree::nearest per query; slots beyond the number of entries: RTG_NONE / +inf
void orc_kd_nearest(const orc_kd *kd, const double *queries3, uint64_t n_queries, uint32_t k, uint32_t *out_index,
                    double *out_distance, int threads) {
    parallel_for(n_queries, threads, [&](uint64_t q) {
        std::vector<gen::KdNeighbor> nb = gen::kd_nearest(kd->tree, mk(queries3[q * 3], queries3[q * 3 + 1], queries3[q * 3 + 2]), k);
        for (uint32_t j = 0; j < k; j++) {
            out_index[q * k + j] = j < nb.size() ? nb[j].value : RTG_NONE;
            out_distance[q * k + j] = j < nb.size() ? nb[j].distance : INF;
        }
    });
}
// what the reference's test compares `nearest` with (tree/kd_tree.rs:262-264): every distance, sorted (by
// distance under real_cmp, then by value)
void orc_kd_nearest_brute(const double *positions3, uint64_t n, const double *queries3, uint64_t n_queries, uint32_t k,
                          uint32_t *out_index, double *out_distance, int threads) {
    parallel_for(n_queries, threads, [&](uint64_t q) {
        V3 query = mk(queries3[q * 3], queries3[q * 3 + 1], queries3[q * 3 + 2]);
        std::vector<gen::KdNeighbor> all(n);
        for (uint64_t i = 0; i < n; i++)
            all[i] = gen::KdNeighbor{length(query - mk(positions3[i * 3], positions3[i * 3 + 1], positions3[i * 3 + 2])), (uint32_t)i};
        size_t keep = std::min<size_t>(k, n);
        std::partial_sort(all.begin(), all.begin() + keep, all.end(), [](const gen::KdNeighbor &a, const gen::KdNeighbor &b) {
            int c = real_cmp(a.distance, b.distance);
            return c != 0 ? c < 0 : a.value < b.value;
        });
        for (uint32_t j = 0; j < k; j++) {
            out_index[q * k + j] = j < keep ? all[j].value : RTG_NONE;
            out_distance[q * k + j] = j < keep ? all[j].distance : INF;
        }
    });
}

// the photon pass of Renderer::render (render/renderer.rs:116-138) with the directions supplied
int orc_photons_emit(const orc_scene *s, const rtg_light *lights, uint32_t n_lights, const double *dirs2,
                     uint64_t count_per_light, uint32_t max_depth, rtg_photon *out, uint64_t capacity, uint64_t *n_photons) {
    uint64_t total = 0;
    std::vector<PathPoint> paths;       // one buffer for all rays
    for (uint32_t l = 0; l < n_lights; l++) {
        const V3 origin = mk(lights[l].position[0], lights[l].position[1], lights[l].position[2]);
        const V3 color = mk(lights[l].color[0], lights[l].color[1], lights[l].color[2]);
        for (uint64_t i = 0; i < count_per_light; i++) {
            const double *dir = dirs2 + (l * count_per_light + i) * 2;
            gen::Vec3T<double> n = gen::zenith_y_into_normal<double>(dir[0], dir[1]);
            paths.clear();
            Counters c;
            all_specular_rec(*s, Ray{origin, mk(n[0], n[1], n[2])}, 1.0, 0, max_depth, paths, c);
            for (const PathPoint &pp : paths) {
                if (pp.depth == 0) continue;
                if (total < capacity) {
                    rtg_photon &o = out[total];
                    std::memset(&o, 0, sizeof(o));
                    V3 light = color * pp.attenuation;
                    for (int k = 0; k < 3; k++) {
                        o.position[k] = pp.point.position[k];
                        o.origin[k] = origin[k];
                        o.light[k] = light[k];
                        o.surface_object[k] = RTG_NONE;
                        o.surface_leaf[k] = RTG_NONE;
                    }
                    o.dir[0] = dir[0];
                    o.dir[1] = dir[1];
                    o.n_bounces = pp.depth;
                    o.modes = pp.modes;
                    o.light_index = l;
                    uint32_t at = pp.parent;                     // ancestors, deepest first
                    for (uint32_t d = pp.depth; d-- > 0;) {
                        const PathPoint &a = paths[at];
                        if (d < 3) {
                            const Manifold &m = a.point.manifold;
                            o.surface_object[d] = m.array[m.len - 1];
                            o.surface_leaf[d] = m.len == 2 ? m.array[0] : RTG_NONE;
                        }
                        at = a.parent;
                    }
                }
                total++;
            }
        }
    }
    *n_photons = total;
    return total <= capacity ? 0 : -1;
}

// compute_indirect_irrad (render/renderer.rs:206-265) per point; photons indexed as given to orc_kd_create
int orc_indirect_irrad(const orc_scene *s, const orc_kd *kd, const rtg_photon *photons, const rtg_point *points,
                       uint64_t n_points, uint32_t photon_samples, uint32_t newton_steps, double newton_epsilon,
                       uint32_t max_depth, double *rgb_out, int threads) {
    parallel_for(n_points, threads, [&](uint64_t i) {
        V3 c = gen::compute_indirect_irrad(*s, kd->tree, photons, points[i], photon_samples, newton_steps, newton_epsilon, max_depth);
        for (int k = 0; k < 3; k++) rgb_out[i * 3 + k] = c[k];
    });
    return 0;
}
// One photon's corrected landing point, for inspection: the Newton iteration and the final full trace of
// compute_indirect_irrad for photon `ph` against point `p`.  out: dir[2] after the steps, landed (0/1), position v[3],
// d/da[3], d/db[3], residual separation after each step (2 * newton_steps, NaN when the step did not run).
void orc_photon_adjust(const orc_scene *s, const rtg_photon *ph, const rtg_point *p, uint32_t newton_steps, uint32_t max_depth,
                       double *out12, double *out_seps) {
    using namespace gen;
    double dir[2] = {ph->dir[0], ph->dir[1]};
    std::vector<OptManifold> fm;
    std::vector<int> modes;
    for (uint32_t i = 0; i < ph->n_bounces; i++) {
        fm.push_back(OptManifold{true, manifold_of(ph->surface_object[i], ph->surface_leaf[i])});
        modes.push_back((ph->modes >> i) & 1u);
    }
    fm.push_back(OptManifold{true, Manifold{{p->manifold[0], p->manifold[1]}, p->manifold_len}});
    const Vec3T<Der2> origin = vcast<Der2>(mk(ph->origin[0], ph->origin[1], ph->origin[2]));
    for (uint32_t k = 0; k < 2 * newton_steps; k++) out_seps[k] = QNAN;
    for (uint32_t step = 0; step < newton_steps; step++) {
        RayT<Der2> ray{origin, zenith_y_into_normal(der_var(dir[0], 0), der_var(dir[1], 1))};
        std::vector<PathT<Der2>> hit;
        std::vector<Manifold> om;
        std::vector<int> omodes;
        all_specular_filtered(*s, ray, der_from(1.0), fm.data(), fm.size(), modes.data(), modes.size(), true, max_depth, om, omodes, hit);
        if (hit.empty()) break;
        Der2 sep[2] = {hit[0].point.mp[0] - der_from(p->manifold_point[0]), hit[0].point.mp[1] - der_from(p->manifold_point[1])};
        out_seps[step * 2] = sep[0].v;
        out_seps[step * 2 + 1] = sep[1].v;
        Mat<2> jac = {{{sep[0].d[0], sep[0].d[1]}, {sep[1].d[0], sep[1].d[1]}}};
        Mat<2> inv = inverse2(jac);
        const double sv[2] = {sep[0].v, sep[1].v};
        double st[2];
        matvec(inv, sv, st);
        dir[0] = dir[0] - st[0];
        dir[1] = dir[1] - st[1];
    }
    RayT<Der2> ray{origin, zenith_y_into_normal(der_var(dir[0], 0), der_var(dir[1], 1))};
    std::vector<PathT<Der2>> real_photon;
    std::vector<Manifold> om;
    std::vector<int> omodes;
    all_specular_filtered(*s, ray, der_from(1.0), (const OptManifold *)nullptr, 0, modes.data(), modes.size(), true, max_depth, om,
                          omodes, real_photon);
    out12[0] = dir[0];
    out12[1] = dir[1];
    out12[2] = real_photon.empty() ? 0.0 : 1.0;
    for (int k = 0; k < 3; k++) {
        out12[3 + k] = real_photon.empty() ? QNAN : real_photon[0].point.position[k].v;
        out12[6 + k] = real_photon.empty() ? QNAN : real_photon[0].point.position[k].d[0];
        out12[9 + k] = real_photon.empty() ? QNAN : real_photon[0].point.position[k].d[1];
    }
}

// raytrace_pixel (render/renderer.rs:266-283) with the photon map in place, per pixel of the tile
int orc_render_frame_photons(const orc_scene *s, const rtg_view *view, const rtg_light *lights, uint32_t n_lights,
                             const orc_kd *kd, const rtg_photon *photons, uint32_t photon_samples, uint32_t newton_steps,
                             double newton_epsilon, uint32_t width, uint32_t height, rtg_rect tile, uint32_t max_depth,
                             double *out_rgb, uint8_t *out_rgba8, int threads) {
    uint64_t n = (uint64_t)tile.w * tile.h;
    parallel_for(n, threads, [&](uint64_t i) {
        uint32_t x = tile.x0 + (uint32_t)(i % tile.w), y = tile.y0 + (uint32_t)(i / tile.w);
        double sxy[2];
        pixel_ndc(x, y, width, height, sxy);
        Ray ray = view_get_ray(*view, sxy);
        Counters c;
        std::vector<PathPoint> paths;
        all_specular_rec(*s, ray, 1.0, 0, max_depth, paths, c);
        V3 total = mk(0.0, 0.0, 0.0);
        for (const PathPoint &pp : paths) {
            rtg_point pt;
            point_to_c(true, pp.point, nullptr, &pt);
            V3 indirect = gen::compute_indirect_irrad(*s, kd->tree, photons, pt, photon_samples, newton_steps, newton_epsilon, 4);
            V3 irrad = (indirect + compute_direct_irrad(*s, lights, n_lights, pp.point.position, pp.point.inter_normal, c)) +
                       mk(0.0, 0.0, 0.0);
            total = total + map_mul(irrad, pp.point.material.diffuse) * pp.attenuation;
        }
        if (out_rgb)
            for (int k = 0; k < 3; k++) out_rgb[i * 3 + k] = total[k];
        if (out_rgba8) {
            for (int k = 0; k < 3; k++) out_rgba8[i * 4 + k] = pq8(total[k]);
            out_rgba8[i * 4 + 3] = 255;
        }
    });
    return 0;
}

} // extern "C"
