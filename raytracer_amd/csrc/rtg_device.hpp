// rtg_device.hpp -- device-side geometry, intersection and traversal for gfx950.
//
// All arithmetic is IEEE binary64 with contraction off (the translation unit is
// built with -ffp-contract=off) and follows the operation order of the
// reference renderer so that hit ids and hit times are bit-identical to the
// CPU renderer.  Citations are relative to the reference's src/ directory.
//
// Layout of the data this code walks (built by rtg_api.cpp):
//   DevNode  64 B, 16-B aligned  -> 4 x global_load_dwordx4 per node visit
//   DevTri   80 B, 16-B aligned  -> 5 x global_load_dwordx4 per triangle test
//   DevObject                    -> wave-uniform, read through scalar loads
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtg.h"
#include "rtg_pair_slots.hpp"

namespace rtg {

constexpr double kInf = __builtin_huge_val();
constexpr double kPi = 3.14159265358979323846264338327950288;

// ---------------------------------------------------------------- layouts --
struct alignas(16) DevNode {
    double lo[3];
    double hi[3];
    uint32_t link;       // stack layout: first child index; threaded layout: skip (miss) link
    uint32_t nchild;     // stack layout: bit31 = last sibling, bits 0..30 child count; threaded layout: pass link
                         // (position of the first child, or the skip link again when the node has no children)
    uint32_t leaf_begin;
    uint32_t leaf_count;
};
static_assert(sizeof(DevNode) == 64, "node must be 4 x 16 B");

struct alignas(16) DevTri {
    double p[9];
    uint32_t source;     // index of the triangle in the caller's leaf array (== leaf index)
    uint32_t pad;
};
static_assert(sizeof(DevTri) == 80, "triangle must be 5 x 16 B");

// Threaded nodes name both successors by position, so the array may be in any
// order: `skip` is where a walk goes on when the box test fails (the node after
// the subtree in the reference's visit order, or n_nodes at the end), the pass
// link where it goes on when it passes.  The host and device builders place the
// children of a node side by side and the root last (DevMesh::troot): a visit of
// a passed node's second child then finds it in the 64-byte half line the first
// child came with.  The pairs under the parents of the top kThreadedTopLevels
// levels come first, level by level and in pre-order within a level -- the part
// of the tree every ray walks is one compact block --, the pairs of all deeper
// parents follow in pre-order of the parents.  Trees in which a node has leaves
// and children at once keep the pre-order array.
// (kThreadedTopLevels and the placement rules below the top block: rtg_pair_slots.hpp)
//
// 32-byte companion of a threaded node for the single-precision pre-test: the
// box rounded to f32 with the two planes of an axis side by side (lo.x, hi.x,
// lo.y, hi.y, lo.z, hi.z: one packed f32 operation handles an axis), the skip
// link, and one word that is either the pass link of a node without leaves
// (bit 31 clear), the node's only leaf (bit 31 set, bit 30 clear: a leaf node
// goes on to its skip link), or the mark of any other node (bits 31 and 30 set:
// several leaves, or leaves and children -- leaf range and pass link are read
// from the full node).
struct alignas(16) HotNode {
    float planes[6];
    uint32_t skip;
    uint32_t next;
};
static_assert(sizeof(HotNode) == 32, "hot node must be 2 x 16 B");
constexpr uint32_t kHotLeaf = 0x80000000u, kHotRange = 0xC0000000u, kHotIndexMask = 0x3FFFFFFFu;
__host__ __device__ inline uint32_t hot_next_word(uint32_t skip, uint32_t pass, uint32_t leaf_begin, uint32_t leaf_count) {
    if (leaf_count == 0) return pass;
    if (leaf_count == 1 && pass == skip && leaf_begin < 0x40000000u) return kHotLeaf | leaf_begin;
    return kHotRange;
}

// Everything the persistent kernel reads of a mesh sits in one allocation so a
// lane keeps one pointer per mesh:
//   [MeshSlabHeader 64 B][HotNode x n_nodes][threaded DevNode x n_nodes][DevTri x n_leaves]
// hot node i at +64 + 32 i, threaded node i at +64 + 32 n + 64 i, triangle l at
// +64 + 96 n + 80 l (all 16-byte aligned); hot node and threaded node of one
// tree node share their position i.
struct alignas(16) MeshSlabHeader {
    const rtg_trispline *patches;     // non-null: leaves are facets of trispline patches
    const rtg_patch_leaf *leaf_info;
    uint32_t newton_steps;
    uint32_t pad[11];
};
static_assert(sizeof(MeshSlabHeader) == 64, "slab header must be 64 B");

struct DevMesh {
    const DevNode *nodes;        // reference order (children after descendants, root last)
    const DevNode *tnodes;       // threaded with pass and skip links, walk starts at troot
    const DevTri *tris;
    const double *normals;       // 9 per leaf or nullptr
    uint32_t n_nodes;
    uint32_t root;
    uint32_t depth;
    uint32_t fast_div_ok;        // every box coordinate is 0 or has 2^-600 <= |c| <= 2^400
    const uint8_t *slab;              // MeshSlabHeader + hot nodes + threaded nodes + triangles (tnodes, tris point into it)
    float coord_max;                  // largest |box coordinate|
    uint32_t f32_ok;                  // every box coordinate is 0 or has 2^-40 <= |c| <= 2^40
    const rtg_trispline *patches;     // non-null: leaves are facets of trispline patches
    const rtg_patch_leaf *leaf_info;  // per leaf: patch index and corner parameters
    uint32_t newton_steps;
    uint32_t troot;                   // position of the root in the threaded arrays
};

struct DevMaterial {
    double diffuse[3];
    double n1, n2;
    int32_t has_dielectric;
    int32_t pad;
};

struct DevObject {
    int32_t kind;
    int32_t index;       // position in the caller's object list (the id reported in hits)
    double reverse[16];
    double forward_norm[9];
    DevMaterial material, material2;
    double sphere_center[3];
    double sphere_radius;
    double plane_position[3];
    double plane_tan1[3];
    double plane_tan2[3];
    DevMesh mesh;
};

struct DevScene {
    const DevObject *objects;
    uint32_t n_objects;
    uint32_t traversal;          // 0 = short stack, 1 = threaded
    uint32_t *overflow;          // global stack overflow slab (may be null when unused)
    uint32_t overflow_levels;    // levels available per thread in the slab
    uint32_t overflow_stride;    // threads the slab was sized for
    uint32_t has_patches;        // some object is a trispline patch set: launch the PATCH kernel variants
    uint32_t any_f32_ok;         // some mesh allows the single-precision box pre-test
    // The persistent kernel walks only the meshes / patch sets that have nodes (same order, `index` = id); spheres
    // and planes are answered by a streaming pass and merged in when a ray retires.
    const DevObject *mesh_objects;
    uint32_t n_mesh_objects;
    uint32_t n_prims;            // spheres + planes in `objects`
    uint32_t wide_mesh;          // some mesh has 2^27 nodes or more: 32-bit byte offsets into its node arrays overflow
};

// ------------------------------------------------------------------- math --
struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 mk(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 ld3(const double *p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator/(V3 a, double s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
// Vector::dot (math/vec.rs:36-38): products summed left to right from 0.0.
__device__ __forceinline__ double dot(V3 a, V3 b) { return ((0.0 + a.x * b.x) + a.y * b.y) + a.z * b.z; }
// Vec3::cross (math/vec.rs:199-203)
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ double length(V3 a) { return __builtin_sqrt(dot(a, a)); }
__device__ __forceinline__ V3 normalize(V3 a) { return a / length(a); }   // math/vec.rs:43-45
__device__ __forceinline__ bool is_nan(double x) { return x != x; }
__device__ __forceinline__ bool is_finite(double x) { return __builtin_fabs(x) < kInf; }

struct Ray {
    V3 o, d;
};
__device__ __forceinline__ V3 ray_pos(const Ray &r, double t) { return r.o + r.d * t; }        // geo/ray.rs:21-23
__device__ __forceinline__ Ray new_bounce(V3 o, V3 d) { return Ray{o + d * 1e-5, d}; }         // geo/ray.rs:10,16-18

// Mat4 * (p, w) restricted to the first three rows (math/mat.rs:155-160,180-183):
// each row is a 4-term dot folded from 0.0; the w term is kept because x + r3*0.0
// is not always x.
__device__ __forceinline__ V3 mat4_apply(const double *m, V3 p, double w) {
    V3 r;
    r.x = (((0.0 + m[0] * p.x) + m[1] * p.y) + m[2] * p.z) + m[3] * w;
    r.y = (((0.0 + m[4] * p.x) + m[5] * p.y) + m[6] * p.z) + m[7] * w;
    r.z = (((0.0 + m[8] * p.x) + m[9] * p.y) + m[10] * p.z) + m[11] * w;
    return r;
}
__device__ __forceinline__ V3 mat3_apply(const double *m, V3 p) {
    V3 r;
    r.x = ((0.0 + m[0] * p.x) + m[1] * p.y) + m[2] * p.z;
    r.y = ((0.0 + m[3] * p.x) + m[4] * p.y) + m[5] * p.z;
    r.z = ((0.0 + m[6] * p.x) + m[7] * p.y) + m[8] * p.z;
    return r;
}

// ------------------------------------------------------------- primitives --
// Bounds::raycast (geo/bounds.rs:80-90) followed by the caller's
// `intersect(&holder.interval())` (tree/bvh.rs:121, render/object.rs:73-75).
// An axis whose two plane times are not both finite is skipped; intervals are
// open (strict <).  `bound` is the holder's current best time (+inf when
// empty); a NaN bound fails every comparison, as NaN-propagating minimum does.
__device__ __forceinline__ bool box_hit(V3 lo, V3 hi, const Ray &r, double bound) {
    double mn = -kInf, mx = kInf;
    bool ok = true;
    {
        double a = (lo.x - r.o.x) / r.d.x, b = (hi.x - r.o.x) / r.d.x;
        if (is_finite(a) && is_finite(b)) {
            mn = __builtin_fmax(mn, __builtin_fmin(a, b));
            mx = __builtin_fmin(mx, __builtin_fmax(a, b));
            ok = ok && (mn < mx);
        }
    }
    {
        double a = (lo.y - r.o.y) / r.d.y, b = (hi.y - r.o.y) / r.d.y;
        if (is_finite(a) && is_finite(b)) {
            mn = __builtin_fmax(mn, __builtin_fmin(a, b));
            mx = __builtin_fmin(mx, __builtin_fmax(a, b));
            ok = ok && (mn < mx);
        }
    }
    {
        double a = (lo.z - r.o.z) / r.d.z, b = (hi.z - r.o.z) / r.d.z;
        if (is_finite(a) && is_finite(b)) {
            mn = __builtin_fmax(mn, __builtin_fmin(a, b));
            mx = __builtin_fmin(mx, __builtin_fmax(a, b));
            ok = ok && (mn < mx);
        }
    }
    double lo0 = __builtin_fmax(mn, 0.0);
    return ok && (lo0 < mx) && (lo0 < bound);
}

// Triangle::raycast with manifold == false (geo/triangle.rs:54-82): returns the
// hit time only; the rest of the TrianglePoint is rebuilt for the winner.
__device__ __forceinline__ bool tri_hit_time(V3 v0, V3 v1, V3 v2, const Ray &r, double &t_out) {
    V3 e01 = v1 - v0, e12 = v2 - v1, e20 = v0 - v2;
    V3 nt = cross(e01, e12);
    double denom = dot(nt, nt);
    bool ok = !(dot(nt, r.d) > 0.0);
    double t = dot(v0 - r.o, nt) / dot(r.d, nt);
    ok = ok && !(t < 0.0);
    V3 q = ray_pos(r, t);
    double c0 = dot(cross(e01, q - v0), nt) / denom;
    double c1 = dot(cross(e12, q - v1), nt) / denom;
    double c2 = dot(cross(e20, q - v2), nt) / denom;
    ok = ok && !(c0 < 0.0) && !(c1 < 0.0) && !(c2 < 0.0);
    t_out = t;
    return ok;
}

// Full TrianglePoint for a leaf already known to be the winner.
struct TriPoint {
    double t;
    V3 bary, q, n;
};
__device__ __forceinline__ TriPoint tri_point(V3 v0, V3 v1, V3 v2, const Ray &r) {
    V3 e01 = v1 - v0, e12 = v2 - v1, e20 = v0 - v2;
    V3 nt = cross(e01, e12);
    double denom = dot(nt, nt);
    TriPoint p;
    p.t = dot(v0 - r.o, nt) / dot(r.d, nt);
    p.q = ray_pos(r, p.t);
    double c0 = dot(cross(e01, p.q - v0), nt) / denom;
    double c1 = dot(cross(e12, p.q - v1), nt) / denom;
    double c2 = dot(cross(e20, p.q - v2), nt) / denom;
    p.bary = mk(c1, c2, c0);   // barycenter[(i + 2) % 3] = c_i
    p.n = normalize(nt);
    return p;
}

// roots 0.0.7 find_roots_quadratic, consumed as in geo/sphere.rs:89-93: the
// smallest root >= 0.  Returns false when there is none.
__device__ __forceinline__ bool quadratic_first_nonneg(double a2, double a1, double a0, double &t_out) {
    double r0 = 0.0, r1 = 0.0;
    int n = 0;
    if (a2 == 0.0) {
        if (a1 == 0.0) {
            if (a0 == 0.0) {
                r0 = 0.0;
                n = 1;
            }
        } else {
            r0 = -a0 / a1;
            n = 1;
        }
    } else {
        double disc = a1 * a1 - 4.0 * a2 * a0;
        if (disc < 0.0) {
            n = 0;
        } else {
            double a2x2 = 2.0 * a2;
            if (disc == 0.0) {
                r0 = -a1 / a2x2;
                n = 1;
            } else {
                double sq = __builtin_sqrt(disc);
                double same, diff;
                if (a1 < 0.0) {
                    same = -a1 + sq;
                    diff = -a1 - sq;
                } else {
                    same = -a1 - sq;
                    diff = -a1 + sq;
                }
                double x1, x2;
                if (__builtin_fabs(same) > __builtin_fabs(a2x2)) {
                    double a0x2 = 2.0 * a0;
                    if (__builtin_fabs(diff) > __builtin_fabs(a2x2)) {
                        x1 = a0x2 / same;
                        x2 = a0x2 / diff;
                    } else {
                        x1 = a0x2 / same;
                        x2 = same / a2x2;
                    }
                } else {
                    x1 = diff / a2x2;
                    x2 = same / a2x2;
                }
                if (x1 < x2) {
                    r0 = x1;
                    r1 = x2;
                } else {
                    r0 = x2;
                    r1 = x1;
                }
                n = 2;
            }
        }
    }
    // filter(|x| x >= 0).min_by(real_cmp): first minimum wins; roots that pass
    // the filter are never NaN.
    bool have = false;
    double t = 0.0;
    if (n >= 1 && r0 >= 0.0) {
        have = true;
        t = r0;
    }
    if (n >= 2 && r1 >= 0.0) {
        if (!have || r1 < t) {
            have = true;
            t = r1;
        }
    }
    t_out = t;
    return have;
}

// Sphere::raycast (geo/sphere.rs:81-108): hit time only.
__device__ __forceinline__ bool sphere_hit_time(V3 c, double rad, const Ray &r, double &t_out) {
    V3 eo = r.o - c;
    double a = dot(r.d, r.d);
    double b = 2.0 * dot(eo, r.d);
    double cc = dot(eo, eo) - rad * rad;
    return quadratic_first_nonneg(a, b, cc, t_out);
}

// `x.round() as i64` (render/plane_object.rs:37): half away from zero,
// saturating, NaN -> 0; returned as the wrapped unsigned value.
__device__ __forceinline__ uint64_t round_to_i64_bits(double x) {
    double r = __builtin_round(x);
    if (r != r) return 0;
    if (r >= 9223372036854775808.0) return 0x7FFFFFFFFFFFFFFFull;
    if (r <= -9223372036854775808.0) return 0x8000000000000000ull;
    return (uint64_t)(long long)r;
}

// PlaneObject::raycast (render/plane_object.rs:23-49): hit time only.
__device__ __forceinline__ bool plane_hit_time(V3 p, V3 tan1, V3 tan2, const Ray &r, double &t_out) {
    V3 n = cross(tan1, tan2);
    double t = dot(p - r.o, n) / dot(r.d, n);
    t_out = t;
    return !(t < 0.0);
}

// -------------------------------------------------------- trispline patches --
// Patch evaluation (trispline.rs:161-181, glam::DVec3 arithmetic: products and
// sums strictly left to right) and this library's ray/patch intersection (see
// rtg_patch_set in rtg.h).
struct Patch {
    V3 m00, m01, m02, m03, m10, m11, m12, m20, m21, m30;
};
__device__ __forceinline__ Patch load_patch(const rtg_trispline *p) {
    const double *d = &p->m[0][0];
    Patch s;
    s.m00 = ld3(d);
    s.m01 = ld3(d + 3);
    s.m02 = ld3(d + 6);
    s.m03 = ld3(d + 9);
    s.m10 = ld3(d + 12);
    s.m11 = ld3(d + 15);
    s.m12 = ld3(d + 18);
    s.m20 = ld3(d + 21);
    s.m21 = ld3(d + 24);
    s.m30 = ld3(d + 27);
    return s;
}
__device__ __forceinline__ V3 patch_position(const Patch &s, double t, double u) {
    return s.m30 * (t * t * t) + s.m20 * (t * t) + s.m10 * t + s.m00 + s.m21 * (t * t * u) + s.m11 * (t * u) +
           s.m01 * u + s.m12 * (t * u * u) + s.m02 * (u * u) + s.m03 * (u * u * u);
}
__device__ __forceinline__ V3 patch_ddt(const Patch &s, double t, double u) {
    return s.m30 * (3.0 * t * t) + s.m20 * (2.0 * t) + s.m10 + s.m21 * (2.0 * t * u) + s.m11 * u + s.m12 * (u * u);
}
__device__ __forceinline__ V3 patch_ddu(const Patch &s, double t, double u) {
    return s.m21 * (t * t) + s.m11 * t + s.m01 + s.m12 * (t * 2.0 * u) + s.m02 * (2.0 * u) + s.m03 * (3.0 * u * u);
}
// glam normalize: v * (1 / length), dot without the leading 0.0
__device__ __forceinline__ V3 glam_normalize(V3 a) {
    double l2 = (a.x * a.x + a.y * a.y) + a.z * a.z;
    return a * (1.0 / __builtin_sqrt(l2));
}
struct PatchHit {
    double tau, t, u;
    V3 normal;
};
// Full Triangle::raycast (manifold == false) including the barycentrics.
__device__ __forceinline__ bool tri_hit_full(V3 v0, V3 v1, V3 v2, const Ray &r, double &t_out, V3 &bary) {
    V3 e01 = v1 - v0, e12 = v2 - v1, e20 = v0 - v2;
    V3 nt = cross(e01, e12);
    double denom = dot(nt, nt);
    bool ok = !(dot(nt, r.d) > 0.0);
    double t = dot(v0 - r.o, nt) / dot(r.d, nt);
    ok = ok && !(t < 0.0);
    V3 q = ray_pos(r, t);
    double c0 = dot(cross(e01, q - v0), nt) / denom;
    double c1 = dot(cross(e12, q - v1), nt) / denom;
    double c2 = dot(cross(e20, q - v2), nt) / denom;
    ok = ok && !(c0 < 0.0) && !(c1 < 0.0) && !(c2 < 0.0);
    t_out = t;
    bary = mk(c1, c2, c0);
    return ok;
}
template <bool WANT_NORMAL>
__device__ __forceinline__ bool patch_leaf_hit(const rtg_trispline *patches, const rtg_patch_leaf *info,
                                               uint32_t newton_steps, V3 v0, V3 v1, V3 v2, const Ray &r,
                                               PatchHit &out) {
    double tf;
    V3 bary;
    if (!tri_hit_full(v0, v1, v2, r, tf, bary)) return false;
    Patch sp = load_patch(patches + info->patch);
    double t = ((0.0 + bary.x * info->tu[0][0]) + bary.y * info->tu[1][0]) + bary.z * info->tu[2][0];
    double u = ((0.0 + bary.x * info->tu[0][1]) + bary.y * info->tu[1][1]) + bary.z * info->tu[2][1];
    double tau = tf;
    V3 c = neg(r.d);
    for (uint32_t it = 0; it < newton_steps; it++) {
        V3 a = patch_ddt(sp, t, u), b = patch_ddu(sp, t, u);
        V3 res = ray_pos(r, tau) - patch_position(sp, t, u);
        V3 bc = cross(b, c);
        double det = dot(a, bc);
        double dt = dot(res, bc) / det;
        double du = dot(a, cross(res, c)) / det;
        double dtau = dot(a, cross(b, res)) / det;
        t = t + dt;
        u = u + du;
        tau = tau + dtau;
    }
    V3 f = patch_position(sp, t, u) - ray_pos(r, tau);
    double tol2 = 1e-20 * (dot(sp.m10, sp.m10) + dot(sp.m01, sp.m01));
    if (!(t >= 0.0 && u >= 0.0 && t + u <= 1.0 && tau >= 0.0 && dot(f, f) <= tol2)) return false;
    out.tau = tau;
    out.t = t;
    out.u = u;
    if (WANT_NORMAL) out.normal = glam_normalize(cross(patch_ddt(sp, t, u), patch_ddu(sp, t, u)));
    return true;
}

// ---------------------------------------------------------------- holder --
// RaycastPointHolder (render/object.rs:29-31,60-79) reduced to what decides the
// result: whether a point is held, its time and its leaf.
struct Holder {
    bool has;
    double t;
    uint32_t leaf;
};
__device__ __forceinline__ void holder_add(Holder &h, bool some, double t, uint32_t leaf) {
    if (some) {
        if (h.has) {
            if (t < h.t) {
                h.t = t;
                h.leaf = leaf;
            }
        } else {
            h.has = true;
            h.t = t;
            h.leaf = leaf;
        }
    }
}
__device__ __forceinline__ double holder_bound(const Holder &h) { return h.has ? h.t : kInf; }

struct Counter {
    uint32_t nodes, tris;
};

__device__ __forceinline__ void load_node(const DevNode *p, V3 &lo, V3 &hi, uint4 &tail) {
    const double2 *q = reinterpret_cast<const double2 *>(p);
    double2 a = q[0], b = q[1], c = q[2];
    tail = reinterpret_cast<const uint4 *>(p)[3];
    lo = mk(a.x, a.y, b.x);
    hi = mk(b.y, c.x, c.y);
}
__device__ __forceinline__ void load_tri(const DevTri *p, V3 &v0, V3 &v1, V3 &v2) {
    const double2 *q = reinterpret_cast<const double2 *>(p);
    double2 a = q[0], b = q[1], c = q[2], d = q[3];
    double e = p->p[8];
    v0 = mk(a.x, a.y, b.x);
    v1 = mk(b.y, c.x, c.y);
    v2 = mk(d.x, d.y, e);
}

template <bool COUNT, bool PATCH>
__device__ __forceinline__ void test_leaves(const DevMesh &m, uint32_t begin, uint32_t count,
                                            const Ray &r, Holder &h, Counter &ctr) {
    for (uint32_t i = 0; i < count; i++) {
        V3 v0, v1, v2;
        load_tri(m.tris + begin + i, v0, v1, v2);
        double t;
        bool some;
        if (PATCH && m.patches) {
            PatchHit ph;
            some = patch_leaf_hit<false>(m.patches, m.leaf_info + begin + i, m.newton_steps, v0, v1, v2, r, ph);
            t = ph.tau;
        } else {
            some = tri_hit_time(v0, v1, v2, r, t);
        }
        holder_add(h, some, t, begin + i);
        if (COUNT) ctr.tris++;
    }
}

// -------------------------------------------------------------- traversal --
// Short-stack traversal of the reference-order node array, in the exact visit
// order of Bvh::raycast_rec (tree/bvh.rs:118-133): box test, own leaves in
// index order, children in index order.  One stack entry per tree level: the
// next sibling still to be visited at that level.  The four newest entries live
// in VGPRs, older ones in LDS (kLdsLevels per lane, [level][lane] so a wave's
// access is conflict-free), the rest in a global slab.
constexpr int kRegLevels = 4;
constexpr int kLdsLevels = 28;
constexpr int kBlock = 256;

struct ShortStack {
    uint32_t s0, s1, s2, s3;
    int n;
    uint32_t *lds;         // this lane's column: lds[level * kBlock]
    uint32_t *glob;        // this lane's column in the overflow slab
    uint32_t gstride;

    __device__ __forceinline__ void spill_store(int level, uint32_t v) {
        if (level < kLdsLevels)
            lds[level * kBlock] = v;
        else
            glob[(size_t)(level - kLdsLevels) * gstride] = v;
    }
    __device__ __forceinline__ uint32_t spill_load(int level) {
        if (level < kLdsLevels) return lds[level * kBlock];
        return glob[(size_t)(level - kLdsLevels) * gstride];
    }
    __device__ __forceinline__ void push(uint32_t v) {
        if (n >= kRegLevels) spill_store(n - kRegLevels, s3);
        s3 = s2;
        s2 = s1;
        s1 = s0;
        s0 = v;
        n++;
    }
    __device__ __forceinline__ void pop() {
        s0 = s1;
        s1 = s2;
        s2 = s3;
        n--;
        if (n >= kRegLevels) s3 = spill_load(n - kRegLevels);
    }
};

template <bool COUNT, bool PATCH>
__device__ __forceinline__ void traverse_stack(const DevMesh &m, const Ray &r, Holder &h, Counter &ctr,
                                               uint32_t *lds_col, uint32_t *glob_col, uint32_t gstride) {
    ShortStack st;
    st.s0 = st.s1 = st.s2 = st.s3 = 0;
    st.n = 0;
    st.lds = lds_col;
    st.glob = glob_col;
    st.gstride = gstride;
    st.push(m.root);
    while (st.n > 0) {
        uint32_t cur = st.s0;
        V3 lo, hi;
        uint4 tail;
        load_node(m.nodes + cur, lo, hi, tail);
        if (COUNT) ctr.nodes++;
        // this level's entry: advance to the next sibling or retire the level
        if (tail.y & 0x80000000u)
            st.pop();
        else
            st.s0 = cur + 1;
        if (box_hit(lo, hi, r, holder_bound(h))) {
            test_leaves<COUNT, PATCH>(m, tail.z, tail.w, r, h, ctr);
            if (tail.y & 0x7FFFFFFFu) st.push(tail.x);
        }
    }
}

// Threaded traversal: a node whose box test passes is followed by its pass link
// (its first child, or whatever follows a childless node); a node that fails
// jumps over its subtree through the skip link.  Same visit order as the
// recursion, no stack.
template <bool COUNT, bool PATCH>
__device__ __forceinline__ void traverse_threaded(const DevMesh &m, const Ray &r, Holder &h, Counter &ctr) {
    const uint32_t end = m.n_nodes;
    uint32_t cur = end ? m.troot : 0u;
    while (cur < end) {
        V3 lo, hi;
        uint4 tail;
        load_node(m.tnodes + cur, lo, hi, tail);
        if (COUNT) ctr.nodes++;
        if (box_hit(lo, hi, r, holder_bound(h))) {
            test_leaves<COUNT, PATCH>(m, tail.z, tail.w, r, h, ctr);
            cur = tail.y;
        } else {
            cur = tail.x;
        }
    }
}

// ------------------------------------------------------------ scene level --
struct SceneHit {
    bool have;
    double t;
    uint32_t object;
    uint32_t leaf;
};

struct TraceCtx {
    uint32_t *lds_col;
    uint32_t *glob_col;
    uint32_t gstride;
};

// TransformObject::raycast's ray (render/transform_object.rs:20,
// geo/transform.rs:69-77): position and tangent through `reverse`; the
// direction is not renormalised, so `time` is shared between the two spaces.
__device__ __forceinline__ Ray inner_ray(const DevObject &o, const Ray &outer) {
    return Ray{mat4_apply(o.reverse, outer.o, 1.0), mat4_apply(o.reverse, outer.d, 0.0)};
}

// SceneObject::raycast(ray, None) (render/scene_object.rs:19-36) reduced to the
// winner's identity: every object in order, `.min()` keeps the first minimum
// under real_cmp, where NaN sorts below every number (math/scalar.rs:46-60).
template <bool COUNT, bool PATCH>
__device__ __forceinline__ SceneHit scene_closest(const DevScene &s, const Ray &ray, const TraceCtx &tc,
                                                  Counter &ctr) {
    SceneHit best;
    best.have = false;
    best.t = kInf;
    best.object = RTG_NONE;
    best.leaf = RTG_NONE;
    for (uint32_t i = 0; i < s.n_objects; i++) {
        const DevObject &o = s.objects[i];
        Ray in = inner_ray(o, ray);
        bool some = false;
        double t = 0.0;
        uint32_t leaf = RTG_NONE;
        if (o.kind == RTG_SPHERE) {
            some = sphere_hit_time(ld3(o.sphere_center), o.sphere_radius, in, t);
        } else if (o.kind == RTG_PLANE) {
            some = plane_hit_time(ld3(o.plane_position), ld3(o.plane_tan1), ld3(o.plane_tan2), in, t);
        } else if (o.kind == RTG_MESH || o.kind == RTG_PATCHES) {
            if (o.mesh.n_nodes != 0) {
                Holder h;
                h.has = false;
                h.t = kInf;
                h.leaf = RTG_NONE;
                if (s.traversal == 1)
                    traverse_threaded<COUNT, PATCH>(o.mesh, in, h, ctr);
                else
                    traverse_stack<COUNT, PATCH>(o.mesh, in, h, ctr, tc.lds_col, tc.glob_col, tc.gstride);
                some = h.has;
                t = h.t;
                leaf = h.leaf;
            }
        }
        if (some) {
            bool better = !best.have || (!is_nan(best.t) && (is_nan(t) || best.t > t));
            if (better) {
                best.have = true;
                best.t = t;
                best.object = i;
                best.leaf = leaf;
            }
        }
    }
    return best;
}

// Rebuilds the winner's full RaycastPoint (render/object.rs:18-27) exactly as
// the reference would have produced it on the way up: primitive point in the
// object's space, then TransformObject's position / normal fix-up
// (render/transform_object.rs:22-27) and SceneObject's manifold push.
template <bool PATCH>
__device__ __forceinline__ void scene_point(const DevScene &s, const Ray &ray, const SceneHit &hit, rtg_point &out) {
    out.time = hit.have ? hit.t : kInf;
    out.manifold[0] = 0;
    out.manifold[1] = 0;
    out.manifold_len = 0;
    out.material_slot = 0;
    for (int k = 0; k < 3; k++) {
        out.position[k] = 0.0;
        out.inter_normal[k] = 0.0;
        out.geo_normal[k] = 0.0;
    }
    out.manifold_point[0] = 0.0;
    out.manifold_point[1] = 0.0;
    if (!hit.have) return;
    const DevObject &o = s.objects[hit.object];
    Ray in = inner_ray(o, ray);
    V3 ni, ng;
    double mp0, mp1;
    uint32_t slot = hit.object * 2;
    if (o.kind == RTG_SPHERE) {
        V3 c = ld3(o.sphere_center);
        V3 position = ray_pos(in, hit.t);
        V3 norm = normalize(position - c);
        ni = norm;
        ng = norm;
        mp0 = norm.x;
        mp1 = norm.y;
        out.manifold[0] = hit.object;
        out.manifold_len = 1;
    } else if (o.kind == RTG_PLANE) {
        V3 p = ld3(o.plane_position), t1 = ld3(o.plane_tan1), t2 = ld3(o.plane_tan2);
        V3 n = cross(t1, t2);
        V3 x = ray_pos(in, hit.t);
        double m1 = dot(x - p, t1);
        double m2 = dot(x - p, t2);
        uint64_t sum = round_to_i64_bits(m1 * 10.0) + round_to_i64_bits(m2 * 10.0);
        if ((sum & 1ull) == 0) slot += 1;   // i64 `% 2 == 0` is the low bit
        ni = n;
        ng = n;
        mp0 = m1;
        mp1 = m2;
        out.manifold[0] = hit.object;
        out.manifold_len = 1;
    } else {
        V3 v0, v1, v2;
        load_tri(o.mesh.tris + hit.leaf, v0, v1, v2);
        TriPoint tp;
        if (PATCH && o.mesh.patches) {
            PatchHit ph;
            ph.t = 0.0;
            ph.u = 0.0;
            ph.normal = mk(0.0, 0.0, 0.0);
            patch_leaf_hit<true>(o.mesh.patches, o.mesh.leaf_info + hit.leaf, o.mesh.newton_steps, v0, v1, v2, in, ph);
            tp.n = ph.normal;
            tp.bary = mk(ph.t, ph.u, 0.0);
        } else {
            tp = tri_point(v0, v1, v2, in);
        }
        ng = tp.n;
        if (o.mesh.normals && !(PATCH && o.mesh.patches)) {
            const double *nn = o.mesh.normals + (size_t)hit.leaf * 9;
            V3 n0 = ld3(nn), n1 = ld3(nn + 3), n2 = ld3(nn + 6);
            ni = normalize((n0 * tp.bary.x + n1 * tp.bary.y) + n2 * tp.bary.z);
        } else {
            ni = tp.n;
        }
        mp0 = tp.bary.x;
        mp1 = tp.bary.y;
        out.manifold[0] = hit.leaf;
        out.manifold[1] = hit.object;
        out.manifold_len = 2;
    }
    V3 pos = ray_pos(ray, hit.t);
    V3 fi = normalize(mat3_apply(o.forward_norm, ni));   // geo/transform.rs:78-80
    V3 fg = normalize(mat3_apply(o.forward_norm, ng));
    out.position[0] = pos.x;
    out.position[1] = pos.y;
    out.position[2] = pos.z;
    out.inter_normal[0] = fi.x;
    out.inter_normal[1] = fi.y;
    out.inter_normal[2] = fi.z;
    out.geo_normal[0] = fg.x;
    out.geo_normal[1] = fg.y;
    out.geo_normal[2] = fg.z;
    out.manifold_point[0] = mp0;
    out.manifold_point[1] = mp1;
    out.material_slot = slot;
}

__device__ __forceinline__ const DevMaterial &material_of_slot(const DevScene &s, uint32_t slot) {
    const DevObject &o = s.objects[slot >> 1];
    return (slot & 1u) ? o.material2 : o.material;
}

// ---------------------------------------------------------------- camera --
// render_pixel's NDC (render/renderer.rs:157-159) and View::get_ray
// (geo/view.rs:13-19) with tan(fov/2) supplied by the host.
__device__ __forceinline__ Ray primary_ray(const rtg_view &v, uint32_t x, uint32_t y, uint32_t w, uint32_t h) {
    double sx = ((double)x - ((double)w - 1.0) / 2.0) / (double)w;
    double sy = -((double)y - ((double)h - 1.0) / 2.0) / (double)h;
    double dx = sx * v.tan_half_fov[0];
    double dy = sy * v.tan_half_fov[1];
    V3 d = normalize(mk(dx, dy, -1.0));
    return Ray{mk(v.orig[0], v.orig[1], v.orig[2]), d};
}

// --------------------------------------------------------------- shading --
// One light of compute_direct_irrad (render/renderer.rs:184-205) up to its shadow ray: the displacement to the
// light, the Lambert cosine against the interpolated normal, and the bounce ray whose closest hit decides
// occlusion.  Returns false when the light is behind the surface (no query is made then).
struct ShadowQuery {
    Ray ray;
    double dis, dis2, d;
};
__device__ __forceinline__ bool shadow_query(const rtg_light &light, V3 position, V3 inter_normal, ShadowQuery &q) {
    V3 disp = ld3(light.position) - position;
    q.dis2 = dot(disp, disp);
    q.dis = __builtin_sqrt(q.dis2);
    V3 dir = disp / q.dis;
    q.d = dot(dir, inter_normal);
    if (q.d < 0.0) return false;
    q.ray = new_bounce(position, dir);
    return true;
}
// The light's contribution once the occluder (closest hit along the shadow ray) is known.
__device__ __forceinline__ V3 add_light(V3 lighting, const rtg_light &light, const ShadowQuery &q, bool occ_have,
                                        double occ_t) {
    if (occ_have && occ_t < q.dis) return lighting;
    return lighting + (ld3(light.color) * q.d) / (4.0 * kPi * q.dis2);
}

// compute_direct_irrad (render/renderer.rs:184-205): one closest-hit shadow
// query per light, Lambert term over 4*pi*dis^2.
template <bool COUNT, bool PATCH>
__device__ __forceinline__ V3 direct_irrad(const DevScene &s, const rtg_light *lights, uint32_t n_lights,
                                           V3 position, V3 inter_normal, const TraceCtx &tc, Counter &ctr) {
    V3 lighting = mk(1.0, 1.0, 1.0) * 0.0;
    for (uint32_t i = 0; i < n_lights; i++) {
        ShadowQuery q;
        if (!shadow_query(lights[i], position, inter_normal, q)) continue;
        SceneHit occ = scene_closest<COUNT, PATCH>(s, q.ray, tc, ctr);
        lighting = add_light(lighting, lights[i], q, occ.have, occ.t);
    }
    return lighting;
}

// Dielectric::new (render/dielectric.rs:13-52).
struct Dielectric {
    double reflectance;
    V3 reflect, refract;
    bool has_refract;
};
__device__ __forceinline__ Dielectric dielectric_new(V3 inc, V3 signed_norm, double n1, double n2) {
    inc = normalize(inc);
    double d = dot(inc, signed_norm);
    V3 norm;
    double cos_i, n_i, n_t;
    if (d < 0.0) {
        norm = signed_norm;
        cos_i = -d;
        n_i = n1;
        n_t = n2;
    } else {
        norm = neg(signed_norm);
        cos_i = d;
        n_i = n2;
        n_t = n1;
    }
    double eta = n_i / n_t;
    double s2i = 1.0 - cos_i * cos_i;
    // f64::maximum(x, 0.0): NaN propagates
    s2i = is_nan(s2i) ? s2i : (s2i > 0.0 ? s2i : 0.0);
    double s2t = eta * eta * s2i;
    Dielectric r;
    r.refract = mk(0.0, 0.0, 0.0);
    if (s2t >= 1.0) {
        r.has_refract = false;
        r.reflectance = 1.0;
    } else {
        double cos_t = __builtin_sqrt(1.0 - s2t);
        r.has_refract = true;
        r.refract = inc * eta + norm * (eta * cos_i - cos_t);
        double r_par = ((n_t * cos_i) - (n_i * cos_t)) / ((n_t * cos_i) + (n_i * cos_t));
        double r_perp = ((n_i * cos_i) - (n_t * cos_t)) / ((n_i * cos_i) + (n_t * cos_t));
        r.reflectance = (r_par * r_par + r_perp * r_perp) / 2.0;
    }
    r.reflect = inc - (norm * 2.0) * dot(inc, norm);
    return r;
}

// smpte2048_encode (geo/color.rs:5-17) of c/80 (render/image.rs:39), quantised
// to 8 bits: round(255 * clamp01(e)).
__device__ __forceinline__ uint8_t pq8(double c) {
    const double M1 = 1305.0 / 8192.0, M2 = 2523.0 / 32.0;
    const double C1 = 107.0 / 128.0, C2 = 2413.0 / 128.0, C3 = 2392.0 / 128.0;
    double x = c / 80.0;
    double p = pow(x, M1);
    double e = pow((C1 + C2 * p) / (1.0 + C3 * p), M2);
    if (!(e > 0.0)) e = 0.0;
    if (e > 1.0) e = 1.0;
    return (uint8_t)__builtin_floor(255.0 * e + 0.5);
}

// One entry of the k-d tree.  The tree is implicit in the array (tree/seq_tree.rs:36-49): the root of a range is
// its element at len / 2, the two subtrees are the ranges left and right of it.
struct alignas(16) KdNode {
    double pos[3];
    double lo[3], hi[3];      // bounds of the subtree (KdEntry::bounds)
    uint32_t value;           // index of the photon in the caller's order
    uint32_t axis;
};
static_assert(sizeof(KdNode) == 80, "k-d node must be 5 x 16 B");

} // namespace rtg
