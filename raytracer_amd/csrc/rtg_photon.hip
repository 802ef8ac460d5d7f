// rtg_photon.hip -- the photon map on the device: nearest photons from a k-d tree, and the caustic correction of
// `compute_indirect_irrad` (render/renderer.rs:206-265) in forward-mode dual numbers.
//
// Work is laid out as one lane per (surface point, photon sample) pair:
//   1. k_kd_nearest      `KdTree::nearest` (tree/kd_tree.rs:115-154): the recursion with an explicit stack
//   2. k_newton          newton_steps corrections of the photon's emission direction; every bounce evaluates the one
//                        surface the photon met (the manifold filter), so no tree is walked
//   3. per bounce level  k_final_rays -> closest hit of the whole scene (the persistent kernel) -> k_final_bounce:
//                        the corrected photon traced for real; the hit found in plain doubles is re-evaluated in dual
//                        numbers (a dual number's value part never depends on its derivative parts, so the values --
//                        and every decision taken on them -- are the ones of a plain trace)
//   4. k_indirect_sum    one photon per light, light / (4 pi |dP/da x dP/db|), summed in light order
// Citations are relative to the reference's src/ directory.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_scan.hpp>

#include "rtg_launch.hpp"

namespace rtg {
namespace {

// ---- Der<2> (math/scalar.rs:72-128, 222-342) ------------------------------------------------------------
struct D2 {
    double v, a, b;
};
__device__ __forceinline__ D2 dconst(double v) { return D2{v, 0.0, 0.0}; }
__device__ __forceinline__ D2 operator+(D2 x, D2 y) { return D2{x.v + y.v, x.a + y.a, x.b + y.b}; }
__device__ __forceinline__ D2 operator-(D2 x, D2 y) { return D2{x.v - y.v, x.a - y.a, x.b - y.b}; }
__device__ __forceinline__ D2 operator*(D2 x, D2 y) { return D2{x.v * y.v, x.v * y.a + x.a * y.v, x.v * y.b + x.b * y.v}; }
__device__ __forceinline__ D2 operator/(D2 x, D2 y) {
    return D2{x.v / y.v, (y.v * x.a - x.v * y.a) / (y.v * y.v), (y.v * x.b - x.v * y.b) / (y.v * y.v)};
}
__device__ __forceinline__ D2 operator-(D2 x) { return D2{-x.v, x.a * -1.0, x.b * -1.0}; }
// derived PartialOrd: lexicographic over (v, d[0], d[1]); a NaN that decides leaves the pair unordered
__device__ __forceinline__ int d_cmp(D2 x, D2 y) {
    if (x.v < y.v) return -1;
    if (x.v > y.v) return 1;
    if (!(x.v == y.v)) return 2;
    if (x.a < y.a) return -1;
    if (x.a > y.a) return 1;
    if (!(x.a == y.a)) return 2;
    if (x.b < y.b) return -1;
    if (x.b > y.b) return 1;
    if (!(x.b == y.b)) return 2;
    return 0;
}
__device__ __forceinline__ bool d_lt(D2 x, D2 y) { return d_cmp(x, y) == -1; }
__device__ __forceinline__ bool d_gt(D2 x, D2 y) { return d_cmp(x, y) == 1; }
__device__ __forceinline__ bool d_ge(D2 x, D2 y) {
    int c = d_cmp(x, y);
    return c == 0 || c == 1;
}
__device__ __forceinline__ bool d_is_zero(D2 x) { return x.v == 0.0 && x.a == 0.0 && x.b == 0.0; }
__device__ __forceinline__ D2 d_sqrt(D2 x) {
    double k = 1.0 / (2.0 * __builtin_sqrt(x.v));
    return D2{__builtin_sqrt(x.v), x.a * k, x.b * k};
}
__device__ __forceinline__ D2 d_abs(D2 x) { return x.v < 0.0 ? -x : x; }
__device__ __forceinline__ D2 d_maximum(D2 x, D2 y) {          // scalar.rs:312-322
    if (x.v != x.v) return x;
    if (y.v != y.v) return y;
    return x.v <= y.v ? y : x;
}
// real_cmp on the values (scalar.rs:333-335, 46-60): NaN below every number
__device__ __forceinline__ int real_cmp(double a, double b) {
    bool an = a != a, bn = b != b;
    if (an) return bn ? 0 : -1;
    if (bn) return 1;
    return a < b ? -1 : (a > b ? 1 : 0);
}

struct D3 {
    D2 x, y, z;
};
__device__ __forceinline__ D3 d3(D2 x, D2 y, D2 z) { return D3{x, y, z}; }
__device__ __forceinline__ D3 d3const(V3 a) { return D3{dconst(a.x), dconst(a.y), dconst(a.z)}; }
__device__ __forceinline__ V3 d3val(const D3 &a) { return mk(a.x.v, a.y.v, a.z.v); }
__device__ __forceinline__ D3 operator+(const D3 &a, const D3 &b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ D3 operator-(const D3 &a, const D3 &b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ D3 operator*(const D3 &a, D2 s) { return d3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ D3 operator/(const D3 &a, D2 s) { return d3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ D3 operator-(const D3 &a) { return d3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ D2 ddot(const D3 &a, const D3 &b) { return ((dconst(0.0) + a.x * b.x) + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ D2 dlength(const D3 &a) { return d_sqrt(ddot(a, a)); }
__device__ __forceinline__ D3 dnormalize(const D3 &a) { return a / dlength(a); }
__device__ __forceinline__ D3 dcross(const D3 &a, const D3 &b) {
    return d3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
struct RayD {
    D3 o, d;
};
__device__ __forceinline__ D3 dray_pos(const RayD &r, D2 t) { return r.o + r.d * t; }
__device__ __forceinline__ RayD dbounce(const D3 &o, const D3 &d) { return RayD{o + d * dconst(1e-5), d}; }

// Mat4 * (p, w), rows 0..2, the matrix cast to Der (geo/transform.rs:60-80, math/mat.rs:155-160,180-183)
__device__ __forceinline__ D3 dmat4_apply(const double *m, const D3 &p, double w) {
    const D2 v[4] = {p.x, p.y, p.z, dconst(w)};
    D2 out[3];
    for (int i = 0; i < 3; i++) {
        D2 acc = dconst(0.0);
        for (int j = 0; j < 4; j++) acc = acc + dconst(m[i * 4 + j]) * v[j];
        out[i] = acc;
    }
    return d3(out[0], out[1], out[2]);
}
__device__ __forceinline__ D3 dforward_norm(const double *m, const D3 &n) {
    const D2 v[3] = {n.x, n.y, n.z};
    D2 out[3];
    for (int i = 0; i < 3; i++) {
        D2 acc = dconst(0.0);
        for (int j = 0; j < 3; j++) acc = acc + dconst(m[i * 3 + j]) * v[j];
        out[i] = acc;
    }
    return dnormalize(d3(out[0], out[1], out[2]));
}

// ZenithY::into_normal (geo/sphere.rs:29-34) of ZenithY::as_input: zenith carries d/da = 1, y carries d/db = 1
__device__ __forceinline__ D3 zenith_y_into_normal(double zenith, double y) {
    D2 z{zenith, 1.0, 0.0}, yy{y, 0.0, 1.0};
    D2 radius = d_sqrt(dconst(1.0) - yy * yy);
    double sn = sin(zenith), cs = cos(zenith);
    D2 c{cs, z.a * -sn, z.b * -sn}, s{sn, z.a * cs, z.b * cs};
    return d3(c * radius, yy, s * radius);
}

// roots 0.0.7 find_roots_quadratic over Der, consumed as in geo/sphere.rs:89-93
__device__ __forceinline__ bool dquadratic_first_nonneg(D2 a2, D2 a1, D2 a0, D2 &t_out) {
    const D2 zero = dconst(0.0);
    D2 r0 = zero, r1 = zero;
    int n = 0;
    if (d_is_zero(a2)) {
        if (d_is_zero(a1)) {
            if (d_is_zero(a0)) {
                r0 = zero;
                n = 1;
            }
        } else {
            r0 = -a0 / a1;
            n = 1;
        }
    } else {
        D2 disc = a1 * a1 - dconst(4.0) * a2 * a0;
        if (!d_lt(disc, zero)) {
            D2 a2x2 = dconst(2.0) * a2;
            if (d_is_zero(disc)) {
                r0 = -a1 / a2x2;
                n = 1;
            } else {
                D2 sq = d_sqrt(disc);
                D2 same, diff;
                if (d_lt(a1, zero)) {
                    same = -a1 + sq;
                    diff = -a1 - sq;
                } else {
                    same = -a1 - sq;
                    diff = -a1 + sq;
                }
                D2 x1, x2;
                if (d_gt(d_abs(same), d_abs(a2x2))) {
                    D2 a0x2 = dconst(2.0) * a0;
                    if (d_gt(d_abs(diff), d_abs(a2x2))) {
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
                if (d_lt(x1, x2)) {
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
    bool have = false;
    D2 t = zero;
    if (n >= 1 && d_ge(r0, zero)) {
        have = true;
        t = r0;
    }
    if (n >= 2 && d_ge(r1, zero)) {
        if (!have || real_cmp(r1.v, t.v) < 0) {
            have = true;
            t = r1;
        }
    }
    t_out = t;
    return have;
}

// What the correction needs of a RaycastPoint<Der<2>> (render/object.rs:18-27)
struct PointD {
    D2 t;
    D3 position, geo_normal;
    D2 mp0, mp1;
    bool second_material;
};

// TransformObject<AnyObject>::raycast over Der (render/transform_object.rs:17-28, any_object.rs:15-23) for one
// object; `leaf` names the triangle of a mesh (Bvh::raycast with a manifold, tree/bvh.rs:108-111: that triangle
// alone, tested without the rejections, geo/triangle.rs:54-82 with manifold == true).
__device__ __forceinline__ bool object_eval(const DevObject &o, uint32_t leaf, uint64_t n_leaves, const RayD &outer, PointD &out) {
    RayD in{dmat4_apply(o.reverse, outer.o, 1.0), dmat4_apply(o.reverse, outer.d, 0.0)};
    const D2 zero = dconst(0.0);
    D3 gn;
    out.second_material = false;
    if (o.kind == RTG_SPHERE) {                                       // geo/sphere.rs:81-108
        D3 c = d3const(ld3(o.sphere_center));
        D3 eo = in.o - c;
        D2 a = ddot(in.d, in.d);
        D2 b = dconst(2.0) * ddot(eo, in.d);
        D2 cc = ddot(eo, eo) - dconst(o.sphere_radius * o.sphere_radius);
        D2 t;
        if (!dquadratic_first_nonneg(a, b, cc, t)) return false;
        D3 position = dray_pos(in, t);
        D3 norm = dnormalize(position - c);
        out.t = t;
        gn = norm;
        out.mp0 = norm.x;
        out.mp1 = norm.y;
    } else if (o.kind == RTG_PLANE) {                                 // render/plane_object.rs:23-49
        V3 t1 = ld3(o.plane_tan1), t2 = ld3(o.plane_tan2);
        D3 p = d3const(ld3(o.plane_position));
        D3 n = d3const(cross(t1, t2));
        D2 t = ddot(p - in.o, n) / ddot(in.d, n);
        if (d_lt(t, zero)) return false;
        D3 x = dray_pos(in, t);
        D2 m1 = ddot(x - p, d3const(t1));
        D2 m2 = ddot(x - p, d3const(t2));
        uint64_t sum = round_to_i64_bits(m1.v * 10.0) + round_to_i64_bits(m2.v * 10.0);
        out.second_material = (sum & 1ull) == 0;
        out.t = t;
        gn = n;
        out.mp0 = m1;
        out.mp1 = m2;
    } else if (o.kind == RTG_MESH) {
        if (o.mesh.patches || leaf == RTG_NONE || (uint64_t)leaf >= n_leaves) return false;
        V3 p0, p1, p2;
        load_tri(o.mesh.tris + leaf, p0, p1, p2);
        D3 v0 = d3const(p0), v1 = d3const(p1), v2 = d3const(p2);
        D3 nt = dcross(v1 - v0, v2 - v1);
        D2 denom = ddot(nt, nt);
        D2 t = ddot(v0 - in.o, nt) / ddot(in.d, nt);
        D3 q = dray_pos(in, t);
        D2 c0 = ddot(dcross(v1 - v0, q - v0), nt) / denom;
        D2 c1 = ddot(dcross(v2 - v1, q - v1), nt) / denom;
        out.t = t;
        gn = dnormalize(nt);
        out.mp0 = c1;                                                 // barycenter[(i + 2) % 3] = c_i
        (void)c0;
        D2 c2 = ddot(dcross(v0 - v2, q - v2), nt) / denom;
        out.mp1 = c2;
    } else {
        return false;
    }
    out.position = dray_pos(outer, out.t);
    out.geo_normal = dforward_norm(o.forward_norm, gn);
    return true;
}

// Dielectric::new over Der (render/dielectric.rs:13-52): the outgoing direction of the requested mode
__device__ __forceinline__ bool dielectric_dir(const D3 &inc_in, const D3 &signed_norm, double n1, double n2, bool refract, D3 &out) {
    const D2 zero = dconst(0.0), one = dconst(1.0);
    D3 inc = dnormalize(inc_in);
    D2 dt = ddot(inc, signed_norm);
    D3 norm;
    D2 cos_i, n_i, n_t;
    if (d_lt(dt, zero)) {
        norm = signed_norm;
        cos_i = -dt;
        n_i = dconst(n1);
        n_t = dconst(n2);
    } else {
        norm = -signed_norm;
        cos_i = dt;
        n_i = dconst(n2);
        n_t = dconst(n1);
    }
    if (!refract) {
        out = inc - norm * dconst(2.0) * ddot(inc, norm);
        return true;
    }
    D2 eta = n_i / n_t;
    D2 s2i = d_maximum(one - cos_i * cos_i, zero);
    D2 s2t = eta * eta * s2i;
    if (d_ge(s2t, one)) return false;
    D2 cos_t = d_sqrt(one - s2t);
    out = inc * eta + norm * (eta * cos_i - cos_t);
    return true;
}

__device__ __forceinline__ const DevMaterial &material_of(const DevObject &o, const PointD &p) {
    return p.second_material ? o.material2 : o.material;
}

// ---- nearest photons -----------------------------------------------------------------------------------
__device__ __forceinline__ bool rc_less(double a, double b) { return real_cmp(a, b) < 0; }

template <int K>
__global__ __launch_bounds__(128) void k_kd_nearest(const KdNode *__restrict__ nodes, uint32_t n, const double *__restrict__ queries3,
                                                    uint64_t n_queries, uint32_t k, uint32_t stack_cap,
                                                    uint32_t *__restrict__ out_index, double *__restrict__ out_distance) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_queries) return;
    const V3 query = ld3(queries3 + q * 3);
    // CappedHeap (util/capped_heap.rs) kept as a sorted array: the heap's top is its last element
    double hd[K];
    uint32_t hi[K];
#pragma unroll
    for (int j = 0; j < K; j++) {
        hd[j] = kInf;
        hi[j] = RTG_NONE;
    }
    uint32_t size = 0;
    auto heap_push = [&](double dist, uint32_t value) {
        if (size == k) {                       // push_pop (util/binary_heap.rs:81-93): above the top: dropped
            double top = kInf;
#pragma unroll
            for (int j = 0; j < K; j++)
                if ((uint32_t)j + 1 == size) top = hd[j];
            if (real_cmp(dist, top) > 0) return;
            size--;
        }
        uint32_t p = 0;
#pragma unroll
        for (int j = 0; j < K; j++)
            if ((uint32_t)j < size && !rc_less(dist, hd[j])) p++;
#pragma unroll
        for (int j = K - 1; j >= 1; j--) {
            if ((uint32_t)j == p) {
                hd[j] = dist;
                hi[j] = value;
            } else if ((uint32_t)j > p) {
                hd[j] = hd[j - 1];
                hi[j] = hi[j - 1];
            }
        }
        if (p == 0) {
            hd[0] = dist;
            hi[0] = value;
        }
        size++;
    };
    auto heap_top = [&]() {
        double top = -kInf;                    // peek().map_or(NEG_INFINITY, ..)
#pragma unroll
        for (int j = 0; j < K; j++)
            if ((uint32_t)j + 1 == size) top = hd[j];
        return top;
    };
    // nearest_rec (tree/kd_tree.rs:122-154) with its own stack: a frame is (begin, len) plus whether the first
    // subtree has been searched already.  A balanced tree over n entries is floor(log2 n) + 1 levels deep and a
    // level holds one suspended frame, the innermost one two: `stack_cap` frames per lane, kept in LDS (frame f of
    // lane t at f * blockDim + t: no bank conflicts).  As a per-lane array the stack is indexed dynamically and lives
    // in scratch memory, which for a full GPU of waves does not fit L2 and moved 4.6 GB per 2 M queries.
    extern __shared__ unsigned long long kd_stack[];
    const uint32_t lane_slot = threadIdx.x, stride = blockDim.x;
    const int kStack = (int)stack_cap;
    int sp = 0;
    kd_stack[lane_slot] = (unsigned long long)n << 32;
    sp = 1;
    while (sp > 0 && k != 0) {
        sp--;
        const unsigned long long frame = kd_stack[(uint32_t)sp * stride + lane_slot];
        const uint32_t begin = (uint32_t)frame;
        const uint32_t raw = (uint32_t)(frame >> 32);
        const bool resumed = (raw & 0x80000000u) != 0;
        const uint32_t len = raw & 0x7FFFFFFFu;
        if (len == 0) continue;
        const uint32_t c = begin + len / 2;
        const KdNode *node = nodes + c;
        const V3 pos = ld3(node->pos);
        const uint32_t axis = node->axis;
        const double qa = axis == 0 ? query.x : (axis == 1 ? query.y : query.z);
        const double pa = axis == 0 ? pos.x : (axis == 1 ? pos.y : pos.z);
        const bool left_first = qa < pa;
        const uint32_t l_begin = begin, l_len = len / 2, r_begin = c + 1, r_len = len - len / 2 - 1;
        if (!resumed) {
            if (size == k) {
                // Bounds::distance (geo/bounds.rs:41-55)
                const V3 lo = ld3(node->lo), hi3 = ld3(node->hi);
                double total = 0.0;
                {
                    double d = query.x < lo.x ? lo.x - query.x : (query.x > hi3.x ? query.x - hi3.x : 0.0);
                    total += d * d;
                }
                {
                    double d = query.y < lo.y ? lo.y - query.y : (query.y > hi3.y ? query.y - hi3.y : 0.0);
                    total += d * d;
                }
                {
                    double d = query.z < lo.z ? lo.z - query.z : (query.z > hi3.z ? query.z - hi3.z : 0.0);
                    total += d * d;
                }
                if (__builtin_sqrt(total) > heap_top()) continue;
            }
            if (sp + 2 > kStack) break;       // cannot happen for a balanced tree; never write past the stack
            kd_stack[(uint32_t)sp * stride + lane_slot] = ((unsigned long long)(len | 0x80000000u) << 32) | begin;
            sp++;
            kd_stack[(uint32_t)sp * stride + lane_slot] =
                ((unsigned long long)(left_first ? l_len : r_len) << 32) | (left_first ? l_begin : r_begin);
            sp++;
        } else {
            heap_push(length(query - pos), node->value);
            bool more = size < k;
            if (!more) more = __builtin_fabs(qa - pa) < heap_top();
            if (more) {
                kd_stack[(uint32_t)sp * stride + lane_slot] =
                    ((unsigned long long)(left_first ? r_len : l_len) << 32) | (left_first ? r_begin : l_begin);
                sp++;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < K; j++)
        if ((uint32_t)j < k) {
            out_index[q * k + j] = (uint32_t)j < size ? hi[j] : RTG_NONE;
            out_distance[q * k + j] = (uint32_t)j < size ? hd[j] : kInf;
        }
}

// ---- the photon pass: directions -> rays, paths -> photons --------------------------------------------------
// Ray::new(light.sphere.orig(), dir.into_normal()) (render/renderer.rs:123, geo/sphere.rs:29-34) with the cosine and
// sine of the zenith taken on the host (`staged` holds cos, sin, zenith, y per direction): the square root and the
// two products round on the device as they do on the host.
__global__ __launch_bounds__(256) void k_photon_rays(const double *__restrict__ staged, uint32_t n, const rtg_light *__restrict__ lights,
                                                     uint64_t first_ray, uint64_t count_per_light, double *__restrict__ rays6,
                                                     double *__restrict__ dirs2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double2 cs = reinterpret_cast<const double2 *>(staged)[(size_t)i * 2];
    const double2 zy = reinterpret_cast<const double2 *>(staged)[(size_t)i * 2 + 1];
    const rtg_light &light = lights[(first_ray + i) / count_per_light];
    const double radius = __builtin_sqrt(1.0 - zy.y * zy.y);
    double2 *w = reinterpret_cast<double2 *>(rays6 + (size_t)i * 6);
    w[0] = make_double2(light.position[0], light.position[1]);
    w[1] = make_double2(light.position[2], cs.x * radius);
    w[2] = make_double2(zy.y, cs.y * radius);
    reinterpret_cast<double2 *>(dirs2)[i] = zy;
}
// The same with the cosine and sine taken here (option "photon_trig" = device): `dirs2_in` holds zenith, y.
__global__ __launch_bounds__(256) void k_photon_rays_trig(const double *__restrict__ dirs2_in, uint32_t n,
                                                          const rtg_light *__restrict__ lights, uint64_t first_ray,
                                                          uint64_t count_per_light, double *__restrict__ rays6,
                                                          double *__restrict__ dirs2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double2 zy = reinterpret_cast<const double2 *>(dirs2_in)[i];
    const rtg_light &light = lights[(first_ray + i) / count_per_light];
    const double radius = __builtin_sqrt(1.0 - zy.y * zy.y);
    const double c = cos(zy.x), s = sin(zy.x);
    double2 *w = reinterpret_cast<double2 *>(rays6 + (size_t)i * 6);
    w[0] = make_double2(light.position[0], light.position[1]);
    w[1] = make_double2(light.position[2], c * radius);
    w[2] = make_double2(zy.y, s * radius);
    reinterpret_cast<double2 *>(dirs2)[i] = zy;
}
// The same for directions made here: Sphere::fibonacci_sphere (src/geo/sphere.rs:52-80) for ray `first_ray + i` of the
// pass, i.e. direction k = ray mod count_per_light of light ray / count_per_light, with the two draws per direction
// taken from a SplitMix64 stream seeded per light (seed + l * gamma; direction k uses draws 2k and 2k + 1, whose
// state is the seed advanced 2k times) -- the stream and the arithmetic of the host library's generator, so both
// make the same (zenith, y) bits: the remainder is the exact one (one fused multiply-add after the rounded
// quotient, corrected by at most one step), everything else a single rounded operation.
__device__ __forceinline__ uint64_t splitmix64_next(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void k_photon_rays_fibonacci(uint32_t n, const rtg_light *__restrict__ lights, uint64_t first_ray,
                                                               uint64_t count_per_light, uint64_t seed, double *__restrict__ rays6,
                                                               double *__restrict__ dirs2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t ray = first_ray + i, l = ray / count_per_light, k = ray - l * count_per_light;
    const double phi = kPi * (3.0 - __builtin_sqrt(5.0));
    const double d = 2.0 * kPi / phi;
    const double dydi = 2.0 / ((double)count_per_light + d);
    const double dydj = dydi * d;
    uint64_t s = seed + l * 0x9E3779B97F4A7C15ull + 2 * k * 0x9E3779B97F4A7C15ull;
    const double u = (double)(splitmix64_next(s) >> 11) * (1.0 / 9007199254740992.0);
    const double v = (double)(splitmix64_next(s) >> 11) * (1.0 / 9007199254740992.0);
    const double fi = (double)k + u;
    const double x = phi * fi, y2pi = 2.0 * kPi;
    double q = __builtin_floor(x / y2pi);
    double zenith = __builtin_fma(-q, y2pi, x);
    if (zenith < 0.0) {
        q -= 1.0;
        zenith = __builtin_fma(-q, y2pi, x);
    } else if (zenith >= y2pi) {
        q += 1.0;
        zenith = __builtin_fma(-q, y2pi, x);
    }
    const double y = -1.0 + dydi * fi + dydj * v;
    const rtg_light &light = lights[l];
    const double radius = __builtin_sqrt(1.0 - y * y);
    const double c = cos(zenith), sn = sin(zenith);
    double2 *w = reinterpret_cast<double2 *>(rays6 + (size_t)i * 6);
    w[0] = make_double2(light.position[0], light.position[1]);
    w[1] = make_double2(light.position[2], c * radius);
    w[2] = make_double2(y, sn * radius);
    reinterpret_cast<double2 *>(dirs2)[i] = make_double2(zenith, y);
}
// Only a ray whose first hit lies on a refracting material has a path tree, and only paths below the root make
// photons (renderer.rs:124-127): the pass traces every direction's first hit, keeps the rays whose hit object has such
// a material (the object's two materials are not told apart here: keeping too many only costs time) in their order,
// and builds path records for those alone.  kept_index[j] is the batch index of kept ray j.
__global__ __launch_bounds__(256) void k_photon_first_flags(DevScene s, const rtg_hit *__restrict__ hits, uint32_t n,
                                                            uint32_t *__restrict__ flags) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t object = hits[i].object;
    bool keep = false;
    if (object < s.n_objects) {
        const DevObject &o = s.objects[object];
        keep = o.material.has_dielectric != 0 || o.material2.has_dielectric != 0;
    }
    flags[i] = keep ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_photon_keep(const double *__restrict__ rays6, const uint32_t *__restrict__ flags,
                                                     const uint32_t *__restrict__ offsets, uint32_t n,
                                                     double *__restrict__ kept_rays6, uint32_t *__restrict__ kept_index) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    const uint32_t j = offsets[i];
    const double2 *src = reinterpret_cast<const double2 *>(rays6) + (size_t)i * 3;
    double2 *dst = reinterpret_cast<double2 *>(kept_rays6) + (size_t)j * 3;
    dst[0] = src[0];
    dst[1] = src[1];
    dst[2] = src[2];
    kept_index[j] = i;
}
__global__ __launch_bounds__(256) void k_photon_flags(const rtg_path *__restrict__ paths, uint32_t n, uint32_t *__restrict__ flags) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = paths[i].depth >= 1 ? 1u : 0u;      // `if path.modes.len() == 0 { return None }`
}
// render/renderer.rs:129-136; `first_ray` is the batch's first ray among all the rays of the pass, path ray r is ray
// ray_index[r] of the batch (`dirs2` holds the batch's directions)
__global__ __launch_bounds__(256) void k_photon_write(const rtg_path *__restrict__ paths, uint32_t n,
                                                      const uint32_t *__restrict__ offsets, const double *__restrict__ dirs2,
                                                      const uint32_t *__restrict__ ray_index,
                                                      const rtg_light *__restrict__ lights, uint64_t first_ray,
                                                      uint64_t count_per_light, uint64_t room, rtg_photon *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rtg_path &p = paths[i];
    if (p.depth < 1) return;
    const uint64_t slot = offsets[i];
    if (slot >= room) return;
    const uint32_t in_batch = ray_index[p.ray];
    const uint64_t ray = first_ray + in_batch;
    const uint32_t l = (uint32_t)(ray / count_per_light);
    rtg_photon ph;
    for (int k = 0; k < 3; k++) {
        ph.position[k] = p.point.position[k];
        ph.origin[k] = lights[l].position[k];
        ph.light[k] = lights[l].color[k] * p.attenuation;
        ph.surface_object[k] = RTG_NONE;
        ph.surface_leaf[k] = RTG_NONE;
    }
    ph.dir[0] = dirs2[(size_t)in_batch * 2];
    ph.dir[1] = dirs2[(size_t)in_batch * 2 + 1];
    ph.n_bounces = p.depth;
    ph.modes = p.modes;
    ph.light_index = l;
    ph._pad = 0;
    uint32_t at = p.parent;                                   // ancestors, deepest first
    for (uint32_t d = p.depth; d-- > 0 && at < n;) {
        const rtg_path &a = paths[at];
        const uint32_t len = a.point.manifold_len;
        const uint32_t object = len == 2 ? a.point.manifold[1] : a.point.manifold[0];
        const uint32_t leaf = len == 2 ? a.point.manifold[0] : RTG_NONE;
#pragma unroll
        for (uint32_t k = 0; k < 3; k++)      // constant indices keep the record in registers
            if (k == d) {
                ph.surface_object[k] = object;
                ph.surface_leaf[k] = leaf;
            }
        at = a.parent;
    }
    out[slot] = ph;
}

// ---- the correction ------------------------------------------------------------------------------------------
struct Pairs {
    const DevObject *objects;
    uint32_t n_objects;
    const uint64_t *object_leaves;     // triangles per object (0 for spheres and planes)
    const rtg_photon *photons;
    uint32_t n_photons;
    const uint32_t *nearest;           // n_points x samples photon indices, RTG_NONE where there are fewer photons
    const rtg_point *points;
    uint64_t n_points;
    uint32_t samples;
};

__device__ __forceinline__ bool surface_of_point(const rtg_point &p, uint32_t &object, uint32_t &leaf) {
    if (p.manifold_len == 1) {
        object = p.manifold[0];
        leaf = RTG_NONE;
        return true;
    }
    if (p.manifold_len == 2) {
        object = p.manifold[1];
        leaf = p.manifold[0];
        return true;
    }
    return false;
}

// The Newton iteration of compute_indirect_irrad (render/renderer.rs:211-229): dir2[e] = the corrected ZenithY.
// A pair without a photon, or whose point is a miss, gets NaN.
// 256 registers, no scratch: builds held to 168 or 128 registers (three or four waves per SIMD) spill 260 / 396 bytes
// per lane and run 24 % / 50 % longer
__global__ __launch_bounds__(128) void k_newton(Pairs pr, uint32_t newton_steps, double *__restrict__ dir2) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= pr.n_points * pr.samples) return;
    const uint64_t pi = e / pr.samples;
    const uint32_t which = pr.nearest[e];
    const rtg_point &pt = pr.points[pi];
    uint32_t p_object = 0, p_leaf = 0;
    if (which >= pr.n_photons || !surface_of_point(pt, p_object, p_leaf) || p_object >= pr.n_objects) {
        dir2[e * 2] = __builtin_nan("");
        dir2[e * 2 + 1] = __builtin_nan("");
        return;
    }
    const rtg_photon &ph = pr.photons[which];
    double z = ph.dir[0], y = ph.dir[1];
    const D3 origin = d3const(ld3(ph.origin));
    const uint32_t n_bounces = ph.n_bounces < 3 ? ph.n_bounces : 3;
    for (uint32_t step = 0; step < newton_steps; step++) {
        RayD ray{origin, zenith_y_into_normal(z, y)};
        bool ok = true;
        PointD hit;
        for (uint32_t b = 0; b <= n_bounces && ok; b++) {
            const bool last = b == n_bounces;
            const uint32_t object = last ? p_object : ph.surface_object[b];
            const uint32_t leaf = last ? p_leaf : ph.surface_leaf[b];
            if (object >= pr.n_objects) {
                ok = false;
                break;
            }
            const DevObject &o = pr.objects[object];
            ok = object_eval(o, leaf, pr.object_leaves[object], ray, hit);
            if (!ok || last) break;
            const DevMaterial &m = material_of(o, hit);
            if (!m.has_dielectric) {
                ok = false;
                break;
            }
            D3 out;
            ok = dielectric_dir(ray.d, hit.geo_normal, m.n1, m.n2, ((ph.modes >> b) & 1u) != 0, out);
            if (ok) ray = dbounce(hit.position, out);
        }
        if (!ok) break;                                       // `None => break`
        // sep = manifold_point - p.manifold_point; dir -= jacobian(sep)^-1 * value(sep)
        const D2 s0 = hit.mp0 - dconst(pt.manifold_point[0]), s1 = hit.mp1 - dconst(pt.manifold_point[1]);
        const double ja = s0.a, jb = s0.b, jc = s1.a, jd = s1.b;
        const double inv = 1.0 / (ja * jd - jb * jc);         // Mat2::inverse (math/mat.rs:67-70)
        const double i00 = jd * inv, i01 = (-jb) * inv, i10 = (-jc) * inv, i11 = ja * inv;
        const double st0 = (0.0 + i00 * s0.v) + i01 * s1.v;
        const double st1 = (0.0 + i10 * s0.v) + i11 * s1.v;
        z = z - st0;
        y = y - st1;
    }
    dir2[e * 2] = z;
    dir2[e * 2 + 1] = y;
}

// The corrected photon traced for real (render/renderer.rs:246-247): per pair 18 doubles of dual-number ray state
// (origin and direction, each component v, d/da, d/db; state[k * n_pairs + e]) and the plain ray the scene is asked
// about.  A pair that is over gets a ray that meets nothing (NaN origin, zero direction).
__device__ __forceinline__ void store_ray_state(double *state, uint64_t n_pairs, uint64_t e, const RayD &r) {
    const D2 c[6] = {r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z};
    for (int k = 0; k < 6; k++) {
        state[(uint64_t)(k * 3 + 0) * n_pairs + e] = c[k].v;
        state[(uint64_t)(k * 3 + 1) * n_pairs + e] = c[k].a;
        state[(uint64_t)(k * 3 + 2) * n_pairs + e] = c[k].b;
    }
}
__device__ __forceinline__ RayD load_ray_state(const double *state, uint64_t n_pairs, uint64_t e) {
    D2 c[6];
    for (int k = 0; k < 6; k++)
        c[k] = D2{state[(uint64_t)(k * 3 + 0) * n_pairs + e], state[(uint64_t)(k * 3 + 1) * n_pairs + e],
                  state[(uint64_t)(k * 3 + 2) * n_pairs + e]};
    return RayD{d3(c[0], c[1], c[2]), d3(c[3], c[4], c[5])};
}
__device__ __forceinline__ void store_plain_ray(double *rays6, uint64_t e, bool alive, const RayD &r) {
    double *w = rays6 + e * 6;
    const double nan = __builtin_nan("");
    w[0] = alive ? r.o.x.v : nan;
    w[1] = alive ? r.o.y.v : nan;
    w[2] = alive ? r.o.z.v : nan;
    w[3] = alive ? r.d.x.v : 0.0;
    w[4] = alive ? r.d.y.v : 0.0;
    w[5] = alive ? r.d.z.v : 0.0;
}
// status[e]: 0 = over without a landing, 1 = in flight, 2 = landed (landing[e * 9 ..] = position v, d/da, d/db)
__global__ __launch_bounds__(256) void k_final_start(Pairs pr, const double *__restrict__ dir2, double *__restrict__ state,
                                                     double *__restrict__ rays6, uint32_t *__restrict__ status) {
    const uint64_t n_pairs = pr.n_points * pr.samples;
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_pairs) return;
    const uint32_t which = pr.nearest[e];
    const double z = dir2[e * 2], y = dir2[e * 2 + 1];
    RayD ray{};
    bool alive = which < pr.n_photons && !is_nan(z) && !is_nan(y);
    if (alive) {
        ray = RayD{d3const(ld3(pr.photons[which].origin)), zenith_y_into_normal(z, y)};
        store_ray_state(state, n_pairs, e, ray);
    }
    store_plain_ray(rays6, e, alive, ray);
    status[e] = alive ? 1u : 0u;
}
// One level of raytrace_all_specular_rec (render/renderer.rs:296-378) with no manifold filter and the photon's own
// modes: `hits` are the closest hits of the level's plain rays.
__global__ __launch_bounds__(128) void k_final_bounce(Pairs pr, uint32_t level, const rtg_hit *__restrict__ hits,
                                                      double *__restrict__ state, double *__restrict__ rays6,
                                                      uint32_t *__restrict__ status, double *__restrict__ landing) {
    const uint64_t n_pairs = pr.n_points * pr.samples;
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_pairs) return;
    if (status[e] != 1u) return;
    const rtg_photon &ph = pr.photons[pr.nearest[e]];
    const uint32_t n_bounces = ph.n_bounces < 3 ? ph.n_bounces : 3;
    const rtg_hit h = hits[e];
    RayD ray = load_ray_state(state, n_pairs, e);
    bool alive = h.object < pr.n_objects;
    uint32_t result = 0;
    if (alive) {
        const DevObject &o = pr.objects[h.object];
        PointD hit;
        alive = object_eval(o, h.leaf, pr.object_leaves[h.object], ray, hit);
        if (alive && level == n_bounces) {                    // filter_modes is empty: this is the photon
            const D2 c[3] = {hit.position.x, hit.position.y, hit.position.z};
            for (int k = 0; k < 3; k++) {
                landing[e * 9 + k] = c[k].v;
                landing[e * 9 + 3 + k] = c[k].a;
                landing[e * 9 + 6 + k] = c[k].b;
            }
            result = 2;
            alive = false;
        } else if (alive) {
            const DevMaterial &m = material_of(o, hit);
            D3 out;
            alive = m.has_dielectric != 0 && dielectric_dir(ray.d, hit.geo_normal, m.n1, m.n2, ((ph.modes >> level) & 1u) != 0, out);
            if (alive) {
                ray = dbounce(hit.position, out);
                store_ray_state(state, n_pairs, e, ray);
                result = 1;
            }
        }
    }
    store_plain_ray(rays6, e, alive, ray);
    status[e] = result;
}

// render/renderer.rs:249-264: of the corrected photons that land within epsilon of the point the last one per light
// counts (HashMap::insert); the sum runs over lights in ascending index (the reference's HashMap has no order).
constexpr int kMaxSamples = 16;
__global__ __launch_bounds__(256) void k_indirect_sum(Pairs pr, const uint32_t *__restrict__ status,
                                                      const double *__restrict__ landing, double newton_epsilon,
                                                      double *__restrict__ rgb) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pr.n_points) return;
    const rtg_point &pt = pr.points[i];
    const V3 p_position = ld3(pt.position);
    uint32_t light_of[kMaxSamples];
    bool counts[kMaxSamples];
    for (uint32_t j = 0; j < pr.samples; j++) {
        const uint64_t e = i * pr.samples + j;
        counts[j] = false;
        light_of[j] = RTG_NONE;
        if (status[e] != 2u) continue;
        const V3 landed = ld3(landing + e * 9);
        if (!(length(landed - p_position) < newton_epsilon)) continue;
        counts[j] = true;
        light_of[j] = pr.photons[pr.nearest[e]].light_index;
    }
    for (uint32_t j = 0; j < pr.samples; j++)                 // a later photon of the same light replaces this one
        for (uint32_t l = j + 1; l < pr.samples; l++)
            if (counts[j] && counts[l] && light_of[l] == light_of[j]) counts[j] = false;
    V3 total = mk(0.0, 0.0, 0.0);
    uint32_t after = 0;
    bool first = true;
    for (;;) {                                                // ascending light index
        uint32_t best = RTG_NONE;
        for (uint32_t j = 0; j < pr.samples; j++)
            if (counts[j] && (first || light_of[j] > after) && (best == RTG_NONE || light_of[j] < light_of[best])) best = j;
        if (best == RTG_NONE) break;
        const uint64_t e = i * pr.samples + best;
        const V3 d0 = ld3(landing + e * 9 + 3), d1 = ld3(landing + e * 9 + 6);
        const double area = 4.0 * kPi * length(cross(d0, d1));
        total = total + ld3(pr.photons[pr.nearest[e]].light) / area;
        after = light_of[best];
        first = false;
    }
    rgb[i * 3] = total.x;
    rgb[i * 3 + 1] = total.y;
    rgb[i * 3 + 2] = total.z;
}

// ---- a frame with the photon term ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_camera_rays(rtg_view view, uint32_t width, uint32_t height, rtg_rect tile,
                                                     double *__restrict__ rays6) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)tile.w * tile.h) return;
    const uint32_t py = (uint32_t)(i / tile.w), px = (uint32_t)(i - (uint64_t)py * tile.w);
    const Ray r = primary_ray(view, tile.x0 + px, tile.y0 + py, width, height);
    double *w = rays6 + i * 6;
    w[0] = r.o.x;
    w[1] = r.o.y;
    w[2] = r.o.z;
    w[3] = r.d.x;
    w[4] = r.d.y;
    w[5] = r.d.z;
}
// raytrace_pixel's sum (render/renderer.rs:268-278): a pixel's paths are adjacent records, in the order the
// recursion emitted them, the first of them at depth 0; path_rgb[j] = irrad.map_mul(diffuse) of path j.
__global__ __launch_bounds__(256) void k_frame_sum_paths(const rtg_path *__restrict__ paths, uint32_t n_paths,
                                                         const double *__restrict__ path_rgb, double *__restrict__ out_rgb) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_paths || paths[i].depth != 0) return;
    const uint32_t ray = paths[i].ray;
    V3 total = mk(0.0, 0.0, 0.0);
    for (uint32_t j = i; j < n_paths && paths[j].ray == ray; j++) total = total + ld3(path_rgb + (size_t)j * 3) * paths[j].attenuation;
    out_rgb[(size_t)ray * 3] = total.x;
    out_rgb[(size_t)ray * 3 + 1] = total.y;
    out_rgb[(size_t)ray * 3 + 2] = total.z;
}
__global__ __launch_bounds__(256) void k_frame_pq8(const double *__restrict__ rgb, uint64_t n, uint8_t *__restrict__ rgba8) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uchar4 c;
    c.x = pq8(rgb[i * 3]);
    c.y = pq8(rgb[i * 3 + 1]);
    c.z = pq8(rgb[i * 3 + 2]);
    c.w = 255;
    reinterpret_cast<uchar4 *>(rgba8)[i] = c;
}

template <int K>
void launch_kd(hipStream_t stream, const void *nodes, uint32_t n, const double *q, uint64_t nq, uint32_t k, uint32_t *oi, double *od) {
    // frames a lane can have suspended: one per level of the tree and two for the innermost (n = 0: the root frame)
    uint32_t levels = 1;
    while (levels < 32 && (n >> levels) != 0) levels++;
    const uint32_t cap = levels + 2;
    hipLaunchKernelGGL(k_kd_nearest<K>, dim3((uint32_t)((nq + 127) / 128)), dim3(128), (size_t)cap * 128 * sizeof(unsigned long long),
                       stream, (const KdNode *)nodes, n, q, nq, k, cap, oi, od);
}

} // namespace

hipError_t launch_kd_nearest(hipStream_t stream, const void *nodes, uint32_t n_nodes, const double *queries3, uint64_t n_queries,
                             uint32_t k, uint32_t *out_index, double *out_distance) {
    if (n_queries == 0 || k == 0) return hipSuccess;
    if (k <= 1)
        launch_kd<1>(stream, nodes, n_nodes, queries3, n_queries, k, out_index, out_distance);
    else if (k <= 2)
        launch_kd<2>(stream, nodes, n_nodes, queries3, n_queries, k, out_index, out_distance);
    else if (k <= 4)
        launch_kd<4>(stream, nodes, n_nodes, queries3, n_queries, k, out_index, out_distance);
    else if (k <= 8)
        launch_kd<8>(stream, nodes, n_nodes, queries3, n_queries, k, out_index, out_distance);
    else
        launch_kd<16>(stream, nodes, n_nodes, queries3, n_queries, k, out_index, out_distance);
    return hipGetLastError();
}

size_t photon_scan_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (size_t)n, rocprim::plus<uint32_t>(),
                                  (hipStream_t) nullptr);
    return bytes;
}

hipError_t launch_photon_first_filter(hipStream_t stream, const DevScene &scene, const rtg_hit *hits, const double *rays6,
                                      uint32_t n, uint32_t *flags, uint32_t *offsets, void *temp, size_t temp_bytes,
                                      double *kept_rays6, uint32_t *kept_index) {
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_photon_first_flags, dim3(blocks), dim3(256), 0, stream, scene, hits, n, flags);
    hipError_t e = rocprim::exclusive_scan(temp, temp_bytes, flags, offsets, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_photon_keep, dim3(blocks), dim3(256), 0, stream, rays6, flags, offsets, n, kept_rays6, kept_index);
    return hipGetLastError();
}

hipError_t launch_photons_from_paths(hipStream_t stream, const rtg_path *paths, uint32_t n_paths, const double *dirs2,
                                     const uint32_t *ray_index, const rtg_light *lights, uint64_t first_ray,
                                     uint64_t count_per_light, uint32_t *flags, uint32_t *offsets, void *temp,
                                     size_t temp_bytes, uint64_t room, rtg_photon *out) {
    if (n_paths == 0) return hipSuccess;
    const uint32_t blocks = (n_paths + 255) / 256;
    hipLaunchKernelGGL(k_photon_flags, dim3(blocks), dim3(256), 0, stream, paths, n_paths, flags);
    hipError_t e = rocprim::exclusive_scan(temp, temp_bytes, flags, offsets, 0u, (size_t)n_paths, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_photon_write, dim3(blocks), dim3(256), 0, stream, paths, n_paths, offsets, dirs2, ray_index, lights,
                       first_ray, count_per_light, room, out);
    return hipGetLastError();
}

hipError_t launch_photon_rays(hipStream_t stream, const double *staged, uint32_t n, const rtg_light *lights, uint64_t first_ray,
                              uint64_t count_per_light, double *rays6, double *dirs2) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_photon_rays, dim3((n + 255) / 256), dim3(256), 0, stream, staged, n, lights, first_ray, count_per_light, rays6,
                       dirs2);
    return hipGetLastError();
}
hipError_t launch_photon_rays_trig(hipStream_t stream, const double *dirs2_in, uint32_t n, const rtg_light *lights,
                                   uint64_t first_ray, uint64_t count_per_light, double *rays6, double *dirs2) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_photon_rays_trig, dim3((n + 255) / 256), dim3(256), 0, stream, dirs2_in, n, lights, first_ray,
                       count_per_light, rays6, dirs2);
    return hipGetLastError();
}
hipError_t launch_photon_rays_fibonacci(hipStream_t stream, uint32_t n, const rtg_light *lights, uint64_t first_ray,
                                        uint64_t count_per_light, uint64_t seed, double *rays6, double *dirs2) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_photon_rays_fibonacci, dim3((n + 255) / 256), dim3(256), 0, stream, n, lights, first_ray, count_per_light,
                       seed, rays6, dirs2);
    return hipGetLastError();
}
hipError_t launch_camera_rays(hipStream_t stream, const rtg_view &view, uint32_t width, uint32_t height, rtg_rect tile, double *rays6) {
    const uint64_t n = (uint64_t)tile.w * tile.h;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_camera_rays, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, view, width, height, tile, rays6);
    return hipGetLastError();
}
hipError_t launch_frame_from_paths(hipStream_t stream, const rtg_path *paths, uint32_t n_paths, const double *path_rgb,
                                   uint64_t n_pixels, double *out_rgb, uint8_t *out_rgba8) {
    if (n_pixels == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(out_rgb, 0, (size_t)n_pixels * 3 * sizeof(double), stream);   // a pixel without paths is black
    if (e != hipSuccess) return e;
    if (n_paths)
        hipLaunchKernelGGL(k_frame_sum_paths, dim3((n_paths + 255) / 256), dim3(256), 0, stream, paths, n_paths, path_rgb, out_rgb);
    if (out_rgba8)
        hipLaunchKernelGGL(k_frame_pq8, dim3((uint32_t)((n_pixels + 255) / 256)), dim3(256), 0, stream, out_rgb, n_pixels, out_rgba8);
    return hipGetLastError();
}

static Pairs pairs_of(const PhotonPairs &p) {
    Pairs pr;
    pr.objects = p.objects;
    pr.n_objects = p.n_objects;
    pr.object_leaves = p.object_leaves;
    pr.photons = p.photons;
    pr.n_photons = p.n_photons;
    pr.nearest = p.nearest;
    pr.points = p.points;
    pr.n_points = p.n_points;
    pr.samples = p.samples;
    return pr;
}

hipError_t launch_photon_newton(hipStream_t stream, const PhotonPairs &p, uint32_t newton_steps, double *dir2) {
    const uint64_t n = p.n_points * p.samples;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_newton, dim3((uint32_t)((n + 127) / 128)), dim3(128), 0, stream, pairs_of(p), newton_steps, dir2);
    return hipGetLastError();
}
hipError_t launch_photon_final_start(hipStream_t stream, const PhotonPairs &p, const double *dir2, double *state, double *rays6,
                                     uint32_t *status) {
    const uint64_t n = p.n_points * p.samples;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_final_start, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, pairs_of(p), dir2, state, rays6, status);
    return hipGetLastError();
}
hipError_t launch_photon_final_bounce(hipStream_t stream, const PhotonPairs &p, uint32_t level, const rtg_hit *hits, double *state,
                                      double *rays6, uint32_t *status, double *landing) {
    const uint64_t n = p.n_points * p.samples;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_final_bounce, dim3((uint32_t)((n + 127) / 128)), dim3(128), 0, stream, pairs_of(p), level, hits, state, rays6,
                       status, landing);
    return hipGetLastError();
}
hipError_t launch_photon_sum(hipStream_t stream, const PhotonPairs &p, const uint32_t *status, const double *landing,
                             double newton_epsilon, double *rgb) {
    if (p.n_points == 0) return hipSuccess;
    hipLaunchKernelGGL(k_indirect_sum, dim3((uint32_t)((p.n_points + 255) / 256)), dim3(256), 0, stream, pairs_of(p), status, landing,
                       newton_epsilon, rgb);
    return hipGetLastError();
}

} // namespace rtg
