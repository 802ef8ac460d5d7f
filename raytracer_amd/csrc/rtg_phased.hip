// rtg_phased.hip -- persistent, phase-scheduled closest-hit kernel for gfx950.
//
// Why: in the one-ray-per-lane kernels a wave runs until its slowest ray is
// done (55 % of the lanes active on the 1M-triangle frame) and executes the
// triangle test in almost every loop iteration for two or three lanes.  Here a
// lane is a small state machine -- IDLE, OBJ (between objects of the scene),
// NODE (one box test of the threaded pre-order tree), TRI (one triangle of a
// leaf range) -- and the wave runs one phase at a time for the lanes that are
// in it:
//   * NODE steps run whenever any lane has a node to test;
//   * TRI steps wait until `tri_min` lanes have a triangle pending (or no lane
//     can do a node step), so the expensive triangle code runs for many lanes
//     at once; a lane with a pending triangle does not advance, so every ray
//     still sees exactly the reference's visit order and best-time pruning
//     (tree/bvh.rs:118-133);
//   * a lane whose ray has seen every object waits in OBJ until `obj_min` lanes
//     are there; one OBJ step then stores the finished hits, hands out new rays
//     (64-ray chunks from per-XCD work counters) and enters the next object
//     (blocks b and b+8 share an XCD and its L2, so each XCD first drains its
//     own contiguous eighth of the frame and only then helps the others).
//
// Rays are prepared a chunk at a time: the wave that takes a chunk generates
// its 64 rays with every lane busy, carries them into the first mesh's space
// and parks them in LDS; a lane that needs a ray copies one slot.
//
// The box test of a visit is decided in single precision on a 32-byte
// companion node whenever a rigorous error bound allows it (PreRay below); the
// few visits that are too close to call, and rays or meshes outside the
// magnitude window of the bound, run the reference's f64 test.  Builds without
// the pre-test replace the six divisions of the f64 test by an exact
// reciprocal form: the direction is constant per ray, so the quotient is
// rebuilt from r = 1/d (one IEEE division per ray and axis) with two FMA
// refinement steps,
//   q0 = x*r; e0 = fma(-q0,d,x); q1 = fma(e0,r,q0); e1 = fma(-q1,d,x); q = fma(e1,r,q1)
// q1 is a faithful quotient, the residual e1 is then exact, and with r the
// correctly rounded reciprocal the last step returns the correctly rounded
// x/d (Markstein's theorem), i.e. the bits an IEEE division returns, provided
// nothing over/underflows on the way.  Rays and meshes whose magnitudes could
// do that (see fast_div_ray_ok / the host's mesh check) take the plain
// division path; results are identical either way.
#include <hip/hip_ext.h>

#include "rtg_device.hpp"
#include "rtg_launch.hpp"

namespace rtg {

#ifndef RTG_ROUNDS
#define RTG_ROUNDS 32
#endif
#ifndef RTG_NODE_BURST
#define RTG_NODE_BURST 16
#endif
#ifndef RTG_PHASED_WAVES
#define RTG_PHASED_WAVES 4
#endif
enum : uint32_t { PH_IDLE = 0, PH_OBJ = 1, PH_NODE = 2, PH_TRI = 3, PH_DONE = 5, PH_SOLVE = 6 };
// PH_DONE: traversal of the current mesh finished, reported at the end of the iteration.  PH_SOLVE (patch scenes):
// the ray hits the flat facet of a patch leaf and waits for the Newton solve, which runs once enough lanes wait.

// Lane mask of a predicate; the intrinsic folds into the compare that produced the predicate.
__device__ __forceinline__ uint64_t wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// the per-lane boolean of a wave-uniform lane mask (free: the mask is used as the condition as it stands)
__device__ __forceinline__ bool lanes_of(uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }

// ---- ray sources ------------------------------------------------------------------
// Every source also names the time below which a hit ends the query early (stop_time): -inf for closest-hit
// queries; a shadow ray only has to know whether anything lies before the light.
// kFiveWaves: the source's lean kernel variant is built for five waves per SIMD.  Coherent rays (camera, shadow)
// gain from the extra latency hiding; an incoherent batch against a tree larger than the caches keeps the memory
// path busy with three waves already, and more rays in flight only make the caches miss more (measured: 64 M
// bench_mesh rays against 10 M triangles, 26.8 ms at three waves, 26.5 ms at four, 28.7 ms at five).
// kLanePairFetch: neighbouring lanes fetch their two node records together, half a record each per load (single-mesh
// builds with the pre-test).  With unrelated rays every lane stands at a node of its own and a load touches 64
// cache lines; the L1 looks lines up one after the other, and that rate -- not the bytes from memory -- is what bounds
// such a batch (64 M bench_mesh rays against 10 M triangles: 26.5 -> 24.2 ms per launch).  Coherent rays share lines
// anyway and would only pay for the exchange (-1 % on the 1080p frame).
struct PrimarySource {
    static constexpr bool kEarlyStop = false;
    static constexpr bool kFiveWaves = true;
    static constexpr bool kLanePairFetch = false;
    __device__ __forceinline__ double stop_time(uint64_t) const { return -kInf; }
    rtg_view view;
    uint32_t width, height;
    rtg_rect tile;
    uint32_t tiles_x;     // 8x8 tiles per row of the rectangle
    // item -> pixel: items are numbered tile by tile, 64 per tile, row-major inside the tile
    __device__ __forceinline__ bool locate(uint64_t item, uint32_t &px, uint32_t &py, uint64_t &out_index) const {
        uint32_t t = (uint32_t)(item >> 6), l = (uint32_t)item & 63u;
        uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
        px = tx * 8 + (l & 7u);
        py = ty * 8 + (l >> 3);
        if (px >= tile.w || py >= tile.h) return false;
        out_index = (uint64_t)py * tile.w + px;
        return true;
    }
    // output slot of an item that make() accepts; `chunk_first` is the output slot of the chunk's first item (the
    // tile's top left pixel), worked out once per 64-item chunk so that no lane divides by the tile count again
    __device__ __forceinline__ uint64_t chunk_origin(uint32_t chunk) const {
        uint32_t ty = chunk / tiles_x, tx = chunk - ty * tiles_x;
        return (uint64_t)(ty * 8u) * tile.w + tx * 8u;
    }
    __device__ __forceinline__ uint64_t slot_of(uint64_t item, uint64_t chunk_first) const {
        uint32_t l = (uint32_t)item & 63u;
        return chunk_first + (uint64_t)(l >> 3) * tile.w + (l & 7u);
    }
    // STRAIGHT: the ray is worked out for every item, inside the tile or not, and the answer says which -- an early
    // return here puts a join into the caller's chunk set-up across which the compiler kept the half-made ray in
    // registers next to the traversal state, and spilled (1080p frame: 44 bytes of scratch per lane and 2 % of the
    // time).  The patch kernels, whose registers are spent on the Newton solve, are 2 % faster with the branch.
    template <bool STRAIGHT = true>
    __device__ __forceinline__ bool make(uint64_t item, Ray &ray, uint64_t &out_index) const {
        uint32_t px, py;
        const bool inside = locate(item, px, py, out_index);
        if (!STRAIGHT && !inside) return false;
        ray = primary_ray(view, tile.x0 + px, tile.y0 + py, width, height);
        return inside;
    }
};
struct BatchSource {
    static constexpr bool kEarlyStop = false;
    static constexpr bool kFiveWaves = false;
    static constexpr bool kLanePairFetch = true;
    __device__ __forceinline__ double stop_time(uint64_t) const { return -kInf; }
    const double *rays6;
    uint64_t n;
    __device__ __forceinline__ uint64_t chunk_origin(uint32_t) const { return 0; }
    __device__ __forceinline__ uint64_t slot_of(uint64_t item, uint64_t) const { return item; }
    template <bool STRAIGHT = true>
    __device__ __forceinline__ bool make(uint64_t item, Ray &ray, uint64_t &out_index) const {
        const bool inside = item < n;                          // n > 0: the launcher sends no empty batch
        const double2 *q = reinterpret_cast<const double2 *>(rays6 + (inside ? item : n - 1) * 6);
        double2 a = q[0], b = q[1], c = q[2];
        ray = Ray{mk(a.x, a.y, b.x), mk(b.y, c.x, c.y)};
        out_index = item;
        if (!inside) return false;
        return true;
    }
};

// Shadow rays of compute_direct_irrad (render/renderer.rs:184-205) for a batch of surface points: item
// l * n_points + p is light l seen from point p; items without a query (no hit at p, light behind the surface)
// produce no ray and leave their output slot untouched.
struct ShadowSource {
    static constexpr bool kEarlyStop = true;
    static constexpr bool kFiveWaves = true;
    static constexpr bool kLanePairFetch = false;
    const rtg_point *points;
    const rtg_light *lights;
    uint64_t n_points;
    uint32_t n_lights;
    // compute_direct_irrad only asks whether the closest hit lies before the light (`occ.t < dis`,
    // render/renderer.rs:184-205).  Within one mesh the first accepted hit below `dis` settles that mesh: the
    // reference's own traversal meets the same hit in the same state, and the holder only ever replaces it by a
    // lower time.  Across objects the scene takes `.min()` under real_cmp (render/scene_object.rs:31-34), where a
    // NaN time -- a zero-area triangle met while its mesh's holder is empty -- orders below every number and makes
    // `occ.t < dis` false.  So a ray that ended its walk of one mesh early still walks the meshes after it, but
    // only up to their first accepted hit (the only one that can be NaN), and every result goes through the scene
    // minimum.
    __device__ __forceinline__ double stop_time(uint64_t item) const {
        uint64_t l = item / n_points, p = item - l * n_points;
        const rtg_point &pt = points[p];
        ShadowQuery q;
        (void)shadow_query(lights[l], ld3(pt.position), ld3(pt.inter_normal), q);
        return q.dis;
    }
    __device__ __forceinline__ uint64_t chunk_origin(uint32_t) const { return 0; }
    __device__ __forceinline__ uint64_t slot_of(uint64_t item, uint64_t) const { return item; }
    // Straight-line on purpose (as the other sources' make): the ray is worked out for every item and the reasons
    // not to trace it are collected in the result -- early returns here put a join into the caller's chunk set-up
    // across which the compiler kept the half-made ray in registers next to the traversal state, and spilled.
    template <bool STRAIGHT = true>
    __device__ __forceinline__ bool make(uint64_t item, Ray &ray, uint64_t &out_index) const {
        uint64_t l = item / n_points, p = item - l * n_points;
        const bool light_ok = l < n_lights;
        const rtg_point &pt = points[p];
        const V3 position = ld3(pt.position), normal = ld3(pt.inter_normal);
        const V3 disp = ld3(lights[light_ok ? l : 0].position) - position;       // shadow_query, without its early return
        const V3 dir = disp / __builtin_sqrt(dot(disp, disp));
        ray = new_bounce(position, dir);
        out_index = item;
        return light_ok && pt.manifold_len != 0 && !(dot(dir, normal) < 0.0);
    }
};

// Shadow rays of one level of the path tree and the traced rays of the next level in one launch: items below
// n_shadow are ShadowSource items, the rest are batch rays; the output slot is the item index, so the hit buffer is
// the occluder table followed by the batch hits.
struct ShadowAndBatchSource {
    static constexpr bool kEarlyStop = true;
    static constexpr bool kFiveWaves = false;
    static constexpr bool kLanePairFetch = false;
    ShadowSource shadow;
    BatchSource batch;
    uint64_t n_shadow;
    __device__ __forceinline__ double stop_time(uint64_t item) const {
        return item < n_shadow ? shadow.stop_time(item) : -kInf;
    }
    __device__ __forceinline__ uint64_t chunk_origin(uint32_t) const { return 0; }
    __device__ __forceinline__ uint64_t slot_of(uint64_t item, uint64_t) const { return item; }
    template <bool STRAIGHT = true>
    __device__ __forceinline__ bool make(uint64_t item, Ray &ray, uint64_t &out_index) const {
        bool ok = item < n_shadow ? shadow.make(item, ray, out_index) : batch.make(item - n_shadow, ray, out_index);
        out_index = item;
        return ok;
    }
};

// ---- exact division by a per-ray constant ------------------------------------------
__device__ __forceinline__ double div_by_const(double x, double d, double r) {
    double q0 = x * r;
    double e0 = __builtin_fma(-q0, d, x);
    double q1 = __builtin_fma(e0, r, q0);
    double e1 = __builtin_fma(-q1, d, x);
    return __builtin_fma(e1, r, q1);
}
// Magnitude windows inside which no intermediate of div_by_const can overflow,
// underflow or lose an exact residual (box coordinates are checked on the host
// with the same origin window).
__device__ __forceinline__ bool in_window(double v, double lo, double hi) {
    double a = __builtin_fabs(v);
    return v == 0.0 || (a >= lo && a <= hi);
}
__device__ __forceinline__ bool fast_div_ray_ok(const Ray &r) {
    const double dlo = 0x1p-100, dhi = 0x1p100, olo = 0x1p-600, ohi = 0x1p400;
    return in_window(r.d.x, dlo, dhi) && in_window(r.d.y, dlo, dhi) && in_window(r.d.z, dlo, dhi) &&
           in_window(r.o.x, olo, ohi) && in_window(r.o.y, olo, ohi) && in_window(r.o.z, olo, ohi);
}

// v_min_f64 / v_max_f64 as the hardware defines them (IEEE-754 minNum/maxNum: a quiet
// NaN operand is ignored).  Written as asm so the compiler does not add a
// canonicalising v_max_f64 x,x in front of every use.
__device__ __forceinline__ double hw_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double hw_max(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Box test for a lane inside the fast-division window.  There every quotient is
// finite unless the direction component is 0, and then both plane times of the
// axis come out as quiet NaN (x*inf is inf or NaN, the residual fma is NaN): the
// reference skips exactly those axes (geo/bounds.rs:85), and min/max ignore the
// NaNs, so no per-axis test is needed.  The per-axis `min < max` checks of
// Interval::intersect collapse into the final one because the lower bound only
// grows and the upper bound only shrinks: max(mn,0) < mx implies every earlier one.
__device__ __forceinline__ bool box_hit_fast(V3 lo, V3 hi, const Ray &r, V3 rcp, double bound) {
    double ax = div_by_const(lo.x - r.o.x, r.d.x, rcp.x), bx = div_by_const(hi.x - r.o.x, r.d.x, rcp.x);
    double ay = div_by_const(lo.y - r.o.y, r.d.y, rcp.y), by = div_by_const(hi.y - r.o.y, r.d.y, rcp.y);
    double az = div_by_const(lo.z - r.o.z, r.d.z, rcp.z), bz = div_by_const(hi.z - r.o.z, r.d.z, rcp.z);
    double mn = hw_max(hw_max(hw_min(ax, bx), hw_min(ay, by)), hw_max(hw_min(az, bz), 0.0));
    // the +inf keeps mx a number when all three axes are skipped (full interval)
    double mx = hw_min(hw_min(hw_max(ax, bx), hw_max(ay, by)), hw_min(hw_max(az, bz), kInf));
    return mn < mx && mn < bound;
}

// Triangle::raycast's accept/reject decision and hit time (geo/triangle.rs:54-82)
// without the three barycentric divisions: c_i = n_i / denom is only compared
// with 0 here, and for denom in [2^-400, 2^400] and n_i either 0 or not below
// 2^-600 in magnitude the quotient cannot underflow, so `c_i < 0` is `n_i < 0`
// (NaN and infinite n_i included).  Outside that window the wave falls back to
// the divisions.  The barycentrics themselves are rebuilt for the winner only.
// The steps are kept in this order by scheduling barriers and the edge vectors are rebuilt where they are used (the
// same subtractions, so the same bits): evaluated all at once the test holds some 75 registers, in sequence about 50,
// which is what lets the kernel run five waves per SIMD.
template <class LoadRay>
__device__ __forceinline__ bool tri_hit_time_signs(V3 v0, V3 v1, V3 v2, LoadRay load_ray, double &t_out) {
    V3 nt = cross(v1 - v0, v2 - v1);
    __builtin_amdgcn_sched_barrier(0);
    double nd, t;
    V3 q;
    {
        const Ray r = load_ray();
        nd = dot(nt, r.d);
        t = dot(v0 - r.o, nt) / nd;          // d.dot(nt) has the same products in the same order as nt.dot(d)
        q = ray_pos(r, t);
    }
    bool ok = !(nd > 0.0) && !(t < 0.0);
    __builtin_amdgcn_sched_barrier(0);
    double denom = dot(nt, nt);
    double n0 = dot(cross(v1 - v0, q - v0), nt);
    __builtin_amdgcn_sched_barrier(0);
    double n1 = dot(cross(v2 - v1, q - v1), nt);
    __builtin_amdgcn_sched_barrier(0);
    double n2 = dot(cross(v0 - v2, q - v2), nt);
    __builtin_amdgcn_sched_barrier(0);
    const double tiny = 0x1p-600;
    bool window = denom >= 0x1p-400 && denom <= 0x1p400 &&
                  (n0 == 0.0 || !(__builtin_fabs(n0) < tiny)) && (n1 == 0.0 || !(__builtin_fabs(n1) < tiny)) &&
                  (n2 == 0.0 || !(__builtin_fabs(n2) < tiny));
    if (wave_ballot(!window) != 0) {
        ok = ok && !(n0 / denom < 0.0) && !(n1 / denom < 0.0) && !(n2 / denom < 0.0);
    } else {
        ok = ok && !(n0 < 0.0) && !(n1 < 0.0) && !(n2 < 0.0);
    }
    t_out = t;
    return ok;
}

__device__ __forceinline__ float hw_minf(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float hw_maxf(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float hw_max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float hw_min3f(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float hw_max0f(float a) {      // max(a, 0), NaN -> 0
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a));
    return r;
}
typedef float vf2 __attribute__((ext_vector_type(2)));
// Single-precision pre-test window (ray side): direction components 0 or in
// [2^-40, 2^40], origin components 0 or in the same band; with the mesh's
// coordinates in that band too, no f32 product below can overflow or go
// subnormal, so every rounding is a relative 2^-24 error.
__device__ __forceinline__ bool pre_ray_ok(const Ray &r) {
    const double lo = 0x1p-40, hi = 0x1p40;
    return in_window(r.d.x, lo, hi) && in_window(r.d.y, lo, hi) && in_window(r.d.z, lo, hi) &&
           in_window(r.o.x, lo, hi) && in_window(r.o.y, lo, hi) && in_window(r.o.z, lo, hi);
}
typedef const uint8_t __attribute__((address_space(1))) *GBytePtr;

// Generic box test with IEEE divisions (any magnitudes).
__device__ __forceinline__ bool box_hit_div(V3 lo, V3 hi, const Ray &r, double bound) {
    double mn = -kInf, mx = kInf;
    bool ok = true;
#define RTG_AXIS(c)                                                   \
    {                                                                 \
        double a = (lo.c - r.o.c) / r.d.c, b = (hi.c - r.o.c) / r.d.c; \
        if (is_finite(a) && is_finite(b)) {                           \
            mn = __builtin_fmax(mn, __builtin_fmin(a, b));            \
            mx = __builtin_fmin(mx, __builtin_fmax(a, b));            \
            ok = ok && (mn < mx);                                     \
        }                                                             \
    }
    RTG_AXIS(x)
    RTG_AXIS(y)
    RTG_AXIS(z)
#undef RTG_AXIS
    double lo0 = __builtin_fmax(mn, 0.0);
    return ok && (lo0 < mx) && (lo0 < bound);
}

typedef const DevNode __attribute__((address_space(1))) *GNodePtr;
typedef const DevTri __attribute__((address_space(1))) *GTriPtr;

typedef double vd2 __attribute__((ext_vector_type(2)));
typedef uint32_t vu4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_node_g(GNodePtr p, V3 &lo, V3 &hi, uint4 &tail) {
    typedef const vd2 __attribute__((address_space(1))) *D2;
    typedef const vu4 __attribute__((address_space(1))) *U4;
    D2 q = (D2)p;
    vd2 a = q[0], b = q[1], c = q[2];
    vu4 t = ((U4)p)[3];
    tail.x = t.x;
    tail.y = t.y;
    tail.z = t.z;
    tail.w = t.w;
    lo = mk(a.x, a.y, b.x);
    hi = mk(b.y, c.x, c.y);
}
__device__ __forceinline__ void load_tri_g(GTriPtr p, V3 &v0, V3 &v1, V3 &v2) {
    typedef const vd2 __attribute__((address_space(1))) *D2;
    D2 q = (D2)p;
    vd2 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
    v0 = mk(a.x, a.y, b.x);
    v1 = mk(b.y, c.x, c.y);
    v2 = mk(d.x, d.y, e.x);
}

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Work hand-out: 8 ranges of 64-item chunks, one per XCD label; a wave drains
// its own range first, then the others in ring order.
constexpr int kCounterStride = 32;   // one 128-B line per counter
struct WorkQueue {
    uint32_t *counters;
    uint32_t total_chunks;
    uint32_t per_range;
    uint32_t range;       // range currently being drained
    uint32_t tried;       // ranges found empty so far
    __device__ __forceinline__ bool grab(uint32_t &chunk) {
        while (tried < 8) {
            uint32_t begin = range * per_range;
            uint32_t end = begin + per_range < total_chunks ? begin + per_range : total_chunks;
            uint32_t got = 0;
            if ((threadIdx.x & 63u) == 0 && begin < end) got = atomicAdd(counters + range * kCounterStride, 1u);
            got = __builtin_amdgcn_readfirstlane(got);
            if (begin < end && begin + got < end) {
                chunk = begin + got;
                return true;
            }
            range = (range + 1) & 7u;
            tried++;
        }
        return false;
    }
};

// The ray of the single-precision box pre-test.  q32 = fma(c32, r32, -(o32 * r32)) differs from the reference's f64
// plane time (c - o) / d by at most E = 2^-24 * (3|c| + 4|o|) * |r| (to first order: c, o and r are each rounded to f32
// once, the product o32 * r32 once, the fma once, and |q| <= (|c| + |o|) |r|).  Minimum and maximum are 1-Lipschitz, so
// the entry and exit times carry the same bound with |c| replaced by B, the mesh's largest |coordinate|, and the
// largest E of the three axes; the difference the pre-test looks at is therefore off by at most 2 E.  The holder's
// bound rounded to f32 is off by 2^-24 |t|: below E whenever t <= 3 (B + |o|) |r|, and a larger t leaves the
// comparison nowhere near a tie.  e2 = 2 E, inflated by 0.2 % for the roundings of this very computation and the
// second-order terms.  Axes with d = 0 get r32 = NaN, so their plane times are NaN and drop out of the min/max exactly
// as in the f64 test.
struct PreRay {
    float rx, ry, rz, nx, ny, nz, e2;
};
// The costly part (three f64 divisions) is done when a chunk's rays are prepared with every lane busy; the rest is a
// dozen f32 operations on the origin and runs where a lane picks its ray up, so only the reciprocals are parked.
__device__ __forceinline__ void pre_ray_reciprocals(const Ray &in, float &rx, float &ry, float &rz) {
    const float qnan = __builtin_nanf("");
    rx = in.d.x == 0.0 ? qnan : (float)(1.0 / in.d.x);
    ry = in.d.y == 0.0 ? qnan : (float)(1.0 / in.d.y);
    rz = in.d.z == 0.0 ? qnan : (float)(1.0 / in.d.z);
}
__device__ __forceinline__ PreRay pre_ray_from_reciprocals(const Ray &in, float rx, float ry, float rz, float coord_max) {
    float ox = (float)in.o.x, oy = (float)in.o.y, oz = (float)in.o.z;
    PreRay p;
    p.rx = rx;
    p.ry = ry;
    p.rz = rz;
    p.nx = -(ox * p.rx);
    p.ny = -(oy * p.ry);
    p.nz = -(oz * p.rz);
    const float b3 = 3.0f * coord_max;
    float ex = in.d.x == 0.0 ? 0.0f : (b3 + 4.0f * __builtin_fabsf(ox)) * __builtin_fabsf(p.rx);
    float ey = in.d.y == 0.0 ? 0.0f : (b3 + 4.0f * __builtin_fabsf(oy)) * __builtin_fabsf(p.ry);
    float ez = in.d.z == 0.0 ? 0.0f : (b3 + 4.0f * __builtin_fabsf(oz)) * __builtin_fabsf(p.rz);
    p.e2 = __builtin_fmaxf(ex, __builtin_fmaxf(ey, ez)) * (0x1p-23f * 1.002f);
    return p;
}
__device__ __forceinline__ PreRay make_pre_ray(const Ray &in, float coord_max) {
    float rx, ry, rz;
    pre_ray_reciprocals(in, rx, ry, rz);
    return pre_ray_from_reciprocals(in, rx, ry, rz, coord_max);
}

// SceneObject::raycast's `.min()` over the objects' hits (render/scene_object.rs:19-36, ordering
// render/object.rs:81-99 with real_cmp, math/scalar.rs:46-60): NaN orders below every number and equals NaN.
__device__ __forceinline__ int real_cmp(double a, double b) {
    if (is_nan(a)) return is_nan(b) ? 0 : -1;
    if (is_nan(b)) return 1;
    return a < b ? -1 : (a > b ? 1 : 0);
}
// A later object replaces the best hit only when its time orders strictly below: the first of equals stays.
__device__ __forceinline__ bool scene_better(bool best_have, double best_t, double t) {
    return !best_have || real_cmp(t, best_t) < 0;
}

// Closest hit among the scene's spheres and planes for every item of a source, with every lane busy.  The
// persistent kernel walks the meshes only and merges this in when a ray retires; in a scene without meshes this
// pass is the whole answer.  Items without a ray write nothing.
template <class Source>
__global__ __launch_bounds__(kBlock) void k_prims(DevScene s, Source src, uint64_t n_items, rtg_hit *out) {
    uint64_t item = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (item >= n_items) return;
    Ray ray;
    uint64_t out_index;
    if (!src.make(item, ray, out_index)) return;
    bool have = false;
    double best_t = kInf;
    uint32_t best = RTG_NONE;
    for (uint32_t j = 0; j < s.n_objects; j++) {
        const DevObject &o = s.objects[j];
        if (o.kind != RTG_SPHERE && o.kind != RTG_PLANE) continue;
        Ray in = inner_ray(o, ray);
        double t = 0.0;
        bool some = o.kind == RTG_SPHERE
                        ? sphere_hit_time(ld3(o.sphere_center), o.sphere_radius, in, t)
                        : plane_hit_time(ld3(o.plane_position), ld3(o.plane_tan1), ld3(o.plane_tan2), in, t);
        if (some && scene_better(have, best_t, t)) {
            have = true;
            best_t = t;
            best = j;
        }
    }
    uint64_t tb = (uint64_t)__double_as_longlong(have ? best_t : kInf);
    uint4 v;
    v.x = best;
    v.y = RTG_NONE;
    v.z = (uint32_t)tb;
    v.w = (uint32_t)(tb >> 32);
    *reinterpret_cast<uint4 *>(out + out_index) = v;
}

struct PhasedParams {
    uint32_t tri_min;     // run a TRI step once this many lanes wait for one
    uint32_t obj_min;     // same for the between-objects step
    uint32_t solve_min;   // same for the Newton solve of patch leaves
};

// STATS builds add per-wave step/lane tallies into work_counters[8*kCounterStride ...] (diagnostics only).
// The kernel walks the scene's meshes (DevScene::mesh_objects); `prim_best`, when given, holds every item's closest
// sphere / plane hit, merged in when the ray retires.  COUNT builds tally node visits and triangle tests per ray;
// KEEP builds (scenes with several meshes) hold the outer ray in LDS for the meshes after the first.  Point records are not built here: a streaming pass rebuilds them from the hit ids
// (k_points_primary / k_points_batch), with every lane busy.
#ifndef RTG_PHASED_BLOCK
#define RTG_PHASED_BLOCK 64
#endif
// Threads per workgroup of the persistent kernel.  Waves never synchronise with each other, so the size only sets
// the granularity at which the hardware hands finished waves' registers and LDS to the next launch.
constexpr int kPhasedBlock = RTG_PHASED_BLOCK;
static_assert(kPhasedBlock % 64 == 0 && kPhasedBlock >= 64 && kPhasedBlock <= 256, "one to four waves per workgroup");

// Waves per SIMD the kernel is compiled for.  The lean variants -- one mesh, triangles, no counters -- fit 96
// registers and 8 KB of LDS per wave, so five waves share a SIMD (20 per CU: all of its 160 KB of LDS); the others
// (several meshes, counting, statistics) keep more state and stay at four; patch scenes run at three (below).
#ifndef RTG_PHASED_WAVES_LEAN
#define RTG_PHASED_WAVES_LEAN 5
#endif
// Patch scenes keep the Newton solve's state as well: at four waves (128 registers) the kernel spills 180 bytes per
// lane and moves 4.5 GB of scratch traffic per 3840x2160 frame; at three (166 registers, 20 bytes) the frame is 8 %
// faster, at two slower again.
#ifndef RTG_PHASED_WAVES_PATCH
#define RTG_PHASED_WAVES_PATCH 3
#endif
constexpr int phased_waves(bool five_ok, bool count, bool patch, bool keep, bool stats) {
    return patch ? RTG_PHASED_WAVES_PATCH : ((!five_ok || count || keep || stats) ? RTG_PHASED_WAVES : RTG_PHASED_WAVES_LEAN);
}

// The ray source, the output arrays and the scene's object list are only looked at between objects (a new chunk, a
// refill, a finished ray).  Read through the kernel-argument segment at those places -- behind a barrier the optimiser
// cannot move loads across -- they do not sit in scalar registers for the whole kernel, which is at the scalar register
// limit: what they displaced there went to vector registers and on to scratch memory (1080p frame: 80 -> 36 bytes
// of scratch per lane, +3 %).  PhasedArgs mirrors k_trace_phased's parameter list.
template <class Source> struct PhasedArgs {
    DevScene s;
    Source src;
    uint64_t n_items;
    uint32_t *work_counters;
    PhasedParams prm;
    rtg_hit *out_hits;
    rtg_ray_counters *out_counters;
    const rtg_hit *prim_best;
};
template <class Source> __device__ __forceinline__ const PhasedArgs<Source> &args_in_kernarg() {
    typedef const char __attribute__((address_space(4))) *KPtr;
    KPtr base = (KPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(base));
    return *(const PhasedArgs<Source> *)(const char *)base;
}

template <class Source, bool COUNT, bool PATCH, bool PRE, bool KEEP, bool STATS = false>
__global__ __launch_bounds__(kPhasedBlock, phased_waves(Source::kFiveWaves, COUNT, PATCH, KEEP, STATS)) void k_trace_phased(DevScene s, Source src, uint64_t n_items,
                                                          uint32_t *work_counters, PhasedParams prm,
                                                          rtg_hit *out_hits, rtg_ray_counters *out_counters,
                                                          const rtg_hit *__restrict__ prim_best) {
    uint32_t st_iter = 0, st_node = 0, st_node_l = 0, st_tri = 0, st_tri_l = 0, st_obj = 0, st_obj_l = 0, st_refill = 0;
    uint32_t st_amb = 0, st_amb_l = 0, st_lines = 0;
    uint64_t tk_node = 0, tk_tri = 0, tk_obj = 0, tk_all = 0, tk0 = 0, tk1 = 0;
    if (STATS) tk_all = __builtin_amdgcn_s_memtime();
    WorkQueue wq;
    wq.counters = work_counters;
    wq.total_chunks = (uint32_t)((n_items + 63) >> 6);
    wq.per_range = (wq.total_chunks + 7) >> 3;
    wq.range = blockIdx.x & 7u;
    wq.tried = 0;
    bool exhausted = false;       // wave-uniform: every work range has been drained
    uint64_t chunk_base = 0;      // wave-uniform
    uint64_t chunk_first = 0;     // output slot of the chunk's first item (wave-uniform)
    uint32_t chunk_left = 0;      // wave-uniform

    // per-lane state
    uint32_t phase = PH_OBJ;      // every lane starts by asking for a ray
    bool has_ray = false;
    // Per-ray state that is only touched between objects lives in LDS ([field][lane], 8-byte fields, so a
    // wave's access is conflict-free): the outer ray, the output slot and the scene-level best hit.
    // One mesh (no KEEP): best time and {leaf : 32 | output slot : 32} -- the object is the mesh whenever a leaf is
    // held.  Several meshes: best time, {leaf : 32 | object : 32}, output slot, outer ray.
    extern __shared__ double cold[];      // 2 * kPhasedBlock doubles, or 9 * kPhasedBlock in KEEP builds
    double *cold_lane = cold + threadIdx.x;
    enum { C_BEST_T = 0, C_BEST_ID, C_OUT, C_OX, C_OY, C_OZ, C_DX, C_DY, C_DZ };
#define COLD(f) cold_lane[(f) * kPhasedBlock]
    // A wave that takes a new 64-item chunk prepares all 64 rays at once (ray generation, transform into the first
    // mesh's space, pre-test ray) with every lane busy and parks them here ([field][slot of the wave]); a lane that
    // later needs a ray only copies one slot.
    __shared__ double stage_d[6 * kPhasedBlock];
    __shared__ uint32_t stage_u[(Source::kEarlyStop ? 6 : 4) * kPhasedBlock];   // reciprocals, flags, (stop time)
#define STG_D(slot, f) stage_d[(f) * kPhasedBlock + (threadIdx.x & ~63u) + (slot)]
#define STG_U(slot, f) stage_u[(f) * kPhasedBlock + (threadIdx.x & ~63u) + (slot)]
    const bool need_outer = KEEP;
    // the first mesh's constants, read once per wave
    const uint32_t first_id = (uint32_t)s.mesh_objects[0].index;
    const GBytePtr first_slab = (GBytePtr)s.mesh_objects[0].mesh.slab + sizeof(MeshSlabHeader);
    const uint32_t first_end = s.mesh_objects[0].mesh.n_nodes;
    const uint32_t first_root = s.mesh_objects[0].mesh.troot;
    const bool first_fast_ok = s.mesh_objects[0].mesh.fast_div_ok != 0;
    const float first_coord_max = s.mesh_objects[0].mesh.coord_max;
    // The ray in the current object's space is only read by the triangle and patch steps and by the exact box test
    // (rare with the pre-test on), so it is parked in LDS ([field][lane]) and read where it is used: twelve registers
    // fewer are live across the node loop.
    // KEEP builds (four waves, 128 registers, the outer ray already in LDS) keep it in registers instead.
    __shared__ double in_lds[KEEP ? 1 : 6 * kPhasedBlock];
    Ray in_reg{};
#define IN_L(f) in_lds[(f) * kPhasedBlock + threadIdx.x]
    auto load_in = [&]() {
        if (KEEP) return in_reg;
        return Ray{mk(IN_L(0), IN_L(1), IN_L(2)), mk(IN_L(3), IN_L(4), IN_L(5))};
    };
    auto store_in = [&](const Ray &r) {
        if (KEEP) {
            in_reg = r;
            return;
        }
        IN_L(0) = r.o.x;
        IN_L(1) = r.o.y;
        IN_L(2) = r.o.z;
        IN_L(3) = r.d.x;
        IN_L(4) = r.d.y;
        IN_L(5) = r.d.z;
    };
    // Builds without the pre-test keep the reciprocal direction for the division-free f64 box test; builds with
    // it spend those registers on the f32 ray instead and use IEEE division wherever the pre-test cannot decide.
    V3 rcp = mk(0, 0, 0);
    bool fast = false;
    Holder h;                     // invariant: t == +inf while empty, so h.t is the box test's bound as it stands
    h.has = false;
    h.t = kInf;
    h.leaf = RTG_NONE;
    uint32_t obj_i = 0, cur = 0, end_v = 0;
    uint32_t leaf_cur = 0;        // next leaf to test (low 30 bits), with the kind bits of HotNode::next; bit 30: the leaf is one
                                  // of a range whose owner `cur` still names (the range's end moves the walk on)
    // With one mesh in the scene (no KEEP) the tree is the same for every lane and every ray: its base and size stay
    // wave-uniform (scalar registers); only scenes with several meshes carry them per lane.
#define SLAB (KEEP ? slab_v : first_slab)
#define END (KEEP ? end_v : first_end)
    GBytePtr slab_v = nullptr;      // the current mesh's hot-node array (slab + 64); everything else is an offset
    bool pre = false;             // this ray/mesh pair may use the single-precision pre-test
    uint64_t m_notpre = ~0ull, m_notfast = ~0ull;   // lanes whose ray may not use the pre-test / the division-free test
    // the full f64 node `index` of the current mesh; the index is laundered so that the address arithmetic stays in
    // the rare branches that need it instead of being hoisted into every node step
    auto full_node = [&](uint32_t index) {
        asm volatile("" : "+v"(index));
        return (GNodePtr)(SLAB + (uint64_t)END * sizeof(HotNode)) + index;
    };
    // the first mesh's hot nodes as a raw buffer (base, size in bytes, 32-bit data)
    const __amdgpu_buffer_rsrc_t hot_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)first_slab, 0, (int)(KEEP ? 0u : first_end * (uint32_t)sizeof(HotNode)), 0x00020000);
    double t_stop = -kInf;        // sources with kEarlyStop: a held hit below this time ends the walk of a mesh
    // single-precision ray for the pre-test: reciprocal direction, -(origin * reciprocal), error bound, and the
    // holder's bound rounded to f32
    float r32x = 0, r32y = 0, r32z = 0, np32x = 0, np32y = 0, np32z = 0, e2 = 0, b32 = 0;
    Counter ctr;
    ctr.nodes = 0;
    ctr.tris = 0;

    // a leaf has been dealt with: next leaf of the node's range (rare), else back to the nodes
    auto leaf_done = [&]() {
        bool more = false;
        if (leaf_cur & 0x40000000u) {    // rare: the owning node (cur) lists several leaves, or has children as well
            typedef const vu4 __attribute__((address_space(1))) *U4;
            vu4 tail = ((U4)((GNodePtr)(SLAB + (uint64_t)END * sizeof(HotNode)) + cur))[3];
            more = (leaf_cur & kHotIndexMask) + 1 < tail.z + tail.w;
            if (!more) cur = tail.y;     // the owner's pass link
        }
        leaf_cur++;
        phase = more ? PH_TRI : (cur >= END ? PH_DONE : PH_NODE);
    };

    for (;;) {
        if (STATS) st_iter++;
        uint64_t m_node = wave_ballot(phase == PH_NODE);
        uint64_t m_tri = wave_ballot(phase == PH_TRI);
        uint64_t m_obj = wave_ballot(phase == PH_OBJ);
        uint64_t m_solve = PATCH ? wave_ballot(phase == PH_SOLVE) : 0;
        if ((m_node | m_tri | m_obj | m_solve) == 0) break;   // every lane is done and no chunk is left

        // ---- between objects: finish the ray, fetch the next one, enter the next object -------
        // Thresholds scale with the lanes that still have a ray: once the work queue is dry a wave drains with few
        // lanes, and making those wait for a full-size batch would only serialise them.
        const uint32_t live = (uint32_t)__popcll(m_node | m_tri | m_solve);
        const uint32_t tri_thr = exhausted ? (live * prm.tri_min + 63u) >> 6 : prm.tri_min;
        const uint32_t solve_thr = exhausted ? (live * prm.solve_min + 63u) >> 6 : prm.solve_min;
        const uint32_t obj_thr = exhausted ? 1u : prm.obj_min;
        if (m_obj != 0 && ((uint32_t)__popcll(m_obj) >= obj_thr || (m_node | m_tri | m_solve) == 0)) {
            if (STATS) {
                st_obj++;
                st_obj_l += (uint32_t)__popcll(m_obj);
                tk0 = __builtin_amdgcn_s_memtime();
            }
            // (1) a ray that has seen every object reports its closest hit
            if (phase == PH_OBJ && has_ray && obj_i >= s.n_mesh_objects) {
                uint64_t best_id = (uint64_t)__double_as_longlong(COLD(C_BEST_ID));
                uint64_t out_index;
                uint32_t object, leaf = (uint32_t)(best_id >> 32);
                if (KEEP) {
                    out_index = (uint64_t)__double_as_longlong(COLD(C_OUT));
                    object = (uint32_t)best_id;
                } else {
                    out_index = (uint32_t)best_id;
                    object = leaf != RTG_NONE ? first_id : RTG_NONE;
                }
                double t = COLD(C_BEST_T);
                if (prim_best) {                  // the closest sphere / plane: lower (time, object index) wins
                    uint4 p = *reinterpret_cast<const uint4 *>(prim_best + out_index);
                    double tp = __longlong_as_double((long long)(((uint64_t)p.w << 32) | p.z));
                    if (p.x != RTG_NONE) {
                        int c = object == RTG_NONE ? -1 : real_cmp(tp, t);
                        if (c < 0 || (c == 0 && p.x < object)) {
                            object = p.x;
                            leaf = RTG_NONE;
                            t = tp;
                        }
                    }
                }
                if (out_hits) {
                    bool have = object != RTG_NONE;
                    uint64_t tb = __double_as_longlong(have ? t : kInf);
                    uint4 v;
                    v.x = object;
                    v.y = have ? leaf : RTG_NONE;
                    v.z = (uint32_t)tb;
                    v.w = (uint32_t)(tb >> 32);
                    *reinterpret_cast<uint4 *>(out_hits + out_index) = v;
                }
                if (COUNT) {
                    out_counters[out_index].nodes_visited = ctr.nodes;
                    out_counters[out_index].tris_tested = ctr.tris;
                }
                has_ray = false;
            }
            // (2) lanes without a ray take the next items of the wave's chunk
            uint64_t m_need = wave_ballot(phase == PH_OBJ && !has_ray);
            while (m_need != 0 && !exhausted) {
                if (STATS) st_refill++;
                if (chunk_left == 0) {
                    uint32_t chunk;
                    if (!wq.grab(chunk)) {
                        exhausted = true;
                        break;
                    }
                    chunk_base = (uint64_t)chunk << 6;
                    chunk_left = 64;
                    const Source &ksrc = args_in_kernarg<Source>().src;
                    chunk_first = ksrc.chunk_origin(chunk);
                    {                       // every lane prepares one ray of the chunk
                        const DevObject &o0 = s.mesh_objects[0];
                        Ray ray;
                        uint64_t out_index;
                        uint32_t flags = 0;
                        if (ksrc.template make<!PATCH>(chunk_base + (threadIdx.x & 63u), ray, out_index)) {
                            Ray in0 = inner_ray(o0, ray);
                            flags = 1u;
                            STG_D(threadIdx.x & 63u, 0) = in0.o.x;
                            STG_D(threadIdx.x & 63u, 1) = in0.o.y;
                            STG_D(threadIdx.x & 63u, 2) = in0.o.z;
                            STG_D(threadIdx.x & 63u, 3) = in0.d.x;
                            STG_D(threadIdx.x & 63u, 4) = in0.d.y;
                            STG_D(threadIdx.x & 63u, 5) = in0.d.z;
                            if (PRE) {
                                if (o0.mesh.f32_ok != 0 && pre_ray_ok(in0)) flags |= 2u;
                                float rx, ry, rz;
                                pre_ray_reciprocals(in0, rx, ry, rz);
                                STG_U(threadIdx.x & 63u, 0) = __float_as_uint(rx);
                                STG_U(threadIdx.x & 63u, 1) = __float_as_uint(ry);
                                STG_U(threadIdx.x & 63u, 2) = __float_as_uint(rz);
                            }
                        }
                        STG_U(threadIdx.x & 63u, 3) = flags;
                        if (Source::kEarlyStop && flags != 0) {
                            uint64_t tb = (uint64_t)__double_as_longlong(ksrc.stop_time(chunk_base + (threadIdx.x & 63u)));
                            STG_U(threadIdx.x & 63u, 4) = (uint32_t)tb;
                            STG_U(threadIdx.x & 63u, 5) = (uint32_t)(tb >> 32);
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                uint32_t n_need = (uint32_t)__popcll(m_need);
                uint32_t rank = lane_rank(m_need);
                uint32_t n_take = n_need < chunk_left ? n_need : chunk_left;
                if (phase == PH_OBJ && !has_ray && rank < n_take) {
                    uint32_t slot = (64u - chunk_left) + rank;
                    uint64_t item = chunk_base + slot;
                    {
                        uint32_t flags = STG_U(slot, 3);
                        if (flags & 1u) {
                            has_ray = true;
                            obj_i = 0;
                            if (Source::kEarlyStop) {
                                t_stop = __longlong_as_double((long long)(((uint64_t)STG_U(slot, 5) << 32) | STG_U(slot, 4)));
                            }
                            const Source &rsrc = args_in_kernarg<Source>().src;
                            uint64_t out_index = rsrc.slot_of(item, chunk_first);
                            if (need_outer) {
                                Ray ray;
                                (void)rsrc.make(item, ray, out_index);
                                COLD(C_OX) = ray.o.x;
                                COLD(C_OY) = ray.o.y;
                                COLD(C_OZ) = ray.o.z;
                                COLD(C_DX) = ray.d.x;
                                COLD(C_DY) = ray.d.y;
                                COLD(C_DZ) = ray.d.z;
                            }
                            COLD(C_BEST_T) = kInf;
                            if (KEEP) {
                                COLD(C_OUT) = __longlong_as_double((long long)out_index);
                                COLD(C_BEST_ID) = __longlong_as_double(-1ll);   // object = leaf = RTG_NONE
                            } else {                                            // leaf = RTG_NONE, output slot
                                COLD(C_BEST_ID) = __longlong_as_double((long long)(0xFFFFFFFF00000000ull | (uint32_t)out_index));
                            }
                            ctr.nodes = 0;
                            ctr.tris = 0;
                            // straight into the first object's tree
                            const Ray in{mk(STG_D(slot, 0), STG_D(slot, 1), STG_D(slot, 2)),
                                         mk(STG_D(slot, 3), STG_D(slot, 4), STG_D(slot, 5))};
                            store_in(in);
                            if (KEEP) {
                                slab_v = first_slab;
                                end_v = first_end;
                            }
                            cur = first_root;
                            h.has = false;
                            h.t = kInf;
                            h.leaf = RTG_NONE;
                            if (!PRE) {
                                fast = first_fast_ok && fast_div_ray_ok(in);
                                rcp = mk(1.0 / in.d.x, 1.0 / in.d.y, 1.0 / in.d.z);
                            }
                            if (PRE) {
                                pre = (flags & 2u) != 0;
                                PreRay pr = pre_ray_from_reciprocals(in, __uint_as_float(STG_U(slot, 0)),
                                                                     __uint_as_float(STG_U(slot, 1)),
                                                                     __uint_as_float(STG_U(slot, 2)), first_coord_max);
                                r32x = pr.rx;
                                r32y = pr.ry;
                                r32z = pr.rz;
                                np32x = pr.nx;
                                np32y = pr.ny;
                                np32z = pr.nz;
                                e2 = pr.e2;
                                b32 = __builtin_inff();          // == (float)h.t: an empty holder keeps t = +inf
                            }
                            phase = PH_NODE;
                        }
                    }
                }
                chunk_left -= n_take;
                m_need = wave_ballot(phase == PH_OBJ && !has_ray);
            }
            if (exhausted && phase == PH_OBJ && !has_ray) phase = PH_IDLE;   // nothing left for this lane
            // (3) a ray that is done with one mesh enters the next one
            if (KEEP && phase == PH_OBJ && has_ray && obj_i < s.n_mesh_objects) {
                const DevObject &o = s.mesh_objects[obj_i];
                const Ray in = inner_ray(o, Ray{mk(COLD(C_OX), COLD(C_OY), COLD(C_OZ)), mk(COLD(C_DX), COLD(C_DY), COLD(C_DZ))});
                store_in(in);
                slab_v = (GBytePtr)o.mesh.slab + sizeof(MeshSlabHeader);
                end_v = o.mesh.n_nodes;
                cur = o.mesh.troot;
                h.has = false;
                h.t = kInf;
                h.leaf = RTG_NONE;
                if (!PRE) {
                    fast = o.mesh.fast_div_ok != 0 && fast_div_ray_ok(in);
                    rcp = mk(1.0 / in.d.x, 1.0 / in.d.y, 1.0 / in.d.z);
                }
                if (PRE) {
                    pre = o.mesh.f32_ok != 0 && pre_ray_ok(in);
                    PreRay pr = make_pre_ray(in, o.mesh.coord_max);
                    r32x = pr.rx;
                    r32y = pr.ry;
                    r32z = pr.rz;
                    np32x = pr.nx;
                    np32y = pr.ny;
                    np32z = pr.nz;
                    e2 = pr.e2;
                    b32 = __builtin_inff();          // == (float)h.t: an empty holder keeps t = +inf
                }
                phase = PH_NODE;
            }
            m_node = wave_ballot(phase == PH_NODE);
            m_tri = wave_ballot(phase == PH_TRI);
            if (PRE) m_notpre = wave_ballot(!pre);
            if (!PRE) m_notfast = wave_ballot(!fast);
            if (STATS) tk_obj += __builtin_amdgcn_s_memtime() - tk0;
        }

        // ---- traversal rounds: box tests in bursts, then the triangle step that ended the burst; the wave stays in
        // this inner loop until enough lanes wait between objects (leaving it is what costs: the outer iteration
        // re-evaluates every phase mask and merges the per-lane flags)
#pragma unroll 1
        for (int round = 0; round < RTG_ROUNDS; round++) {
        // ---- one box test per lane, repeated until a triangle step is due ----------------------
        uint64_t m_done_burst = 0;       // during a burst the lanes' phases live in the masks; written back after it
        // A node step is a load, a wait and some thirty instructions that end in the next load; a triangle step is a
        // long stretch of arithmetic.  The wave is latency-bound, so its node steps go ahead of other waves'
        // triangle steps on the SIMD: the next load is out sooner (+2 % on the 1080p frame, nothing lost elsewhere).
        __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
        for (int rep = 0; rep < RTG_NODE_BURST && m_node != 0; rep++) {
            if (STATS) {
                st_node++;
                st_node_l += (uint32_t)__popcll(m_node);
                // different 128-byte lines the step's node load touches (hot node i sits at byte 64 + 32 i of the slab;
                // lanes without a node read the root's record)
                const uint32_t line = lanes_of(m_node) ? (cur + 2u) >> 2 : 0u;
                uint64_t rest = ~0ull;
                while (rest != 0) {
                    const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)line, __ffsll((unsigned long long)rest) - 1);
                    rest &= ~wave_ballot(line == v);
                    st_lines++;
                }
                tk0 = __builtin_amdgcn_s_memtime();
            }
            // Lane sets are kept as scalar masks and combined with scalar instructions: a vote over a boolean that is
            // not itself a comparison costs two vector instructions, a vote over a comparison none.
            const bool slow_any = PRE || (m_node & m_notfast) != 0;
            const bool pre_all = PRE && (m_node & m_notpre) == 0;
            // The step has no divergent branch on its common path: every lane executes it (a vector instruction costs
            // the same with 46 or 64 lanes enabled), lanes outside m_node read the array's first record -- one more cache
            // line per load -- and keep their state through the selects.  That keeps every mask wave-uniform.
            const bool act = lanes_of(m_node);
            bool go_exact = false;
            uint64_t m_go;
            uint32_t skip, next;       // HotNode::skip, HotNode::next
            if (pre_all) {
                // single-precision pre-test on the 32-byte companion node
                typedef const vu4 __attribute__((address_space(1))) *U4;
                vu4 h0, h1;
                if (KEEP) {
                    U4 hp = (U4)((act ? slab_v : first_slab) + (uint64_t)(act ? cur : 0u) * sizeof(HotNode));
                    h0 = hp[0];
                    h1 = hp[1];
                } else if (Source::kLanePairFetch) {
                    // Lanes 2i and 2i+1 fetch their two records together: both read lane 2i's node in the first load
                    // and lane 2i+1's in the second, the even lane the first 16 bytes and the odd lane the last, and
                    // swap halves afterwards.  A load then touches 32 cache lines instead of 64, which is what the L1
                    // charges for when every lane is at a node of its own.
                    const uint32_t mine = (act ? cur : 0u) * (uint32_t)sizeof(HotNode);
                    const uint32_t theirs = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, true);
                    const bool odd = (threadIdx.x & 1u) != 0;
                    const uint32_t half = odd ? 16u : 0u;
                    const vu4 a = __builtin_amdgcn_raw_buffer_load_b128(hot_rsrc, (odd ? theirs : mine) + half, 0, 0);
                    const vu4 b = __builtin_amdgcn_raw_buffer_load_b128(hot_rsrc, (odd ? mine : theirs) + half, 0, 0);
                    vu4 ax, bx;
                    for (int k = 0; k < 4; k++) {
                        ax[k] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a[k], 0xB1, 0xF, 0xF, true);
                        bx[k] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b[k], 0xB1, 0xF, 0xF, true);
                    }
                    h0 = odd ? bx : a;
                    h1 = odd ? b : ax;
                } else {
                    // one mesh: buffer loads, whose address is a scalar base plus the lane's 32-bit byte offset -- no
                    // 64-bit address arithmetic per lane (the launcher sends trees whose hot nodes span 4 GiB or more
                    // to the KEEP build)
                    const uint32_t off = (act ? cur : 0u) * (uint32_t)sizeof(HotNode);
                    h0 = __builtin_amdgcn_raw_buffer_load_b128(hot_rsrc, off, 0, 0);
                    h1 = __builtin_amdgcn_raw_buffer_load_b128(hot_rsrc, off + 16, 0, 0);
                }
                vf2 cx = {__uint_as_float(h0.x), __uint_as_float(h0.y)};   // lo.x, hi.x
                vf2 cy = {__uint_as_float(h0.z), __uint_as_float(h0.w)};
                vf2 cz = {__uint_as_float(h1.x), __uint_as_float(h1.y)};
                skip = h1.z;
                next = h1.w;
                vf2 qx = __builtin_elementwise_fma(cx, (vf2){r32x, r32x}, (vf2){np32x, np32x});
                vf2 qy = __builtin_elementwise_fma(cy, (vf2){r32y, r32y}, (vf2){np32y, np32y});
                vf2 qz = __builtin_elementwise_fma(cz, (vf2){r32z, r32z}, (vf2){np32z, np32z});
                float mn = hw_max0f(hw_max3f(hw_minf(qx.x, qx.y), hw_minf(qy.x, qy.y), hw_minf(qz.x, qz.y)));
                float mx = hw_min3f(hw_maxf(qx.x, qx.y), hw_maxf(qy.x, qy.y), hw_maxf(qz.x, qz.y));
                // The two margins g1 = mx - mn (the box's interval) and g2 = b32 - mn (the holder's bound) are only
                // ever compared with the same thresholds, and rounding is monotonic, so their minimum is
                // min(mx, b32) - mn: one subtraction.  The minimum is the NaN-propagating one: mx is NaN when every
                // axis is skipped, b32 when the held time is NaN, and both must end up undecided.
                const float g = __builtin_elementwise_minimum(mx, b32) - mn;
                const uint64_t m_yes = wave_ballot(g > e2) & m_node;
                const uint64_t m_no = wave_ballot(g < -e2);
                const uint64_t m_amb = m_node & ~(m_yes | m_no);      // NaNs land here: neither proven
                m_go = m_yes;
                if (m_amb != 0) {
                    if (STATS) {
                        st_amb++;
                        st_amb_l += (uint32_t)__popcll(m_amb);
                    }
                    if (lanes_of(m_amb)) {       // too close to call in single precision: the exact test decides
                        V3 lo, hi;
                        uint4 tail;
                        load_node_g(full_node(cur), lo, hi, tail);
                        go_exact = box_hit_div(lo, hi, load_in(), h.t);
                    }
                    m_go |= wave_ballot(go_exact);
                }
            } else {
                skip = 0;
                next = 0;
                if (act) {
                    V3 lo, hi;
                    uint4 tail;
                    load_node_g(full_node(cur), lo, hi, tail);
                    const Ray in = load_in();
                    go_exact = slow_any ? box_hit_div(lo, hi, in, h.t) : box_hit_fast(lo, hi, in, rcp, h.t);
                    skip = tail.x;
                    next = hot_next_word(tail.x, tail.y, tail.z, tail.w);
                }
                m_go = wave_ballot(go_exact);
            }
            if (COUNT) ctr.nodes += act ? 1u : 0u;
            // branch-free successor: a passed box goes on to the node's pass link (a leaf node's is its skip link), a
            // failed one to the skip link
            const uint64_t m_leafy = wave_ballot((int32_t)next < 0);
            const uint64_t m_to_leaf = m_go & m_leafy;
            const bool to_leaf = lanes_of(m_to_leaf);
            const uint32_t node_index = cur;
            const uint32_t pass = lanes_of(m_leafy) ? skip : next;
            cur = lanes_of(m_go) ? pass : (act ? skip : cur);
            leaf_cur = to_leaf ? next : leaf_cur;
            // rare: a node with several leaves, or with leaves and children
            const uint64_t m_range = m_to_leaf & wave_ballot(next >= kHotRange);
            if (m_range != 0) {
                if (lanes_of(m_range)) {                                     // the range lives in the full node
                    typedef const vu4 __attribute__((address_space(1))) *U4;
                    vu4 t = ((U4)full_node(node_index))[3];
                    leaf_cur = t.z | kHotRange;
                    cur = node_index;            // the walk stays at the owner until its last leaf is done
                }
            }
            const uint64_t m_end = wave_ballot(cur >= END) & m_node & ~m_to_leaf;   // traversal finished (resolved below)
            m_done_burst |= m_end;
            m_tri |= m_to_leaf;
            m_node &= ~(m_to_leaf | m_end);
            if (STATS) tk_node += __builtin_amdgcn_s_memtime() - tk0;
            // go round again until a triangle step is due (finished lanes wait for the end of the burst)
            if ((uint32_t)__popcll(m_tri) >= tri_thr) break;
        }
        __builtin_amdgcn_s_setprio(0);
        phase = lanes_of(m_tri) ? PH_TRI : (lanes_of(m_done_burst) ? PH_DONE : phase);

        // ---- one triangle per waiting lane ----------------------------------------------
        if (m_tri != 0 && ((uint32_t)__popcll(m_tri) >= tri_thr || m_node == 0)) {
            if (STATS) {
                st_tri++;
                st_tri_l += (uint32_t)__popcll(m_tri);
                tk0 = __builtin_amdgcn_s_memtime();
            }
            if (phase == PH_TRI) {
                V3 v0, v1, v2;
                const uint32_t leaf = leaf_cur & kHotIndexMask;
                load_tri_g((GTriPtr)(SLAB + (uint64_t)END * (sizeof(HotNode) + sizeof(DevNode))) + leaf, v0, v1, v2);
                double t;
                bool some;
                bool solve = false;
                if (PATCH && ((const MeshSlabHeader *)(const uint8_t *)(SLAB - sizeof(MeshSlabHeader)))->patches) {
                    // patch leaf: only rays that hit the flat facet go on to the Newton solve, in a step of its own
                    V3 bary;
                    solve = tri_hit_full(v0, v1, v2, load_in(), t, bary);
                    some = false;
                } else {
#ifdef RTG_EXPERIMENT_NOTRI       // occupancy experiment only: no triangle ever hits, the test's registers vanish
                    some = false;
                    t = v0.x;
#else
                    some = tri_hit_time_signs(v0, v1, v2, load_in, t);
#endif
                }
                if (COUNT) ctr.tris++;
                if (solve) {
                    phase = PH_SOLVE;
                } else {
                    holder_add(h, some, t, leaf);
                    if (PRE) b32 = (float)h.t;       // an empty holder keeps t = +inf
                    leaf_done();
                    if (Source::kEarlyStop && h.has && h.t < t_stop) {
                        t_stop = kInf;       // later meshes: any accepted hit with a finite time ends their walk
                        phase = PH_DONE;
                    }
                }
            }
            if (STATS) tk_tri += __builtin_amdgcn_s_memtime() - tk0;
        }

        // ---- patch scenes: Newton solve for the lanes whose ray hits its leaf's facet ---------
        if (PATCH) {
            m_solve = wave_ballot(phase == PH_SOLVE);
            if (m_solve != 0 && ((uint32_t)__popcll(m_solve) >= solve_thr || wave_ballot(phase == PH_NODE) == 0)) {
                if (phase == PH_SOLVE) {
                    V3 v0, v1, v2;
                    const uint32_t leaf = leaf_cur & kHotIndexMask;
                    load_tri_g((GTriPtr)(SLAB + (uint64_t)END * (sizeof(HotNode) + sizeof(DevNode))) + leaf, v0, v1, v2);
                    const MeshSlabHeader *hdr = (const MeshSlabHeader *)(const uint8_t *)(SLAB - sizeof(MeshSlabHeader));
                    PatchHit ph;
                    const Ray in = load_in();
                    bool some = patch_leaf_hit<false>(hdr->patches, hdr->leaf_info + leaf, hdr->newton_steps, v0, v1, v2, in, ph);
                    holder_add(h, some, ph.tau, leaf);
                    if (PRE) b32 = (float)h.t;
                    leaf_done();
                    if (Source::kEarlyStop && h.has && h.t < t_stop) {
                        t_stop = kInf;       // later meshes: any accepted hit with a finite time ends their walk
                        phase = PH_DONE;
                    }
                }
            }
        }

        m_node = wave_ballot(phase == PH_NODE);
        m_tri = wave_ballot(phase == PH_TRI);
        if (PATCH) m_solve = wave_ballot(phase == PH_SOLVE);
        if ((uint32_t)__popcll(wave_ballot(phase == PH_DONE || phase == PH_OBJ)) >= obj_thr ||
            (m_node | m_tri | m_solve) == 0)
            break;
        }   // traversal rounds

        // ---- a finished traversal reports to the scene-level minimum ------------------------
        if (phase == PH_DONE) {
            if (h.has) {
                if (KEEP) {
                    double best_t = COLD(C_BEST_T);
                    bool best_have = (uint32_t)__double_as_longlong(COLD(C_BEST_ID)) != RTG_NONE;
                    if (scene_better(best_have, best_t, h.t)) {
                        uint32_t id = (uint32_t)s.mesh_objects[obj_i].index;
                        COLD(C_BEST_T) = h.t;
                        COLD(C_BEST_ID) = __longlong_as_double((long long)(((uint64_t)h.leaf << 32) | id));
                    }
                } else {                  // the only mesh: its holder is the scene's best
                    uint64_t slot = (uint32_t)__double_as_longlong(COLD(C_BEST_ID));
                    COLD(C_BEST_T) = h.t;
                    COLD(C_BEST_ID) = __longlong_as_double((long long)(((uint64_t)h.leaf << 32) | slot));
                }
            }
            obj_i = obj_i + 1;
            phase = PH_OBJ;
        }
    }
    if (STATS && (threadIdx.x & 63u) == 0) {
        uint32_t *st = work_counters + 8 * kCounterStride;
        atomicAdd(st + 0, st_iter);
        atomicAdd(st + 1, st_node);
        atomicAdd(st + 2, st_node_l);
        atomicAdd(st + 3, st_tri);
        atomicAdd(st + 4, st_tri_l);
        atomicAdd(st + 5, st_obj);
        atomicAdd(st + 6, st_obj_l);
        atomicAdd(st + 7, st_refill);
        atomicAdd(st + 8, 1u);
        atomicAdd(st + 9, st_amb);
        atomicAdd(st + 10, st_amb_l);
        tk1 = __builtin_amdgcn_s_memtime() - tk_all;
        atomicAdd(st + 11, (uint32_t)(tk_node >> 6));   // 64-cycle units
        atomicAdd(st + 12, (uint32_t)(tk_tri >> 6));
        atomicAdd(st + 13, (uint32_t)(tk_obj >> 6));
        atomicAdd(st + 14, (uint32_t)(tk1 >> 6));
        atomicAdd(st + 15, st_lines);
    }
}

#undef SLAB
#undef END
#undef IN_L
#undef COLD
#undef STG_D
#undef STG_U

// ---- self-test of div_by_const against the hardware division ----------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t &st) {
    uint64_t z = (st += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// family 0: random significands, exponents across the whole fast window
// family 1: small-integer numerators and denominators (exact and repeating quotients)
// family 2: x = RN(q*d) +- a few ulps (quotients next to representable numbers and to midpoints)
// family 3: denominators with all-ones / single-bit significands
__global__ void k_selftest_div(uint64_t seed, uint32_t per_thread, uint32_t family, uint64_t *out) {
    uint64_t st = seed ^ ((uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0xD1B54A32D192ED03ull);
    uint64_t bad = 0;
    for (uint32_t i = 0; i < per_thread; i++) {
        uint64_t a = mix64(st), b = mix64(st), c = mix64(st);
        double x, d;
        if (family == 0) {
            int ex = (int)(a % 1001) - 600, ed = (int)(b % 201) - 100;           // 2^-600..2^400, 2^-100..2^100
            x = __longlong_as_double((int64_t)((c & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(ex + 1023) << 52)));
            d = __longlong_as_double((int64_t)((mix64(st) & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(ed + 1023) << 52)));
        } else if (family == 1) {
            x = (double)(int64_t)(a % 200001) - 100000.0;
            d = (double)(int64_t)(b % 4001) - 2000.0;
            if (d == 0.0) d = 3.0;
        } else if (family == 2) {
            double q = __longlong_as_double((int64_t)((a & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1023 + (int)(a >> 60) - 8) << 52)));
            d = __longlong_as_double((int64_t)((b & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1023 + (int)(b >> 60) - 8) << 52)));
            x = q * d;
            int64_t bits = __double_as_longlong(x) + (int64_t)(c % 7) - 3;
            x = __longlong_as_double(bits);
        } else {
            uint64_t mant = (b & 1) ? 0x000FFFFFFFFFFFFFull : (1ull << (b % 52));
            if (b & 2) mant ^= (mix64(st) & 0xFF);
            int ed = (int)((b >> 8) % 201) - 100;
            d = __longlong_as_double((int64_t)((b & 0x8000000000000000ull) | mant | ((uint64_t)(ed + 1023) << 52)));
            int ex = (int)(a % 1001) - 600;
            x = __longlong_as_double((int64_t)((c & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(ex + 1023) << 52)));
        }
        double r = 1.0 / d;
        double fast = div_by_const(x, d, r);
        double ref = x / d;
        if (__double_as_longlong(fast) != __double_as_longlong(ref)) {
            bad++;
            out[1] = (uint64_t)__double_as_longlong(x);
            out[2] = (uint64_t)__double_as_longlong(d);
        }
    }
    if (bad) atomicAdd((unsigned long long *)out, (unsigned long long)bad);
}

hipError_t launch_selftest_div(hipStream_t stream, uint64_t seed, uint32_t blocks, uint32_t per_thread, uint32_t family,
                               uint64_t *out3) {
    hipLaunchKernelGGL(k_selftest_div, dim3(blocks), dim3(256), 0, stream, seed, per_thread, family, out3);
    return hipGetLastError();
}

// ---- launchers -------------------------------------------------------------------------
template <class Source>
static hipError_t launch_phased(hipStream_t stream, const DevScene &s, const Source &src, uint64_t n_items,
                                uint64_t n_out, const PhasedScratch &scratch, const PhasedTuning &tune,
                                rtg_hit *out_hits, rtg_ray_counters *out_counters) {
    if (n_items == 0) return hipSuccess;
    // spheres and planes: one streaming pass; it is the whole answer when the scene has no mesh
    const bool has_mesh = s.n_mesh_objects != 0;
    rtg_hit *prim_out = has_mesh ? scratch.prim_best : out_hits;
    if (has_mesh && s.n_prims != 0 && !prim_out) return hipErrorInvalidValue;
    if ((s.n_prims != 0 || !has_mesh) && prim_out) {
        dim3 pg((uint32_t)((n_items + kBlock - 1) / kBlock)), pb(kBlock);
        hipLaunchKernelGGL((k_prims<Source>), pg, pb, 0, stream, s, src, n_items, prim_out);
    }
    if (!has_mesh) {
        if (out_counters) {
            hipError_t ce = hipMemsetAsync(out_counters, 0, n_out * sizeof(rtg_ray_counters), stream);
            if (ce != hipSuccess) return ce;
        }
        return hipGetLastError();
    }
    const rtg_hit *prim_best = s.n_prims != 0 ? scratch.prim_best : nullptr;
    uint32_t *work_counters = scratch.work_counters;
    hipError_t e = hipMemsetAsync(work_counters, 0, (8 * kCounterStride + 16) * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    uint64_t chunks = (n_items + 63) >> 6;
    // tune.max_blocks counts four-wave groups: the number of resident waves does not depend on the workgroup size
    const uint64_t per_group = 256 / kPhasedBlock;
    uint64_t want = (chunks + (kPhasedBlock / 64) - 1) / (kPhasedBlock / 64);
    // tune.max_blocks == 0: fill the GPU -- 256 CUs x waves per SIMD four-wave groups for the variant that will run
    const uint32_t fill = 256u * (uint32_t)phased_waves(Source::kFiveWaves, out_counters != nullptr, s.has_patches != 0,
                                                        s.n_mesh_objects > 1 || s.wide_mesh != 0 || tune.keep_state != 0,
                                                        tune.stats != 0);
    uint64_t most = (uint64_t)(tune.max_blocks ? tune.max_blocks : fill) * per_group;
    uint32_t grid = (uint32_t)(want < most ? want : most);
    if (grid == 0) grid = 1;
    // the outer ray is needed again when there are several meshes; only then does the dynamic LDS hold it (9 cold
    // fields per lane instead of 3)
    const bool count = out_counters != nullptr;
    const bool keep = s.n_mesh_objects > 1 || s.wide_mesh != 0 || tune.keep_state != 0;
    const bool pre = tune.pretest && s.any_f32_ok;
    PhasedParams prm{tune.tri_min, tune.obj_min, tune.patch_tri_min};
    const size_t cold_bytes = (size_t)(keep ? 9 : 2) * kPhasedBlock * sizeof(double);
    if (!keep && n_out > 0xFFFFFFFFull) return hipErrorInvalidValue;     // single-mesh builds keep a 32-bit output slot
    dim3 g(grid), b(kPhasedBlock);
#define RTG_L(C, X, P, K, S)                                                                                      \
    hipExtLaunchKernelGGL((k_trace_phased<Source, C, X, P, K, S>), g, b, cold_bytes, stream, tune.time_start,    \
                          tune.time_stop, 0, s, src, n_items, work_counters, prm, out_hits, out_counters, prim_best)
#define RTG_L_PRE(C, X, K, S) \
    do {                      \
        if (pre)              \
            RTG_L(C, X, true, K, S);  \
        else                  \
            RTG_L(C, X, false, K, S); \
    } while (0)
#define RTG_L_KEEP(C, X, S) \
    do {                    \
        if (keep)           \
            RTG_L_PRE(C, X, true, S);  \
        else                \
            RTG_L_PRE(C, X, false, S); \
    } while (0)
    if (tune.stats && !count && !keep && !s.has_patches) {
        RTG_L_PRE(false, false, false, true);
    } else if (s.has_patches) {
        if (count) RTG_L_KEEP(true, true, false);
        else RTG_L_KEEP(false, true, false);
    } else {
        if (count) RTG_L_KEEP(true, false, false);
        else RTG_L_KEEP(false, false, false);
    }
#undef RTG_L_KEEP
#undef RTG_L_PRE
#undef RTG_L
    return hipGetLastError();
}

hipError_t launch_render_primary_phased(hipStream_t stream, const DevScene &s, const rtg_view &view, uint32_t width,
                                        uint32_t height, rtg_rect tile, const PhasedScratch &scratch,
                                        const PhasedTuning &tune, rtg_hit *out_hits,
                                        rtg_ray_counters *out_counters) {
    PrimarySource src;
    src.view = view;
    src.width = width;
    src.height = height;
    src.tile = tile;
    src.tiles_x = (tile.w + 7) / 8;
    uint64_t tiles = (uint64_t)src.tiles_x * ((tile.h + 7) / 8);
    return launch_phased(stream, s, src, tiles * 64, (uint64_t)tile.w * tile.h, scratch, tune, out_hits, out_counters);
}

hipError_t launch_trace_phased(hipStream_t stream, const DevScene &s, const double *rays6, uint64_t n,
                               const PhasedScratch &scratch, const PhasedTuning &tune, rtg_hit *out_hits,
                               rtg_ray_counters *out_counters) {
    BatchSource src;
    src.rays6 = rays6;
    src.n = n;
    return launch_phased(stream, s, src, n, n, scratch, tune, out_hits, out_counters);
}

hipError_t launch_shadow_phased(hipStream_t stream, const DevScene &s, const rtg_point *points, uint64_t n_points,
                                const rtg_light *lights, uint32_t n_lights, const double *rays6, uint64_t n_rays,
                                const PhasedScratch &scratch, const PhasedTuning &tune, rtg_hit *out_hits) {
    ShadowSource shadow;
    shadow.points = points;
    shadow.lights = lights;
    shadow.n_points = n_points;
    shadow.n_lights = n_lights;
    if (n_rays == 0)
        return launch_phased(stream, s, shadow, n_points * n_lights, n_points * n_lights, scratch, tune, out_hits, nullptr);
    ShadowAndBatchSource src;
    src.shadow = shadow;
    src.batch.rays6 = rays6;
    src.batch.n = n_rays;
    src.n_shadow = n_points * n_lights;
    return launch_phased(stream, s, src, src.n_shadow + n_rays, src.n_shadow + n_rays, scratch, tune, out_hits, nullptr);
}

} // namespace rtg
