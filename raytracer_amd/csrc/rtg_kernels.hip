// rtg_kernels.hip -- gfx950 kernels of the ray-cast path and their launchers.
//
// One lane = one ray.  A wave64 covers an 8x8 pixel tile, a 256-thread block a
// 16x16 tile, so the lanes of a wave walk neighbouring parts of the BVH.  Blocks
// are dealt round-robin over the 8 XCDs, so block b is mapped to tile
// chunk(b % 8) + b / 8: each XCD (and its private L2) renders one contiguous run
// of tiles.
#include "rtg_device.hpp"
#include "rtg_launch.hpp"

namespace rtg {

__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nwg) {
    // bijective for every nwg: XCD k owns q (+1 for the first r XCDs) consecutive tiles
    uint32_t xcd = b & 7u, idx = b >> 3;
    uint32_t q = nwg >> 3, r = nwg & 7u;
    uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ TraceCtx make_trace_ctx(const DevScene &s, uint32_t *lds, uint64_t global_thread) {
    TraceCtx tc;
    tc.lds_col = lds + threadIdx.x;
    tc.gstride = s.overflow_stride;
    tc.glob_col = s.overflow ? s.overflow + (global_thread % s.overflow_stride) : nullptr;
    return tc;
}

__device__ __forceinline__ void store_hit(rtg_hit *out, const SceneHit &h) {
    // one 16-byte store per lane
    uint64_t tb = __double_as_longlong(h.have ? h.t : kInf);
    uint4 v;
    v.x = h.have ? h.object : RTG_NONE;
    v.y = h.have ? h.leaf : RTG_NONE;
    v.z = (uint32_t)tb;
    v.w = (uint32_t)(tb >> 32);
    *reinterpret_cast<uint4 *>(out) = v;
}

template <bool COUNT, bool POINTS, bool PATCH>
__global__ __launch_bounds__(kBlock) void k_render_primary(DevScene s, rtg_view view, uint32_t width,
                                                            uint32_t height, rtg_rect tile, uint32_t tiles_x,
                                                            uint32_t n_tiles, rtg_hit *out_hits,
                                                            rtg_point *out_points, rtg_ray_counters *out_counters) {
    __shared__ uint32_t lds[kLdsLevels * kBlock];
    uint32_t t = xcd_remap(blockIdx.x, n_tiles);
    uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
    uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t px = tx * 16 + (wave & 1u) * 8 + (lane & 7u);
    uint32_t py = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (px >= tile.w || py >= tile.h) return;
    uint64_t idx = (uint64_t)py * tile.w + px;
    TraceCtx tc = make_trace_ctx(s, lds, (uint64_t)blockIdx.x * kBlock + threadIdx.x);
    Ray ray = primary_ray(view, tile.x0 + px, tile.y0 + py, width, height);
    Counter ctr;
    ctr.nodes = 0;
    ctr.tris = 0;
    SceneHit h = scene_closest<COUNT, PATCH>(s, ray, tc, ctr);
    if (out_hits) store_hit(out_hits + idx, h);
    if (POINTS) scene_point<PATCH>(s, ray, h, out_points[idx]);
    if (COUNT) {
        out_counters[idx].nodes_visited = ctr.nodes;
        out_counters[idx].tris_tested = ctr.tris;
    }
}

template <bool COUNT, bool POINTS, bool PATCH>
__global__ __launch_bounds__(kBlock) void k_trace(DevScene s, const double *__restrict__ rays6, uint64_t n,
                                                   rtg_hit *out_hits, rtg_point *out_points,
                                                   rtg_ray_counters *out_counters) {
    __shared__ uint32_t lds[kLdsLevels * kBlock];
    uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= n) return;
    TraceCtx tc = make_trace_ctx(s, lds, gid);
    // 48 B per ray, 16-B aligned: three 128-bit loads
    const double2 *q = reinterpret_cast<const double2 *>(rays6 + gid * 6);
    double2 a = q[0], b = q[1], c = q[2];
    Ray ray{mk(a.x, a.y, b.x), mk(b.y, c.x, c.y)};
    Counter ctr;
    ctr.nodes = 0;
    ctr.tris = 0;
    SceneHit h = scene_closest<COUNT, PATCH>(s, ray, tc, ctr);
    if (out_hits) store_hit(out_hits + gid, h);
    if (POINTS) scene_point<PATCH>(s, ray, h, out_points[gid]);
    if (COUNT) {
        out_counters[gid].nodes_visited = ctr.nodes;
        out_counters[gid].tris_tested = ctr.tris;
    }
}

// RaycastPoint records for hits that are already known (the persistent kernel reports ids and times only): the
// winner's point is a function of the ray and the hit, so it is rebuilt here with every lane busy.
__device__ __forceinline__ Ray load_ray(const double *rays6, uint64_t i) {
    // 48 B per ray, 16-B aligned: three 128-bit loads
    const double2 *q = reinterpret_cast<const double2 *>(rays6 + i * 6);
    double2 a = q[0], b = q[1], c = q[2];
    return Ray{mk(a.x, a.y, b.x), mk(b.y, c.x, c.y)};
}
__device__ __forceinline__ SceneHit load_hit(const rtg_hit *p) {
    uint4 v = *reinterpret_cast<const uint4 *>(p);
    SceneHit h;
    h.object = v.x;
    h.leaf = v.y;
    h.t = __longlong_as_double((long long)(((uint64_t)v.w << 32) | v.z));
    h.have = v.x != RTG_NONE;
    return h;
}
// A workgroup's point records leave through LDS: a lane's record is 112 bytes, and written from the lane itself a
// store instruction would touch 56 cache lines for 1 KB of data; staged, the workgroup writes its 28 KB as one
// contiguous run, 16 bytes per lane per store.  (Falls back to direct stores when the output is not 16-byte aligned.)
__device__ __forceinline__ void store_points(const rtg_point &pt, bool valid, rtg_point *out_points, uint64_t first,
                                             uint64_t n, uint2 *stage) {
    static_assert(sizeof(rtg_point) == 112, "point record is 7 x 16 B");
    const bool aligned = ((uintptr_t)out_points & 15u) == 0;
    if (!aligned) {
        if (valid) out_points[first + threadIdx.x] = pt;
        return;
    }
    if (valid) {
        uint2 w[14];
        __builtin_memcpy(w, &pt, sizeof(pt));
#pragma unroll
        for (int k = 0; k < 14; k++) stage[threadIdx.x * 14 + k] = w[k];
    }
    __syncthreads();
    const uint64_t here = n - first < (uint64_t)kBlock ? n - first : (uint64_t)kBlock;
    const uint4 *src = reinterpret_cast<const uint4 *>(stage);
    uint4 *dst = reinterpret_cast<uint4 *>(out_points + first);
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const uint32_t i = (uint32_t)k * kBlock + threadIdx.x;
        if (i < here * 7) dst[i] = src[i];
    }
}
template <bool PATCH>
__global__ __launch_bounds__(kBlock) void k_points_primary(DevScene s, rtg_view view, uint32_t width, uint32_t height,
                                                            rtg_rect tile, const rtg_hit *__restrict__ hits,
                                                            rtg_point *out_points) {
    __shared__ __align__(16) uint2 stage[kBlock * 14];
    const uint64_t n = (uint64_t)tile.w * tile.h, first = (uint64_t)blockIdx.x * kBlock, gid = first + threadIdx.x;
    rtg_point pt;
    if (gid < n) {
        uint32_t py = (uint32_t)(gid / tile.w), px = (uint32_t)(gid - (uint64_t)py * tile.w);
        Ray ray = primary_ray(view, tile.x0 + px, tile.y0 + py, width, height);
        scene_point<PATCH>(s, ray, load_hit(hits + gid), pt);
    }
    store_points(pt, gid < n, out_points, first, n, stage);
}
template <bool PATCH>
__global__ __launch_bounds__(kBlock) void k_points_batch(DevScene s, const double *__restrict__ rays6, uint64_t n,
                                                          const rtg_hit *__restrict__ hits, rtg_point *out_points) {
    __shared__ __align__(16) uint2 stage[kBlock * 14];
    const uint64_t first = (uint64_t)blockIdx.x * kBlock, gid = first + threadIdx.x;
    rtg_point pt;
    if (gid < n) scene_point<PATCH>(s, load_ray(rays6, gid), load_hit(hits + gid), pt);
    store_points(pt, gid < n, out_points, first, n, stage);
}

// compute_direct_irrad x diffuse for already-built points (renderer.rs:184-205,270-277).
template <bool PATCH>
__global__ __launch_bounds__(kBlock) void k_shade_direct(DevScene s, const rtg_light *__restrict__ lights,
                                                          uint32_t n_lights, const rtg_point *__restrict__ points,
                                                          uint64_t n, const double *__restrict__ indirect,
                                                          double *rgb_out, uint8_t *rgba8_out) {
    __shared__ uint32_t lds[kLdsLevels * kBlock];
    uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= n) return;
    TraceCtx tc = make_trace_ctx(s, lds, gid);
    const rtg_point &pt = points[gid];
    V3 total = mk(0.0, 0.0, 0.0);
    if (pt.manifold_len != 0) {
        Counter ctr;
        // indirect + direct + ambient (renderer.rs:270-273); without a photon map the indirect term is Color::default()
        V3 irrad = ((indirect ? ld3(indirect + gid * 3) : mk(0.0, 0.0, 0.0)) +
                    direct_irrad<false, PATCH>(s, lights, n_lights, ld3(pt.position), ld3(pt.inter_normal), tc, ctr)) +
                   mk(0.0, 0.0, 0.0);
        const DevMaterial &m = material_of_slot(s, pt.material_slot);
        V3 dm = mk(irrad.x * m.diffuse[0], irrad.y * m.diffuse[1], irrad.z * m.diffuse[2]);
        total = total + dm * 1.0;
    }
    if (rgb_out) {
        rgb_out[gid * 3 + 0] = total.x;
        rgb_out[gid * 3 + 1] = total.y;
        rgb_out[gid * 3 + 2] = total.z;
    }
    if (rgba8_out) {
        uchar4 c;
        c.x = pq8(total.x);
        c.y = pq8(total.y);
        c.z = pq8(total.z);
        c.w = 255;
        reinterpret_cast<uchar4 *>(rgba8_out)[gid] = c;
    }
}

// The same shading with the shadow queries already answered (one closest hit per light and point, found by the
// persistent kernel): occluders[l * n + p].  Pairs without a query were never written and are never read.
__global__ __launch_bounds__(kBlock) void k_shade_resolve(DevScene s, const rtg_light *__restrict__ lights,
                                                           uint32_t n_lights, const rtg_point *__restrict__ points,
                                                           uint64_t n, const rtg_hit *__restrict__ occluders,
                                                           const double *__restrict__ indirect, double *rgb_out,
                                                           uint8_t *rgba8_out) {
    uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= n) return;
    const rtg_point &pt = points[gid];
    V3 total = mk(0.0, 0.0, 0.0);
    if (pt.manifold_len != 0) {
        V3 position = ld3(pt.position), inter_normal = ld3(pt.inter_normal);
        V3 lighting = mk(1.0, 1.0, 1.0) * 0.0;
        for (uint32_t i = 0; i < n_lights; i++) {
            ShadowQuery q;
            if (!shadow_query(lights[i], position, inter_normal, q)) continue;
            uint4 o = *reinterpret_cast<const uint4 *>(occluders + (uint64_t)i * n + gid);
            double occ_t = __longlong_as_double((long long)(((uint64_t)o.w << 32) | o.z));
            lighting = add_light(lighting, lights[i], q, o.x != RTG_NONE, occ_t);
        }
        V3 irrad = ((indirect ? ld3(indirect + gid * 3) : mk(0.0, 0.0, 0.0)) + lighting) + mk(0.0, 0.0, 0.0);
        const DevMaterial &m = material_of_slot(s, pt.material_slot);
        V3 dm = mk(irrad.x * m.diffuse[0], irrad.y * m.diffuse[1], irrad.z * m.diffuse[2]);
        total = total + dm * 1.0;
    }
    if (rgb_out) {
        rgb_out[gid * 3 + 0] = total.x;
        rgb_out[gid * 3 + 1] = total.y;
        rgb_out[gid * 3 + 2] = total.z;
    }
    if (rgba8_out) {
        uchar4 c;
        c.x = pq8(total.x);
        c.y = pq8(total.y);
        c.z = pq8(total.z);
        c.w = 255;
        reinterpret_cast<uchar4 *>(rgba8_out)[gid] = c;
    }
}

// Fused frame: raytrace_pixel (render/renderer.rs:266-283) with the photon term
// empty.  The specular recursion (renderer.rs:296-378, depth cap at :312) is
// unrolled into an explicit stack of pending rays: a dielectric hit schedules
// its refracted ray and continues with the reflected one, which reproduces the
// reference's depth-first output order, so the radiance sum has the same
// operation order.
constexpr int kMaxSpecDepth = 8;
struct Pending {
    Ray ray;
    double attenuation;
    uint32_t depth;
};

// With `only_refracting` set (the point records of the primary hits), the kernel redoes just the pixels whose
// primary hit has a dielectric material -- the ones with more than one path -- and leaves the others as the
// pass-structured renderer wrote them.
template <bool PATCH>
__global__ __launch_bounds__(kBlock) void k_render_frame(DevScene s, rtg_view view,
                                                          const rtg_light *__restrict__ lights, uint32_t n_lights,
                                                          uint32_t width, uint32_t height, rtg_rect tile,
                                                          uint32_t tiles_x, uint32_t n_tiles, uint32_t max_depth,
                                                          double *out_rgb, uint8_t *out_rgba8, rtg_hit *out_hits,
                                                          const rtg_point *__restrict__ only_refracting) {
    __shared__ uint32_t lds[kLdsLevels * kBlock];
    uint32_t t = xcd_remap(blockIdx.x, n_tiles);
    uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
    uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t px = tx * 16 + (wave & 1u) * 8 + (lane & 7u);
    uint32_t py = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (px >= tile.w || py >= tile.h) return;
    uint64_t idx = (uint64_t)py * tile.w + px;
    if (only_refracting) {
        const rtg_point &p0 = only_refracting[idx];
        if (p0.manifold_len == 0 || !material_of_slot(s, p0.material_slot).has_dielectric) return;
    }
    TraceCtx tc = make_trace_ctx(s, lds, (uint64_t)blockIdx.x * kBlock + threadIdx.x);
    Counter ctr;

    Pending stack[kMaxSpecDepth];
    int sp = 0;
    Ray ray = primary_ray(view, tile.x0 + px, tile.y0 + py, width, height);
    double attenuation = 1.0;
    uint32_t depth = 0;
    bool live = max_depth > 0;
    bool first = true;
    V3 total = mk(0.0, 0.0, 0.0);
    while (live) {
        SceneHit h = scene_closest<false, PATCH>(s, ray, tc, ctr);
        if (first) {
            if (out_hits) store_hit(out_hits + idx, h);
            first = false;
        }
        bool cont = false;
        if (h.have) {
            rtg_point pt;
            scene_point<PATCH>(s, ray, h, pt);
            const DevMaterial &m = material_of_slot(s, pt.material_slot);
            V3 irrad = (mk(0.0, 0.0, 0.0) + direct_irrad<false, PATCH>(s, lights, n_lights, ld3(pt.position),
                                                                 ld3(pt.inter_normal), tc, ctr)) +
                       mk(0.0, 0.0, 0.0);
            V3 dm = mk(irrad.x * m.diffuse[0], irrad.y * m.diffuse[1], irrad.z * m.diffuse[2]);
            total = total + dm * attenuation;
            if (m.has_dielectric && depth + 1 < max_depth) {
                Dielectric di = dielectric_new(ray.d, ld3(pt.geo_normal), m.n1, m.n2);
                V3 pos = ld3(pt.position);
                if (di.has_refract && sp < kMaxSpecDepth) {
                    stack[sp].ray = new_bounce(pos, di.refract);
                    stack[sp].attenuation = attenuation * (1.0 - di.reflectance);
                    stack[sp].depth = depth + 1;
                    sp++;
                }
                ray = new_bounce(pos, di.reflect);
                attenuation = attenuation * di.reflectance;
                depth = depth + 1;
                cont = true;
            }
        }
        if (!cont) {
            if (sp > 0) {
                sp--;
                ray = stack[sp].ray;
                attenuation = stack[sp].attenuation;
                depth = stack[sp].depth;
            } else {
                live = false;
            }
        }
    }
    if (first && out_hits) {
        SceneHit none;
        none.have = false;
        none.t = kInf;
        none.object = RTG_NONE;
        none.leaf = RTG_NONE;
        store_hit(out_hits + idx, none);
    }
    if (out_rgb) {
        out_rgb[idx * 3 + 0] = total.x;
        out_rgb[idx * 3 + 1] = total.y;
        out_rgb[idx * 3 + 2] = total.z;
    }
    if (out_rgba8) {
        uchar4 c;
        c.x = pq8(total.x);
        c.y = pq8(total.y);
        c.z = pq8(total.z);
        c.w = 255;
        reinterpret_cast<uchar4 *>(out_rgba8)[idx] = c;
    }
}

// ------------------------------------------------------------- launchers --
static inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

hipError_t launch_render_primary(hipStream_t stream, const DevScene &s, const rtg_view &view, uint32_t width,
                                 uint32_t height, rtg_rect tile, rtg_hit *out_hits, rtg_point *out_points,
                                 rtg_ray_counters *out_counters) {
    uint32_t tiles_x = div_up(tile.w, 16), tiles_y = div_up(tile.h, 16);
    uint32_t n_tiles = tiles_x * tiles_y;
    if (n_tiles == 0) return hipSuccess;
    dim3 grid(n_tiles), block(kBlock);
    bool count = out_counters != nullptr, points = out_points != nullptr;
#define RTG_L(C, P, X) hipLaunchKernelGGL((k_render_primary<C, P, X>), grid, block, 0, stream, s, view, width, height, tile, tiles_x, n_tiles, out_hits, out_points, out_counters)
    if (s.has_patches) {
        if (count && points) RTG_L(true, true, true);
        else if (count) RTG_L(true, false, true);
        else if (points) RTG_L(false, true, true);
        else RTG_L(false, false, true);
    } else {
        if (count && points) RTG_L(true, true, false);
        else if (count) RTG_L(true, false, false);
        else if (points) RTG_L(false, true, false);
        else RTG_L(false, false, false);
    }
#undef RTG_L
    return hipGetLastError();
}

hipError_t launch_trace(hipStream_t stream, const DevScene &s, const double *rays6, uint64_t n, rtg_hit *out_hits,
                        rtg_point *out_points, rtg_ray_counters *out_counters) {
    if (n == 0) return hipSuccess;
    dim3 grid((uint32_t)((n + kBlock - 1) / kBlock)), block(kBlock);
    bool count = out_counters != nullptr, points = out_points != nullptr;
#define RTG_L(C, P, X) hipLaunchKernelGGL((k_trace<C, P, X>), grid, block, 0, stream, s, rays6, n, out_hits, out_points, out_counters)
    if (s.has_patches) {
        if (count && points) RTG_L(true, true, true);
        else if (count) RTG_L(true, false, true);
        else if (points) RTG_L(false, true, true);
        else RTG_L(false, false, true);
    } else {
        if (count && points) RTG_L(true, true, false);
        else if (count) RTG_L(true, false, false);
        else if (points) RTG_L(false, true, false);
        else RTG_L(false, false, false);
    }
#undef RTG_L
    return hipGetLastError();
}

hipError_t launch_shade_direct(hipStream_t stream, const DevScene &s, const rtg_light *lights, uint32_t n_lights,
                               const rtg_point *points, uint64_t n, double *rgb_out, uint8_t *rgba8_out,
                               const double *indirect) {
    if (n == 0) return hipSuccess;
    dim3 grid((uint32_t)((n + kBlock - 1) / kBlock)), block(kBlock);
    if (s.has_patches)
        hipLaunchKernelGGL(k_shade_direct<true>, grid, block, 0, stream, s, lights, n_lights, points, n, indirect, rgb_out, rgba8_out);
    else
        hipLaunchKernelGGL(k_shade_direct<false>, grid, block, 0, stream, s, lights, n_lights, points, n, indirect, rgb_out, rgba8_out);
    return hipGetLastError();
}

hipError_t launch_points_primary(hipStream_t stream, const DevScene &s, const rtg_view &view, uint32_t width,
                                 uint32_t height, rtg_rect tile, const rtg_hit *hits, rtg_point *out_points) {
    uint64_t n = (uint64_t)tile.w * tile.h;
    if (n == 0) return hipSuccess;
    dim3 grid((uint32_t)((n + kBlock - 1) / kBlock)), block(kBlock);
    if (s.has_patches)
        hipLaunchKernelGGL(k_points_primary<true>, grid, block, 0, stream, s, view, width, height, tile, hits, out_points);
    else
        hipLaunchKernelGGL(k_points_primary<false>, grid, block, 0, stream, s, view, width, height, tile, hits, out_points);
    return hipGetLastError();
}

hipError_t launch_points_batch(hipStream_t stream, const DevScene &s, const double *rays6, uint64_t n,
                               const rtg_hit *hits, rtg_point *out_points) {
    if (n == 0) return hipSuccess;
    dim3 grid((uint32_t)((n + kBlock - 1) / kBlock)), block(kBlock);
    if (s.has_patches)
        hipLaunchKernelGGL(k_points_batch<true>, grid, block, 0, stream, s, rays6, n, hits, out_points);
    else
        hipLaunchKernelGGL(k_points_batch<false>, grid, block, 0, stream, s, rays6, n, hits, out_points);
    return hipGetLastError();
}

hipError_t launch_shade_resolve(hipStream_t stream, const DevScene &s, const rtg_light *lights, uint32_t n_lights,
                                const rtg_point *points, uint64_t n, const rtg_hit *occluders, double *rgb_out,
                                uint8_t *rgba8_out, const double *indirect) {
    if (n == 0) return hipSuccess;
    dim3 grid((uint32_t)((n + kBlock - 1) / kBlock)), block(kBlock);
    hipLaunchKernelGGL(k_shade_resolve, grid, block, 0, stream, s, lights, n_lights, points, n, occluders, indirect, rgb_out,
                       rgba8_out);
    return hipGetLastError();
}

hipError_t launch_render_frame(hipStream_t stream, const DevScene &s, const rtg_view &view, const rtg_light *lights,
                               uint32_t n_lights, uint32_t width, uint32_t height, rtg_rect tile, uint32_t max_depth,
                               double *out_rgb, uint8_t *out_rgba8, rtg_hit *out_hits,
                               const rtg_point *only_refracting) {
    uint32_t tiles_x = div_up(tile.w, 16), tiles_y = div_up(tile.h, 16);
    uint32_t n_tiles = tiles_x * tiles_y;
    if (n_tiles == 0) return hipSuccess;
    dim3 grid(n_tiles), block(kBlock);
    if (s.has_patches)
        hipLaunchKernelGGL(k_render_frame<true>, grid, block, 0, stream, s, view, lights, n_lights, width, height, tile,
                           tiles_x, n_tiles, max_depth, out_rgb, out_rgba8, out_hits, only_refracting);
    else
        hipLaunchKernelGGL(k_render_frame<false>, grid, block, 0, stream, s, view, lights, n_lights, width, height, tile,
                           tiles_x, n_tiles, max_depth, out_rgb, out_rgba8, out_hits, only_refracting);
    return hipGetLastError();
}

} // namespace rtg
