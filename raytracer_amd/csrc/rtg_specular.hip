// rtg_specular.hip -- the specular path tree (raytrace_all_specular_rec, render/renderer.rs:296-378) as levels of
// ray batches, so every closest-hit query of a frame runs in the persistent kernel.
//
// Level 0 is the primary rays.  A hit on a dielectric material at depth d (d + 1 < max_depth) spawns the reflected
// ray and, unless reflection is total, the refracted ray of Dielectric::new (render/dielectric.rs:13-52) into level
// d + 1.  The reference sums the radiance of a pixel's paths in depth-first order (hit, then the reflected subtree,
// then the refracted subtree); every ray therefore carries the pre-order index of its node in the pixel's binary
// path tree, its shaded radiance times its attenuation is parked in that slot, and a last pass adds a pixel's slots
// in index order -- the same additions in the same order as the recursion.
#include "rtg_device.hpp"
#include "rtg_launch.hpp"

namespace rtg {

// nodes of a full binary tree with `levels` levels
__device__ __forceinline__ uint32_t subtree_nodes(uint32_t levels) { return (1u << levels) - 1u; }

__device__ __forceinline__ bool refracts(const DevScene &s, const rtg_point &pt) {
    return pt.manifold_len != 0 && material_of_slot(s, pt.material_slot).has_dielectric != 0;
}

__device__ __forceinline__ Ray load_ray6(const double *rays6, uint64_t i) {
    const double2 *q = reinterpret_cast<const double2 *>(rays6 + i * 6);
    double2 a = q[0], b = q[1], c = q[2];
    return Ray{mk(a.x, a.y, b.x), mk(b.y, c.x, c.y)};
}
__device__ __forceinline__ void store_ray6(double *rays6, uint64_t i, const Ray &r) {
    double2 *q = reinterpret_cast<double2 *>(rays6 + i * 6);
    q[0] = make_double2(r.o.x, r.o.y);
    q[1] = make_double2(r.o.z, r.d.x);
    q[2] = make_double2(r.d.y, r.d.z);
}

// How many rays of this level hit a refracting material (each spawns at most two rays).
__global__ __launch_bounds__(kBlock) void k_spec_count(DevScene s, const rtg_point *__restrict__ points, uint64_t n,
                                                        uint32_t *counter) {
    uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    bool yes = gid < n && refracts(s, points[gid]);
    uint64_t m = __builtin_amdgcn_ballot_w64(yes);
    if ((threadIdx.x & 63u) == 0 && m != 0) atomicAdd(counter, (uint32_t)__popcll(m));
}

// Spawns level d + 1 from level d.  Level 0 (`rays6 == nullptr`) regenerates the primary rays and opens a slot group
// per refracting pixel (groups[g] = pixel index); deeper levels inherit group and attenuation from `meta`.
// counters[0]: rays written to the next level, counters[1]: groups opened.
__global__ __launch_bounds__(kBlock) void k_spec_spawn(DevScene s, rtg_view view, uint32_t width, uint32_t height,
                                                        rtg_rect tile, const double *__restrict__ rays6,
                                                        const SpecMeta *__restrict__ meta,
                                                        const rtg_point *__restrict__ points, uint64_t n, uint32_t depth,
                                                        uint32_t max_depth, double *next_rays6, SpecMeta *next_meta,
                                                        uint32_t *groups, uint32_t *counters) {
    uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = gid < n && refracts(s, points[gid]);
    // The two counters are bumped once per wave: the lanes that spawn take consecutive places (a lane with a
    // refracted ray takes two), in lane order.
    const uint64_t spawning = __ballot(active);
    if (spawning == 0) return;
    Ray ray{};
    SpecMeta me{};
    Dielectric di{};
    V3 pos = mk(0.0, 0.0, 0.0);
    if (active) {
        const rtg_point &pt = points[gid];
        const DevMaterial &m = material_of_slot(s, pt.material_slot);
        if (rays6) {
            ray = load_ray6(rays6, gid);
            me = meta[gid];
        } else {
            uint32_t py = (uint32_t)(gid / tile.w), px = (uint32_t)(gid - (uint64_t)py * tile.w);
            ray = primary_ray(view, tile.x0 + px, tile.y0 + py, width, height);
        }
        di = dielectric_new(ray.d, ld3(pt.geo_normal), m.n1, m.n2);
        pos = ld3(pt.position);
    }
    const uint64_t twice = __ballot(active && di.has_refract);
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __ffsll((unsigned long long)spawning) - 1;
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t ray_base = 0, group_base = 0;
    if ((int)lane == leader) {
        ray_base = atomicAdd(counters, (uint32_t)(__popcll(spawning) + __popcll(twice)));
        if (!rays6) group_base = atomicAdd(counters + 1, (uint32_t)__popcll(spawning));
    }
    ray_base = __shfl(ray_base, leader, 64);
    group_base = __shfl(group_base, leader, 64);
    if (!active) return;
    if (!rays6) {
        me.group = group_base + (uint32_t)__popcll(spawning & below);
        me.slot = 0;
        me.attenuation = 1.0;
        groups[me.group] = (uint32_t)gid;
    }
    const uint32_t at = ray_base + (uint32_t)(__popcll(spawning & below) + __popcll(twice & below));
    // children of the node at `depth` with pre-order index `slot`: the reflected subtree comes first
    SpecMeta c;
    c.group = me.group;
    c.slot = me.slot + 1u;
    c.attenuation = me.attenuation * di.reflectance;
    store_ray6(next_rays6, at, new_bounce(pos, di.reflect));
    next_meta[at] = c;
    if (di.has_refract) {
        c.slot = me.slot + 1u + subtree_nodes(max_depth - depth - 1u);
        c.attenuation = me.attenuation * (1.0 - di.reflectance);
        store_ray6(next_rays6, at + 1u, new_bounce(pos, di.refract));
        next_meta[at + 1u] = c;
    }
}

// Parks each ray's shaded radiance times its attenuation in its slot (a ray without a hit contributes nothing).
__global__ __launch_bounds__(kBlock) void k_spec_store(const SpecMeta *__restrict__ meta,
                                                        const rtg_point *__restrict__ points,
                                                        const double *__restrict__ rgb, uint64_t n,
                                                        uint32_t slots_per_group, double *slots) {
    uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= n) return;
    if (points[gid].manifold_len == 0) return;
    SpecMeta me = meta[gid];
    double *o = slots + ((uint64_t)me.group * slots_per_group + me.slot) * 4;
    o[0] = rgb[gid * 3 + 0] * me.attenuation;
    o[1] = rgb[gid * 3 + 1] * me.attenuation;
    o[2] = rgb[gid * 3 + 2] * me.attenuation;
    o[3] = 1.0;                                   // slot in use
}

// Adds a refracting pixel's slots to its primary radiance in pre-order and rewrites the pixel.
__global__ __launch_bounds__(kBlock) void k_spec_sum(const uint32_t *__restrict__ groups, uint32_t n_groups,
                                                      uint32_t slots_per_group, const double *__restrict__ slots,
                                                      double *frame_rgb, double *out_rgb, uint8_t *out_rgba8) {
    uint32_t g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n_groups) return;
    uint64_t pixel = groups[g];
    V3 total = mk(frame_rgb[pixel * 3 + 0], frame_rgb[pixel * 3 + 1], frame_rgb[pixel * 3 + 2]);
    const double *sl = slots + (uint64_t)g * slots_per_group * 4;
    for (uint32_t i = 1; i < slots_per_group; i++)
        if (sl[i * 4 + 3] != 0.0) total = total + mk(sl[i * 4 + 0], sl[i * 4 + 1], sl[i * 4 + 2]);
    if (out_rgb) {
        out_rgb[pixel * 3 + 0] = total.x;
        out_rgb[pixel * 3 + 1] = total.y;
        out_rgb[pixel * 3 + 2] = total.z;
    }
    if (out_rgba8) {
        uchar4 c;
        c.x = pq8(total.x);
        c.y = pq8(total.y);
        c.z = pq8(total.z);
        c.w = 255;
        reinterpret_cast<uchar4 *>(out_rgba8)[pixel] = c;
    }
}

static inline dim3 grid_for(uint64_t n) { return dim3((uint32_t)((n + kBlock - 1) / kBlock)); }

hipError_t launch_spec_count(hipStream_t stream, const DevScene &s, const rtg_point *points, uint64_t n,
                             uint32_t *counter) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_spec_count, grid_for(n), dim3(kBlock), 0, stream, s, points, n, counter);
    return hipGetLastError();
}

hipError_t launch_spec_spawn(hipStream_t stream, const DevScene &s, const rtg_view &view, uint32_t width,
                             uint32_t height, rtg_rect tile, const double *rays6, const SpecMeta *meta,
                             const rtg_point *points, uint64_t n, uint32_t depth, uint32_t max_depth,
                             double *next_rays6, SpecMeta *next_meta, uint32_t *groups, uint32_t *counters) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_spec_spawn, grid_for(n), dim3(kBlock), 0, stream, s, view, width, height, tile, rays6, meta,
                       points, n, depth, max_depth, next_rays6, next_meta, groups, counters);
    return hipGetLastError();
}

hipError_t launch_spec_store(hipStream_t stream, const SpecMeta *meta, const rtg_point *points, const double *rgb,
                             uint64_t n, uint32_t slots_per_group, double *slots) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_spec_store, grid_for(n), dim3(kBlock), 0, stream, meta, points, rgb, n, slots_per_group, slots);
    return hipGetLastError();
}

hipError_t launch_spec_sum(hipStream_t stream, const uint32_t *groups, uint32_t n_groups, uint32_t slots_per_group,
                           const double *slots, double *frame_rgb, double *out_rgb, uint8_t *out_rgba8) {
    if (n_groups == 0) return hipSuccess;
    hipLaunchKernelGGL(k_spec_sum, grid_for(n_groups), dim3(kBlock), 0, stream, groups, n_groups, slots_per_group, slots,
                       frame_rgb, out_rgb, out_rgba8);
    return hipGetLastError();
}

} // namespace rtg
