// rtg_paths.hip -- Renderer::raytrace_all_specular (render/renderer.rs:284-378) for a batch of rays, as path records.
//
// The rays' path trees are traced level by level by the same machinery as a frame's (rtg_specular.hip): level 0 is
// the batch, a dielectric hit at depth d spawns the reflected and the refracted ray into level d + 1, and every ray
// carries the pre-order index of its node in its root ray's binary path tree.  Here each hit becomes a record; the
// records of all levels are then put in the reference's order -- root rays in batch order, and within one root
// ray the order the recursion pushes them, which is the pre-order index -- by one radix sort on (ray, index).
#include <hip/hip_runtime.h>

#include <cstddef>

#include <rocprim/device/device_radix_sort.hpp>

#include "rtg_device.hpp"
#include "rtg_launch.hpp"

namespace rtg {
namespace {

__device__ __forceinline__ uint32_t full_tree_nodes(uint32_t levels) { return (1u << levels) - 1u; }

__global__ __launch_bounds__(kBlock) void k_path_roots(uint64_t n, SpecMeta *meta) {
    uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= n) return;
    SpecMeta m;
    m.group = (uint32_t)gid;
    m.slot = 0;
    m.attenuation = 1.0;
    meta[gid] = m;
}

// One record per ray of the level that hit something, appended in any order; keys[(ray << 8) | slot].
__global__ __launch_bounds__(kBlock) void k_path_emit(const SpecMeta *__restrict__ meta,
                                                       const rtg_point *__restrict__ points, uint64_t n,
                                                       uint32_t max_depth, uint32_t *counter, uint64_t *keys,
                                                       rtg_path *records) {
    uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool emits = gid < n && points[gid].manifold_len != 0;
    // one counter update per wave: its lanes that have a record take consecutive places
    const uint64_t emitting = __ballot(emits);
    if (emitting == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == (uint32_t)(__ffsll((unsigned long long)emitting) - 1)) base = atomicAdd(counter, (uint32_t)__popcll(emitting));
    base = __shfl(base, __ffsll((unsigned long long)emitting) - 1, 64);
    if (!emits) return;
    const uint32_t idx = base + (uint32_t)__popcll(emitting & ((1ull << lane) - 1ull));
    const SpecMeta me = meta[gid];
    // walk down from the root to the node with pre-order index `slot`: the reflected subtree comes first
    uint32_t at = 0, depth = 0, modes = 0, parent = 0xFFFFFFFFu;
    while (at != me.slot) {
        parent = at;
        const uint32_t size = full_tree_nodes(max_depth - depth - 1u);
        if (me.slot - at - 1u < size) {
            at = at + 1u;
        } else {
            at = at + 1u + size;
            modes |= 1u << depth;
        }
        depth++;
    }
    keys[idx] = ((uint64_t)me.group << 8) | me.slot;
    rtg_path r;
    r.ray = me.group;
    r.depth = depth;
    r.modes = modes;
    r.parent = parent;                        // the parent's pre-order index for now; k_path_gather turns it into a position
    r.attenuation = me.attenuation;
    r._pad = 0.0;
    r.point = points[gid];
    records[idx] = r;
}

struct LevelTable {
    const rtg_path *records[8];
    uint32_t first[9];                        // unsorted index of each level's first record
};

__global__ __launch_bounds__(kBlock) void k_path_iota(uint32_t n, uint32_t *v) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) v[i] = i;
}

__global__ __launch_bounds__(kBlock) void k_path_gather(LevelTable table, uint32_t n_levels,
                                                         const uint64_t *__restrict__ sorted_keys,
                                                         const uint32_t *__restrict__ sorted_source, uint32_t n,
                                                         rtg_path *out) {
    uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const uint32_t src = sorted_source[p];
    // the level whose records hold `src`, picked with constant indices: a variable index would make the compiler
    // copy the table to scratch memory
    const rtg_path *records = table.records[0];
    uint32_t first = table.first[0];
#pragma unroll
    for (uint32_t l = 1; l < 8; l++)
        if (l < n_levels && src >= table.first[l]) {
            records = table.records[l];
            first = table.first[l];
        }
    // a record moves as nine 16-byte words; the first holds ray, depth, modes and parent
    static_assert(sizeof(rtg_path) == 144 && offsetof(rtg_path, parent) == 12, "path record layout");
    const uint4 *from = reinterpret_cast<const uint4 *>(records + (src - first));
    uint4 *to = reinterpret_cast<uint4 *>(out + p);
    uint4 head = from[0];
    if (head.w != 0xFFFFFFFFu) {
        // the parent's record sits before this one: lower pre-order index of the same ray
        const uint64_t want = ((uint64_t)head.x << 8) | head.w;
        uint32_t lo = 0, hi = p;
        while (lo < hi) {
            uint32_t mid = lo + (hi - lo) / 2;
            if (sorted_keys[mid] < want)
                lo = mid + 1;
            else
                hi = mid;
        }
        head.w = lo;
    }
    to[0] = head;
#pragma unroll
    for (int k = 1; k < 9; k++) to[k] = from[k];
}

inline dim3 grid_for(uint64_t n) { return dim3((uint32_t)((n + kBlock - 1) / kBlock)); }

} // namespace

hipError_t launch_path_roots(hipStream_t stream, uint64_t n, SpecMeta *meta) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_path_roots, grid_for(n), dim3(kBlock), 0, stream, n, meta);
    return hipGetLastError();
}

hipError_t launch_path_emit(hipStream_t stream, const SpecMeta *meta, const rtg_point *points, uint64_t n,
                            uint32_t max_depth, uint32_t *counter, uint64_t *keys, rtg_path *records) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_path_emit, grid_for(n), dim3(kBlock), 0, stream, meta, points, n, max_depth, counter, keys,
                       records);
    return hipGetLastError();
}

size_t path_sort_temp_bytes(uint32_t n, uint32_t key_bits) {
    size_t bytes = 0;
    uint64_t *k = nullptr;
    uint32_t *v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0u, key_bits, (hipStream_t) nullptr);
    return bytes;
}

hipError_t launch_path_sort_gather(hipStream_t stream, const rtg_path *const *level_records, const uint32_t *level_first,
                                   uint32_t n_levels, uint32_t n, uint32_t key_bits, const uint64_t *keys,
                                   uint64_t *keys_sorted, uint32_t *source, uint32_t *source_sorted, void *temp,
                                   size_t temp_bytes, rtg_path *out) {
    if (n == 0) return hipSuccess;
    LevelTable table{};
    for (uint32_t l = 0; l < n_levels; l++) {
        table.records[l] = level_records[l];
        table.first[l] = level_first[l];
    }
    table.first[n_levels] = n;
    hipLaunchKernelGGL(k_path_iota, grid_for(n), dim3(kBlock), 0, stream, n, source);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_sorted, source, source_sorted, (size_t)n, 0u, key_bits,
                                  stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_path_gather, grid_for(n), dim3(kBlock), 0, stream, table, n_levels, keys_sorted, source_sorted, n,
                       out);
    return hipGetLastError();
}

} // namespace rtg
