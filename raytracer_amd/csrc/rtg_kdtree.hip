// rtg_kdtree.hip -- KdTree::new (src/tree/kd_tree.rs:74-95, src/tree/seq_tree.rs:36-49) on the device.
//
// The reference builds its photon map recursively: bounds of the range, the widest axis (the last of equal
// maxima), select_nth_unstable at len / 2 along that axis, the two halves in turn; the tree is implicit in the
// array order.  Here all ranges of one depth are split together.  Every photon first gets its rank along each
// axis (three device-wide sorts); a level then is: bounds of every range by atomics, one radix sort of
// (range begin, rank along the range's axis) that orders every range at once and leaves finished entries where
// they are, and a pass that finishes the middle entry of every range and hands the rest to the two halves.  The
// selection is unstable in the reference (entries with equal coordinates may land on either side); ranks refine
// the coordinate order, so the arrangement made here is one of those the reference may produce.
//
// Once no range is longer than kFinishMax entries, the remaining levels cost launches, not work: one kernel then
// finishes every range in a workgroup of its own, with the same steps per level (bounds, widest axis, order by rank,
// middle entry) carried out in LDS, so the tree is the one the level-by-level passes make.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "rtg_device.hpp"
#include "rtg_launch.hpp"

namespace rtg {
namespace {

constexpr uint64_t kSign = 0x8000000000000000ull;
constexpr uint32_t kDone = 0xFFFFFFFFu;     // seg_end of a finished entry

// order-preserving integer image of a double (-0.0 below +0.0, as f64::minimum / maximum order them)
__device__ __forceinline__ uint64_t ord_of(double x) {
    uint64_t u = (uint64_t)__double_as_longlong(x);
    return (u & kSign) ? ~u : (u | kSign);
}
__device__ __forceinline__ double ord_back(uint64_t k) {
    uint64_t u = (k & kSign) ? (k ^ kSign) : ~k;
    return __longlong_as_double((long long)u);
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int d) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, d, 64);
    hi = __shfl_xor(hi, d, 64);
    return ((uint64_t)hi << 32) | lo;
}

// Sort keys of one axis; the pass over axis 0 also collects the longest path (summary[0]) and the first photon
// with a NaN coordinate (summary[1]).
__global__ __launch_bounds__(256) void k_kd_axis_keys(const rtg_photon *__restrict__ photons, uint32_t n, int axis,
                                                      uint64_t *__restrict__ keys, uint32_t *__restrict__ ids,
                                                      uint32_t *__restrict__ summary) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rtg_photon &p = photons[i];
    keys[i] = ord_of(p.position[axis]);
    ids[i] = i;
    if (axis == 0) {
        const uint32_t b = p.n_bounces < 3u ? p.n_bounces : 3u;
        if (b > 0) atomicMax(summary, b);
        if (p.position[0] != p.position[0] || p.position[1] != p.position[1] || p.position[2] != p.position[2])
            atomicMin(summary + 1, i);
    }
}

__global__ __launch_bounds__(256) void k_kd_ranks(const uint32_t *__restrict__ sorted_ids, uint32_t n, uint32_t *__restrict__ rank) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) rank[sorted_ids[r]] = r;
}

__global__ __launch_bounds__(256) void k_kd_start(uint32_t n, uint32_t *__restrict__ order, uint32_t *__restrict__ seg_begin,
                                                  uint32_t *__restrict__ seg_end) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    order[p] = p;
    seg_begin[p] = 0;
    seg_end[p] = n;
}

// Bounds of every unfinished range: box_lo / box_hi[begin * 3 + axis], cleared to all-ones / zero before.  At the
// first levels thousands of waves share a range and their atomics on its six words would queue up one behind the
// other, so a workgroup of sixteen waves whose entries all share a range combines them in LDS first.
constexpr uint32_t kBoxThreads = 1024;
__global__ __launch_bounds__(kBoxThreads) void k_kd_boxes(const rtg_photon *__restrict__ photons, uint32_t n, const uint32_t *__restrict__ order,
                                                          const uint32_t *__restrict__ seg_begin, const uint32_t *__restrict__ seg_end,
                                                          unsigned long long *__restrict__ box_lo, unsigned long long *__restrict__ box_hi) {
    __shared__ uint32_t s_range[kBoxThreads / 64];         // the wave's range when all its entries share one, else kDone
    __shared__ uint64_t s_lo[kBoxThreads / 64][3], s_hi[kBoxThreads / 64][3];
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = threadIdx.x >> 6;
    const bool active = p < n && seg_end[p] != kDone;
    const uint64_t lanes = __ballot(active);
    uint32_t b = 0;
    uint64_t w[3] = {~0ull, ~0ull, ~0ull}, v[3] = {0, 0, 0};
    if (active) {
        b = seg_begin[p];
        const rtg_photon &ph = photons[order[p]];
        for (int k = 0; k < 3; k++) w[k] = v[k] = ord_of(ph.position[k]);
    }
    const int first = lanes ? __ffsll((unsigned long long)lanes) - 1 : 0;
    const uint32_t b0 = __shfl(b, first, 64);
    const bool shared_range = lanes != 0 && __ballot(active && b != b0) == 0;
    if (shared_range) {
        for (int d = 32; d > 0; d >>= 1)
            for (int k = 0; k < 3; k++) {
                const uint64_t l = shfl_xor_u64(w[k], d), h = shfl_xor_u64(v[k], d);
                w[k] = l < w[k] ? l : w[k];
                v[k] = h > v[k] ? h : v[k];
            }
    } else if (active) {
        for (int k = 0; k < 3; k++) {
            atomicMin(box_lo + (size_t)b * 3 + k, (unsigned long long)w[k]);
            atomicMax(box_hi + (size_t)b * 3 + k, (unsigned long long)v[k]);
        }
    }
    if ((threadIdx.x & 63u) == 0) {
        s_range[wave] = shared_range ? b0 : kDone;
        for (int k = 0; k < 3; k++) {
            s_lo[wave][k] = w[k];
            s_hi[wave][k] = v[k];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                // runs of waves with the same range: one set of atomics per run
        uint32_t r = kDone;
        uint64_t lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0, 0, 0};
        for (uint32_t q = 0; q <= kBoxThreads / 64; q++) {
            const uint32_t rq = q < kBoxThreads / 64 ? s_range[q] : kDone;
            if (rq != r || q == kBoxThreads / 64) {
                if (r != kDone)
                    for (int k = 0; k < 3; k++) {
                        atomicMin(box_lo + (size_t)r * 3 + k, (unsigned long long)lo[k]);
                        atomicMax(box_hi + (size_t)r * 3 + k, (unsigned long long)hi[k]);
                    }
                r = rq;
                for (int k = 0; k < 3; k++) {
                    lo[k] = ~0ull;
                    hi[k] = 0;
                }
            }
            if (rq != kDone)
                for (int k = 0; k < 3; k++) {
                    lo[k] = s_lo[q][k] < lo[k] ? s_lo[q][k] : lo[k];
                    hi[k] = s_hi[q][k] > hi[k] ? s_hi[q][k] : hi[k];
                }
        }
    }
}

// key = (range begin << rank_bits) | rank along the range's widest axis; a finished entry keeps its place.
__global__ __launch_bounds__(256) void k_kd_keys(uint32_t n, const uint32_t *__restrict__ rank0, const uint32_t *__restrict__ rank1,
                                                 const uint32_t *__restrict__ rank2, const uint32_t *__restrict__ order,
                                                 const uint32_t *__restrict__ seg_begin, const uint32_t *__restrict__ seg_end,
                                                 const unsigned long long *__restrict__ box_lo,
                                                 const unsigned long long *__restrict__ box_hi, uint32_t rank_bits,
                                                 uint64_t *__restrict__ keys, uint32_t *__restrict__ seg_axis) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (seg_end[p] == kDone) {
        keys[p] = (uint64_t)p << rank_bits;
        return;
    }
    const uint32_t b = seg_begin[p];
    double extent[3];
    for (int k = 0; k < 3; k++) extent[k] = ord_back(box_hi[(size_t)b * 3 + k]) - ord_back(box_lo[(size_t)b * 3 + k]);
    int axis = 0;                               // max_by_key: the last of equal maxima (src/tree/kd_tree.rs:84-88)
    for (int k = 1; k < 3; k++)
        if (!(extent[k] < extent[axis])) axis = k;
    const uint32_t id = order[p];
    const uint32_t r = axis == 0 ? rank0[id] : (axis == 1 ? rank1[id] : rank2[id]);
    keys[p] = ((uint64_t)b << rank_bits) | r;
    if (p == b) seg_axis[b] = (uint32_t)axis;
}

// The middle entry of every range becomes its root (bounds = the range's, KdEntry::bounds); the entries left and
// right of it form the two ranges of the next level.
__global__ __launch_bounds__(256) void k_kd_split(const rtg_photon *__restrict__ photons, uint32_t n, const uint32_t *__restrict__ order,
                                                  uint32_t *__restrict__ seg_begin, uint32_t *__restrict__ seg_end,
                                                  const unsigned long long *__restrict__ box_lo,
                                                  const unsigned long long *__restrict__ box_hi, const uint32_t *__restrict__ seg_axis,
                                                  KdNode *__restrict__ nodes) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t e = seg_end[p];
    if (e == kDone) return;
    const uint32_t b = seg_begin[p];
    const uint32_t c = b + (e - b) / 2;         // src/tree/seq_tree.rs:41
    if (p < c) {
        seg_end[p] = c;
    } else if (p > c) {
        seg_begin[p] = c + 1;
    } else {
        const uint32_t id = order[p];
        KdNode node;
        for (int k = 0; k < 3; k++) {
            node.pos[k] = photons[id].position[k];
            node.lo[k] = ord_back(box_lo[(size_t)b * 3 + k]);
            node.hi[k] = ord_back(box_hi[(size_t)b * 3 + k]);
        }
        node.value = id;
        node.axis = seg_axis[b];
        nodes[p] = node;
        seg_begin[p] = p;
        seg_end[p] = kDone;
    }
}

// ---- the last levels, a range per workgroup --------------------------------------------------------------------
constexpr uint32_t kFinishMax = 1024;          // entries of a range the finishing kernel takes (a power of two)
constexpr uint32_t kFinishThreads = 256;
constexpr uint32_t kFinishPer = kFinishMax / kFinishThreads;

// The unfinished ranges, in any order: list[i] = begin of range i, *count = number of ranges.
__global__ __launch_bounds__(256) void k_kd_list_ranges(uint32_t n, const uint32_t *__restrict__ seg_begin,
                                                        const uint32_t *__restrict__ seg_end, uint32_t *__restrict__ list,
                                                        uint32_t *__restrict__ count) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && seg_end[p] != kDone && seg_begin[p] == p) list[atomicAdd(count, 1u)] = p;
}

__global__ __launch_bounds__(kFinishThreads) void k_kd_finish(const rtg_photon *__restrict__ photons,
                                                              const uint32_t *__restrict__ order,
                                                              const uint32_t *__restrict__ rank0, const uint32_t *__restrict__ rank1,
                                                              const uint32_t *__restrict__ rank2,
                                                              const uint32_t *__restrict__ seg_end, const uint32_t *__restrict__ list,
                                                              const uint32_t *__restrict__ count, KdNode *__restrict__ nodes) {
    if (blockIdx.x >= *count) return;
    __shared__ double pos[3][kFinishMax];                  // by entry (the order the range arrived in)
    __shared__ uint32_t id[kFinishMax];                    // by entry
    __shared__ uint64_t key[kFinishMax];                   // by position
    __shared__ uint32_t ent[kFinishMax];                   // by position: the entry that sits there
    __shared__ uint32_t sb[kFinishMax], se[kFinishMax];    // by position: the range the position belongs to
    __shared__ unsigned long long blo[kFinishMax][3], bhi[kFinishMax][3];   // by range begin
    __shared__ uint32_t sax[kFinishMax];                   // by range begin
    const uint32_t base = list[blockIdx.x];
    const uint32_t m = seg_end[base] - base;               // <= kFinishMax (the caller's loop guarantees it)
    const uint32_t t = threadIdx.x;
    for (uint32_t j = 0; j < kFinishPer; j++) {
        const uint32_t p = t + j * kFinishThreads;
        if (p < m) {
            const uint32_t i = order[base + p];
            id[p] = i;
            for (int k = 0; k < 3; k++) pos[k][p] = photons[i].position[k];
            ent[p] = p;
            sb[p] = 0;
            se[p] = m;
        } else {
            ent[p] = p;
            sb[p] = p;
            se[p] = kDone;
        }
    }
    __syncthreads();
    for (uint32_t longest = m; longest > 0; longest /= 2) {
        for (uint32_t j = 0; j < kFinishPer; j++) {        // bounds of every unfinished range
            const uint32_t p = t + j * kFinishThreads;
            if (se[p] != kDone && sb[p] == p)
                for (int k = 0; k < 3; k++) {
                    blo[p][k] = ~0ull;
                    bhi[p][k] = 0ull;
                }
        }
        __syncthreads();
        for (uint32_t j = 0; j < kFinishPer; j++) {
            const uint32_t p = t + j * kFinishThreads;
            if (se[p] != kDone) {
                const uint32_t b = sb[p], e = ent[p];
                for (int k = 0; k < 3; k++) {
                    const unsigned long long v = ord_of(pos[k][e]);
                    atomicMin(&blo[b][k], v);
                    atomicMax(&bhi[b][k], v);
                }
            }
        }
        __syncthreads();
        for (uint32_t j = 0; j < kFinishPer; j++) {        // key = (range begin, rank along the range's widest axis)
            const uint32_t p = t + j * kFinishThreads;
            if (se[p] == kDone) {
                key[p] = p >= m ? ~0ull : (uint64_t)p << 32;
            } else {
                const uint32_t b = sb[p];
                double extent[3];
                for (int k = 0; k < 3; k++) extent[k] = ord_back(bhi[b][k]) - ord_back(blo[b][k]);
                int axis = 0;
                for (int k = 1; k < 3; k++)
                    if (!(extent[k] < extent[axis])) axis = k;
                const uint32_t i = id[ent[p]];
                const uint32_t r = axis == 0 ? rank0[i] : (axis == 1 ? rank1[i] : rank2[i]);
                key[p] = ((uint64_t)b << 32) | r;
                if (p == b) sax[b] = (uint32_t)axis;
            }
        }
        __syncthreads();
        // bitonic sort of (key, entry) over all positions: finished positions and the padding keep their places
        for (uint32_t size = 2; size <= kFinishMax; size <<= 1)
            for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                for (uint32_t j = 0; j < kFinishPer / 2; j++) {
                    const uint32_t q = t + j * kFinishThreads;                    // pair number
                    const uint32_t lo = ((q & ~(stride - 1)) << 1) | (q & (stride - 1));
                    const uint32_t hi = lo | stride;
                    const bool up = (lo & size) == 0;
                    const uint64_t a = key[lo], c = key[hi];
                    if ((a > c) == up) {
                        key[lo] = c;
                        key[hi] = a;
                        const uint32_t x = ent[lo];
                        ent[lo] = ent[hi];
                        ent[hi] = x;
                    }
                }
                __syncthreads();
            }
        for (uint32_t j = 0; j < kFinishPer; j++) {        // the middle entry of every range is its root
            const uint32_t p = t + j * kFinishThreads;
            const uint32_t e = se[p];
            if (e == kDone) continue;
            const uint32_t b = sb[p];
            const uint32_t c = b + (e - b) / 2;
            if (p < c) {
                se[p] = c;
            } else if (p > c) {
                sb[p] = c + 1;
            } else {
                const uint32_t en = ent[p];
                KdNode node;
                for (int k = 0; k < 3; k++) {
                    node.pos[k] = pos[k][en];
                    node.lo[k] = ord_back(blo[b][k]);
                    node.hi[k] = ord_back(bhi[b][k]);
                }
                node.value = id[en];
                node.axis = sax[b];
                nodes[base + p] = node;
                sb[p] = p;
                se[p] = kDone;
            }
        }
        __syncthreads();
    }
}

uint32_t bits_for(uint32_t n) {                 // bits that hold 0 .. n - 1
    uint32_t bits = 1;
    while (bits < 32 && (1ull << bits) < n) bits++;
    return bits;
}
size_t up256(size_t b) { return (b + 255) / 256 * 256; }

struct KdLayout {
    size_t rank[3], order[2], keys[2], seg_begin, seg_end, seg_axis, box_lo, box_hi, summary, finish_count, temp, total;
    size_t temp_bytes;
};
KdLayout kd_layout(uint32_t n) {
    KdLayout l{};
    size_t at = 0;
    auto take = [&](size_t bytes) {
        const size_t here = at;
        at += up256(bytes);
        return here;
    };
    for (int k = 0; k < 3; k++) l.rank[k] = take((size_t)n * 4);
    for (int k = 0; k < 2; k++) l.order[k] = take((size_t)n * 4);
    for (int k = 0; k < 2; k++) l.keys[k] = take((size_t)n * 8);
    l.seg_begin = take((size_t)n * 4);
    l.seg_end = take((size_t)n * 4);
    l.seg_axis = take((size_t)n * 4);
    l.box_lo = take((size_t)n * 24);
    l.box_hi = take((size_t)n * 24);
    l.summary = take(8);
    l.finish_count = take(4);
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (size_t)n, 0u, 64u, (hipStream_t) nullptr);
    l.temp_bytes = bytes;
    l.temp = take(bytes);
    l.total = at;
    return l;
}

} // namespace

size_t kd_build_scratch_bytes(uint32_t n) { return kd_layout(n).total; }

hipError_t build_kd_device(hipStream_t stream, const rtg_photon *photons, uint32_t n, void *scratch, void *nodes, uint32_t **summary,
                           bool finish_in_lds) {
    if (n == 0) return hipSuccess;
    const KdLayout l = kd_layout(n);
    char *base = (char *)scratch;
    uint32_t *rank[3] = {(uint32_t *)(base + l.rank[0]), (uint32_t *)(base + l.rank[1]), (uint32_t *)(base + l.rank[2])};
    uint32_t *order[2] = {(uint32_t *)(base + l.order[0]), (uint32_t *)(base + l.order[1])};
    uint64_t *keys[2] = {(uint64_t *)(base + l.keys[0]), (uint64_t *)(base + l.keys[1])};
    uint32_t *seg_begin = (uint32_t *)(base + l.seg_begin), *seg_end = (uint32_t *)(base + l.seg_end);
    uint32_t *seg_axis = (uint32_t *)(base + l.seg_axis);
    unsigned long long *box_lo = (unsigned long long *)(base + l.box_lo), *box_hi = (unsigned long long *)(base + l.box_hi);
    uint32_t *sum = (uint32_t *)(base + l.summary);
    void *temp = base + l.temp;
    const dim3 grid((n + 255) / 256), block(256);
    hipError_t e;

    static const uint32_t init[2] = {0u, 0xFFFFFFFFu};      // static: the copy below may read it after this call returns
    if ((e = hipMemcpyAsync(sum, init, sizeof(init), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    for (int axis = 0; axis < 3; axis++) {
        hipLaunchKernelGGL(k_kd_axis_keys, grid, block, 0, stream, photons, n, axis, keys[0], order[0], sum);
        size_t temp_bytes = l.temp_bytes;
        if ((e = rocprim::radix_sort_pairs(temp, temp_bytes, keys[0], keys[1], order[0], order[1], (size_t)n, 0u, 64u, stream)) != hipSuccess)
            return e;
        hipLaunchKernelGGL(k_kd_ranks, grid, block, 0, stream, (const uint32_t *)order[1], n, rank[axis]);
    }
    hipLaunchKernelGGL(k_kd_start, grid, block, 0, stream, n, order[0], seg_begin, seg_end);
    const uint32_t rank_bits = bits_for(n);
    int cur = 0;
    uint64_t ranges = 1;                                         // upper bound on the unfinished ranges of the level
    for (uint32_t longest = n; longest > 0; longest /= 2) {      // the longest range of a level halves (rounding down)
        if (finish_in_lds && longest <= kFinishMax) {
            uint32_t *list = (uint32_t *)keys[0], *count = (uint32_t *)(base + l.finish_count);
            if ((e = hipMemsetAsync(count, 0, sizeof(uint32_t), stream)) != hipSuccess) return e;
            hipLaunchKernelGGL(k_kd_list_ranges, grid, block, 0, stream, n, (const uint32_t *)seg_begin, (const uint32_t *)seg_end, list,
                               count);
            hipLaunchKernelGGL(k_kd_finish, dim3((uint32_t)(ranges < n ? ranges : n)), dim3(kFinishThreads), 0, stream, photons,
                               (const uint32_t *)order[cur], (const uint32_t *)rank[0], (const uint32_t *)rank[1],
                               (const uint32_t *)rank[2], (const uint32_t *)seg_end, (const uint32_t *)list, (const uint32_t *)count,
                               (KdNode *)nodes);
            break;
        }
        ranges *= 2;
        if ((e = hipMemsetAsync(box_lo, 0xFF, (size_t)n * 24, stream)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(box_hi, 0x00, (size_t)n * 24, stream)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_kd_boxes, dim3((n + kBoxThreads - 1) / kBoxThreads), dim3(kBoxThreads), 0, stream, photons, n, (const uint32_t *)order[cur], (const uint32_t *)seg_begin,
                           (const uint32_t *)seg_end, box_lo, box_hi);
        hipLaunchKernelGGL(k_kd_keys, grid, block, 0, stream, n, (const uint32_t *)rank[0], (const uint32_t *)rank[1],
                           (const uint32_t *)rank[2], (const uint32_t *)order[cur], (const uint32_t *)seg_begin, (const uint32_t *)seg_end,
                           (const unsigned long long *)box_lo, (const unsigned long long *)box_hi, rank_bits, keys[0], seg_axis);
        size_t temp_bytes = l.temp_bytes;
        if ((e = rocprim::radix_sort_pairs(temp, temp_bytes, keys[0], keys[1], order[cur], order[cur ^ 1], (size_t)n, 0u, 2u * rank_bits,
                                           stream)) != hipSuccess)
            return e;
        cur ^= 1;
        hipLaunchKernelGGL(k_kd_split, grid, block, 0, stream, photons, n, (const uint32_t *)order[cur], seg_begin, seg_end,
                           (const unsigned long long *)box_lo, (const unsigned long long *)box_hi, (const uint32_t *)seg_axis,
                           (KdNode *)nodes);
    }
    if (summary) *summary = sum;
    return hipGetLastError();
}

} // namespace rtg
