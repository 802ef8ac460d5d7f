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

// Bounds of every unfinished range: box_lo / box_hi[begin * 3 + axis], cleared to all-ones / zero before.
__global__ __launch_bounds__(256) void k_kd_boxes(const rtg_photon *__restrict__ photons, uint32_t n, const uint32_t *__restrict__ order,
                                                  const uint32_t *__restrict__ seg_begin, const uint32_t *__restrict__ seg_end,
                                                  unsigned long long *__restrict__ box_lo, unsigned long long *__restrict__ box_hi) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = p < n && seg_end[p] != kDone;
    const uint64_t lanes = __ballot(active);
    if (lanes == 0) return;
    uint32_t b = 0;
    uint64_t w[3] = {~0ull, ~0ull, ~0ull}, v[3] = {0, 0, 0};
    if (active) {
        b = seg_begin[p];
        const rtg_photon &ph = photons[order[p]];
        for (int k = 0; k < 3; k++) w[k] = v[k] = ord_of(ph.position[k]);
    }
    const int first = __ffsll((unsigned long long)lanes) - 1;
    const uint32_t b0 = __shfl(b, first, 64);
    if (__ballot(active && b != b0) == 0) {
        // the wave's entries share a range: one set of atomics for all of them
        for (int d = 32; d > 0; d >>= 1)
            for (int k = 0; k < 3; k++) {
                const uint64_t l = shfl_xor_u64(w[k], d), h = shfl_xor_u64(v[k], d);
                w[k] = l < w[k] ? l : w[k];
                v[k] = h > v[k] ? h : v[k];
            }
        if ((int)(threadIdx.x & 63u) == first)
            for (int k = 0; k < 3; k++) {
                atomicMin(box_lo + (size_t)b0 * 3 + k, (unsigned long long)w[k]);
                atomicMax(box_hi + (size_t)b0 * 3 + k, (unsigned long long)v[k]);
            }
    } else if (active) {
        for (int k = 0; k < 3; k++) {
            atomicMin(box_lo + (size_t)b * 3 + k, (unsigned long long)w[k]);
            atomicMax(box_hi + (size_t)b * 3 + k, (unsigned long long)v[k]);
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

uint32_t bits_for(uint32_t n) {                 // bits that hold 0 .. n - 1
    uint32_t bits = 1;
    while (bits < 32 && (1ull << bits) < n) bits++;
    return bits;
}
size_t up256(size_t b) { return (b + 255) / 256 * 256; }

struct KdLayout {
    size_t rank[3], order[2], keys[2], seg_begin, seg_end, seg_axis, box_lo, box_hi, summary, temp, total;
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

hipError_t build_kd_device(hipStream_t stream, const rtg_photon *photons, uint32_t n, void *scratch, void *nodes, uint32_t **summary) {
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
    for (uint32_t longest = n; longest > 0; longest /= 2) {      // the longest range of a level halves (rounding down)
        if ((e = hipMemsetAsync(box_lo, 0xFF, (size_t)n * 24, stream)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(box_hi, 0x00, (size_t)n * 24, stream)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_kd_boxes, grid, block, 0, stream, photons, n, (const uint32_t *)order[cur], (const uint32_t *)seg_begin,
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
