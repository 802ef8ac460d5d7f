// rtg_sort.hip -- ordering a ray batch for coherence on the device.
//
// The reference traces a batch in the order it is given (src/tree/benchmark.rs:17-28).  Rays that start close
// together and point into the same octant visit the same nodes, so tracing them next to each other raises the cache
// hit rates of an incoherent batch against a tree larger than the caches.  The key of a ray is
//   octant of the direction (3 bits) | 30-bit Morton code of the origin inside the batch's own bounding box,
// pairs (key, ray index) go through one rocPRIM radix sort, the rays are gathered into key order, traced, and the hit
// records scattered back to the caller's order -- answers are per ray, so the order never changes them.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "rtg_launch.hpp"

namespace rtg {
namespace {

constexpr uint64_t kSign = 0x8000000000000000ull;
__device__ inline uint64_t ord_of(double x) {
    uint64_t u = (uint64_t)__double_as_longlong(x);
    return (u & kSign) ? ~u : (u | kSign);
}
__host__ inline double ord_back(uint64_t k) {
    uint64_t u = (k & kSign) ? (k ^ kSign) : ~k;
    double x;
    __builtin_memcpy(&x, &u, 8);
    return x;
}
__device__ inline uint64_t shfl_down_u64(uint64_t v, int d) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_down(lo, d, 64);
    hi = __shfl_down(hi, d, 64);
    return ((uint64_t)hi << 32) | lo;
}

// box[0..2] = min, box[3..5] = max of the finite origin coordinates, as order-preserving integer images
__global__ __launch_bounds__(256) void k_ray_bounds(const double *__restrict__ rays6, uint64_t n, unsigned long long *box) {
    uint64_t lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const double2 *q = reinterpret_cast<const double2 *>(rays6 + i * 6);
        double2 a = q[0], b = q[1];
        const double o[3] = {a.x, a.y, b.x};
        for (int k = 0; k < 3; k++)
            if (is_finite(o[k])) {
                uint64_t w = ord_of(o[k]);
                lo[k] = w < lo[k] ? w : lo[k];
                hi[k] = w > hi[k] ? w : hi[k];
            }
    }
    for (int d = 32; d > 0; d >>= 1)
        for (int k = 0; k < 3; k++) {
            uint64_t l = shfl_down_u64(lo[k], d), h = shfl_down_u64(hi[k], d);
            lo[k] = l < lo[k] ? l : lo[k];
            hi[k] = h > hi[k] ? h : hi[k];
        }
    if ((threadIdx.x & 63u) == 0)
        for (int k = 0; k < 3; k++) {
            atomicMin(box + k, (unsigned long long)lo[k]);
            atomicMax(box + 3 + k, (unsigned long long)hi[k]);
        }
}

__device__ inline uint32_t spread10(uint32_t v) {      // 10 bits -> every third bit
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void k_ray_sort_keys(const double *__restrict__ rays6, uint64_t n,
                                                        const unsigned long long *__restrict__ box,
                                                        uint64_t *__restrict__ keys, uint32_t *__restrict__ index) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double2 *q = reinterpret_cast<const double2 *>(rays6 + i * 6);
    double2 a = q[0], b = q[1], c = q[2];
    const double o[3] = {a.x, a.y, b.x}, d[3] = {b.y, c.x, c.y};
    uint32_t cell[3];
    for (int k = 0; k < 3; k++) {
        uint64_t lo_k = box[k], hi_k = box[3 + k];
        uint64_t ul = (lo_k & kSign) ? (lo_k ^ kSign) : ~lo_k, uh = (hi_k & kSign) ? (hi_k ^ kSign) : ~hi_k;
        double lo = __longlong_as_double((long long)ul), hi = __longlong_as_double((long long)uh);
        double ext = hi - lo;
        double t = ext > 0.0 ? (o[k] - lo) / ext : 0.0;
        if (!(t > 0.0)) t = 0.0;              // also NaN
        if (t > 1.0) t = 1.0;
        uint32_t v = (uint32_t)(t * 1023.0);
        cell[k] = v > 1023u ? 1023u : v;
    }
    uint64_t morton = ((uint64_t)spread10(cell[0]) << 2) | ((uint64_t)spread10(cell[1]) << 1) | spread10(cell[2]);
    uint64_t oct = (d[0] < 0.0 ? 1u : 0u) | (d[1] < 0.0 ? 2u : 0u) | (d[2] < 0.0 ? 4u : 0u);
    keys[i] = (oct << 30) | morton;
    index[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void k_gather_rays(const double *__restrict__ rays6, const uint32_t *__restrict__ perm,
                                                      uint64_t n, double *__restrict__ out6) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double2 *q = reinterpret_cast<const double2 *>(rays6 + (uint64_t)perm[i] * 6);
    double2 a = q[0], b = q[1], c = q[2];
    double2 *w = reinterpret_cast<double2 *>(out6 + i * 6);
    w[0] = a;
    w[1] = b;
    w[2] = c;
}

__global__ __launch_bounds__(256) void k_scatter_results(const rtg_hit *__restrict__ hits, const rtg_ray_counters *__restrict__ ctr,
                                                          const uint32_t *__restrict__ perm, uint64_t n,
                                                          rtg_hit *__restrict__ out_hits, rtg_ray_counters *__restrict__ out_ctr) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t to = perm[i];
    if (out_hits) reinterpret_cast<uint4 *>(out_hits)[to] = reinterpret_cast<const uint4 *>(hits)[i];
    if (out_ctr) reinterpret_cast<uint2 *>(out_ctr)[to] = reinterpret_cast<const uint2 *>(ctr)[i];
}

constexpr unsigned kKeyBits = 33;

} // namespace

size_t ray_sort_temp_bytes(uint64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                    (uint32_t *)nullptr, (size_t)n, 0u, kKeyBits, (hipStream_t) nullptr);
    return bytes;
}

hipError_t launch_ray_sort(hipStream_t stream, const double *rays6, uint64_t n, const RaySortScratch &w,
                           double *sorted_rays6) {
    if (n == 0) return hipSuccess;
    // bounds of the origins: min images start at all ones, max images at zero
    hipError_t e = hipMemsetAsync(w.box, 0xFF, 3 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(w.box + 3, 0, 3 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(k_ray_bounds, dim3(blocks < 4096u ? blocks : 4096u), dim3(256), 0, stream, rays6, n, w.box);
    hipLaunchKernelGGL(k_ray_sort_keys, dim3(blocks), dim3(256), 0, stream, rays6, n, w.box, w.keys, w.index);
    size_t temp_bytes = w.temp_bytes;
    e = rocprim::radix_sort_pairs(w.temp, temp_bytes, w.keys, w.keys_sorted, w.index, w.perm, (size_t)n, 0u, kKeyBits, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_gather_rays, dim3(blocks), dim3(256), 0, stream, rays6, w.perm, n, sorted_rays6);
    return hipGetLastError();
}

hipError_t launch_ray_unsort(hipStream_t stream, const rtg_hit *hits, const rtg_ray_counters *counters, const uint32_t *perm,
                             uint64_t n, rtg_hit *out_hits, rtg_ray_counters *out_counters) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_results, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, hits, counters, perm, n,
                       out_hits, out_counters);
    return hipGetLastError();
}

} // namespace rtg
