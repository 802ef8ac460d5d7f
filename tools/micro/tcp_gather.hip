// Microbenchmark: cost of gathering 32-byte records with 128-bit loads when (A) every lane fetches its own record
// with two loads, or (B) lane pairs fetch each record together (adjacent 16-byte halves in one instruction).
// Build: hipcc --offload-arch=gfx950 -O3 -o tcp_gather tools/micro/tcp_gather.hip ; run: ./tcp_gather
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t vu4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 4) void k_gather(const vu4 *__restrict__ recs, uint32_t n_recs, uint32_t iters,
                                                    uint32_t *out) {
    uint32_t lane = threadIdx.x & 63u;
    uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    uint32_t st = gid * 2654435761u + 12345u;
    uint32_t st_pair = (gid & ~1u) * 2654435761u + 12345u;     // state of the even lane of the pair
    uint32_t st_odd = (gid | 1u) * 2654435761u + 12345u;
    vu4 acc = {0, 0, 0, 0};
    for (uint32_t i = 0; i < iters; i++) {
        if (MODE == 0) {            // own record, two loads
            st = st * 1664525u + 1013904223u;
            uint32_t idx = (st >> 8) % n_recs;
            vu4 a = recs[(size_t)idx * 2], b = recs[(size_t)idx * 2 + 1];
            acc ^= a ^ b;
        } else if (MODE == 1) {     // pair-cooperative: both lanes read the even lane's record, then the odd lane's
            st_pair = st_pair * 1664525u + 1013904223u;
            st_odd = st_odd * 1664525u + 1013904223u;
            uint32_t ia = (st_pair >> 8) % n_recs, ib = (st_odd >> 8) % n_recs;
            vu4 a = recs[(size_t)ia * 2 + (lane & 1u)], b = recs[(size_t)ib * 2 + (lane & 1u)];
            acc ^= a ^ b;
        } else if (MODE == 2) {     // one 16-byte load per lane from its own record (half the requests of mode 0)
            st = st * 1664525u + 1013904223u;
            uint32_t idx = (st >> 8) % n_recs;
            vu4 a = recs[(size_t)idx * 2];
            acc ^= a;
        } else if (MODE == 4) {     // dependent chain: the next record index comes out of the loaded record (two loads)
            uint32_t idx = (st >> 8) % n_recs;
            vu4 a = recs[(size_t)idx * 2], b = recs[(size_t)idx * 2 + 1];
            acc ^= a ^ b;
            st = st * 1664525u + (a.x ^ b.y);
        } else {                    // quad-cooperative 64-byte records: four lanes read one record's four quarters
            st_pair = (st_pair * 1664525u + 1013904223u);
            uint32_t base = ((gid & ~3u) * 2654435761u + 12345u) + i * 40503u;
            uint32_t idx = ((base * 1664525u + 1013904223u) >> 8) % (n_recs / 2);
            vu4 a = recs[(size_t)idx * 4 + (lane & 3u)];
            acc ^= a;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[gid] = 1;
}

int main() {
    const uint32_t n_max = 32u << 20;  // 32M records x 32 B = 1 GB
    std::vector<uint32_t> h((size_t)n_max * 8);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u);
    vu4 *d;
    uint32_t *out;
    if (hipMalloc(&d, h.size() * 4) != hipSuccess || hipMalloc(&out, 1024 * 256 * 4) != hipSuccess) return 1;
    if (hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
    const uint32_t iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const uint32_t sizes[4] = {256u, 65536u, 2u << 20, 32u << 20};   // 8 KB (L1), 2 MB (L2), 64 MB (MALL), 1 GB (HBM)
    for (int si = 0; si < 4; si++)
    for (int mode = 0; mode < 5; mode++) {
        const uint32_t n_recs = sizes[si];
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(1024), dim3(256), 0, 0, d, n_recs, iters, out);
            if (mode == 1) hipLaunchKernelGGL(k_gather<1>, dim3(1024), dim3(256), 0, 0, d, n_recs, iters, out);
            if (mode == 2) hipLaunchKernelGGL(k_gather<2>, dim3(1024), dim3(256), 0, 0, d, n_recs, iters, out);
            if (mode == 4) hipLaunchKernelGGL(k_gather<4>, dim3(1024), dim3(256), 0, 0, d, n_recs, iters, out);
            if (mode == 3) hipLaunchKernelGGL(k_gather<3>, dim3(1024), dim3(256), 0, 0, d, n_recs, iters, out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            double lane_loads = 1024.0 * 256 * iters * (mode == 0 || mode == 1 || mode == 4 ? 2 : 1);
            if (rep == 1)
                if (mode == 4) printf("          dependent chain: %.0f clk per step (2.4 GHz)\n", ms * 1e-3 * 2.4e9 / iters);
                printf("%8u records, mode %d: %.3f ms, %.2f lane-loads/clk/CU (2.4 GHz, 256 CUs), %.1f GB/s\n", n_recs, mode, ms,
                       lane_loads / (ms * 1e-3) / 2.4e9 / 256, lane_loads * 16 / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
