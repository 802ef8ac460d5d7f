// hwid.hip -- where the waves of a launch land: for every one-wave workgroup the XCD, shader engine, shader array and
// compute unit it runs on (the wave's hardware id registers), to check what a kernel assumes about the placement of
// blockIdx (blockIdx & 7 = XCD) and to see how the compute units of one XCD are numbered.
//   hipcc --offload-arch=gfx950 -O2 -o hwid hwid.hip && ./hwid
#include <hip/hip_runtime.h>

#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void k_ids(uint32_t *out) {
    // s_getreg_b32: simm16 = (size - 1) << 11 | offset << 6 | register id;  HW_ID = 4, XCC_ID = 20
    uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    // keep the wave resident for a moment so that the launch spreads over the chip
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000) {
    }
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
    }
}

int main() {
    const int blocks = 4096;
    uint32_t *d = nullptr;
    if (hipMalloc(&d, blocks * 2 * sizeof(uint32_t)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_ids, dim3(blocks), dim3(64), 0, 0, d);
    std::vector<uint32_t> h(blocks * 2);
    if (hipMemcpy(h.data(), d, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    std::map<uint32_t, std::set<uint32_t>> xcc_of_residue, cus_of_xcc;
    std::map<uint32_t, int> waves_of_cu;
    for (int b = 0; b < blocks; b++) {
        uint32_t hw = h[b * 2], xcc = h[b * 2 + 1] & 0xF;
        uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        xcc_of_residue[b & 7].insert(xcc);
        cus_of_xcc[xcc].insert(se * 100 + sh * 10 * 2 + cu);
        waves_of_cu[xcc * 1000 + se * 100 + sh * 20 + cu]++;
    }
    for (auto &kv : xcc_of_residue) {
        printf("blockIdx & 7 = %u -> XCC_ID", kv.first);
        for (uint32_t x : kv.second) printf(" %u", x);
        printf("\n");
    }
    for (auto &kv : cus_of_xcc) {
        printf("XCC %u: %zu compute units (se*100 + sh*20 + cu):", kv.first, kv.second.size());
        for (uint32_t c : kv.second) printf(" %u", c);
        printf("\n");
    }
    printf("raw HW_ID of block 0: 0x%08x, XCC_ID register: 0x%08x\n", h[0], h[1]);
    (void)hipFree(d);
    return 0;
}
