// rtg_static.hip -- rtg_render_primary_static: the primary-ray entry point for callers that cannot allocate.
//
// rtg_render_primary keeps its device scratch (the persistent kernel's work counters and, for scenes that mix
// meshes with spheres or planes, one closest sphere/plane hit per pixel) in buffers the context allocates on
// demand.  Code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc cannot take that
// path, so this variant keeps the same scratch in a file-scope device array of fixed size.  Each GPU the library
// runs on holds its own copy of the array, at its own address, until the process ends; the address is looked up
// once per device when the first context on that device is created.  Every call on a GPU shares that GPU's array:
// calls that overlap on one GPU (two contexts, two streams) race.
#include <hip/hip_runtime.h>

#include <mutex>

#include "rtg_internal.hpp"
#include "rtg_launch.hpp"

namespace {

constexpr size_t kCounterBytes = 2048;                                      // (8 * 32 + 16) u32, rounded up
constexpr size_t kPrimBestBytes = (size_t)RTG_STATIC_MAX_PIXELS * sizeof(rtg_hit);
constexpr size_t kScratchBytes = kCounterBytes + kPrimBestBytes;            // 32 MiB + 2 KiB
static_assert((8 * 32 + 16) * sizeof(uint32_t) <= kCounterBytes, "work counters outgrew their slice");

static __device__ __align__(256) unsigned char scratch[kScratchBytes];

// In host code `scratch` is only a handle for the symbol, not a device address: the per-device address comes from
// hipGetSymbolAddress with that device current, and it is the only thing host code below ever uses.
constexpr int kMaxDevices = 64;
std::mutex g_lock;
unsigned char *g_scratch_of_device[kMaxDevices];   // zero-initialised: not looked up yet

} // namespace

namespace rtg {

hipError_t static_scratch_init(int device, unsigned char **out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> guard(g_lock);
    if (g_scratch_of_device[device]) {
        *out = g_scratch_of_device[device];
        return hipSuccess;
    }
    int current = -1;
    hipError_t e = hipGetDevice(&current);
    if (e != hipSuccess) return e;
    if (current != device) return hipErrorInvalidDevice;      // the caller binds the device first
    unsigned char *ptr = nullptr;
    e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!ptr) return hipErrorInvalidSymbol;
    g_scratch_of_device[device] = ptr;
    *out = ptr;
    return hipSuccess;
}

} // namespace rtg

extern "C" int rtg_render_primary_static(rtg_scene *scene, const rtg_view *view, uint32_t width, uint32_t height,
                                         rtg_rect tile, rtg_hit *out_hits) try {
    using namespace rtg;
    if (!scene || !scene->ctx) return api_fail(nullptr, RTG_E_INVALID, "null scene");
    rtg_ctx *ctx = scene->ctx;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return api_fail(ctx, RTG_E_HIP, "hipSetDevice(%d): %s", ctx->device, hipGetErrorString(e));
    if (!view) return api_fail(ctx, RTG_E_INVALID, "rtg_render_primary_static: view is null");
    if (width == 0 || height == 0) return api_fail(ctx, RTG_E_INVALID, "rtg_render_primary_static: empty image");
    if ((uint64_t)tile.x0 + tile.w > width || (uint64_t)tile.y0 + tile.h > height)
        return api_fail(ctx, RTG_E_INVALID, "rtg_render_primary_static: tile [%u+%u, %u+%u] outside %ux%u", tile.x0,
                        tile.w, tile.y0, tile.h, width, height);
    const uint64_t n = (uint64_t)tile.w * tile.h;
    if (n == 0) return RTG_OK;
    if (n > RTG_STATIC_MAX_PIXELS)
        return api_fail(ctx, RTG_E_LIMIT,
                        "rtg_render_primary_static: %llu pixels exceed the static scratch's %u (use rtg_render_primary, or "
                        "split the tile)",
                        (unsigned long long)n, (unsigned)RTG_STATIC_MAX_PIXELS);
    if (!out_hits) return api_fail(ctx, RTG_E_INVALID, "rtg_render_primary_static: out_hits is null");
    if (!api_is_device_pointer(out_hits))
        return api_fail(ctx, RTG_E_INVALID,
                        "rtg_render_primary_static: out_hits must be device memory (a host buffer would need a staging "
                        "allocation)");
    {
        hipPointerAttribute_t attr{};          // a query, not an allocation
        if (hipPointerGetAttributes(&attr, out_hits) != hipSuccess) {
            (void)hipGetLastError();
            return api_fail(ctx, RTG_E_INVALID, "rtg_render_primary_static: out_hits is not a known device address");
        }
        if (attr.device != ctx->device)
            return api_fail(ctx, RTG_E_INVALID, "rtg_render_primary_static: out_hits lives on GPU %d, the scene on GPU %d",
                            attr.device, ctx->device);
    }
    if ((uintptr_t)out_hits % 16 != 0)
        return api_fail(ctx, RTG_E_INVALID, "rtg_render_primary_static: out_hits must be 16-byte aligned");
    unsigned char *base = ctx->static_scratch;      // looked up under the lock when the context was made
    if (!base) return api_fail(ctx, RTG_E_HIP, "rtg_render_primary_static: no static scratch on device %d", ctx->device);

    DevScene ds = scene->dev;           // always the persistent kernel: the per-lane traversals may need a stack slab
    ds.traversal = 1;
    ds.overflow = nullptr;
    ds.overflow_levels = 0;
    ds.overflow_stride = 1;
    PhasedScratch ps;
    ps.work_counters = (uint32_t *)base;                    // cleared by the launcher before every launch
    ps.prim_best = (rtg_hit *)(base + kCounterBytes);       // every slot a retiring ray reads is written first
    PhasedTuning tune = ctx->tune;
    tune.time_start = tune.time_stop = nullptr;             // no event pair: creating one allocates
    tune.stats = 0;
    e = launch_render_primary_phased(ctx->stream, ds, *view, width, height, tile, ps, tune, out_hits, nullptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return api_fail(ctx, RTG_E_HIP, "rtg_render_primary_static: launch: %s", hipGetErrorString(e));
    }
    return RTG_OK;
} catch (...) {
    return rtg::api_caught(scene ? scene->ctx : nullptr, "rtg_render_primary_static");
}
