// rtg_internal.hpp -- what the translation units behind include/rtg.h share: the opaque handle types and the
// error / pointer helpers.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <map>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "rtg_launch.hpp"

struct RtgDeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
};

struct rtg_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int traversal = 2;
    rtg::PhasedTuning tune;
    std::string err;
    std::map<std::string, RtgDeviceBuffer> scratch;
    // option "time_kernels": execution windows of the persistent kernel launches behind rtg_render_primary and
    // rtg_trace, summed by rtg_ctx_kernel_time
    bool time_kernels = false;
    bool sort_rays = false;      // option "sort_rays": rtg_trace orders the batch for coherence before tracing it
    // option "comm_loopback": a communicator created on this context is a real RCCL communicator even with one rank,
    // and the root's own pieces travel by ncclSend / ncclRecv to itself (the several-rank machinery on one GPU)
    bool comm_loopback = false;
    // this GPU's copy of the fixed scratch array behind rtg_render_primary_static, looked up when the context is made
    unsigned char *static_scratch = nullptr;
    // option "photon_trig": the photon directions' cosines and sines come from the device's own routines instead of
    // the host's libm (ZenithY::into_normal, src/geo/sphere.rs:29-34)
    bool photon_trig_device = false;
    // option "photon_batch": log2 of the photon directions handled per batch.  A batch keeps 140 bytes per direction in
    // scratch memory (rays, first hits, the packed rays that can make photons) and costs a fixed number of launches
    // and a few waits for the host; the path records are made for at most 2^20 of the packed rays at a time.
    uint32_t photon_batch_log2 = 23;
    // option "kd_finish": the photon map's k-d tree build hands ranges of at most 1024 photons to one workgroup each,
    // which finishes them in LDS in a single launch (same tree as the level-by-level passes; off for comparison).
    bool kd_finish_in_lds = true;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed, idle_events;
    // rtg_render_frame renders a frame of an opaque scene as two row bands on two streams (option "frame_bands"):
    // `side` is an internal context (own stream and scratch) for the second band, joined back by events
    uint32_t frame_bands = 2;
    rtg_ctx *side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

struct rtg_mesh {
    int device = 0;                  // a mesh may outlive the context that made it: only the GPU is remembered
    rtg::DevMesh dev{};
    void *d_nodes = nullptr, *d_slab = nullptr, *d_normals = nullptr;
    void *d_patches = nullptr, *d_leaf_info = nullptr;
    void *d_leaf_source = nullptr;   // built on the device: the caller's triangle index of every leaf
    bool is_patches = false;
    double bounds[6] = {0, 0, 0, 0, 0, 0};
    uint64_t n_leaves = 0;
};

struct rtg_scene {
    rtg_ctx *ctx = nullptr;          // calls on the scene run on this context: it must outlive the scene's last call
    int device = 0;                  // rtg_scene_destroy only needs the GPU
    rtg::DevScene dev{};
    void *d_objects = nullptr, *d_mesh_objects = nullptr;
    uint32_t max_depth = 0;
    bool any_dielectric = false;   // some material refracts: frames need the specular path tree
    bool has_mesh = false;         // some object is a mesh or patch set: shading is worth doing as passes of the
                                   // persistent kernel (sphere/plane-only scenes are cheaper in one fused launch)
    std::vector<uint64_t> object_leaves;   // per object: triangles of its mesh (0 for spheres and planes)
};


namespace rtg {
// Records `code`'s message on the context (or, without one, for rtg_last_error(NULL) on this thread); returns code.
int api_fail(rtg_ctx *ctx, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
// For the handler of an entry point's function-try-block: turns the exception in flight into a code and a message
// (std::bad_alloc -> RTG_E_NOMEM).  Nothing may unwind through the C ABI: the host on the other side is not C++.
int api_caught(rtg_ctx *ctx, const char *function) noexcept;
// Runs f(begin, end) over [0, n) split into up to 16 ranges, each on its own host thread when threads can be had
// (inline otherwise); returns after all ranges are done.  f must not throw.
template <class F> void api_parallel_ranges(size_t n, size_t min_per_thread, F f);
// True when `p` can be dereferenced by a kernel on the current device.
bool api_is_device_pointer(const void *p);
// A named device buffer of the context, grown on demand and kept for the next call.
int api_scratch(rtg_ctx *ctx, const char *name, size_t bytes, void **out);
// Device view of an input: a device pointer as it is, host memory uploaded into the named scratch buffer.
int api_input(rtg_ctx *ctx, const char *name, const void *user, size_t bytes, const void **dev);
// Makes the scene's GPU current; RTG_E_INVALID for a null scene.
int api_bind_scene(rtg_scene *scene);
// Closest hits of `n` device-resident rays into device memory, through the context's traversal (what rtg_trace
// does between its input and output staging); asynchronous on the context's stream.
int api_trace_device(rtg_scene *scene, const double *d_rays6, uint64_t n, rtg_hit *d_hits);
// (indirect + direct + ambient) x diffuse for device-resident points (what rtg_shade_direct computes, with the photon
// map's term `d_indirect`, 3 doubles per point or null, added first); asynchronous on the context's stream.
int api_shade_points(rtg_scene *scene, const rtg_light *d_lights, uint32_t n_lights, const rtg_point *d_points, uint64_t n,
                     const double *d_indirect, double *d_rgb);

template <class F> void api_parallel_ranges(size_t n, size_t min_per_thread, F f) {
    unsigned t = std::thread::hardware_concurrency();
    if (t > 16) t = 16;
    if (min_per_thread && n / min_per_thread < t) t = (unsigned)(n / min_per_thread);
    if (t < 2) {
        f((size_t)0, n);
        return;
    }
    std::vector<std::thread> pool;
    size_t done = 0;                        // ranges [0, done) have a thread
    try {
        pool.reserve(t - 1);
        for (unsigned w = 0; w + 1 < t; w++) {
            const size_t b = n / t * w + std::min<size_t>(w, n % t), e = n / t * (w + 1) + std::min<size_t>(w + 1, n % t);
            pool.emplace_back([=]() { f(b, e); });
            done = e;
        }
    } catch (...) {
        // no more threads to be had: the rest runs here
    }
    f(done, n);
    for (auto &th : pool) th.join();
}
} // namespace rtg
