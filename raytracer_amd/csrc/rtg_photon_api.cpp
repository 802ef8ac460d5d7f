// rtg_photon_api.cpp -- the photon-map entry points of include/rtg.h: the photon pass, the k-d tree, nearest
// photons and the caustic term of a surface point (src/render/renderer.rs:116-138, 206-265; src/tree/kd_tree.rs).
//
// Host-side duties only.  The photon pass turns the caller's emission directions into rays with the host's libm
// (as the reference does) and hands them to rtg_trace_paths in batches; the k-d tree is built on the device
// (rtg_kdtree.hip) from the records the caller hands over, host or device memory.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <limits>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "rtg_internal.hpp"
#include "rtg_launch.hpp"

using namespace rtg;

struct rtg_photon_map {
    rtg_ctx *ctx = nullptr;          // calls on the map run on this context: it must outlive the map's last call
    int device = 0;
    void *d_nodes = nullptr;         // k-d tree, 80 bytes per photon
    void *d_photons = nullptr;       // the caller's records in the caller's order
    uint64_t n = 0;
    uint32_t max_bounces = 0;        // the longest photon path: levels the corrected photons need to be traced for
};

namespace {

#define HIP_TRY(ctx, expr)                                                                            \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            (void)hipGetLastError();                                                                  \
            return api_fail(ctx, e_ == hipErrorOutOfMemory ? RTG_E_NOMEM : RTG_E_HIP, "%s: %s", #expr, \
                            hipGetErrorString(e_));                                                   \
        }                                                                                             \
    } while (0)

// sin and cos as two separate libm calls, as rustc emits them for `ZenithY::into_normal`; a compiler that merges
// the pair into sincos() changes the last place of about one direction in a thousand (glibc).
double (*volatile libm_sin)(double) = ::sin;
double (*volatile libm_cos)(double) = ::cos;

int check_map_call(rtg_photon_map *map) {
    if (!map || !map->ctx) return api_fail(nullptr, RTG_E_INVALID, "null photon map");
    hipError_t e = hipSetDevice(map->device);
    if (e != hipSuccess) return api_fail(map->ctx, RTG_E_HIP, "hipSetDevice(%d): %s", map->device, hipGetErrorString(e));
    return RTG_OK;
}

// copies `bytes` from device scratch to the caller's buffer (host or device) on the context's stream
int copy_out(rtg_ctx *ctx, void *user, const void *dev, size_t bytes) {
    if (bytes == 0) return RTG_OK;
    HIP_TRY(ctx, hipMemcpyAsync(user, dev, bytes, api_is_device_pointer(user) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                                ctx->stream));
    return RTG_OK;
}

int check_irrad_args(rtg_scene *scene, rtg_photon_map *map, uint32_t photon_samples) {
    rtg_ctx *ctx = scene->ctx;
    if (!map) return api_fail(ctx, RTG_E_INVALID, "photon map is null");
    if (map->device != scene->device) return api_fail(ctx, RTG_E_INVALID, "the photon map lives on another GPU");
    if (photon_samples > 16) return api_fail(ctx, RTG_E_LIMIT, "photon_samples is at most 16");
    if (scene->dev.has_patches) return api_fail(ctx, RTG_E_UNSUPPORTED, "the photon term does not support scenes with patch sets");
    return RTG_OK;
}

// compute_indirect_irrad for device-resident points into device memory (3 doubles per point); asynchronous on the
// scene's stream.
int indirect_device(rtg_scene *scene, rtg_photon_map *map, const rtg_point *d_points, uint64_t n_points, uint32_t photon_samples,
                    uint32_t newton_steps, double newton_epsilon, double *d_rgb) {
    rtg_ctx *ctx = scene->ctx;
    int rc;
    if (n_points == 0) return RTG_OK;
    const uint64_t n_pairs = n_points * photon_samples;
    if (n_pairs > 0xFFFFFFFFull) return api_fail(ctx, RTG_E_LIMIT, "more than 2^32-1 (point, photon sample) pairs in one call");
    if (photon_samples == 0 || map->n == 0) {
        HIP_TRY(ctx, hipMemsetAsync(d_rgb, 0, (size_t)n_points * 3 * sizeof(double), ctx->stream));
        return RTG_OK;
    }
    void *queries = nullptr, *nearest = nullptr, *dist = nullptr, *dir2 = nullptr, *state = nullptr, *rays = nullptr, *hits = nullptr,
         *status = nullptr, *landing = nullptr, *leaves = nullptr;
    if ((rc = api_scratch(ctx, "irrad_queries", (size_t)n_points * 3 * sizeof(double), &queries)) != RTG_OK) return rc;
    if ((rc = api_scratch(ctx, "irrad_nearest", (size_t)n_pairs * sizeof(uint32_t), &nearest)) != RTG_OK) return rc;
    if ((rc = api_scratch(ctx, "irrad_distance", (size_t)n_pairs * sizeof(double), &dist)) != RTG_OK) return rc;
    if ((rc = api_scratch(ctx, "irrad_dir", (size_t)n_pairs * 2 * sizeof(double), &dir2)) != RTG_OK) return rc;
    if ((rc = api_scratch(ctx, "irrad_state", (size_t)n_pairs * 18 * sizeof(double), &state)) != RTG_OK) return rc;
    if ((rc = api_scratch(ctx, "irrad_rays", (size_t)n_pairs * 6 * sizeof(double), &rays)) != RTG_OK) return rc;
    if ((rc = api_scratch(ctx, "irrad_hits", (size_t)n_pairs * sizeof(rtg_hit), &hits)) != RTG_OK) return rc;
    if ((rc = api_scratch(ctx, "irrad_status", (size_t)n_pairs * sizeof(uint32_t), &status)) != RTG_OK) return rc;
    if ((rc = api_scratch(ctx, "irrad_landing", (size_t)n_pairs * 9 * sizeof(double), &landing)) != RTG_OK) return rc;
    if ((rc = api_scratch(ctx, "irrad_leaves", std::max<size_t>(1, scene->object_leaves.size()) * sizeof(uint64_t), &leaves)) != RTG_OK)
        return rc;
    if (!scene->object_leaves.empty())
        HIP_TRY(ctx, hipMemcpyAsync(leaves, scene->object_leaves.data(), scene->object_leaves.size() * sizeof(uint64_t),
                                    hipMemcpyHostToDevice, ctx->stream));
    // the points' positions are the queries: a strided copy out of the point records
    HIP_TRY(ctx, hipMemcpy2DAsync(queries, 3 * sizeof(double), (const char *)d_points + offsetof(rtg_point, position),
                                  sizeof(rtg_point), 3 * sizeof(double), (size_t)n_points, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, launch_kd_nearest(ctx->stream, map->d_nodes, (uint32_t)map->n, (const double *)queries, n_points, photon_samples,
                                   (uint32_t *)nearest, (double *)dist));
    PhotonPairs pr;
    pr.objects = scene->dev.objects;
    pr.n_objects = scene->dev.n_objects;
    pr.object_leaves = (const uint64_t *)leaves;
    pr.photons = (const rtg_photon *)map->d_photons;
    pr.n_photons = (uint32_t)map->n;
    pr.nearest = (const uint32_t *)nearest;
    pr.points = d_points;
    pr.n_points = n_points;
    pr.samples = photon_samples;
    HIP_TRY(ctx, launch_photon_newton(ctx->stream, pr, newton_steps, (double *)dir2));
    HIP_TRY(ctx, launch_photon_final_start(ctx->stream, pr, (const double *)dir2, (double *)state, (double *)rays, (uint32_t *)status));
    for (uint32_t level = 0; level <= map->max_bounces; level++) {   // a photon lands after its last bounce
        if ((rc = api_trace_device(scene, (const double *)rays, n_pairs, (rtg_hit *)hits)) != RTG_OK) return rc;
        HIP_TRY(ctx, launch_photon_final_bounce(ctx->stream, pr, level, (const rtg_hit *)hits, (double *)state, (double *)rays,
                                                (uint32_t *)status, (double *)landing));
    }
    HIP_TRY(ctx, launch_photon_sum(ctx->stream, pr, (const uint32_t *)status, (const double *)landing, newton_epsilon, d_rgb));
    return RTG_OK;
}

// Where the photons of a pass go: the caller's buffer (host or device) of `capacity` records, or a device buffer
// held here that grows as the batches arrive.
struct PhotonSink {
    rtg_photon *user = nullptr;
    uint64_t capacity = 0;
    bool grow = false;
    void *held = nullptr;
    uint64_t held_capacity = 0;
    ~PhotonSink() {
        if (held) (void)hipFree(held);
    }
};

int check_emit_args(rtg_ctx *ctx, const rtg_light *lights, uint32_t n_lights, const double *dirs2, uint64_t count_per_light,
                    uint32_t max_depth) {
    if (max_depth < 1 || max_depth > 4) return api_fail(ctx, RTG_E_INVALID, "photon pass: max_depth must be 1..4");
    if ((uint64_t)n_lights * count_per_light == 0) return RTG_OK;
    if (!lights || !dirs2) return api_fail(ctx, RTG_E_INVALID, "photon pass: lights / dirs2 is null");
    if (api_is_device_pointer(lights)) return api_fail(ctx, RTG_E_INVALID, "photon pass: lights is host memory");
    if (api_is_device_pointer(dirs2) && !ctx->photon_trig_device)
        return api_fail(ctx, RTG_E_INVALID, "photon pass: dirs2 is host memory (the directions' sin / cos are taken on the host; "
                                            "option photon_trig = 2 takes them on the device and accepts device memory)");
    if (count_per_light > (1ull << 40) || (uint64_t)n_lights * count_per_light > (1ull << 40))
        return api_fail(ctx, RTG_E_LIMIT, "photon pass: too many directions");
    return RTG_OK;
}

// renderer.rs:116-138 for checked arguments; *n_total receives the number of photons the pass makes.
// `fibonacci_seed` (dirs2 == nullptr): the directions are Sphere::fibonacci_sphere's, made on the device batch by batch.
int emit_photons(rtg_scene *scene, const rtg_light *lights, uint32_t n_lights, const double *dirs2, uint64_t count_per_light,
                 uint32_t max_depth, PhotonSink &sink, uint64_t *n_total, const uint64_t *fibonacci_seed = nullptr) {
    rtg_ctx *ctx = scene->ctx;
    int rc;
    *n_total = 0;
    const uint64_t n_rays = (uint64_t)n_lights * count_per_light;
    if (n_rays == 0) return RTG_OK;
    const void *d_lights = nullptr;
    if ((rc = api_input(ctx, "photon_lights", lights, (size_t)n_lights * sizeof(rtg_light), &d_lights)) != RTG_OK) return rc;
    const uint64_t per_ray = (1ull << max_depth) - 1;        // paths a ray can have
    // ZenithY::into_normal (src/geo/sphere.rs:29-34): the cosine and the sine of every zenith come from the host's
    // libm, as two calls, spread over the host's cores; they go to the device together with the direction itself
    // (32 bytes per direction through pinned memory, against 64 for finished rays and directions from pageable
    // memory, which made the pass wait for PCIe); a kernel finishes the rays.  The next batch is prepared while the
    // GPU traces this one.
    struct Pinned {
        double *p = nullptr;
        ~Pinned() {
            if (p) (void)hipHostFree(p);
        }
    } staging[2];
    // With the option "photon_trig" on the device, only zenith and y travel (16 bytes per direction, staged through the
    // same pinned buffers; nothing at all when the caller's directions are device memory) and the kernel that finishes
    // the rays takes cosine and sine itself.
    const bool trig_device = ctx->photon_trig_device || fibonacci_seed != nullptr;
    const bool dirs_on_device = fibonacci_seed != nullptr || (trig_device && api_is_device_pointer(dirs2));
    // directions per batch (option "photon_batch"); directions that come from the host are staged batch by batch while
    // the GPU traces the batch before, which wants more and smaller batches
    const uint64_t batch = std::min<uint64_t>(1ull << ctx->photon_batch_log2, dirs_on_device ? ~0ull : 1ull << 20);
    const uint64_t paths_batch = std::min<uint64_t>(batch, 1ull << 20);   // rays per rtg_trace_paths call (2.3 GB of path records at depth 4)
    const size_t per_dir = trig_device ? 2 : 4;
    const size_t batch_doubles = (size_t)std::min(batch, n_rays) * per_dir;
    if (!dirs_on_device) {
        HIP_TRY(ctx, hipHostMalloc((void **)&staging[0].p, batch_doubles * sizeof(double), hipHostMallocDefault));
        if (n_rays > batch) HIP_TRY(ctx, hipHostMalloc((void **)&staging[1].p, batch_doubles * sizeof(double), hipHostMallocDefault));
    }
    auto make_batch = [&](uint64_t first, double *dst) {
        if (dirs_on_device) return;
        if (trig_device) {
            api_parallel_ranges((size_t)std::min(batch, n_rays - first), 1 << 16, [=](size_t begin, size_t end) {
                std::memcpy(dst + begin * 2, dirs2 + (first + begin) * 2, (end - begin) * 2 * sizeof(double));
            });
            return;
        }
        api_parallel_ranges((size_t)std::min(batch, n_rays - first), 4096, [=](size_t begin, size_t end) {
            for (size_t i = begin; i < end; i++) {
                const double zenith = dirs2[(first + i) * 2], y = dirs2[(first + i) * 2 + 1];
                double *w = dst + i * 4;
                w[0] = libm_cos(zenith);
                w[1] = libm_sin(zenith);
                w[2] = zenith;
                w[3] = y;
            }
        });
    };
    struct Joined {
        std::thread thread;
        ~Joined() {
            if (thread.joinable()) thread.join();
        }
    };
    make_batch(0, staging[0].p);
    uint64_t total = 0;
    for (uint64_t first = 0, b = 0; first < n_rays; first += batch, b++) {
        const uint64_t m = std::min(batch, n_rays - first);
        const double *staged = staging[b & 1].p;
        Joined ahead;                                        // joined at the end of the iteration, however it ends
        if (first + batch < n_rays && !dirs_on_device) {
            double *dst = staging[(b + 1) & 1].p;
            const uint64_t next_first = first + batch;
            try {
                ahead.thread = std::thread([&make_batch, next_first, dst]() { make_batch(next_first, dst); });
            } catch (const std::system_error &) {
                make_batch(next_first, dst);
            }
        }
        void *d_staged = nullptr, *d_rays = nullptr, *d_dirs = nullptr;
        if (!dirs_on_device && (rc = api_scratch(ctx, "photon_staged", (size_t)m * per_dir * sizeof(double), &d_staged)) != RTG_OK) return rc;
        if ((rc = api_scratch(ctx, "photon_rays", (size_t)m * 6 * sizeof(double), &d_rays)) != RTG_OK) return rc;
        if ((rc = api_scratch(ctx, "photon_dirs", (size_t)m * 2 * sizeof(double), &d_dirs)) != RTG_OK) return rc;
        if (!dirs_on_device)
            HIP_TRY(ctx, hipMemcpyAsync(d_staged, staged, (size_t)m * per_dir * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        if (fibonacci_seed)
            HIP_TRY(ctx, launch_photon_rays_fibonacci(ctx->stream, (uint32_t)m, (const rtg_light *)d_lights, first, count_per_light,
                                                      *fibonacci_seed, (double *)d_rays, (double *)d_dirs));
        else if (trig_device)
            HIP_TRY(ctx, launch_photon_rays_trig(ctx->stream, dirs_on_device ? dirs2 + first * 2 : (const double *)d_staged, (uint32_t)m,
                                                 (const rtg_light *)d_lights, first, count_per_light, (double *)d_rays, (double *)d_dirs));
        else
            HIP_TRY(ctx, launch_photon_rays(ctx->stream, (const double *)d_staged, (uint32_t)m, (const rtg_light *)d_lights, first,
                                            count_per_light, (double *)d_rays, (double *)d_dirs));
        // The rays that can make photons at all: first hits of the whole batch (hit records only), then the rays whose
        // hit lies on a refracting object, packed in their order.  In the reference's scenes that is a few percent of
        // the directions, and only they go through the path records.
        void *d_hits = nullptr, *first_flags = nullptr, *first_offsets = nullptr, *first_temp = nullptr, *d_kept = nullptr,
             *d_index = nullptr;
        const size_t first_temp_bytes = photon_scan_temp_bytes((uint32_t)m);
        if ((rc = api_scratch(ctx, "photon_first_hits", (size_t)m * sizeof(rtg_hit), &d_hits)) != RTG_OK) return rc;
        if ((rc = api_scratch(ctx, "photon_first_flags", (size_t)m * sizeof(uint32_t), &first_flags)) != RTG_OK) return rc;
        if ((rc = api_scratch(ctx, "photon_first_offsets", (size_t)m * sizeof(uint32_t), &first_offsets)) != RTG_OK) return rc;
        if ((rc = api_scratch(ctx, "photon_first_temp", first_temp_bytes, &first_temp)) != RTG_OK) return rc;
        if ((rc = api_scratch(ctx, "photon_kept_rays", (size_t)m * 6 * sizeof(double), &d_kept)) != RTG_OK) return rc;
        if ((rc = api_scratch(ctx, "photon_kept_index", (size_t)m * sizeof(uint32_t), &d_index)) != RTG_OK) return rc;
        if ((rc = api_trace_device(scene, (const double *)d_rays, m, (rtg_hit *)d_hits)) != RTG_OK) return rc;
        HIP_TRY(ctx, launch_photon_first_filter(ctx->stream, scene->dev, (const rtg_hit *)d_hits, (const double *)d_rays, (uint32_t)m,
                                                (uint32_t *)first_flags, (uint32_t *)first_offsets, first_temp, first_temp_bytes,
                                                (double *)d_kept, (uint32_t *)d_index));
        uint32_t last[2] = {0, 0};                           // offsets[n - 1], flags[n - 1]
        HIP_TRY(ctx, hipMemcpyAsync(&last[0], (const uint32_t *)first_offsets + (m - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&last[1], (const uint32_t *)first_flags + (m - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const uint64_t kept = (uint64_t)last[0] + last[1];
        for (uint64_t part = 0; part < kept; part += paths_batch) {
            const uint64_t k = std::min(paths_batch, kept - part);
            void *d_paths = nullptr;
            if ((rc = api_scratch(ctx, "photon_paths", (size_t)(k * per_ray) * sizeof(rtg_path), &d_paths)) != RTG_OK) return rc;
            uint64_t n_paths = 0;
            if ((rc = rtg_trace_paths(scene, (const double *)d_kept + part * 6, k, max_depth, (rtg_path *)d_paths, k * per_ray, &n_paths)) != RTG_OK)
                return rc;
            if (n_paths == 0) continue;
            void *flags = nullptr, *offsets = nullptr, *temp = nullptr, *d_out = nullptr;
            const size_t temp_bytes = photon_scan_temp_bytes((uint32_t)n_paths);
            if ((rc = api_scratch(ctx, "photon_flags", (size_t)n_paths * sizeof(uint32_t), &flags)) != RTG_OK) return rc;
            if ((rc = api_scratch(ctx, "photon_offsets", (size_t)n_paths * sizeof(uint32_t), &offsets)) != RTG_OK) return rc;
            if ((rc = api_scratch(ctx, "photon_scan_temp", temp_bytes, &temp)) != RTG_OK) return rc;
            // a batch writes only what still fits the caller's room
            const uint64_t room = sink.grow ? n_paths : (sink.capacity > total ? std::min<uint64_t>(sink.capacity - total, n_paths) : 0);
            if ((rc = api_scratch(ctx, "photon_out", (size_t)std::max<uint64_t>(room, 1) * sizeof(rtg_photon), &d_out)) != RTG_OK) return rc;
            HIP_TRY(ctx, launch_photons_from_paths(ctx->stream, (const rtg_path *)d_paths, (uint32_t)n_paths, (const double *)d_dirs,
                                                   (const uint32_t *)d_index + part, (const rtg_light *)d_lights, first, count_per_light,
                                                   (uint32_t *)flags, (uint32_t *)offsets, temp, temp_bytes, room, (rtg_photon *)d_out));
            HIP_TRY(ctx, hipMemcpyAsync(&last[0], (const uint32_t *)offsets + (n_paths - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(&last[1], (const uint32_t *)flags + (n_paths - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            const uint64_t made = (uint64_t)last[0] + last[1];
            if (sink.grow) {
                if (total + made > sink.held_capacity) {
                    const uint64_t want = std::max<uint64_t>({total + made, sink.held_capacity * 2, (uint64_t)1 << 16});
                    void *bigger = nullptr;
                    HIP_TRY(ctx, hipMalloc(&bigger, (size_t)want * sizeof(rtg_photon)));
                    hipError_t e = total ? hipMemcpyAsync(bigger, sink.held, (size_t)total * sizeof(rtg_photon), hipMemcpyDeviceToDevice, ctx->stream)
                                         : hipSuccess;
                    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                    if (e != hipSuccess) {
                        (void)hipFree(bigger);
                        (void)hipGetLastError();
                        return api_fail(ctx, RTG_E_HIP, "photon pass: %s", hipGetErrorString(e));
                    }
                    if (sink.held) (void)hipFree(sink.held);
                    sink.held = bigger;
                    sink.held_capacity = want;
                }
                HIP_TRY(ctx, hipMemcpyAsync((rtg_photon *)sink.held + total, d_out, (size_t)made * sizeof(rtg_photon), hipMemcpyDeviceToDevice,
                                            ctx->stream));
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the scratch is reused by the next batch
            } else if (total + made <= sink.capacity) {
                if ((rc = copy_out(ctx, sink.user + total, d_out, (size_t)made * sizeof(rtg_photon))) != RTG_OK) return rc;
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the scratch is reused by the next batch
            }
            total += made;
        }
    }
    *n_total = total;
    return RTG_OK;
}

// KdTree::new over `n` photon records already on the context's GPU; the records (hipMalloc'ed) belong to the map
// from here on, or are freed when the call fails.
int map_over(rtg_ctx *ctx, void *d_photons, uint64_t n_photons, rtg_photon_map **out) {
    rtg_photon_map *m = new (std::nothrow) rtg_photon_map();
    if (!m) {
        if (d_photons) (void)hipFree(d_photons);
        return api_fail(ctx, RTG_E_NOMEM, "out of host memory");
    }
    m->ctx = ctx;
    m->device = ctx->device;
    m->n = n_photons;
    m->d_photons = d_photons;
    if (n_photons == 0) {
        *out = m;
        return RTG_OK;
    }
    const uint32_t n = (uint32_t)n_photons;
    void *scratch = nullptr;
    int rc = api_scratch(ctx, "kd_build", kd_build_scratch_bytes(n), &scratch);
    if (rc != RTG_OK) {
        rtg_photon_map_destroy(m);
        return rc;
    }
    uint32_t *d_summary = nullptr;
    uint32_t summary[2] = {0, 0xFFFFFFFFu};
    hipError_t e = hipMalloc(&m->d_nodes, (size_t)n * sizeof(KdNode));
    if (e == hipSuccess) e = build_kd_device(ctx->stream, (const rtg_photon *)m->d_photons, n, scratch, m->d_nodes, &d_summary, ctx->kd_finish_in_lds);
    if (e == hipSuccess) e = hipMemcpyAsync(summary, d_summary, sizeof(summary), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rtg_photon_map_destroy(m);
        return api_fail(ctx, e == hipErrorOutOfMemory ? RTG_E_NOMEM : RTG_E_HIP, "photon map: %s", hipGetErrorString(e));
    }
    if (summary[1] != 0xFFFFFFFFu) {
        // the reference refuses NaN keys (NotNan::try_from(..).unwrap(), src/tree/kd_tree.rs:81-82)
        rtg_photon_map_destroy(m);
        return api_fail(ctx, RTG_E_INVALID, "photon map: photon %u has a NaN position", summary[1]);
    }
    m->max_bounces = summary[0];
    *out = m;
    return RTG_OK;
}

} // namespace

extern "C" {

int rtg_photons_emit(rtg_scene *scene, const rtg_light *lights, uint32_t n_lights, const double *dirs2,
                     uint64_t count_per_light, uint32_t max_depth, rtg_photon *out_photons, uint64_t capacity,
                     uint64_t *n_photons) try {
    int rc = api_bind_scene(scene);
    if (rc != RTG_OK) return rc;
    rtg_ctx *ctx = scene->ctx;
    if (!n_photons) return api_fail(ctx, RTG_E_INVALID, "rtg_photons_emit: n_photons is null");
    *n_photons = 0;
    if ((rc = check_emit_args(ctx, lights, n_lights, dirs2, count_per_light, max_depth)) != RTG_OK) return rc;
    if (capacity && !out_photons) return api_fail(ctx, RTG_E_INVALID, "rtg_photons_emit: out_photons is null");
    PhotonSink sink;
    sink.user = out_photons;
    sink.capacity = capacity;
    uint64_t total = 0;
    if ((rc = emit_photons(scene, lights, n_lights, dirs2, count_per_light, max_depth, sink, &total)) != RTG_OK) return rc;
    *n_photons = total;
    if (total > capacity)
        return api_fail(ctx, RTG_E_LIMIT, "rtg_photons_emit: %llu photons, room for %llu", (unsigned long long)total,
                        (unsigned long long)capacity);
    return RTG_OK;
} catch (...) {
    return api_caught(scene ? scene->ctx : nullptr, "rtg_photons_emit");
}

int rtg_photon_map_create(rtg_ctx *ctx, const rtg_photon *photons, uint64_t n_photons, rtg_photon_map **out) try {
    if (!ctx) return api_fail(nullptr, RTG_E_INVALID, "null ctx");
    if (!out) return api_fail(ctx, RTG_E_INVALID, "rtg_photon_map_create: out is null");
    *out = nullptr;
    if (n_photons && !photons) return api_fail(ctx, RTG_E_INVALID, "rtg_photon_map_create: photons is null");
    if (n_photons >= (1ull << 31)) return api_fail(ctx, RTG_E_LIMIT, "rtg_photon_map_create: at most 2^31 - 1 photons");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void *d_photons = nullptr;
    if (n_photons) {
        HIP_TRY(ctx, hipMalloc(&d_photons, (size_t)n_photons * sizeof(rtg_photon)));
        hipError_t e = hipMemcpyAsync(d_photons, photons, (size_t)n_photons * sizeof(rtg_photon),
                                      api_is_device_pointer(photons) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            (void)hipFree(d_photons);
            (void)hipGetLastError();
            return api_fail(ctx, RTG_E_HIP, "rtg_photon_map_create: %s", hipGetErrorString(e));
        }
    }
    return map_over(ctx, d_photons, n_photons, out);
} catch (...) {
    return api_caught(ctx, "rtg_photon_map_create");
}

int rtg_photon_map_emit(rtg_scene *scene, const rtg_light *lights, uint32_t n_lights, const double *dirs2, uint64_t count_per_light,
                        uint32_t max_depth, rtg_photon_map **out, uint64_t *n_photons) try {
    int rc = api_bind_scene(scene);
    if (rc != RTG_OK) return rc;
    rtg_ctx *ctx = scene->ctx;
    if (!out) return api_fail(ctx, RTG_E_INVALID, "rtg_photon_map_emit: out is null");
    *out = nullptr;
    if (n_photons) *n_photons = 0;
    if ((rc = check_emit_args(ctx, lights, n_lights, dirs2, count_per_light, max_depth)) != RTG_OK) return rc;
    PhotonSink sink;
    sink.grow = true;
    uint64_t total = 0;
    if ((rc = emit_photons(scene, lights, n_lights, dirs2, count_per_light, max_depth, sink, &total)) != RTG_OK) return rc;
    if (total >= (1ull << 31)) return api_fail(ctx, RTG_E_LIMIT, "rtg_photon_map_emit: %llu photons, at most 2^31 - 1", (unsigned long long)total);
    void *d_photons = total ? sink.held : nullptr;       // map_over owns the records from here on
    if (total) sink.held = nullptr;
    if ((rc = map_over(ctx, d_photons, total, out)) != RTG_OK) return rc;
    if (n_photons) *n_photons = total;
    return RTG_OK;
} catch (...) {
    return api_caught(scene ? scene->ctx : nullptr, "rtg_photon_map_emit");
}

int rtg_photon_map_emit_fibonacci(rtg_scene *scene, const rtg_light *lights, uint32_t n_lights, uint64_t count_per_light,
                                  uint64_t seed, uint32_t max_depth, rtg_photon_map **out, uint64_t *n_photons) try {
    int rc = api_bind_scene(scene);
    if (rc != RTG_OK) return rc;
    rtg_ctx *ctx = scene->ctx;
    if (!out) return api_fail(ctx, RTG_E_INVALID, "rtg_photon_map_emit_fibonacci: out is null");
    *out = nullptr;
    if (n_photons) *n_photons = 0;
    if (max_depth < 1 || max_depth > 4) return api_fail(ctx, RTG_E_INVALID, "photon pass: max_depth must be 1..4");
    if ((uint64_t)n_lights * count_per_light != 0) {
        if (!lights || api_is_device_pointer(lights)) return api_fail(ctx, RTG_E_INVALID, "photon pass: lights is null or not host memory");
        if (count_per_light > (1ull << 40) || (uint64_t)n_lights * count_per_light > (1ull << 40))
            return api_fail(ctx, RTG_E_LIMIT, "photon pass: too many directions");
    }
    PhotonSink sink;
    sink.grow = true;
    uint64_t total = 0;
    if ((rc = emit_photons(scene, lights, n_lights, nullptr, count_per_light, max_depth, sink, &total, &seed)) != RTG_OK) return rc;
    if (total >= (1ull << 31))
        return api_fail(ctx, RTG_E_LIMIT, "rtg_photon_map_emit_fibonacci: %llu photons, at most 2^31 - 1", (unsigned long long)total);
    void *d_photons = total ? sink.held : nullptr;       // map_over owns the records from here on
    if (total) sink.held = nullptr;
    if ((rc = map_over(ctx, d_photons, total, out)) != RTG_OK) return rc;
    if (n_photons) *n_photons = total;
    return RTG_OK;
} catch (...) {
    return api_caught(scene ? scene->ctx : nullptr, "rtg_photon_map_emit_fibonacci");
}

void rtg_photon_map_destroy(rtg_photon_map *map) {
    if (!map) return;
    (void)hipSetDevice(map->device);
    if (map->d_nodes) (void)hipFree(map->d_nodes);
    if (map->d_photons) (void)hipFree(map->d_photons);
    delete map;
}

int rtg_photon_map_nearest(rtg_photon_map *map, const double *queries3, uint64_t n_queries, uint32_t k, uint32_t *out_index,
                           double *out_distance) try {
    int rc = check_map_call(map);
    if (rc != RTG_OK) return rc;
    rtg_ctx *ctx = map->ctx;
    if (k > 16) return api_fail(ctx, RTG_E_LIMIT, "rtg_photon_map_nearest: k is at most 16");
    if (n_queries == 0 || k == 0) return RTG_OK;
    if (!queries3) return api_fail(ctx, RTG_E_INVALID, "rtg_photon_map_nearest: queries3 is null");
    if (!out_index && !out_distance) return api_fail(ctx, RTG_E_INVALID, "rtg_photon_map_nearest: no output requested");
    if (n_queries > 0xFFFFFFFFull / 16) return api_fail(ctx, RTG_E_LIMIT, "rtg_photon_map_nearest: too many queries for one call");
    const void *d_q = nullptr;
    if ((rc = api_input(ctx, "kd_queries", queries3, (size_t)n_queries * 3 * sizeof(double), &d_q)) != RTG_OK) return rc;
    const size_t slots = (size_t)n_queries * k;
    void *d_i = nullptr, *d_d = nullptr;
    const bool i_dev = out_index && api_is_device_pointer(out_index), d_dev = out_distance && api_is_device_pointer(out_distance);
    if (i_dev)
        d_i = out_index;
    else if ((rc = api_scratch(ctx, "kd_index", slots * sizeof(uint32_t), &d_i)) != RTG_OK)
        return rc;
    if (d_dev)
        d_d = out_distance;
    else if ((rc = api_scratch(ctx, "kd_distance", slots * sizeof(double), &d_d)) != RTG_OK)
        return rc;
    HIP_TRY(ctx, launch_kd_nearest(ctx->stream, map->d_nodes, (uint32_t)map->n, (const double *)d_q, n_queries, k, (uint32_t *)d_i,
                                   (double *)d_d));
    bool staged = false;
    if (out_index && !i_dev) {
        HIP_TRY(ctx, hipMemcpyAsync(out_index, d_i, slots * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        staged = true;
    }
    if (out_distance && !d_dev) {
        HIP_TRY(ctx, hipMemcpyAsync(out_distance, d_d, slots * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        staged = true;
    }
    if (staged) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RTG_OK;
} catch (...) {
    return api_caught(map ? map->ctx : nullptr, "rtg_photon_map_nearest");
}

int rtg_photon_map_order(rtg_photon_map *map, uint32_t *out_index) try {
    int rc = check_map_call(map);
    if (rc != RTG_OK) return rc;
    rtg_ctx *ctx = map->ctx;
    if (map->n == 0) return RTG_OK;
    if (!out_index) return api_fail(ctx, RTG_E_INVALID, "rtg_photon_map_order: out_index is null");
    const bool on_device = api_is_device_pointer(out_index);
    HIP_TRY(ctx, hipMemcpy2DAsync(out_index, sizeof(uint32_t), (const char *)map->d_nodes + offsetof(KdNode, value), sizeof(KdNode),
                                  sizeof(uint32_t), (size_t)map->n, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                                  ctx->stream));
    if (!on_device) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RTG_OK;
} catch (...) {
    return api_caught(map ? map->ctx : nullptr, "rtg_photon_map_order");
}

int rtg_indirect_irrad(rtg_scene *scene, rtg_photon_map *map, const rtg_point *points, uint64_t n_points, uint32_t photon_samples,
                       uint32_t newton_steps, double newton_epsilon, double *rgb_out) try {
    int rc = api_bind_scene(scene);
    if (rc != RTG_OK) return rc;
    rtg_ctx *ctx = scene->ctx;
    if ((rc = check_irrad_args(scene, map, photon_samples)) != RTG_OK) return rc;
    if (n_points == 0) return RTG_OK;
    if (!points || !rgb_out) return api_fail(ctx, RTG_E_INVALID, "rtg_indirect_irrad: points / rgb_out is null");
    const void *d_points = nullptr;
    if ((rc = api_input(ctx, "irrad_points", points, (size_t)n_points * sizeof(rtg_point), &d_points)) != RTG_OK) return rc;
    const bool out_dev = api_is_device_pointer(rgb_out);
    void *d_rgb = rgb_out;
    if (!out_dev && (rc = api_scratch(ctx, "irrad_rgb", (size_t)n_points * 3 * sizeof(double), &d_rgb)) != RTG_OK) return rc;
    if ((rc = indirect_device(scene, map, (const rtg_point *)d_points, n_points, photon_samples, newton_steps, newton_epsilon,
                              (double *)d_rgb)) != RTG_OK)
        return rc;
    if (!out_dev) {
        HIP_TRY(ctx, hipMemcpyAsync(rgb_out, d_rgb, (size_t)n_points * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RTG_OK;
} catch (...) {
    return api_caught(scene ? scene->ctx : nullptr, "rtg_indirect_irrad");
}

int rtg_render_frame_photons(rtg_scene *scene, const rtg_view *view, const rtg_light *lights, uint32_t n_lights,
                             rtg_photon_map *map, uint32_t photon_samples, uint32_t newton_steps, double newton_epsilon,
                             uint32_t width, uint32_t height, rtg_rect tile, uint32_t max_depth, double *out_rgb,
                             uint8_t *out_rgba8) try {
    int rc = api_bind_scene(scene);
    if (rc != RTG_OK) return rc;
    rtg_ctx *ctx = scene->ctx;
    if ((rc = check_irrad_args(scene, map, photon_samples)) != RTG_OK) return rc;
    if (!view) return api_fail(ctx, RTG_E_INVALID, "rtg_render_frame_photons: view is null");
    if (n_lights && !lights) return api_fail(ctx, RTG_E_INVALID, "rtg_render_frame_photons: lights is null");
    if (width == 0 || height == 0) return api_fail(ctx, RTG_E_INVALID, "rtg_render_frame_photons: empty image");
    if ((uint64_t)tile.x0 + tile.w > width || (uint64_t)tile.y0 + tile.h > height)
        return api_fail(ctx, RTG_E_INVALID, "rtg_render_frame_photons: tile outside the image");
    if (max_depth < 1 || max_depth > 4) return api_fail(ctx, RTG_E_INVALID, "rtg_render_frame_photons: max_depth must be 1..4");
    const uint64_t n = (uint64_t)tile.w * tile.h;
    if (n == 0) return RTG_OK;
    if (!out_rgb && !out_rgba8) return api_fail(ctx, RTG_E_INVALID, "rtg_render_frame_photons: no output requested");
    const uint64_t per_ray = (1ull << max_depth) - 1;
    if (n * per_ray * (uint64_t)std::max(1u, std::max(n_lights, photon_samples)) > 0xFFFFFFFFull)
        return api_fail(ctx, RTG_E_LIMIT, "rtg_render_frame_photons: tile too large for one call (render it in bands)");
    if ((out_rgb && api_is_device_pointer(out_rgb) && (uintptr_t)out_rgb % 16 != 0) ||
        (out_rgba8 && api_is_device_pointer(out_rgba8) && (uintptr_t)out_rgba8 % 4 != 0))
        return api_fail(ctx, RTG_E_INVALID, "rtg_render_frame_photons: misaligned device output");

    const void *d_lights = nullptr;
    if ((rc = api_input(ctx, "frame_lights", lights, (size_t)n_lights * sizeof(rtg_light), &d_lights)) != RTG_OK) return rc;
    void *rays = nullptr, *paths = nullptr, *points = nullptr, *indirect = nullptr, *path_rgb = nullptr, *frame = nullptr, *c8 = nullptr;
    if ((rc = api_scratch(ctx, "frame_rays", (size_t)n * 6 * sizeof(double), &rays)) != RTG_OK) return rc;
    if ((rc = api_scratch(ctx, "frame_paths", (size_t)(n * per_ray) * sizeof(rtg_path), &paths)) != RTG_OK) return rc;
    HIP_TRY(ctx, launch_camera_rays(ctx->stream, *view, width, height, tile, (double *)rays));
    uint64_t n_paths = 0;
    if ((rc = rtg_trace_paths(scene, (const double *)rays, n, max_depth, (rtg_path *)paths, n * per_ray, &n_paths)) != RTG_OK) return rc;
    if (n_paths) {
        if ((rc = api_scratch(ctx, "frame_path_points", (size_t)n_paths * sizeof(rtg_point), &points)) != RTG_OK) return rc;
        if ((rc = api_scratch(ctx, "frame_indirect", (size_t)n_paths * 3 * sizeof(double), &indirect)) != RTG_OK) return rc;
        if ((rc = api_scratch(ctx, "frame_path_rgb", (size_t)n_paths * 3 * sizeof(double), &path_rgb)) != RTG_OK) return rc;
        // the paths' RaycastPoints, packed: what the irradiance terms are evaluated at (renderer.rs:270-273)
        HIP_TRY(ctx, hipMemcpy2DAsync(points, sizeof(rtg_point), (const char *)paths + offsetof(rtg_path, point), sizeof(rtg_path),
                                      sizeof(rtg_point), (size_t)n_paths, hipMemcpyDeviceToDevice, ctx->stream));
        if ((rc = indirect_device(scene, map, (const rtg_point *)points, n_paths, photon_samples, newton_steps, newton_epsilon,
                                  (double *)indirect)) != RTG_OK)
            return rc;
        if ((rc = api_shade_points(scene, (const rtg_light *)d_lights, n_lights, (const rtg_point *)points, n_paths,
                                   (const double *)indirect, (double *)path_rgb)) != RTG_OK)
            return rc;
    }
    const bool rgb_dev = out_rgb && api_is_device_pointer(out_rgb), c8_dev = out_rgba8 && api_is_device_pointer(out_rgba8);
    frame = rgb_dev ? (void *)out_rgb : nullptr;
    if (!frame && (rc = api_scratch(ctx, "frame_rgb_out", (size_t)n * 3 * sizeof(double), &frame)) != RTG_OK) return rc;
    c8 = c8_dev ? (void *)out_rgba8 : nullptr;
    if (out_rgba8 && !c8 && (rc = api_scratch(ctx, "frame_rgba8_out", (size_t)n * 4, &c8)) != RTG_OK) return rc;
    HIP_TRY(ctx, launch_frame_from_paths(ctx->stream, (const rtg_path *)paths, (uint32_t)n_paths, (const double *)path_rgb, n,
                                         (double *)frame, (uint8_t *)c8));
    bool staged = false;
    if (out_rgb && !rgb_dev) {
        HIP_TRY(ctx, hipMemcpyAsync(out_rgb, frame, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        staged = true;
    }
    if (out_rgba8 && !c8_dev) {
        HIP_TRY(ctx, hipMemcpyAsync(out_rgba8, c8, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        staged = true;
    }
    if (staged) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RTG_OK;
} catch (...) {
    return api_caught(scene ? scene->ctx : nullptr, "rtg_render_frame_photons");
}

} // extern "C"
