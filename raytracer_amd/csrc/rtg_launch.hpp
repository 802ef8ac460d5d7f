// rtg_launch.hpp -- host-callable launchers of the kernels in rtg_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "rtg_device.hpp"

namespace rtg {

hipError_t launch_render_primary(hipStream_t stream, const DevScene &s, const rtg_view &view, uint32_t width,
                                 uint32_t height, rtg_rect tile, rtg_hit *out_hits, rtg_point *out_points,
                                 rtg_ray_counters *out_counters);
hipError_t launch_trace(hipStream_t stream, const DevScene &s, const double *rays6, uint64_t n, rtg_hit *out_hits,
                        rtg_point *out_points, rtg_ray_counters *out_counters);
// `indirect` (3 doubles per point, or null) is the photon map's term of the irradiance sum.
hipError_t launch_shade_direct(hipStream_t stream, const DevScene &s, const rtg_light *lights, uint32_t n_lights,
                               const rtg_point *points, uint64_t n, double *rgb_out, uint8_t *rgba8_out,
                               const double *indirect = nullptr);
hipError_t launch_render_frame(hipStream_t stream, const DevScene &s, const rtg_view &view, const rtg_light *lights,
                               uint32_t n_lights, uint32_t width, uint32_t height, rtg_rect tile, uint32_t max_depth,
                               double *out_rgb, uint8_t *out_rgba8, rtg_hit *out_hits,
                               const rtg_point *only_refracting = nullptr);

// Persistent phase-scheduled kernels (rtg_phased.hip).
struct PhasedTuning {
    uint32_t tri_min = 20;
    uint32_t patch_tri_min = 24;   // lanes that must wait for a patch leaf's Newton solve before one runs
    uint32_t obj_min = 4;
    uint32_t max_blocks = 0;     // persistent grid in four-wave groups; 0 = as many as the GPU holds at once
    uint32_t pretest = 1;        // single-precision box pre-test with exact fallback
    uint32_t stats = 0;          // diagnostics build: tally steps and lanes per phase
    uint32_t keep_state = 0;     // run the several-meshes build (per-lane tree base, outer ray kept) on any scene
    // when set, the persistent kernel's launch records its execution window (the dispatch's own begin and end
    // timestamps, what a profiler reports as the kernel's duration) in these events
    hipEvent_t time_start = nullptr, time_stop = nullptr;
};
// Device scratch of one persistent launch: the work counters (8 * 32 + 16 u32) and, for scenes that mix meshes
// with spheres or planes, one rtg_hit per output slot for the closest sphere / plane hit.
struct PhasedScratch {
    uint32_t *work_counters = nullptr;
    rtg_hit *prim_best = nullptr;
};
hipError_t launch_render_primary_phased(hipStream_t stream, const DevScene &s, const rtg_view &view, uint32_t width,
                                        uint32_t height, rtg_rect tile, const PhasedScratch &scratch,
                                        const PhasedTuning &tune, rtg_hit *out_hits, rtg_ray_counters *out_counters);
hipError_t launch_trace_phased(hipStream_t stream, const DevScene &s, const double *rays6, uint64_t n,
                               const PhasedScratch &scratch, const PhasedTuning &tune, rtg_hit *out_hits,
                               rtg_ray_counters *out_counters);
// RaycastPoint records rebuilt from hit ids (one lane per ray, streaming): the rays are regenerated from the view
// or read from the batch.
hipError_t launch_points_primary(hipStream_t stream, const DevScene &s, const rtg_view &view, uint32_t width,
                                 uint32_t height, rtg_rect tile, const rtg_hit *hits, rtg_point *out_points);
hipError_t launch_points_batch(hipStream_t stream, const DevScene &s, const double *rays6, uint64_t n,
                               const rtg_hit *hits, rtg_point *out_points);

// Closest hit along the shadow ray of every (light, point) pair that has one, out_hits[l * n_points + p], and in
// the same launch the closest hits of `n_rays` batch rays, out_hits[n_lights * n_points + i].
hipError_t launch_shadow_phased(hipStream_t stream, const DevScene &s, const rtg_point *points, uint64_t n_points,
                                const rtg_light *lights, uint32_t n_lights, const double *rays6, uint64_t n_rays,
                                const PhasedScratch &scratch, const PhasedTuning &tune, rtg_hit *out_hits);
// compute_direct_irrad x diffuse from occluders found by launch_shadow_phased.
hipError_t launch_shade_resolve(hipStream_t stream, const DevScene &s, const rtg_light *lights, uint32_t n_lights,
                                const rtg_point *points, uint64_t n, const rtg_hit *occluders, double *rgb_out,
                                uint8_t *rgba8_out, const double *indirect = nullptr);

// Specular path tree as levels of ray batches (rtg_specular.hip).
struct SpecMeta {
    uint32_t group;        // slot group of the refracting pixel the ray descends from
    uint32_t slot;         // pre-order index of the ray's node in that pixel's path tree
    double attenuation;
};
hipError_t launch_spec_count(hipStream_t stream, const DevScene &s, const rtg_point *points, uint64_t n,
                             uint32_t *counter);
hipError_t launch_spec_spawn(hipStream_t stream, const DevScene &s, const rtg_view &view, uint32_t width,
                             uint32_t height, rtg_rect tile, const double *rays6, const SpecMeta *meta,
                             const rtg_point *points, uint64_t n, uint32_t depth, uint32_t max_depth,
                             double *next_rays6, SpecMeta *next_meta, uint32_t *groups, uint32_t *counters);
hipError_t launch_spec_store(hipStream_t stream, const SpecMeta *meta, const rtg_point *points, const double *rgb,
                             uint64_t n, uint32_t slots_per_group, double *slots);
hipError_t launch_spec_sum(hipStream_t stream, const uint32_t *groups, uint32_t n_groups, uint32_t slots_per_group,
                           const double *slots, double *frame_rgb, double *out_rgb, uint8_t *out_rgba8);

// BvhForest::new(triangles).subdivide() + Bvh::new on the device (rtg_build.hip), for 2 <= n <= 2^30 triangles
// already in device memory.  On success the four device allocations belong to the caller (hipFree).
struct BuiltMesh {
    void *d_slab = nullptr;          // MeshSlabHeader (zeroed) + hot nodes + threaded nodes + triangles
    void *d_nodes = nullptr;         // reference-order DevNode x n_nodes
    void *d_leaf_source = nullptr;   // u32 x n: the caller's triangle index of every leaf
    void *d_normals = nullptr;       // 9 doubles per leaf in leaf order, or nullptr
    uint32_t n_nodes = 0, root = 0, troot = 0, depth = 0, fast_div_ok = 0, f32_ok = 0;   // troot: root's threaded position
    double coord_max = 0.0;
    double bounds[6] = {0, 0, 0, 0, 0, 0};
};
hipError_t build_subdivide_device(hipStream_t stream, const rtg_tri *d_tris, uint32_t n, const double *d_normals_in,
                                  BuiltMesh *out, std::string &err);

// by_leaf[l] = by_facet[leaf_source[l]]: per-facet patch records put into leaf order after a device build.
hipError_t launch_gather_leaf_info(hipStream_t stream, const rtg_patch_leaf *by_facet, const uint32_t *leaf_source,
                                   uint32_t n, rtg_patch_leaf *by_leaf);

// Path records of a ray batch (rtg_paths.hip): root meta, one record per hit of a level, and the final ordering
// by (ray, pre-order index) with parents resolved to positions.
hipError_t launch_path_roots(hipStream_t stream, uint64_t n, SpecMeta *meta);
hipError_t launch_path_emit(hipStream_t stream, const SpecMeta *meta, const rtg_point *points, uint64_t n,
                            uint32_t max_depth, uint32_t *counter, uint64_t *keys, rtg_path *records);
size_t path_sort_temp_bytes(uint32_t n, uint32_t key_bits);
hipError_t launch_path_sort_gather(hipStream_t stream, const rtg_path *const *level_records, const uint32_t *level_first,
                                   uint32_t n_levels, uint32_t n, uint32_t key_bits, const uint64_t *keys,
                                   uint64_t *keys_sorted, uint32_t *source, uint32_t *source_sorted, void *temp,
                                   size_t temp_bytes, rtg_path *out);

// Ray batches put into coherence order on the device (rtg_sort.hip): key = direction octant | 30-bit Morton code of
// the origin inside the batch's bounding box, one radix sort of (key, index) pairs, rays gathered into key order;
// afterwards the results are scattered back with the same permutation.
struct RaySortScratch {
    unsigned long long *box = nullptr;     // 6 words
    uint64_t *keys = nullptr, *keys_sorted = nullptr;
    uint32_t *index = nullptr, *perm = nullptr;
    void *temp = nullptr;
    size_t temp_bytes = 0;
};
size_t ray_sort_temp_bytes(uint64_t n);
hipError_t launch_ray_sort(hipStream_t stream, const double *rays6, uint64_t n, const RaySortScratch &scratch,
                           double *sorted_rays6);
hipError_t launch_ray_unsort(hipStream_t stream, const rtg_hit *hits, const rtg_ray_counters *counters, const uint32_t *perm,
                             uint64_t n, rtg_hit *out_hits, rtg_ray_counters *out_counters);

// The photon map (rtg_photon.hip).  `nodes` is the k-d tree as 80-byte records in implicit-median order; k <= 16.
hipError_t launch_kd_nearest(hipStream_t stream, const void *nodes, uint32_t n_nodes, const double *queries3,
                             uint64_t n_queries, uint32_t k, uint32_t *out_index, double *out_distance);
// KdTree::new on the device (rtg_kdtree.hip): `nodes` receives n 80-byte records in implicit-median order.  `scratch`
// holds kd_build_scratch_bytes(n) bytes; *summary points at two words inside it, valid once the stream has drained:
// the longest photon path (capped at 3) and the index of the first photon with a NaN coordinate (or 0xFFFFFFFF).
size_t kd_build_scratch_bytes(uint32_t n);
// finish_in_lds: ranges of at most 1024 photons are finished by one workgroup each in a single launch (same tree).
hipError_t build_kd_device(hipStream_t stream, const rtg_photon *photons, uint32_t n, void *scratch, void *nodes, uint32_t **summary,
                           bool finish_in_lds = true);
// The rays of a batch of emission directions and their (zenith, y) pairs from `staged` (cos zenith, sin zenith,
// zenith, y per direction, device memory).
hipError_t launch_photon_rays(hipStream_t stream, const double *staged, uint32_t n, const rtg_light *lights, uint64_t first_ray,
                              uint64_t count_per_light, double *rays6, double *dirs2);
hipError_t launch_photon_rays_trig(hipStream_t stream, const double *dirs2_in, uint32_t n, const rtg_light *lights,
                                   uint64_t first_ray, uint64_t count_per_light, double *rays6, double *dirs2);
hipError_t launch_photon_rays_fibonacci(hipStream_t stream, uint32_t n, const rtg_light *lights, uint64_t first_ray,
                                        uint64_t count_per_light, uint64_t seed, double *rays6, double *dirs2);
// The rays of a batch that can make photons (their first hit lies on an object with a refracting material), packed
// in their order: flags + exclusive scan (offsets[i] = kept rays before ray i; the count is offsets[n-1] + flags[n-1])
// + copy; kept_index[j] = batch index of kept ray j.
hipError_t launch_photon_first_filter(hipStream_t stream, const DevScene &scene, const rtg_hit *hits, const double *rays6,
                                      uint32_t n, uint32_t *flags, uint32_t *offsets, void *temp, size_t temp_bytes,
                                      double *kept_rays6, uint32_t *kept_index);
// Photons of one batch of path records: flags + exclusive scan (offsets[i] = photons before path i) + records
// written at out[offsets[i]] while that is below `room`.  Path ray r is ray ray_index[r] of the batch `dirs2` and
// `first_ray` describe.
size_t photon_scan_temp_bytes(uint32_t n);
hipError_t launch_photons_from_paths(hipStream_t stream, const rtg_path *paths, uint32_t n_paths, const double *dirs2,
                                     const uint32_t *ray_index, const rtg_light *lights, uint64_t first_ray,
                                     uint64_t count_per_light, uint32_t *flags, uint32_t *offsets, void *temp,
                                     size_t temp_bytes, uint64_t room, rtg_photon *out);
// The (surface point, photon sample) pairs the caustic correction works on; pair e = point * samples + sample.
struct PhotonPairs {
    const DevObject *objects = nullptr;
    uint32_t n_objects = 0;
    const uint64_t *object_leaves = nullptr;
    const rtg_photon *photons = nullptr;
    uint32_t n_photons = 0;
    const uint32_t *nearest = nullptr;
    const rtg_point *points = nullptr;
    uint64_t n_points = 0;
    uint32_t samples = 0;
};
hipError_t launch_photon_newton(hipStream_t stream, const PhotonPairs &p, uint32_t newton_steps, double *dir2);
hipError_t launch_photon_final_start(hipStream_t stream, const PhotonPairs &p, const double *dir2, double *state, double *rays6,
                                     uint32_t *status);
hipError_t launch_photon_final_bounce(hipStream_t stream, const PhotonPairs &p, uint32_t level, const rtg_hit *hits, double *state,
                                      double *rays6, uint32_t *status, double *landing);
hipError_t launch_photon_sum(hipStream_t stream, const PhotonPairs &p, const uint32_t *status, const double *landing,
                             double newton_epsilon, double *rgb);

// A frame from path records: camera rays of a tile, and the per-pixel sum of path_rgb[j] * attenuation over a
// pixel's paths (pixels without paths are black), optionally PQ-encoded to 8 bits.
hipError_t launch_camera_rays(hipStream_t stream, const rtg_view &view, uint32_t width, uint32_t height, rtg_rect tile, double *rays6);
hipError_t launch_frame_from_paths(hipStream_t stream, const rtg_path *paths, uint32_t n_paths, const double *path_rgb,
                                   uint64_t n_pixels, double *out_rgb, uint8_t *out_rgba8);

// Looks up this GPU's copy of the fixed scratch array behind rtg_render_primary_static (rtg_static.hip); called
// with `device` current, once per device (later calls return the address found then).  *out is the device address.
hipError_t static_scratch_init(int device, unsigned char **out);

hipError_t launch_selftest_div(hipStream_t stream, uint64_t seed, uint32_t blocks, uint32_t per_thread, uint32_t family,
                               uint64_t *out3);

} // namespace rtg
