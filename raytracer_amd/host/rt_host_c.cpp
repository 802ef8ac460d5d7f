// rt_host_c.cpp -- C wrappers of the host library (see rt_host_c.h).
#include "rt_host_c.h"

#include <map>
#include <string>

#include <algorithm>
#include <cstring>
#include <new>
#include <numeric>

#include "rt_bvh.hpp"
#include "rt_mesh.hpp"
#include "rt_patch.hpp"
#include "rt_scene.hpp"

struct rth_bvh {
    rt::Bvh bvh;
};

extern "C" {

rth_bvh *rth_bvh_build_subdivide(const rtg_tri *tris, uint64_t n, const double *vertex_normals, int threads) {
    rth_bvh *b = new rth_bvh();
    b->bvh = rt::build_subdivide(tris, n, vertex_normals, threads);
    return b;
}
rth_bvh *rth_bvh_build(const rtg_tri *tris, uint64_t n, const double *vertex_normals, int threads, int strategy) {
    rth_bvh *b = new rth_bvh();
    if (strategy == 1)
        b->bvh = rt::build_contract(tris, n, vertex_normals, threads);
    else if (strategy == 2)
        b->bvh = rt::build_hybrid(tris, n, vertex_normals, threads);
    else
        b->bvh = rt::build_subdivide(tris, n, vertex_normals, threads);
    return b;
}
void rth_bvh_destroy(rth_bvh *b) { delete b; }
void rth_bvh_view(const rth_bvh *b, rtg_bvh *out) { *out = b->bvh.view(); }
const uint32_t *rth_bvh_leaf_source(const rth_bvh *b) { return b->bvh.leaf_source.data(); }
uint32_t rth_bvh_depth(const rth_bvh *b) { return b->bvh.depth; }

void rth_heightfield(uint32_t nx, uint32_t ny, uint64_t seed, double scale, rtg_tri *out) {
    rt::heightfield(nx, ny, seed, scale, out);
}
void rth_bench_mesh_rays(const double lo_hi[6], uint64_t n, uint64_t seed, double *rays6) {
    rt::bench_mesh_rays(lo_hi, n, seed, rays6);
}
void rth_sort_rays(const double lo_hi[6], const double *rays6, uint64_t n, uint32_t *perm, double *sorted) {
    std::vector<uint64_t> keys((size_t)n);
    for (uint64_t i = 0; i < n; i++) keys[i] = rt::ray_sort_key(lo_hi, rays6 + i * 6);
    std::iota(perm, perm + n, 0u);
    std::stable_sort(perm, perm + n, [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
    if (sorted)
        for (uint64_t i = 0; i < n; i++) std::memcpy(sorted + i * 6, rays6 + (size_t)perm[i] * 6, 6 * sizeof(double));
}

void rth_trispline_from_normal_hermite(const double in30[30], rtg_trispline *out) {
    rt::NormalHermiteTrispline nh;
    rt::D3 *f[10] = {&nh.p0, &nh.p1, &nh.p2, &nh.p01, &nh.p12, &nh.p20, &nh.n0, &nh.n1, &nh.n2, &nh.control};
    for (int i = 0; i < 10; i++) *f[i] = rt::D3{in30[i * 3], in30[i * 3 + 1], in30[i * 3 + 2]};
    *out = rt::trispline_from(nh);
}
void rth_trispline_eval(const rtg_trispline *sp, double t, double u, double out12[12]) {
    rt::D3 v[4] = {rt::trispline_position(*sp, t, u), rt::trispline_ddt(*sp, t, u), rt::trispline_ddu(*sp, t, u),
                   rt::trispline_normal(*sp, t, u)};
    for (int i = 0; i < 4; i++) {
        out12[i * 3] = v[i].x;
        out12[i * 3 + 1] = v[i].y;
        out12[i * 3 + 2] = v[i].z;
    }
}
void rth_patches_tessellate(const rtg_trispline *patches, uint64_t n, uint32_t k, rtg_tri *out_tris,
                            rtg_patch_leaf *out_info) {
    rt::tessellate(patches, n, k, out_tris, out_info);
}
void rth_patch_terrain(uint32_t nx, uint32_t ny, uint64_t seed, double scale, rtg_trispline *out) {
    std::vector<rtg_trispline> p = rt::patch_terrain(nx, ny, seed, scale);
    std::memcpy(out, p.data(), p.size() * sizeof(rtg_trispline));
}

void rth_transform_build(double scale, const double translate[3], rtg_transform *out) {
    rt::TransformBuilder tb;
    tb.scale(scale).translate(translate[0], translate[1], translate[2]);
    *out = tb.build().t;
}
void rth_scene_builder(uint64_t time, rtg_object objects[2], rtg_light lights[2], rtg_view *view) {
    rt::SceneBuilder sb;
    sb.time = time;
    std::vector<rt::Object> objs = sb.scene_object();
    objects[0] = objs[0].c;
    objects[1] = objs[1].c;
    std::vector<rt::Light> ls = sb.lights();
    for (int i = 0; i < 2; i++) {
        lights[i].position[0] = ls[i].sphere.orig.x;
        lights[i].position[1] = ls[i].sphere.orig.y;
        lights[i].position[2] = ls[i].sphere.orig.z;
        lights[i].color[0] = ls[i].color.x;
        lights[i].color[1] = ls[i].color.y;
        lights[i].color[2] = ls[i].color.z;
    }
    *view = sb.view().c();
}
int rth_render_scene_builder(rtg_ctx *ctx, uint64_t time, uint32_t width, uint32_t height, uint32_t max_depth,
                             double *out_rgb, rtg_hit *out_hits) try {
    rt::SceneBuilder sb;
    sb.time = time;
    rt::Scene scene = sb.scene();
    scene.size[0] = width;
    scene.size[1] = height;
    rt::Renderer r(ctx, std::move(scene));
    r.max_depth = max_depth;
    int rc = r.render();
    if (rc != RTG_OK) return rc;
    if (out_rgb) std::memcpy(out_rgb, r.radiosity().data(), r.radiosity().size() * sizeof(double));
    if (out_hits) std::memcpy(out_hits, r.hits().data(), r.hits().size() * sizeof(rtg_hit));
    return RTG_OK;
} catch (const std::bad_alloc &) {
    return RTG_E_NOMEM;          // nothing unwinds into the caller, which need not be C++
} catch (...) {
    return RTG_E_INTERNAL;
}
int rth_render_scene_builder_photons(rtg_ctx *ctx, uint64_t time, uint32_t width, uint32_t height, uint32_t max_depth,
                                     uint64_t photon_count, uint64_t seed, int photons_on_host, double *out_rgb,
                                     uint64_t *n_photons, double seconds[3]) try {
    rt::SceneBuilder sb;
    sb.time = time;
    rt::Scene scene = sb.scene();
    scene.size[0] = width;
    scene.size[1] = height;
    scene.photon_count = photon_count;
    rt::Renderer r(ctx, std::move(scene));
    r.max_depth = max_depth;
    r.photon_pass = true;
    r.photons_on_host = photons_on_host != 0;
    r.seed = seed;
    int rc = r.render();
    if (n_photons) *n_photons = r.photons();
    if (seconds) {
        seconds[0] = r.seconds_directions();
        seconds[1] = r.seconds_photons();
        seconds[2] = r.seconds_pixels();
    }
    if (rc != RTG_OK) return rc;
    if (out_rgb) std::memcpy(out_rgb, r.radiosity().data(), r.radiosity().size() * sizeof(double));
    return RTG_OK;
} catch (const std::bad_alloc &) {
    return RTG_E_NOMEM;          // nothing unwinds into the caller, which need not be C++
} catch (...) {
    return RTG_E_INTERNAL;
}
// upper bound of one Radiance picture's bytes (sizing calls): the text header and, per row, the scanline marker and
// four components that run-length coding can at worst double
static uint64_t hdr_bytes_max(uint32_t width, uint32_t height) {
    return 96 + (uint64_t)height * (4 + 4 * (2 * (uint64_t)width + 2));
}
uint64_t rth_to_hdr(const double *rgb, uint32_t width, uint32_t height, uint8_t *out, uint64_t cap) {
    if (!out) return hdr_bytes_max(width, height);      // sizing call: nothing is encoded
    std::vector<uint8_t> b = rt::to_hdr(rgb, width, height);
    if (out && cap >= b.size()) std::memcpy(out, b.data(), b.size());
    return b.size();
}
void rth_fibonacci_sphere(uint64_t count, uint64_t seed, double *dirs3) { rt::fibonacci_sphere(count, seed, dirs3); }
void rth_fibonacci_zenith_y(uint64_t count, uint64_t seed, double *dirs2) { rt::fibonacci_zenith_y(count, seed, dirs2); }
uint64_t rth_images_tar(const double *rgb, uint32_t width, uint32_t height, double seconds, const uint32_t *cost,
                        uint8_t *out, uint64_t cap) {
    if (!out)                                           // sizing call: three members (header block + padded bytes), two end blocks
        return 3 * (512 + (hdr_bytes_max(width, height) + 511) / 512 * 512) + 1024;
    std::map<std::string, std::vector<uint8_t>> images;                    // Renderer::images (renderer.rs:379-385)
    const size_t n = (size_t)width * height * 3;
    images["image.hdr"] = rt::to_hdr(rgb, width, height);
    std::vector<double> pq(n);
    rt::pq_encode(rgb, n, pq.data());
    images["image.pq.hdr"] = rt::to_hdr(pq.data(), width, height);
    std::vector<double> perf = rt::perf_image(seconds, cost, width, height);
    images["perf.hdr"] = rt::to_hdr(perf.data(), width, height);
    std::vector<uint8_t> b = rt::images_tar(images);
    if (out && cap >= b.size()) std::memcpy(out, b.data(), b.size());
    return b.size();
}

int rth_render_images(rtg_ctx *ctx, const rtg_object *objects, uint32_t n_objects, const rtg_light *lights,
                      uint32_t n_lights, const double view_orig[3], const double fov[2], uint32_t width, uint32_t height,
                      uint32_t max_depth, double *out_rgb, rtg_hit *out_hits, uint32_t *out_cost, double *out_seconds,
                      uint8_t *out_tar, uint64_t cap, uint64_t *tar_bytes) try {
    if (!ctx || (n_objects && !objects) || (n_lights && !lights) || !view_orig || !fov) return RTG_E_INVALID;
    rt::Scene scene;
    scene.size[0] = width;
    scene.size[1] = height;
    scene.view = rt::View::from_fov(rt::Vec3{view_orig[0], view_orig[1], view_orig[2]}, fov[0], fov[1]);
    for (uint32_t i = 0; i < n_lights; i++)
        scene.lights.push_back(rt::Light{rt::Sphere{rt::Vec3{lights[i].position[0], lights[i].position[1], lights[i].position[2]}, 1.0},
                                         rt::Color{lights[i].color[0], lights[i].color[1], lights[i].color[2]}});
    for (uint32_t i = 0; i < n_objects; i++) {
        rt::Object o;
        o.c = objects[i];                                 // meshes stay the caller's (no shared handle taken)
        scene.scene_object.push_back(o);
    }
    rt::Renderer r(ctx, std::move(scene));
    r.max_depth = max_depth;
    int rc = r.render();
    if (rc != RTG_OK) return rc;
    if (out_rgb) std::memcpy(out_rgb, r.radiosity().data(), r.radiosity().size() * sizeof(double));
    if (out_hits) std::memcpy(out_hits, r.hits().data(), r.hits().size() * sizeof(rtg_hit));
    if (out_cost) std::memcpy(out_cost, r.pixel_cost().data(), r.pixel_cost().size() * sizeof(uint32_t));
    if (out_seconds) *out_seconds = r.seconds_pixels();
    std::vector<uint8_t> tar = rt::images_tar(r.images());
    if (tar_bytes) *tar_bytes = tar.size();
    if (out_tar) {
        if (cap < tar.size()) return RTG_E_LIMIT;
        std::memcpy(out_tar, tar.data(), tar.size());
    }
    return RTG_OK;
} catch (const std::bad_alloc &) {
    return RTG_E_NOMEM;
} catch (...) {
    return RTG_E_INTERNAL;
}

} // extern "C"
