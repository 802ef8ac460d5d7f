/* rt_host_c.h -- C entry points of the host-side library (librt_host.so) for
 * callers without a C++ toolchain (the Python tests and bench).  Thin wrappers
 * over rt_bvh.hpp / rt_mesh.hpp / rt_scene.hpp. */
#ifndef RT_HOST_C_H
#define RT_HOST_C_H
#include "../../include/rtg.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_bvh rth_bvh;

/* BvhForest::new(tris).subdivide() + Bvh::new (src/tree/bvh_subdivide.rs, src/tree/bvh.rs:80-106) */
rth_bvh *rth_bvh_build_subdivide(const rtg_tri *tris, uint64_t n, const double *vertex_normals, int threads);
/* strategy 0: BvhForest::subdivide, 1: BvhForest::contract, 2: BvhForest::hybrid. */
rth_bvh *rth_bvh_build(const rtg_tri *tris, uint64_t n, const double *vertex_normals, int threads, int strategy);
void rth_bvh_destroy(rth_bvh *b);
void rth_bvh_view(const rth_bvh *b, rtg_bvh *out);
const uint32_t *rth_bvh_leaf_source(const rth_bvh *b);
uint32_t rth_bvh_depth(const rth_bvh *b);

void rth_heightfield(uint32_t nx, uint32_t ny, uint64_t seed, double scale, rtg_tri *out);
/* Sphere::fibonacci_sphere + ZenithY::into_normal (src/geo/sphere.rs:27-32,48-80): the photon pass's directions. */
void rth_fibonacci_sphere(uint64_t count, uint64_t seed, double *dirs3);
void rth_fibonacci_zenith_y(uint64_t count, uint64_t seed, double *dirs2);
void rth_bench_mesh_rays(const double lo_hi[6], uint64_t n, uint64_t seed, double *rays6);
/* Stable sort of rays by (direction octant, Morton code of origin); perm[i] = source index of sorted ray i. */
void rth_sort_rays(const double lo_hi[6], const double *rays6, uint64_t n, uint32_t *perm, double *sorted_rays6);

/* trispline patches (src/trispline.rs): in30 = p0,p1,p2,p01,p12,p20,n0,n1,n2,control */
void rth_trispline_from_normal_hermite(const double in30[30], rtg_trispline *out);
/* out12 = position, ddt, ddu, normal */
void rth_trispline_eval(const rtg_trispline *sp, double t, double u, double out12[12]);
void rth_patches_tessellate(const rtg_trispline *patches, uint64_t n, uint32_t k, rtg_tri *out_tris, rtg_patch_leaf *out_info);
/* 2*nx*ny patches over the heightfield terrain */
void rth_patch_terrain(uint32_t nx, uint32_t ny, uint64_t seed, double scale, rtg_trispline *out);

/* TransformBuilder::new().scale(s).translate(x,y,z).build() (src/geo/transform.rs:33-49) */
void rth_transform_build(double scale, const double translate[3], rtg_transform *out);
/* SceneBuilder{time} (src/lib.rs:42-121): objects[2] = sphere, plane; lights[2]; view. */
void rth_scene_builder(uint64_t time, rtg_object objects[2], rtg_light lights[2], rtg_view *view);

/* Renderer over the default scene plus optional extra mesh objects: renders `width` x `height`
 * and returns the radiosity (3 doubles per pixel) and primary hits. */
int rth_render_scene_builder(rtg_ctx *ctx, uint64_t time, uint32_t width, uint32_t height, uint32_t max_depth,
                             double *out_rgb, rtg_hit *out_hits);
/* The same with the photon pass of Renderer::render (src/render/renderer.rs:116-140): `photon_count` Fibonacci
 * directions per light from `seed` (the reference seeds from entropy), the photon map, and the pixel pass with the
 * photon term.  photons_on_host != 0: directions and their sin / cos on the host (the CPU renderer's photons bit for
 * bit); 0: the whole pass on the GPU.  seconds[3] (may be NULL): directions, photon pass + k-d tree, pixel pass. */
int rth_render_scene_builder_photons(rtg_ctx *ctx, uint64_t time, uint32_t width, uint32_t height, uint32_t max_depth,
                                     uint64_t photon_count, uint64_t seed, int photons_on_host, double *out_rgb,
                                     uint64_t *n_photons, double seconds[3]);
/* Radiance .hdr bytes of an RGB double image as ImageBuilder::to_hdr writes them (src/render/image.rs:44-71);
 * returns the byte count; called with out == NULL it returns an upper bound to size the buffer with. */
uint64_t rth_to_hdr(const double *rgb, uint32_t width, uint32_t height, uint8_t *out, uint64_t cap);
/* The three images of Renderer::images (image.hdr, image.pq.hdr, perf.hdr; src/render/renderer.rs:379-385) of an RGB
 * double frame rendered in `seconds`, framed as the tar archive the reference's server returns
 * (src/bin/server.rs:19-30); `cost` (may be NULL: evenly) is the per-pixel work the frame's time is shared out by in
 * perf.hdr.  Returns the byte count (out == NULL: an upper bound). */
uint64_t rth_images_tar(const double *rgb, uint32_t width, uint32_t height, double seconds, const uint32_t *cost,
                        uint8_t *out, uint64_t cap);
/* Renderer::new(scene).render() and .images() (src/render/renderer.rs:106-154,379-385) over a caller-made scene on
 * the GPU, the images framed as the server's tar archive: radiosity (3 doubles per pixel), primary hits, the
 * per-pixel cost behind perf.hdr (nodes visited + triangles tested by the camera ray), the pixel pass's wall time and
 * the archive.  `view_orig` and `fov` are the eye and the view's two opening angles (View::from_fov); meshes named by the objects stay the
 * caller's.  Every output may be NULL; RTG_E_LIMIT when `cap` is too small for the archive (*tar_bytes says how
 * much it needs). */
int rth_render_images(rtg_ctx *ctx, const rtg_object *objects, uint32_t n_objects, const rtg_light *lights,
                      uint32_t n_lights, const double view_orig[3], const double fov[2], uint32_t width, uint32_t height,
                      uint32_t max_depth, double *out_rgb, rtg_hit *out_hits, uint32_t *out_cost, double *out_seconds,
                      uint8_t *out_tar, uint64_t cap, uint64_t *tar_bytes);

#ifdef __cplusplus
}
#endif
#endif
