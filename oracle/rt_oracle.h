/*
 * rt_oracle.h -- C API of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  The oracle is a plain-double C++ restatement of
 * the reference renderer's ray-cast path (nathdobson/raytracer, Rust).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 * the product library (librtg.so) neither links nor calls it.
 *
 * PARITY UNPINNED for the ray routines: the Rust reference cannot be built
 * here (no rustc/cargo, src/mesh/ missing from the snapshot) and it ships no
 * test or fixture for ray/box, ray/triangle, ray/sphere, ray/plane, traversal
 * or shading.  The oracle is pinned only by the reference's own unit tests
 * that touch the path (src/math/mat.rs:223-253 matrix inverses,
 * src/tree/bv_kd_tree.rs:80-95 union + surface area, src/geo/color.rs:19-30
 * PQ round trip) and by hand-derived known answers.  The one output the
 * reference keeps of the path, saved/sphere.jpg (its default scene, photon
 * term included), is a lossy picture: the oracle's frame of that scene agrees
 * with it pixel for pixel in structure and in where the light falls
 * (tests/test_oracle_picture.py), which no bit-exact claim can rest on.
 *
 * Data formats (rtg_bvh_node, rtg_tri, rtg_object, rtg_view, rtg_light,
 * rtg_hit, rtg_point, ...) are the PODs of include/rtg.h so the same buffers
 * can be handed to the oracle and to the GPU library.  In an rtg_object given
 * to the oracle, `mesh` holds an orc_bvh* instead of an rtg_mesh*.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rtg.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_bvh orc_bvh;
typedef struct orc_scene orc_scene;

/* ---- scalar / small known-answer entry points -------------------------- */
double orc_minimum(double a, double b);
double orc_maximum(double a, double b);
int orc_real_cmp(double a, double b);
double orc_dot3(const double a[3], const double b[3]);
void orc_cross(const double a[3], const double b[3], double out[3]);
void orc_normalize3(const double a[3], double out[3]);
void orc_mat2_inverse(const double m[4], double out[4]);
void orc_mat3_inverse(const double m[9], double out[9]);
void orc_mat4_inverse(const double m[16], double out[16]);
void orc_mat4_mul(const double a[16], const double b[16], double out[16]);
void orc_mat3_mul(const double a[9], const double b[9], double out[9]);
void orc_mat2_mul(const double a[4], const double b[4], double out[4]);
void orc_mat4_from_scale_rotation_translation(const double scale[3], const double translate[3],
                                              double out[16]);
void orc_transform_from_forward(const double forward[16], rtg_transform *out);
void orc_bounds_union(const double a[6], const double b[6], double out[6]);
double orc_bounds_surface_area(const double b[6]);
double orc_smpte2048_encode(double x);
double orc_smpte2048_decode(double x);
uint8_t orc_pq8(double c);
/* returns the number of roots (0,1,2) written ascending */
int orc_find_roots_quadratic(double a2, double a1, double a0, double out[2]);

/* ---- primitives -------------------------------------------------------- */
/* returns 1 and writes (lo,hi) when the slab test yields Some(interval) */
int orc_bounds_raycast(const double box_lo_hi[6], const double ray6[6], double out_lo_hi[2]);
/* out: t, bary[3], q[3], mp[2], normal[3] (12 doubles) */
int orc_triangle_raycast(const double tri9[9], const double ray6[6], int manifold, double out12[12]);
int orc_sphere_raycast(const double center[3], double radius, const double ray6[6], rtg_point *out);
int orc_plane_raycast(const double position[3], const double tan1[3], const double tan2[3],
                      const double ray6[6], rtg_point *out, int *out_material2);
void orc_view_get_ray(const rtg_view *view, const double s[2], double out_ray6[6]);
void orc_pixel_ndc(uint32_t x, uint32_t y, uint32_t width, uint32_t height, double out_s[2]);
/* Dielectric::new: out = reflectance, reflect[3], has_refract, refract[3] (8 doubles) */
void orc_dielectric(const double inc[3], const double signed_norm[3], double n1, double n2,
                    double out8[8]);

/* ---- BVH --------------------------------------------------------------- */
/* BvhForest::new(tris).subdivide() then Bvh::new(&tree). */
orc_bvh *orc_bvh_build_subdivide(const rtg_tri *tris, uint64_t n, const double *vertex_normals);
/* strategy 0: subdivide; 1: BvhForest::contract (tree/bvh_cluster.rs:43-48); 2: BvhForest::hybrid
 * (tree/bvh_joint.rs:3-8).  The pairing is evaluated by its definition in O(n^3): small inputs only. */
orc_bvh *orc_bvh_build(const rtg_tri *tris, uint64_t n, const double *vertex_normals, int strategy);
/* Wrap already-flattened arrays (copied). */
orc_bvh *orc_bvh_from_flat(const rtg_bvh *flat);
void orc_bvh_destroy(orc_bvh *b);
uint64_t orc_bvh_n_nodes(const orc_bvh *b);
uint64_t orc_bvh_n_leaves(const orc_bvh *b);
uint32_t orc_bvh_root(const orc_bvh *b);
const rtg_bvh_node *orc_bvh_nodes(const orc_bvh *b);
const rtg_tri *orc_bvh_leaves(const orc_bvh *b);
const double *orc_bvh_vertex_normals(const orc_bvh *b);
/* source triangle index of each leaf (leaf order = traversal order) */
const uint32_t *orc_bvh_leaf_source(const orc_bvh *b);
uint32_t orc_bvh_depth(const orc_bvh *b);
/* Bvh::raycast(ray, None) in the mesh's own space; returns 1 on hit */
int orc_bvh_raycast(const orc_bvh *b, const double ray6[6], rtg_hit *out, rtg_ray_counters *ctr);
/* every leaf tested in index order with the holder's strict-< rule (no boxes) */
int orc_bvh_raycast_brute(const orc_bvh *b, const double ray6[6], rtg_hit *out);

/* ---- trispline patches (src/trispline.rs) -------------------------------- */
/* in30: p0,p1,p2,p01,p12,p20,n0,n1,n2,control (NormalHermiteTrispline, trispline.rs:24-36) */
void orc_trispline_from_normal_hermite(const double in30[30], rtg_trispline *out);
/* out12 = position, ddt, ddu, normal at (t,u) */
void orc_trispline_eval(const rtg_trispline *sp, double t, double u, double out12[12]);
/* k*k flat facets per patch with their corner parameters */
void orc_patches_tessellate(const rtg_trispline *patches, uint64_t n, uint32_t k, rtg_tri *out_tris,
                            rtg_patch_leaf *out_info);
/* turns a facet BVH into a patch mesh; ps->leaf_info is in leaf order */
void orc_bvh_set_patches(orc_bvh *b, const rtg_patch_set *ps);

/* ---- scene / renderer -------------------------------------------------- */
orc_scene *orc_scene_create(const rtg_scene_desc *desc);
void orc_scene_destroy(orc_scene *s);
int orc_render_primary(const orc_scene *s, const rtg_view *view, uint32_t width, uint32_t height,
                       rtg_rect tile, rtg_hit *out_hits, rtg_point *out_points,
                       rtg_ray_counters *out_counters, int threads);
int orc_trace(const orc_scene *s, const double *rays6, uint64_t n, rtg_hit *out_hits,
              rtg_point *out_points, rtg_ray_counters *out_counters, int threads);
/* raytrace_all_specular(&ray, &[], None) per ray (renderer.rs:284-378), paths concatenated in ray order. */
int orc_trace_paths(const orc_scene *s, const double *rays6, uint64_t n, uint32_t max_depth, rtg_path *out,
                    uint64_t capacity, uint64_t *n_paths);
int orc_shade_direct(const orc_scene *s, const rtg_light *lights, uint32_t n_lights,
                     const rtg_point *points, uint64_t n, double *rgb_out, uint8_t *rgba8_out,
                     int threads);
int orc_render_frame(const orc_scene *s, const rtg_view *view, const rtg_light *lights,
                     uint32_t n_lights, uint32_t width, uint32_t height, rtg_rect tile,
                     uint32_t max_depth, double *out_rgb, uint8_t *out_rgba8, rtg_hit *out_hits,
                     int threads);

/* ---- photon map: k-d tree, Der<2>, caustic correction (PARITY UNPINNED beyond the k-d tree and Der tests) ---- */
typedef struct orc_kd orc_kd;
/* op: 0 add, 1 sub, 2 mul, 3 div, 4 neg, 5 sqrt, 6 sin, 7 cos, 8 abs, 9 minimum, 10 maximum; Der<2> as (v, d0, d1) */
void orc_der2_op(int op, const double a[3], const double b[3], double out[3]);
int orc_der2_partial_cmp(const double a[3], const double b[3]);
void orc_der2_known_answer(double out[3]);
void orc_zenith_y_into_normal(const double dir2[2], double out[3]);
void orc_fibonacci_sphere(uint64_t count, const double *uniform2, double *out_dirs2);
orc_kd *orc_kd_create(const double *positions3, uint64_t n);
void orc_kd_destroy(orc_kd *kd);
int orc_kd_check(const orc_kd *kd);
void orc_kd_nearest(const orc_kd *kd, const double *queries3, uint64_t n_queries, uint32_t k, uint32_t *out_index,
                    double *out_distance, int threads);
void orc_kd_nearest_brute(const double *positions3, uint64_t n, const double *queries3, uint64_t n_queries, uint32_t k,
                          uint32_t *out_index, double *out_distance, int threads);
int orc_photons_emit(const orc_scene *s, const rtg_light *lights, uint32_t n_lights, const double *dirs2,
                     uint64_t count_per_light, uint32_t max_depth, rtg_photon *out, uint64_t capacity, uint64_t *n_photons);
int orc_indirect_irrad(const orc_scene *s, const orc_kd *kd, const rtg_photon *photons, const rtg_point *points,
                       uint64_t n_points, uint32_t photon_samples, uint32_t newton_steps, double newton_epsilon,
                       uint32_t max_depth, double *rgb_out, int threads);
int orc_render_frame_photons(const orc_scene *s, const rtg_view *view, const rtg_light *lights, uint32_t n_lights,
                             const orc_kd *kd, const rtg_photon *photons, uint32_t photon_samples, uint32_t newton_steps,
                             double newton_epsilon, uint32_t width, uint32_t height, rtg_rect tile, uint32_t max_depth,
                             double *out_rgb, uint8_t *out_rgba8, int threads);
void orc_photon_adjust(const orc_scene *s, const rtg_photon *ph, const rtg_point *p, uint32_t newton_steps, uint32_t max_depth,
                       double *out12, double *out_seps);

#ifdef __cplusplus
}
#endif
#endif
