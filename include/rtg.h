/*
 * rtg.h -- C ABI of the MI355X (gfx950) ray-cast / BVH-traversal path.
 *
 * This is the drop-in boundary for the per-pixel ray-cast path of
 * nathdobson/raytracer.  The reference has no FFI of its own; its seams are the
 * Rust generics `Object::raycast` (src/render/object.rs:33-35), the frame API
 * `Scene`/`Renderer` (src/render/renderer.rs:44-53,106,115,379) and the flat
 * `Bvh` (src/tree/bvh.rs:14-24,80,107,134).  Every entry point below names the
 * reference interface it replaces.  A Rust `extern "C"` block can bind this
 * header 1:1: plain pointers, sizes and PODs only, no C++ or torch types.
 *
 * Conventions
 *  - every function returns 0 (RTG_OK) or a negative RTG_E_* code; the message
 *    is available from rtg_last_error().  The reference panics instead
 *    (e.g. src/render/object.rs:45-46, src/tree/bvh.rs:109); panics must not
 *    cross a C ABI, so they become error codes.
 *    No C++ exception leaves the library either: every entry point stops them
 *    at the boundary (RTG_E_NOMEM for a failed host allocation, RTG_E_INTERNAL
 *    otherwise).
 *  - "no hit" is not an error: object = leaf = RTG_NONE, t = +inf.
 *  - all arithmetic is IEEE binary64 without contraction, in the reference's
 *    operation order, so hit ids and hit times are bit-identical to the CPU
 *    renderer's.
 *  - input arrays are caller-owned host memory, copied at create time.  Output
 *    pointers may be host pointers (the call synchronises and copies back) or
 *    device pointers on the context's GPU (the call is asynchronous on the
 *    context's stream).
 *  - one rtg_ctx per GPU; calls on one ctx are serialised by the caller
 *    (mirrors the single `&mut self` caller of Renderer::render); distinct
 *    contexts may be driven from distinct host threads.
 *  - lifetimes: a scene runs its calls on the context that created it, so
 *    destroy scenes before their context.  A mesh only remembers its GPU: it
 *    may outlive the context that created it, be shared by scenes of other
 *    contexts on the same GPU, and must outlive every scene that uses it.
 */
#ifndef RTG_H
#define RTG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTG_OK 0
#define RTG_E_INVALID (-1)     /* bad argument / malformed tree           */
#define RTG_E_HIP (-2)         /* a HIP runtime call failed               */
#define RTG_E_NOMEM (-3)       /* host or device allocation failed        */
#define RTG_E_NODEVICE (-4)    /* no usable gfx950 device                 */
#define RTG_E_UNSUPPORTED (-5) /* valid request the build does not cover  */
#define RTG_E_LIMIT (-6)       /* size beyond a documented maximum        */
#define RTG_E_INTERNAL (-7)    /* a C++ exception other than bad_alloc was stopped at the boundary */

#define RTG_NONE 0xFFFFFFFFu

typedef struct rtg_ctx rtg_ctx;
typedef struct rtg_mesh rtg_mesh;
typedef struct rtg_scene rtg_scene;

/* ---- plain data ------------------------------------------------------- */

/* One flattened BVH node: `BvhEntry` (src/tree/bvh.rs:14-18).  Children and
 * leaves are index ranges [begin,end) into the node / leaf arrays. */
typedef struct {
    double lo[3];
    double hi[3];
    uint32_t child_begin, child_end;
    uint32_t leaf_begin, leaf_end;
} rtg_bvh_node;

/* One leaf: the three vertex positions of a `TriVerts` / `Triangle`
 * (src/geo/triangle.rs:7-9). */
typedef struct {
    double p[3][3];
} rtg_tri;

/* A host-flattened `Bvh` (src/tree/bvh.rs:20-24).  `vertex_normals` is either
 * NULL or n_leaves x 3 x 3 doubles used for the interpolated normal. */
typedef struct {
    const rtg_bvh_node *nodes;
    uint64_t n_nodes;
    const rtg_tri *leaves;
    uint64_t n_leaves;
    const double *vertex_normals;
    uint32_t root;
} rtg_bvh;

/* One cubic triangular patch, `Trispline` (src/trispline.rs:55-66): the ten
 * power-basis coefficients of position(t,u) = sum m_ij t^i u^j in the order
 * m00, m01, m02, m03, m10, m11, m12, m20, m21, m30; domain t,u >= 0, t+u <= 1. */
typedef struct {
    double m[10][3];
} rtg_trispline;

/* What a BVH leaf of a patch mesh stands for: a flat facet of patch `patch`
 * whose three corners sit at the (t,u) parameters `tu` (the facet's vertex
 * positions are the rtg_tri of the same leaf). */
typedef struct {
    double tu[3][2];
    uint32_t patch;
    uint32_t _pad;
} rtg_patch_leaf;

/* A set of patches behind a facet BVH.  The reference evaluates patches but has
 * no ray/patch intersection (src/trispline.rs:161-181), so the intersection is
 * this library's definition: a leaf's flat facet is tested like any triangle
 * (src/geo/triangle.rs:54-82); when it is hit, (t,u,time) start from the
 * facet's barycentric interpolation and hit time and take `newton_steps`
 * Newton steps on position(t,u) = orig + dir*time (Jacobian from ddt, ddu,
 * -dir, solved by Cramer's rule); the hit is accepted when the final point is
 * inside the patch domain, time >= 0 and the squared residual is at most
 * 1e-20 * (|m10|^2 + |m01|^2).  Normals are normalize(ddt x ddu). */
typedef struct {
    const rtg_trispline *patches;
    uint64_t n_patches;
    const rtg_patch_leaf *leaf_info; /* one per BVH leaf, in leaf order */
    uint32_t newton_steps;
} rtg_patch_set;

/* `Material` (src/render/material.rs:3-7). */
typedef struct {
    double diffuse[3];
    double n1, n2;          /* dielectric indices, used when has_dielectric */
    int32_t has_dielectric;
    int32_t _pad;
} rtg_material;

/* `Transform` (src/geo/transform.rs:16-21), the three matrices the ray path
 * reads; row-major. */
typedef struct {
    double forward[16];
    double reverse[16];
    double forward_norm[9];
} rtg_transform;

enum { RTG_SPHERE = 0, RTG_MESH = 1, RTG_PLANE = 2, RTG_PATCHES = 3 };

/* One `TransformObject<AnyObject>` (src/render/transform_object.rs:6-9,
 * src/render/any_object.rs:9-13). */
typedef struct {
    int32_t kind;
    int32_t _pad;
    rtg_transform transform;
    rtg_material material;   /* sphere, mesh, patches; plane: material1 */
    rtg_material material2;  /* plane only (src/render/plane_object.rs:8-14) */
    double sphere_center[3]; /* `Sphere` (src/geo/sphere.rs:14-17) */
    double sphere_radius;
    double plane_position[3];
    double plane_tan1[3];
    double plane_tan2[3];
    rtg_mesh *mesh;          /* kind MESH / PATCHES: a handle from rtg_mesh_create */
} rtg_object;

/* `SceneObject` (src/render/scene_object.rs:9-11): objects in vector order. */
typedef struct {
    const rtg_object *objects;
    uint32_t n_objects;
} rtg_scene_desc;

/* `View` (src/geo/view.rs:6-9) with tan(fov/2) evaluated by the host libm. */
typedef struct {
    double orig[3];
    double tan_half_fov[2];
} rtg_view;

/* `Light` (src/render/renderer.rs:38-42): sphere.orig and color; the light
 * radius is stored by the reference but never read by shading. */
typedef struct {
    double position[3];
    double color[3];
} rtg_light;

typedef struct {
    uint32_t x0, y0, w, h;
} rtg_rect;

/* Compact hit id: `Manifold{array:[leaf, object]}` + `time`
 * (src/render/object.rs:12-16,20). */
typedef struct {
    uint32_t object;
    uint32_t leaf;
    double t;
} rtg_hit;

/* Full `RaycastPoint` (src/render/object.rs:18-27).  manifold_len == 0 marks
 * "no hit" (every scene-level hit has at least the object index pushed). */
typedef struct {
    double time;
    double position[3];
    double inter_normal[3];
    double geo_normal[3];
    double manifold_point[2];
    uint32_t manifold[2];
    uint32_t manifold_len;
    uint32_t material_slot; /* object*2 + (1 if the plane's material2) */
} rtg_point;

/* Per-ray traversal work, for the algorithmic-bytes figure. */
typedef struct {
    uint32_t nodes_visited;
    uint32_t tris_tested;
} rtg_ray_counters;

/* One `SpecularPath` (src/render/renderer.rs:93-103): the hit a ray of a batch reaches after `depth` specular
 * bounces.  `modes` bit i is set when bounce i was `SpecularMode::Refract`, clear for `Reflect`
 * (`modes.len()` = depth); `attenuation` is the product of the Fresnel factors along the way; the reference's
 * `manifolds` are the `point.manifold` of the ancestors, found by following `parent` (index in the same output
 * array of the path that ends on the surface this one bounced off last; RTG_NONE at depth 0). */
typedef struct {
    uint32_t ray;
    uint32_t depth;
    uint32_t modes;
    uint32_t parent;
    double attenuation;
    double _pad;
    rtg_point point;
} rtg_path;

/* ---- context ---------------------------------------------------------- */

/* No reference counterpart (the reference is CPU-only): binds a HIP device. */
int rtg_ctx_create(int hip_device, rtg_ctx **out);
void rtg_ctx_destroy(rtg_ctx *ctx);
/* Message for the last failing call on this ctx (ctx == NULL: last failing
 * create on this thread).  Replaces the reference's panic messages. */
const char *rtg_last_error(rtg_ctx *ctx);
/* Launch on a caller-owned hipStream_t (NULL restores the ctx's own stream). */
int rtg_ctx_set_stream(rtg_ctx *ctx, void *hip_stream);
void *rtg_ctx_get_stream(rtg_ctx *ctx);
int rtg_ctx_synchronize(rtg_ctx *ctx);
/* Options (no reference counterpart; tuning and A/B switches):
 *   "traversal" 0 = one ray per lane, short stack (VGPR window + LDS spill + global overflow)
 *               1 = one ray per lane, threaded pre-order tree (skip links, no stack)
 *               2 = persistent phase-scheduled waves over the threaded tree
 *   "tri_min" / "obj_min"               scheduling thresholds of traversal 2 (lanes)
 *   "patch_tri_min"                     lanes waiting for a patch leaf's Newton solve before one runs
 *   "max_blocks"                        persistent grid size of traversal 2 in four-wave groups (default: what the
 *                                       GPU holds at once, 1280 for single-mesh scenes, 1024 otherwise)
 *   "pretest" 2 = on (default), 1 = off: single-precision box pre-test of traversal 2
 *   "stats" 2 = tally steps per phase in traversal 2 (diagnostics), 1 = off
 *   "comm_loopback" 2 = a communicator created on the context afterwards goes through RCCL even with one rank (it
 *                       needs an id then) and the root's own pieces travel by ncclSend / ncclRecv to itself: the
 *                       several-rank machinery on a single GPU (diagnostics); 1 = off (default)
 *   "keep_state" 2 = traversal 2 runs the build made for scenes with several meshes (per-lane tree address, outer ray
 *                    kept) on any scene -- what a mesh of 2^27 nodes or more gets by itself; 1 = off (default)
 *   "frame_bands" 2 = rtg_render_frame issues an opaque scene's frame as two row bands on two streams, so one
 *                     band's launches run while the other's drain (default), 1 = one band
 *   "sort_rays" 2 = rtg_trace (traversal 2) puts the batch into coherence order on the device first (key = direction
 *                   octant | 30-bit Morton code of the origin; one radix sort), traces it and returns the answers in
 *                   the caller's order; 1 = off (default: for one trace of a batch the sort costs more than it saves)
 *   "photon_trig" 2 = the photon pass takes the cosine and sine of every direction's zenith with the GPU's own
 *                     routines (ZenithY::into_normal, src/geo/sphere.rs:29-34): directions may then be device memory
 *                     and nothing but 16 bytes per direction crosses PCIe; photon positions agree with host mode to
 *                     about 1e-15 relative, not bit for bit (the last place of sin / cos differs between libraries,
 *                     as it does between the platforms the reference runs on); 1 = host libm (default)
 *   "photon_batch" 12..24 = log2 of the photon directions handled per batch (default 23): a batch holds 140 bytes per
 *                     direction in scratch memory and costs a fixed number of launches and waits for the host
 *   "kd_finish" 2 = the photon map's k-d tree build finishes ranges of at most 1024 photons in one launch, a workgroup
 *                   per range (default); 1 = every level as device-wide passes (same tree, slower: for comparison)
 *   "time_kernels" 2 = record each persistent-kernel launch's execution window (rtg_ctx_kernel_time), 1 = off */
int rtg_ctx_set_option(rtg_ctx *ctx, const char *key, int64_t value);
/* With option "time_kernels" = 2 every persistent-kernel launch behind rtg_render_primary / rtg_trace records its
 * execution window (the dispatch's own begin and end timestamps: the duration a profiler reports for the kernel,
 * also while launches of other contexts share the GPU).  This call waits for the context's stream, returns the summed
 * windows and the number of launches since the previous call, and starts over. */
int rtg_ctx_kernel_time(rtg_ctx *ctx, double *total_ms, uint64_t *launches);
/* Diagnostics for traversal 2 after a launch with "stats" on: loop iterations,
 * node steps, node lanes, tri steps, tri lanes, obj steps, obj lanes, refills,
 * waves (no reference counterpart). */
int rtg_ctx_debug_stats(rtg_ctx *ctx, uint32_t out[16]);

/* Self-test (no reference counterpart): compares the reciprocal-based quotient
 * the box test of traversal 2 uses with the hardware IEEE division on n_pairs
 * generated (x, d) pairs of one family (0 random, 1 small integers, 2 quotients
 * next to representable numbers, 3 special denominators).  out[0] = number of
 * pairs whose quotients differ in any bit, out[1], out[2] = bits of one such pair. */
int rtg_selftest_division(rtg_ctx *ctx, uint64_t seed, uint64_t n_pairs, uint32_t family, uint64_t out[3]);

/* Restates `Transform::from(Mat4)` (src/geo/transform.rs:51-58) with the
 * inverse formulas of src/math/mat.rs:80-87,100-154 for callers that only hold
 * the forward matrix. */
int rtg_transform_from_forward(const double forward[16], rtg_transform *out);

/* ---- geometry --------------------------------------------------------- */

/* Uploads a flattened `Bvh`; replaces holding an `Arc<Bvh>`
 * (src/render/mesh_object.rs:10-13).  The tree is validated (ranges in
 * bounds, acyclic, every node reachable at most once). */
int rtg_mesh_create(rtg_ctx *ctx, const rtg_bvh *bvh, rtg_mesh **out);
/* (every node must be reachable from `root`, as in a tree made by Bvh::new) */
void rtg_mesh_destroy(rtg_mesh *mesh);
/* A facet BVH whose leaves stand for trispline patches (object kind
 * RTG_PATCHES); `bvh` is the flattened tree over the facets, built like any
 * mesh.  No reference counterpart for the intersection (see rtg_patch_set). */
int rtg_mesh_create_patches(rtg_ctx *ctx, const rtg_bvh *bvh, const rtg_patch_set *patches, rtg_mesh **out);
/* Builds the tree on the GPU: replaces `BvhForest::new(&tris).subdivide()` followed by `Bvh::new`
 * (src/tree/bvh.rs:48-58,80-106; src/tree/bvh_subdivide.rs:8-43).  The tree is the one those functions produce,
 * node for node (same splits, same leaf order, same boxes), so hit leaf indices mean the same as with a
 * host-flattened tree; `rtg_mesh_read_bvh` returns it.  `triangles` (and the optional `vertex_normals`,
 * n x 3 x 3 doubles) may be host or device memory and are not retained.  Every vertex must be finite (the
 * reference panics on a NaN sort key, bvh_subdivide.rs:28): RTG_E_INVALID otherwise. */
int rtg_mesh_build(rtg_ctx *ctx, const rtg_tri *triangles, uint64_t n_triangles, const double *vertex_normals,
                   rtg_mesh **out);
/* rtg_mesh_build for a patch set: `facets` are the flat facets in any order and `patches->leaf_info[i]` describes
 * facet i; the records follow their facets into leaf order. */
int rtg_mesh_build_patches(rtg_ctx *ctx, const rtg_tri *facets, uint64_t n_facets, const rtg_patch_set *patches,
                           rtg_mesh **out);
/* Node and leaf counts, root index (`Bvh::root`) and depth of a mesh's flattened tree; any pointer may be NULL. */
int rtg_mesh_sizes(const rtg_mesh *mesh, uint64_t *n_nodes, uint64_t *n_leaves, uint32_t *root, uint32_t *depth);
/* Copies a mesh's flattened tree back to host arrays sized by rtg_mesh_sizes (blocking; waits for the GPU's
 * outstanding work; failures are reported through rtg_last_error(NULL)): `nodes` in the order of
 * `Bvh::nodes`, `leaves` in the order of `Bvh::leaves` (src/tree/bvh.rs:20-24), `leaf_source[l]` = index of leaf
 * l's triangle in the array given to rtg_mesh_build (the identity for rtg_mesh_create).  Any pointer may be NULL. */
int rtg_mesh_read_bvh(const rtg_mesh *mesh, rtg_bvh_node *nodes, rtg_tri *leaves, uint32_t *leaf_source);
/* `Bvh::bounds` (src/tree/bvh.rs:134-136): lo[3], hi[3] of the root. */
int rtg_mesh_bounds(const rtg_mesh *mesh, double out_lo_hi[6]);

/* Replaces `SceneObject::new` (src/render/scene_object.rs:13-17).  The meshes may come from any context of the
 * same GPU: contexts are how a caller keeps several frames in flight (one stream each), and they can share one
 * uploaded tree. */
int rtg_scene_create(rtg_ctx *ctx, const rtg_scene_desc *desc, rtg_scene **out);
void rtg_scene_destroy(rtg_scene *scene);

/* ---- the hot path ----------------------------------------------------- */

/* Primary rays of a pixel rectangle: replaces the pixel pass
 * `pixels.into_par_iter().map(render_pixel)` (src/render/renderer.rs:141-160)
 * down to `scene_object.raycast(ray, None)` (renderer.rs:316).  Outputs are
 * row-major over the tile: index (y - y0) * w + (x - x0).  out_points and
 * out_counters may be NULL. */
int rtg_render_primary(rtg_scene *scene, const rtg_view *view, uint32_t width, uint32_t height,
                       rtg_rect tile, rtg_hit *out_hits, rtg_point *out_points,
                       rtg_ray_counters *out_counters);

/* rtg_render_primary for callers that cannot allocate.  It has no counterpart in the reference; it is for callers
 * that cannot call hipMalloc: code under HIP graph capture, tight serving loops, harnesses that forbid allocation.
 * Same hit ids and times as rtg_render_primary, with these differences: `out_hits` must be device memory on the
 * scene's GPU (16-byte aligned); the tile may hold at most RTG_STATIC_MAX_PIXELS pixels (RTG_E_LIMIT beyond);
 * always the default (persistent) traversal; the call neither allocates nor frees device or host memory and is
 * asynchronous on the context's stream.  The kernels' scratch (work counters, closest sphere/plane hit per pixel:
 * 32 MiB + 2 KiB) is a fixed array inside the library, one copy per GPU, shared by every call on that GPU: calls
 * that overlap on one GPU (two contexts or streams) race -- serialise them.  (The HIP runtime itself sets a queue up
 * on the first kernel launch of a stream, the kernels' private segment included; that happens once per stream,
 * whichever entry point launches first, so a caller that must not see it warms the stream up with one small call.) */
#define RTG_STATIC_MAX_PIXELS (1u << 21)
int rtg_render_primary_static(rtg_scene *scene, const rtg_view *view, uint32_t width, uint32_t height,
                              rtg_rect tile, rtg_hit *out_hits);

/* Batched `Object::raycast(ray, None)` (src/render/object.rs:33-35,
 * src/render/scene_object.rs:19-36) for arbitrary rays: rays6 holds
 * orig.xyz, dir.xyz per ray.  out_points / out_counters may be NULL. */
int rtg_trace(rtg_scene *scene, const double *rays6, uint64_t n_rays, rtg_hit *out_hits,
              rtg_point *out_points, rtg_ray_counters *out_counters);

/* Batched `Renderer::raytrace_all_specular(&ray, &[], None)` (src/render/renderer.rs:284-378): every hit of every
 * ray's specular path tree (reflection and refraction at each dielectric hit, at most `max_depth` surface
 * interactions per path; the reference's constant is 4, renderer.rs:312).  Paths are written in ray order and,
 * within a ray, in the order the recursion emits them.  What the photon pass (renderer.rs:119-138) and, with
 * photons enabled, the pixel pass (renderer.rs:266-278) iterate over.  `capacity` is the room in out_paths;
 * *n_paths receives the number of paths; when it exceeds `capacity` nothing is written and RTG_E_LIMIT is
 * returned (n_rays * (2^max_depth - 1) always suffices). */
int rtg_trace_paths(rtg_scene *scene, const double *rays6, uint64_t n_rays, uint32_t max_depth,
                    rtg_path *out_paths, uint64_t capacity, uint64_t *n_paths);

/* `compute_direct_irrad` x `material.diffuse` (src/render/renderer.rs:184-205,
 * 270-277) for n hit points (one shadow ray per light, closest-hit query).
 * rgb_out: 3 doubles per point, rgba8_out (optional): PQ-encoded 8-bit
 * (src/geo/color.rs:15-17 applied to c/80, src/render/image.rs:39). */
int rtg_shade_direct(rtg_scene *scene, const rtg_light *lights, uint32_t n_lights,
                     const rtg_point *points, uint64_t n_points, double *rgb_out,
                     uint8_t *rgba8_out);

/* Fused frame: ray generation, closest hit, the specular path tree up to
 * `max_depth` surface interactions (src/render/renderer.rs:284-378; depth 1 =
 * primary hit only, the reference uses 4) and direct shading, i.e.
 * `raytrace_pixel` (renderer.rs:266-283) with the photon term disabled.
 * Any of out_rgb / out_rgba8 / out_hits may be NULL. */
int rtg_render_frame(rtg_scene *scene, const rtg_view *view, const rtg_light *lights,
                     uint32_t n_lights, uint32_t width, uint32_t height, rtg_rect tile,
                     uint32_t max_depth, double *out_rgb, uint8_t *out_rgba8, rtg_hit *out_hits);

/* ---- photon map (indirect light through specular surfaces) -------------- */

/* One stored photon: `Photon` and the position of its `KdEntry` (src/render/renderer.rs:63-71,129-136): a path
 * from a light that met at least one specular surface before it landed. */
typedef struct {
    double position[3];         /* path.raycast_point.position: the kd-tree key */
    double origin[3];           /* Photon::origin: the light's centre */
    double dir[2];              /* Photon::dir, `ZenithY` (zenith, y) of the emission direction (src/geo/sphere.rs:19-33) */
    double light[3];            /* light.color * path.attenuation */
    uint32_t surface_object[3]; /* Photon::manifold: the surfaces the path bounced off, in order -- object index ... */
    uint32_t surface_leaf[3];   /* ... and mesh leaf index (RTG_NONE for spheres and planes) */
    uint32_t n_bounces;         /* Photon::modes.len() == manifold.len(): 1 .. max_depth - 1 (at most 3) */
    uint32_t modes;             /* bit i set: bounce i was SpecularMode::Refract */
    uint32_t light_index;
    uint32_t _pad;
} rtg_photon;

typedef struct rtg_photon_map rtg_photon_map;

/* The photon pass of `Renderer::render` (src/render/renderer.rs:116-138): for every light l and every direction
 * dirs2[(l * count_per_light + i) * 2 ..] (a `ZenithY`; the reference draws them with `Sphere::fibonacci_sphere`
 * from an entropy-seeded generator, so the caller supplies them) the ray from the light's centre is followed
 * through `raytrace_all_specular` and every hit behind at least one specular bounce becomes a photon.  Directions
 * are turned into rays on the host (`ZenithY::into_normal` uses the platform's sin/cos, as the reference does); the
 * paths are traced on the GPU.  max_depth is the reference's constant 4 (at most 4).  out_photons may be
 * host or device memory.  When more than `capacity` photons result, RTG_E_LIMIT is returned and the buffer's contents
 * are unspecified (the batches that still fitted have been written); *n_photons is set either way, so a second call
 * with that much room succeeds. */
int rtg_photons_emit(rtg_scene *scene, const rtg_light *lights, uint32_t n_lights, const double *dirs2,
                     uint64_t count_per_light, uint32_t max_depth, rtg_photon *out_photons, uint64_t capacity,
                     uint64_t *n_photons);

/* `KdTree::new(photons)` (src/tree/kd_tree.rs:74-95): a balanced k-d tree over the photons' positions, kept on the
 * GPU together with the photon records.  `photons` is host or device memory and is copied; the tree is built on
 * the GPU (the reference's selection is unstable, so which of several photons with equal coordinates becomes a
 * range's root is as unspecified here as there).  A NaN position is refused (RTG_E_INVALID), as the reference's
 * `NotNan` keys are. */
int rtg_photon_map_create(rtg_ctx *ctx, const rtg_photon *photons, uint64_t n_photons, rtg_photon_map **out);
/* The photon pass and `KdTree::new` in one call, `self.photons = KdTree::new(photons)` of `Renderer::render`
 * (src/render/renderer.rs:116-140): rtg_photons_emit followed by rtg_photon_map_create, with the photon records
 * never leaving the GPU (a map's photon indices, as rtg_photon_map_nearest reports them, count the records in the
 * order rtg_photons_emit would have written them).  *n_photons (may be NULL) receives the number of photons. */
int rtg_photon_map_emit(rtg_scene *scene, const rtg_light *lights, uint32_t n_lights, const double *dirs2,
                        uint64_t count_per_light, uint32_t max_depth, rtg_photon_map **out, uint64_t *n_photons);
/* The same with the directions made on the GPU as well: `Sphere::fibonacci_sphere(photon_count, rng)` per light
 * (src/geo/sphere.rs:52-80, src/render/renderer.rs:116-118), batch by batch, so that nothing of the pass crosses
 * PCIe.  The reference draws the per-direction jitter from an entropy-seeded generator (renderer.rs:112); here it
 * comes from a SplitMix64 stream seeded with `seed` + l * 0x9E3779B97F4A7C15 for light l, direction k taking draws
 * 2k and 2k + 1 -- the (zenith, y) pairs are bit for bit those the host generator of this build makes from the same
 * seed, and their cosines and sines are the GPU's (as with option "photon_trig" = 2). */
int rtg_photon_map_emit_fibonacci(rtg_scene *scene, const rtg_light *lights, uint32_t n_lights, uint64_t count_per_light,
                                  uint64_t seed, uint32_t max_depth, rtg_photon_map **out, uint64_t *n_photons);
void rtg_photon_map_destroy(rtg_photon_map *map);

/* `KdTree::nearest(&p, k)` (src/tree/kd_tree.rs:115-154) for a batch of query points: out_index[q * k + j] is the
 * index (in the array given to rtg_photon_map_create) of the j-th nearest photon of query q, out_distance its
 * distance, ascending; slots beyond the number of photons hold RTG_NONE / +inf.  k <= 16.  Which of several
 * photons at exactly the same distance is reported is not defined (in the reference it depends on the heap). */
int rtg_photon_map_nearest(rtg_photon_map *map, const double *queries3, uint64_t n_queries, uint32_t k,
                           uint32_t *out_index, double *out_distance);
/* The tree as the reference holds it: `KdTree` is its entry array in tree order (src/tree/kd_tree.rs:74-95,
 * src/tree/seq_tree.rs:36-49: the root of a range sits in its middle, the two halves on either side).  out_index[p]
 * (n_photons values, host or device memory) is the index of the photon at position p of that array. */
int rtg_photon_map_order(rtg_photon_map *map, uint32_t *out_index);

/* `compute_indirect_irrad` (src/render/renderer.rs:206-265) for n surface points: for each of the photon_samples
 * nearest photons the emission direction is corrected by newton_steps Newton steps so that the photon's path --
 * followed through the same surfaces with the same reflect / refract choices, in forward-mode dual numbers
 * (`Der<2>`, src/math/scalar.rs:72-128) -- lands on the point's own surface coordinates; the corrected photon is
 * traced through the whole scene, kept if it lands within newton_epsilon of the point, and contributes
 * light / (4 pi |dP/da x dP/db|).  One photon per light (the last of the nearest ones), summed in light order.
 * rgb_out: 3 doubles per point (zeros for points with manifold_len 0).  Scenes with patch sets are not supported.
 * sin / cos of the direction are the GPU's, so results agree with a CPU evaluation to rounding, not bit for bit. */
int rtg_indirect_irrad(rtg_scene *scene, rtg_photon_map *map, const rtg_point *points, uint64_t n_points,
                       uint32_t photon_samples, uint32_t newton_steps, double newton_epsilon, double *rgb_out);

/* `raytrace_pixel` (src/render/renderer.rs:266-283) with the photon term: per pixel of the tile, the sum over the
 * camera ray's specular paths of (indirect + direct + ambient) x diffuse x attenuation -- rtg_render_frame plus
 * rtg_indirect_irrad at every path's hit point.  max_depth 1..4 (the reference's constant is 4).  Pixels whose paths
 * receive no photon are bit-identical to rtg_render_frame's.  Any one of out_rgb / out_rgba8 may be NULL. */
int rtg_render_frame_photons(rtg_scene *scene, const rtg_view *view, const rtg_light *lights, uint32_t n_lights,
                             rtg_photon_map *map, uint32_t photon_samples, uint32_t newton_steps, double newton_epsilon,
                             uint32_t width, uint32_t height, rtg_rect tile, uint32_t max_depth, double *out_rgb,
                             uint8_t *out_rgba8);

/* ---- several GPUs ----------------------------------------------------- */

/* The path has no exchange during compute: the scene is replicated and pixels / frames are independent
 * (src/render/renderer.rs:147-149 maps over `&self`; the reference itself fans frames out to HTTP workers and
 * collects their tar files, scripts/request_all.sh:2, src/bin/server.rs:19-33).  One process (or host thread) and
 * one rtg_ctx per GPU; the only data that moves is finished output, gathered to one GPU per frame with RCCL
 * point-to-point transfers over xGMI.  RCCL is bound at run time; a one-rank communicator needs none. */
typedef struct rtg_comm rtg_comm;
#define RTG_COMM_ID_BYTES 128
/* Rank 0 makes the id (RCCL's unique id) and hands it to the other ranks by any channel the host has. */
int rtg_comm_unique_id(uint8_t id[RTG_COMM_ID_BYTES]);
/* Collective over the n_ranks callers; `id` may be NULL when n_ranks is 1.  The communicator belongs to `ctx`'s GPU
 * and must be destroyed before the context. */
int rtg_comm_create(rtg_ctx *ctx, const uint8_t id[RTG_COMM_ID_BYTES], int n_ranks, int rank, rtg_comm **out);
/* Wraps an ncclComm_t the host already has (not destroyed by rtg_comm_destroy).  The RCCL that made it must be the
 * copy this process already carried when the library first bound RCCL (RTG_E_UNSUPPORTED otherwise), and it must
 * report n_ranks ranks. */
int rtg_comm_adopt(rtg_ctx *ctx, void *nccl_comm, int n_ranks, int rank, rtg_comm **out);
void rtg_comm_destroy(rtg_comm *comm);
/* What a communicator is: its rank count and this caller's rank as given at creation, and the rank count RCCL's own
 * communicator reports (ncclCommCount; 0 when a single rank gathers by device copies and no RCCL communicator exists).
 * No reference counterpart: it lets a multi-GPU driver (the reference's fan-out script, scripts/request_all.sh:2)
 * record that the transport really spans the ranks it believes it does.  Any output may be NULL. */
int rtg_comm_info(const rtg_comm *comm, int *n_ranks, int *rank, int *rccl_ranks);

/* One piece of a gathered frame: `bytes` bytes that rank `rank` holds at `send_offset` of its send buffer and that
 * land at `recv_offset` of the root's receive buffer (a full-width strip of rows lands where it sits in the frame,
 * so the gather assembles the image). */
typedef struct {
    uint32_t rank;
    uint32_t _pad;
    uint64_t send_offset;
    uint64_t recv_offset;
    uint64_t bytes;
} rtg_piece;

/* Gathers finished output to rank `root`: replaces collecting the workers' responses (scripts/request_all.sh:2).
 * Every rank passes the same piece list; a rank sends its own pieces in list order, the root receives them in
 * list order and copies its own pieces on the device.  `send` / `recv` are device memory (`recv` is only read on
 * the root).  Enqueued on `hip_stream` (NULL: the context's stream), so it overlaps work on other streams. */
int rtg_gather_tiles(rtg_comm *comm, int root, const void *send, void *recv, const rtg_piece *pieces,
                     uint32_t n_pieces, void *hip_stream);

/* Library build info: "rtg <version> gfx950". */
const char *rtg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RTG_H */
