// rt_scene.hpp -- host-side scene description and frame API over the C ABI.
//
// C++ stand-in for the Rust host code that stays on the CPU: it keeps the
// reference's names and argument meaning (SceneBuilder, Scene, Renderer,
// TransformBuilder, View, Light, Material, SphereObject, PlaneObject,
// MeshObject) so a port of the reference's bins reads the same, and drives the
// GPU only through include/rtg.h.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rtg.h"
#include "rt_bvh.hpp"

namespace rt {

struct Vec3 {
    double x = 0, y = 0, z = 0;
};
using Color = Vec3;

struct Mat4 {
    double m[4][4];
    static Mat4 identity();
    static Mat4 from_scale(Vec3 s);                       // math/mat.rs:161-163
    static Mat4 from_translation(Vec3 t);                 // math/mat.rs:165-171
    static Mat4 from_scale_rotation_translation(Vec3 scale, Vec3 translation);   // math/mat.rs:172-177 (R = I)
    Mat4 mul(const Mat4 &rhs) const;                      // math/mat.rs:185-190
};

// geo/transform.rs:8-58
struct Transform {
    rtg_transform t;
    static Transform from(const Mat4 &forward);
    static Transform default_();
};
struct TransformBuilder {
    Vec3 translate_{0, 0, 0}, scale_{1, 1, 1};
    TransformBuilder &scale(double amount);
    TransformBuilder &translate(double x, double y, double z);
    TransformBuilder &rotate();
    Transform build() const;
};

// render/material.rs:3-7
struct Material {
    Color diffuse;
    bool has_dielectric = false;
    double n1 = 0, n2 = 0;
    rtg_material c() const;
};

// geo/view.rs:6-12
struct View {
    Vec3 orig;
    double fov[2];
    static View from_fov(Vec3 orig, double fov_x, double fov_y);
    rtg_view c() const;   // evaluates tan(fov/2) with the host libm
};

// geo/sphere.rs:14-17, render/renderer.rs:38-42
struct Sphere {
    Vec3 orig;
    double rad = 0;
};
struct Light {
    Sphere sphere;
    Color color;
};

// A mesh resident on one GPU: the `Arc<Bvh>` of render/mesh_object.rs:10-13.
struct GpuMesh {
    rtg_mesh *handle = nullptr;
    Bvh bvh;
    ~GpuMesh();
};

// One TransformObject<AnyObject> (render/transform_object.rs, any_object.rs).
struct Object {
    rtg_object c{};
    std::shared_ptr<GpuMesh> mesh;
    static Object sphere(const Transform &t, Sphere s, Material m);                          // SphereObject::new
    static Object plane(const Transform &t, Vec3 position, Vec3 tan1, Vec3 tan2, Material m1, Material m2);  // PlaneObject::new
    static Object model(const Transform &t, std::shared_ptr<GpuMesh> mesh, Material m);      // MeshObject::new
};

// render/renderer.rs:44-53.  The photon fields take effect when the Renderer's
// photon pass is switched on.
struct Scene {
    uint32_t size[2] = {300, 300};
    View view;
    std::vector<Light> lights;
    std::vector<Object> scene_object;
    uint64_t photon_count = 0;
    uint32_t photon_samples = 0;
    uint32_t newton_steps = 0;
    double newton_epsilon = 0;
};

// lib.rs:42-121
struct SceneBuilder {
    uint64_t time = 0;
    Material material() const;
    Object sphere() const;
    Object plane() const;
    View view() const;
    std::vector<Light> lights() const;
    std::vector<Object> scene_object() const;
    Scene scene() const;
    Object make_mesh(std::shared_ptr<GpuMesh> mesh, const Transform &t) const;
};

// render/renderer.rs:55-61,105-154,379-385
class Renderer {
  public:
    Renderer(rtg_ctx *ctx, Scene scene);          // Renderer::new
    ~Renderer();
    // Renderer::render on the GPU: the pixel pass, after the photon pass (renderer.rs:116-140) when `photon_pass`
    // is set, in which case every pixel carries the photon map's term as well.
    int render();
    std::map<std::string, std::vector<uint8_t>> images() const;   // image.hdr, image.pq.hdr, perf.hdr
    const std::vector<double> &radiosity() const { return rgb_; }
    const std::vector<rtg_hit> &hits() const { return hits_; }
    // per pixel: tree nodes visited + triangles tested by the camera ray (what perf.hdr shares the frame's time by)
    const std::vector<uint32_t> &pixel_cost() const { return cost_; }
    bool perf_per_pixel = true;                   // false: no counting pass, perf.hdr spreads the time evenly
    const char *error() const;
    uint32_t max_depth = 4;                       // renderer.rs:312
    bool photon_pass = false;
    // true: the photon directions are made and turned into rays on the host (libm's sin / cos: bit for bit the
    // CPU renderer's photons); false: all of the photon pass runs on the GPU (same directions, the GPU's sin / cos)
    bool photons_on_host = false;
    uint64_t seed = 1;                            // stands for the entropy-seeded generator of renderer.rs:112
    uint64_t photons() const { return n_photons_; }
    // wall time of the last render's parts: directions (the caller's side in the reference), photon pass + k-d
    // tree, pixel pass
    double seconds_directions() const { return seconds_dirs_; }
    double seconds_photons() const { return seconds_photons_; }
    double seconds_pixels() const { return seconds_; }

  private:
    rtg_ctx *ctx_;
    Scene scene_;
    rtg_scene *gpu_scene_ = nullptr;
    std::vector<double> rgb_;
    std::vector<rtg_hit> hits_;
    std::vector<uint32_t> cost_;
    int pixel_costs();
    double seconds_ = 0, seconds_dirs_ = 0, seconds_photons_ = 0;
    rtg_photon_map *photon_map_ = nullptr;
    uint64_t n_photons_ = 0;
};

std::shared_ptr<GpuMesh> upload_mesh(rtg_ctx *ctx, Bvh bvh);

// Radiance RGBE (.hdr) writer for a row-major RGB double image; NaN pixels
// become the reference's checkerboard (render/image.rs:44-66); run-length coded scanlines as the `image` crate's
// HdrEncoder writes them.
std::vector<uint8_t> to_hdr(const double *rgb, uint32_t width, uint32_t height);
// RGB image of the per-pixel times x 3000 (renderer.rs:152): `seconds` shared out in proportion to 1 + cost[i].
std::vector<double> perf_image(double seconds, const uint32_t *cost, uint32_t width, uint32_t height);
// The tar archive of Renderer::images() as the reference's server frames it (bin/server.rs:19-30).
std::vector<uint8_t> images_tar(const std::map<std::string, std::vector<uint8_t>> &images);
double smpte2048_encode(double x);   // geo/color.rs:15-17
void pq_encode(const double *rgb, size_t n, double *out);   // render/image.rs:33-43 over n channel values

} // namespace rt
