// rt_scene.cpp -- see rt_scene.hpp.
#include "rt_scene.hpp"

#include "rt_mesh.hpp"

#include <chrono>
#include <cmath>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

namespace rt {

static const double PI = 3.14159265358979323846264338327950288;

Mat4 Mat4::identity() {
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.m[i][j] = i == j ? 1.0 : 0.0;
    return r;
}
Mat4 Mat4::from_scale(Vec3 s) {
    Mat4 r;
    std::memset(r.m, 0, sizeof(r.m));   // Mat4::from_mat3(Mat3::from_diagonal(scale))
    r.m[0][0] = s.x;
    r.m[1][1] = s.y;
    r.m[2][2] = s.z;
    r.m[3][3] = 1.0;
    return r;
}
Mat4 Mat4::from_translation(Vec3 t) {
    Mat4 r = identity();
    r.m[0][3] = t.x;
    r.m[1][3] = t.y;
    r.m[2][3] = t.z;
    return r;
}
// self applied to every column of rhs; each entry is a dot folded from 0.0.
Mat4 Mat4::mul(const Mat4 &rhs) const {
    Mat4 r;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s = s + m[i][k] * rhs.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
Mat4 Mat4::from_scale_rotation_translation(Vec3 scale, Vec3 translation) {
    return from_scale(scale).mul(identity()).mul(from_translation(translation));
}

Transform Transform::from(const Mat4 &forward) {
    Transform t;
    rtg_transform_from_forward(&forward.m[0][0], &t.t);
    return t;
}
Transform Transform::default_() { return TransformBuilder().build(); }
TransformBuilder &TransformBuilder::scale(double amount) {
    scale_ = Vec3{amount, amount, amount};
    return *this;
}
TransformBuilder &TransformBuilder::translate(double x, double y, double z) {
    translate_ = Vec3{x, y, z};
    return *this;
}
TransformBuilder &TransformBuilder::rotate() { return *this; }
Transform TransformBuilder::build() const {
    return Transform::from(Mat4::from_scale_rotation_translation(scale_, translate_));
}

rtg_material Material::c() const {
    rtg_material m;
    std::memset(&m, 0, sizeof(m));
    m.diffuse[0] = diffuse.x;
    m.diffuse[1] = diffuse.y;
    m.diffuse[2] = diffuse.z;
    m.has_dielectric = has_dielectric ? 1 : 0;
    m.n1 = n1;
    m.n2 = n2;
    return m;
}

View View::from_fov(Vec3 orig, double fov_x, double fov_y) {
    View v;
    v.orig = orig;
    v.fov[0] = fov_x;
    v.fov[1] = fov_y;
    return v;
}
rtg_view View::c() const {
    rtg_view v;
    v.orig[0] = orig.x;
    v.orig[1] = orig.y;
    v.orig[2] = orig.z;
    v.tan_half_fov[0] = std::tan(fov[0] / 2.0);   // geo/view.rs:14-15
    v.tan_half_fov[1] = std::tan(fov[1] / 2.0);
    return v;
}

GpuMesh::~GpuMesh() {
    if (handle) rtg_mesh_destroy(handle);
}
std::shared_ptr<GpuMesh> upload_mesh(rtg_ctx *ctx, Bvh bvh) {
    auto m = std::make_shared<GpuMesh>();
    m->bvh = std::move(bvh);
    rtg_bvh v = m->bvh.view();
    if (rtg_mesh_create(ctx, &v, &m->handle) != RTG_OK) return nullptr;
    return m;
}

static void set3(double *d, Vec3 v) {
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
}
Object Object::sphere(const Transform &t, Sphere s, Material m) {
    Object o;
    std::memset(&o.c, 0, sizeof(o.c));
    o.c.kind = RTG_SPHERE;
    o.c.transform = t.t;
    o.c.material = m.c();
    set3(o.c.sphere_center, s.orig);
    o.c.sphere_radius = s.rad;
    return o;
}
Object Object::plane(const Transform &t, Vec3 position, Vec3 tan1, Vec3 tan2, Material m1, Material m2) {
    Object o;
    std::memset(&o.c, 0, sizeof(o.c));
    o.c.kind = RTG_PLANE;
    o.c.transform = t.t;
    o.c.material = m1.c();
    o.c.material2 = m2.c();
    set3(o.c.plane_position, position);
    set3(o.c.plane_tan1, tan1);
    set3(o.c.plane_tan2, tan2);
    return o;
}
Object Object::model(const Transform &t, std::shared_ptr<GpuMesh> mesh, Material m) {
    Object o;
    std::memset(&o.c, 0, sizeof(o.c));
    o.c.kind = RTG_MESH;
    o.c.transform = t.t;
    o.c.material = m.c();
    o.c.mesh = mesh ? mesh->handle : nullptr;
    o.mesh = std::move(mesh);
    return o;
}

// ---- lib.rs:42-121 -------------------------------------------------------------
Material SceneBuilder::material() const {                      // lib.rs:47-49
    return Material{Color{1.0 * 0.0, 1.0 * 0.0, 1.0 * 0.0}, true, 1.0, 1.5};
}
Object SceneBuilder::make_mesh(std::shared_ptr<GpuMesh> mesh, const Transform &t) const {   // lib.rs:50-54
    return Object::model(t, std::move(mesh), material());
}
Object SceneBuilder::sphere() const {                          // lib.rs:68-77
    return Object::sphere(Transform::default_(), Sphere{Vec3{0.0, -0.2, 0.0}, 0.2},
                          Material{Color{0.0, 0.0, 0.0}, true, 1.0, 1.5});
}
Object SceneBuilder::plane() const {                           // lib.rs:78-86
    return Object::plane(Transform::from(Mat4::identity()), Vec3{0.0, -0.5, 0.0}, Vec3{0.0, 0.0, -1.0},
                         Vec3{-1.0, 0.0, 0.0}, Material{Color{1.0, 1.0, 1.0}, false, 0, 0},
                         Material{Color{0.1, 0.1, 0.1}, false, 0, 0});
}
View SceneBuilder::view() const {                              // lib.rs:87-89
    return View::from_fov(Vec3{0.0, 0.0, 1.0}, PI * 0.5, PI * 0.5);
}
std::vector<Light> SceneBuilder::lights() const {              // lib.rs:90-102
    const double lightz = 1.0, intensity = 100.0, lightdis = 1.0;
    // two separate libm calls, as rustc emits them (a merged sincos() can round differently in the last place)
    static double (*volatile libm_sin)(double) = ::sin;
    static double (*volatile libm_cos)(double) = ::cos;
    const double c = libm_cos((double)time / 100.0 * PI * 2.0);
    const double s = libm_sin((double)time / 100.0 * PI * 2.0);
    return {Light{Sphere{Vec3{lightdis * c, lightdis * s, lightz}, 1.0}, Color{0.0, 0.0, intensity}},
            Light{Sphere{Vec3{-lightdis * c, lightdis * s, lightz}, 1.0}, Color{0.0, intensity, 0.0}}};
}
std::vector<Object> SceneBuilder::scene_object() const { return {sphere(), plane()}; }   // lib.rs:103-108
Scene SceneBuilder::scene() const {                            // lib.rs:109-120
    Scene s;
    s.size[0] = 300;
    s.size[1] = 300;
    s.view = view();
    s.lights = lights();
    s.scene_object = scene_object();
    s.photon_count = 10000000;
    s.photon_samples = 3;
    s.newton_steps = 5;
    s.newton_epsilon = 0.00001;
    return s;
}

// ---- Renderer ---------------------------------------------------------------------
Renderer::Renderer(rtg_ctx *ctx, Scene scene) : ctx_(ctx), scene_(std::move(scene)) {}
Renderer::~Renderer() {
    if (photon_map_) rtg_photon_map_destroy(photon_map_);
    if (gpu_scene_) rtg_scene_destroy(gpu_scene_);
}
const char *Renderer::error() const { return rtg_last_error(ctx_); }

int Renderer::render() {
    if (!gpu_scene_) {
        std::vector<rtg_object> objs;
        for (const Object &o : scene_.scene_object) objs.push_back(o.c);
        rtg_scene_desc d{objs.data(), (uint32_t)objs.size()};
        int rc = rtg_scene_create(ctx_, &d, &gpu_scene_);
        if (rc != RTG_OK) return rc;
    }
    const uint32_t w = scene_.size[0], h = scene_.size[1];
    std::vector<rtg_light> lights;
    for (const Light &l : scene_.lights) {
        rtg_light x;
        set3(x.position, l.sphere.orig);
        set3(x.color, l.color);
        lights.push_back(x);
    }
    rgb_.assign((size_t)w * h * 3, 0.0);
    hits_.assign((size_t)w * h, rtg_hit{RTG_NONE, RTG_NONE, 0.0});
    rtg_view v = scene_.view.c();
    auto since = [](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
    };
    if (!photon_pass) {
        auto t0 = std::chrono::steady_clock::now();
        int rc = rtg_render_frame(gpu_scene_, &v, lights.data(), (uint32_t)lights.size(), w, h, rtg_rect{0, 0, w, h},
                                  max_depth, rgb_.data(), nullptr, hits_.data());
        seconds_ = since(t0);
        if (rc == RTG_OK) rc = pixel_costs();
        return rc;
    }
    // renderer.rs:116-140: a Fibonacci sphere of directions per light, every path behind a specular bounce a photon,
    // the photons in a k-d tree
    auto t0 = std::chrono::steady_clock::now();
    if (photon_map_) {
        rtg_photon_map_destroy(photon_map_);
        photon_map_ = nullptr;
    }
    int rc;
    if (photons_on_host) {
        // not a vector: its elements would be zeroed by one thread first (320 MB at the reference's default counts),
        // and the pages are better touched by the threads that fill them
        std::unique_ptr<double[]> dirs(new double[lights.size() * (size_t)scene_.photon_count * 2]);
        for (size_t l = 0; l < lights.size(); l++)
            fibonacci_zenith_y(scene_.photon_count, seed + 0x9E3779B97F4A7C15ull * l, dirs.get() + l * (size_t)scene_.photon_count * 2);
        seconds_dirs_ = since(t0);
        t0 = std::chrono::steady_clock::now();
        rc = rtg_photon_map_emit(gpu_scene_, lights.data(), (uint32_t)lights.size(), dirs.get(), scene_.photon_count, max_depth,
                                 &photon_map_, &n_photons_);
    } else {
        // the same directions (same seeds, same bits) made on the GPU batch by batch, their cosines and sines the GPU's
        seconds_dirs_ = 0.0;
        rc = rtg_photon_map_emit_fibonacci(gpu_scene_, lights.data(), (uint32_t)lights.size(), scene_.photon_count, seed, max_depth,
                                           &photon_map_, &n_photons_);
    }
    seconds_photons_ = since(t0);
    if (rc != RTG_OK) return rc;
    // renderer.rs:141-153: the pixel pass
    t0 = std::chrono::steady_clock::now();
    rc = rtg_render_frame_photons(gpu_scene_, &v, lights.data(), (uint32_t)lights.size(), photon_map_, scene_.photon_samples,
                                  scene_.newton_steps, scene_.newton_epsilon, w, h, rtg_rect{0, 0, w, h}, max_depth, rgb_.data(),
                                  nullptr);
    if (rc == RTG_OK) rc = rtg_render_primary(gpu_scene_, &v, w, h, rtg_rect{0, 0, w, h}, hits_.data(), nullptr, nullptr);
    seconds_ = since(t0);
    if (rc == RTG_OK) rc = pixel_costs();
    return rc;
}

// The reference times every pixel on its own (renderer.rs:156,174) and shows the times as `perf.hdr`.  A GPU frame
// has one time for all its pixels; what it does know per pixel is the work of the camera ray: tree nodes visited and
// triangles tested (the counting build of the traversal kernel, one more primary pass outside the timed part).
int Renderer::pixel_costs() {
    const uint32_t w = scene_.size[0], h = scene_.size[1];
    cost_.assign((size_t)w * h, 0u);
    if (!perf_per_pixel) return RTG_OK;
    std::vector<rtg_ray_counters> ctr((size_t)w * h);
    rtg_view v = scene_.view.c();
    int rc = rtg_render_primary(gpu_scene_, &v, w, h, rtg_rect{0, 0, w, h}, hits_.data(), nullptr, ctr.data());
    if (rc != RTG_OK) return rc;
    for (size_t i = 0; i < ctr.size(); i++) cost_[i] = ctr[i].nodes_visited + ctr[i].tris_tested;
    return RTG_OK;
}

double smpte2048_encode(double x) {
    const double M1 = 1305.0 / 8192.0, M2 = 2523.0 / 32.0;
    const double C1 = 107.0 / 128.0, C2 = 2413.0 / 128.0, C3 = 2392.0 / 128.0;
    const double xm = std::pow(x, M1);          // the same value in both places of render/image.rs:36-37
    return std::pow((C1 + C2 * xm) / (1.0 + C3 * xm), M2);
}

// Rows [y0, y1) of a picture, shared among the host's cores: f(row range) for every range.  The workers are started
// once and kept: a frame's three small pictures are four such jobs, and starting fifteen threads for each of them
// cost more than the encoding itself.
namespace {
class RowWorkers {
  public:
    static RowWorkers &get() {
        static RowWorkers pool;
        return pool;
    }
    unsigned size() const { return (unsigned)workers_.size() + 1; }
    // f(first row, end row) over `parts` consecutive ranges of [0, height), on the callers' thread and the workers
    void run(uint32_t height, unsigned parts, const std::function<void(uint32_t, uint32_t)> &f) {
        std::lock_guard<std::mutex> one_job(job_lock_);
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = &f;
            height_ = height;
            parts_ = parts;
            next_.store(0);
            left_ = parts;
            generation_++;
        }
        wake_.notify_all();
        work();
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return left_ == 0; });
        job_ = nullptr;
    }

  private:
    RowWorkers() {
        unsigned threads = std::thread::hardware_concurrency();
        threads = threads > 16 ? 16 : threads;
        try {
            for (unsigned t = 1; t < threads; t++) workers_.emplace_back([this] { loop(); });
        } catch (...) {
            // no more threads to be had: the callers' thread does what the missing workers would have done
        }
    }
    ~RowWorkers() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        wake_.notify_all();
        for (auto &w : workers_) w.join();
    }
    void work() {
        for (;;) {
            const unsigned part = next_.fetch_add(1);
            if (part >= parts_) return;
            const uint32_t y0 = (uint32_t)((uint64_t)height_ * part / parts_);
            const uint32_t y1 = (uint32_t)((uint64_t)height_ * (part + 1) / parts_);
            (*job_)(y0, y1);
            std::lock_guard<std::mutex> g(m_);
            if (--left_ == 0) done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                wake_.wait(g, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
            }
            work();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, job_lock_;
    std::condition_variable wake_, done_;
    const std::function<void(uint32_t, uint32_t)> *job_ = nullptr;
    uint32_t height_ = 0;
    unsigned parts_ = 0, left_ = 0;
    std::atomic<unsigned> next_{0};
    uint64_t generation_ = 0;
    bool stop_ = false;
};
} // namespace

template <class F> static void over_rows(uint32_t height, size_t work_per_row, F f) {
    if ((size_t)height * work_per_row < (1u << 16) || height < 2) {
        f(0u, height);
        return;
    }
    RowWorkers &pool = RowWorkers::get();
    const unsigned parts = std::min<unsigned>(pool.size(), height);
    if (parts < 2) {
        f(0u, height);
        return;
    }
    pool.run(height, parts, std::function<void(uint32_t, uint32_t)>(f));
}

// The Radiance picture the reference's ImageBuilder::write_to produces through the `image` crate's HdrEncoder
// (render/image.rs:44-66; image 0.24.1, restated from the crate's published source -- the crate is not part of the
// reference's tree, so its bytes are unpinned here): header with the encoder's comment line, shared-exponent RGBE
// pixels, and for widths of 8 to 32768 pixels one "new RLE" scanline per row -- the marker 2, 2, width high, width
// low, then the row's red, green, blue and exponent bytes each compressed on their own; other widths are written
// flat, four bytes per pixel.
static void to_rgbe8(const double *p, uint32_t x, uint32_t y, uint8_t px[4]) {
    double c[3] = {p[0], p[1], p[2]};
    if (c[0] != c[0] || c[1] != c[1] || c[2] != c[2]) {      // image.rs:52-58
        double v = ((x + y) % 2 == 0) ? 0.0 : 1.0;
        c[0] = c[1] = c[2] = v;
    }
    float f[3] = {(float)c[0], (float)c[1], (float)c[2]};
    float mx = std::fmax(f[0], std::fmax(f[1], f[2]));
    px[0] = px[1] = px[2] = px[3] = 0;
    if (mx > 0.0f) {
        int e = (int)std::floor(std::log2(mx)) + 1;
        float mul = std::ldexp(1.0f, e);
        for (int k = 0; k < 3; k++) {
            float q = std::trunc(f[k] / mul * 256.0f);
            px[k] = (uint8_t)(q < 0.0f ? 0.0f : (q > 255.0f ? 255.0f : q));
        }
        px[3] = (uint8_t)(e + 128);
    }
}

// The encoder's run-length code for one component of a scanline: the bytes are cut into runs of equal bytes (at
// most 127 long; a run counts as one from three bytes on) and the stretches between them (at most 128 long, shorter
// runs folded in); a run is written as 128 + length and the byte, a stretch as its length and its bytes.
static void rle_compress(const uint8_t *data, size_t n, std::vector<uint8_t> &rle) {
    if (n == 0) {
        rle.push_back(0);
        return;
    }
    struct Piece {
        int kind;          // 0 none, 1 run, 2 stretch
        uint8_t value;
        size_t idx, len;
    };
    size_t cur = 0;
    auto next_piece = [&]() {
        if (cur == n) return Piece{0, 0, 0, 0};
        const uint8_t cv = data[cur];
        size_t crun = 0;
        while (cur + crun < n && data[cur + crun] == cv && crun < 127) crun++;
        Piece r = crun > 2 ? Piece{1, cv, cur, crun} : Piece{2, cv, cur, crun};
        cur += crun;
        return r;
    };
    auto emit = [&](const Piece &q) {
        if (q.kind == 1) {
            rle.push_back((uint8_t)(128 + q.len));
            rle.push_back(q.value);
        } else {
            rle.push_back((uint8_t)q.len);
            rle.insert(rle.end(), data + q.idx, data + q.idx + q.len);
        }
    };
    Piece prev{0, 0, 0, 0};
    for (;;) {
        Piece held = prev;
        prev.kind = 0;
        if (held.kind == 1) {
            emit(held);
        } else if (held.kind == 2) {
            Piece nx = next_piece();
            if (nx.kind == 2) {
                const size_t clen = held.len + nx.len;
                if (clen == 128) {
                    emit(Piece{2, 0, held.idx, clen});
                } else if (clen > 128) {
                    prev = Piece{2, 0, held.idx + 128, clen - 128};
                    emit(Piece{2, 0, held.idx, 128});
                } else {
                    prev = Piece{2, 0, held.idx, clen};
                }
            } else {                 // a run follows (kept for the next turn), or the data ends
                prev = nx;
                emit(held);
                if (nx.kind == 0) return;
            }
        } else {
            Piece nx = next_piece();
            if (nx.kind == 0) return;
            if (nx.kind == 1)
                emit(nx);
            else
                prev = nx;
        }
    }
}

std::vector<uint8_t> to_hdr(const double *rgb, uint32_t width, uint32_t height) {
    const std::string head = "#?RADIANCE\n# Rust HDR encoder\nFORMAT=32-bit_rle_rgbe\n\n-Y " + std::to_string(height) + " +X " +
                             std::to_string(width) + "\n";
    if (width < 8 || width > 32768) {
        std::vector<uint8_t> out(head.size() + (size_t)width * height * 4);
        std::memcpy(out.data(), head.data(), head.size());
        uint8_t *pixels = out.data() + head.size();
        for (uint32_t y = 0; y < height; y++)
            for (uint32_t x = 0; x < width; x++)
                to_rgbe8(rgb + ((size_t)y * width + x) * 3, x, y, pixels + ((size_t)y * width + x) * 4);
        return out;
    }
    std::vector<std::vector<uint8_t>> rows(height);
    over_rows(height, (size_t)width * 4, [&rows, rgb, width](uint32_t y0, uint32_t y1) {
        std::vector<uint8_t> comp[4];
        for (auto &c : comp) c.resize(width);
        for (uint32_t y = y0; y < y1; y++) {
            for (uint32_t x = 0; x < width; x++) {
                uint8_t px[4];
                to_rgbe8(rgb + ((size_t)y * width + x) * 3, x, y, px);
                for (int k = 0; k < 4; k++) comp[k][x] = px[k];
            }
            std::vector<uint8_t> &row = rows[y];
            row.reserve((size_t)width * 4 + 64);
            row.push_back(2);
            row.push_back(2);
            row.push_back((uint8_t)(width / 256));
            row.push_back((uint8_t)(width % 256));
            for (int k = 0; k < 4; k++) rle_compress(comp[k].data(), width, row);
        }
    });
    size_t total = head.size();
    for (const auto &r : rows) total += r.size();
    std::vector<uint8_t> out;
    out.reserve(total);
    out.insert(out.end(), head.begin(), head.end());
    for (const auto &r : rows) out.insert(out.end(), r.begin(), r.end());
    return out;
}

// The frame's time spread over its pixels in proportion to 1 + cost (cost == nullptr: evenly), x 3000 as the
// reference scales its per-pixel seconds (renderer.rs:152): the perf image.
std::vector<double> perf_image(double seconds, const uint32_t *cost, uint32_t width, uint32_t height) {
    const size_t n = (size_t)width * height;
    std::vector<double> perf(n * 3);
    double total = 0.0;
    for (size_t i = 0; i < n; i++) total += 1.0 + (cost ? (double)cost[i] : 0.0);
    for (size_t i = 0; i < n; i++) {
        const double v = seconds * ((1.0 + (cost ? (double)cost[i] : 0.0)) / total) * 3000.0;
        perf[i * 3] = perf[i * 3 + 1] = perf[i * 3 + 2] = v;
    }
    return perf;
}

// ImageBuilder::smpte2048_encode (render/image.rs:33-43): every channel / 80 through the PQ curve.
void pq_encode(const double *rgb, size_t n, double *out) {
    over_rows((uint32_t)((n + 4095) / 4096), 4096, [=](uint32_t b0, uint32_t b1) {
        const size_t end = std::min(n, (size_t)b1 * 4096);
        for (size_t i = (size_t)b0 * 4096; i < end; i++) out[i] = smpte2048_encode(rgb[i] / 80.0);
    });
}

std::map<std::string, std::vector<uint8_t>> Renderer::images() const {      // renderer.rs:379-385
    std::map<std::string, std::vector<uint8_t>> r;
    const uint32_t w = scene_.size[0], h = scene_.size[1];
    if (rgb_.empty()) return r;
    r["image.hdr"] = to_hdr(rgb_.data(), w, h);
    std::vector<double> pq(rgb_.size());
    pq_encode(rgb_.data(), rgb_.size(), pq.data());                                        // image.rs:39
    r["image.pq.hdr"] = to_hdr(pq.data(), w, h);
    // renderer.rs:152: per-pixel time x 3000 -- here the frame's time shared out by the camera rays' work
    std::vector<double> perf = perf_image(seconds_, cost_.empty() ? nullptr : cost_.data(), w, h);
    r["perf.hdr"] = to_hdr(perf.data(), w, h);
    return r;
}

// The archive the reference's server answers with (bin/server.rs:19-30): one GNU-format header per image (path,
// size, mode 0o700, modification time 0, checksum), the bytes padded to 512, two empty blocks at the end.
std::vector<uint8_t> images_tar(const std::map<std::string, std::vector<uint8_t>> &images) {
    std::vector<uint8_t> out;
    auto octal = [](uint8_t *field, size_t width, uint64_t value) {          // width-1 octal digits and a NUL
        for (size_t i = width - 1; i-- > 0;) {
            field[i] = (uint8_t)('0' + (value & 7));
            value >>= 3;
        }
        field[width - 1] = 0;
    };
    for (const auto &kv : images) {
        uint8_t h[512] = {0};
        std::memcpy(h, kv.first.data(), std::min<size_t>(kv.first.size(), 100));
        octal(h + 100, 8, 0700);
        octal(h + 124, 12, kv.second.size());
        octal(h + 136, 12, 0);
        std::memcpy(h + 257, "ustar ", 6);
        h[263] = ' ';
        h[264] = 0;
        std::memset(h + 148, ' ', 8);
        uint32_t sum = 0;
        for (int i = 0; i < 512; i++) sum += h[i];
        octal(h + 148, 8, sum);
        out.insert(out.end(), h, h + 512);
        out.insert(out.end(), kv.second.begin(), kv.second.end());
        out.resize((out.size() + 511) / 512 * 512, 0);
    }
    out.resize(out.size() + 1024, 0);
    return out;
}

} // namespace rt
