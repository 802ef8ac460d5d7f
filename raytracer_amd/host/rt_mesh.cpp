// rt_mesh.cpp -- see rt_mesh.hpp.
#include "rt_mesh.hpp"

#include <cmath>
#include <thread>
#include <vector>

namespace rt {
namespace {

uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
double lattice(int64_t ix, int64_t iy, uint64_t seed) {
    uint64_t s = seed ^ ((uint64_t)ix * 0xD1B54A32D192ED03ull) ^ ((uint64_t)iy * 0x8CB92BA72F3D8DD7ull);
    uint64_t h = splitmix64(s);
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}
double value_noise(double x, double y, uint64_t seed) {
    double fx = std::floor(x), fy = std::floor(y);
    int64_t ix = (int64_t)fx, iy = (int64_t)fy;
    double tx = x - fx, ty = y - fy;
    double sx = tx * tx * (3.0 - 2.0 * tx), sy = ty * ty * (3.0 - 2.0 * ty);
    double a = lattice(ix, iy, seed), b = lattice(ix + 1, iy, seed);
    double c = lattice(ix, iy + 1, seed), d = lattice(ix + 1, iy + 1, seed);
    double ab = a + (b - a) * sx, cd = c + (d - c) * sx;
    return ab + (cd - ab) * sy;
}
double fbm(double x, double y, uint64_t seed) {
    double amp = 0.5, sum = 0.0, f = 1.0;
    for (int o = 0; o < 5; o++) {
        sum += amp * (value_noise(x * f, y * f, seed + (uint64_t)o * 1315423911ull) - 0.5);
        amp *= 0.5;
        f *= 2.17;
    }
    return sum;
}
struct P3 {
    double x, y, z;
};
P3 vertex(uint32_t i, uint32_t j, uint32_t nx, uint32_t ny, uint64_t seed, double scale) {
    double u = (double)i / (double)nx, v = (double)j / (double)ny;
    double x = (u - 0.5) * 3.2;
    double z = 0.5 - 2.5 * v;
    double y = -0.5 + 1.7 * v * v + 0.22 * fbm(u * 37.0, v * 23.0, seed);
    // fixed small rotations about y then x (cos/sin of 0.15 and 0.07 as literals)
    const double cy = 0.98877107793604224, sy = 0.14943813247359922;
    const double cx = 0.99755100025327959, sx = 0.069942847337532768;
    double x1 = cy * x + sy * z, z1 = -sy * x + cy * z;
    double y2 = cx * y - sx * z1, z2 = sx * y + cx * z1;
    return P3{x1 * scale, y2 * scale, z2 * scale};
}
void put(rtg_tri &t, P3 a, P3 b, P3 c) {
    t.p[0][0] = a.x; t.p[0][1] = a.y; t.p[0][2] = a.z;
    t.p[1][0] = b.x; t.p[1][1] = b.y; t.p[1][2] = b.z;
    t.p[2][0] = c.x; t.p[2][1] = c.y; t.p[2][2] = c.z;
}
} // namespace

void heightfield(uint32_t nx, uint32_t ny, uint64_t seed, double scale, rtg_tri *out) {
    std::vector<P3> row0(nx + 1), row1(nx + 1);
    for (uint32_t i = 0; i <= nx; i++) row0[i] = vertex(i, 0, nx, ny, seed, scale);
    size_t k = 0;
    for (uint32_t j = 0; j < ny; j++) {
        for (uint32_t i = 0; i <= nx; i++) row1[i] = vertex(i, j + 1, nx, ny, seed, scale);
        for (uint32_t i = 0; i < nx; i++) {
            put(out[k++], row0[i], row0[i + 1], row1[i + 1]);
            put(out[k++], row0[i], row1[i + 1], row1[i]);
        }
        row0.swap(row1);
    }
}

void bench_mesh_rays(const double b[6], uint64_t n, uint64_t seed, double *rays6) {
    uint64_t s = seed;
    auto uni = [&s]() { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); };
    for (uint64_t r = 0; r < n; r++) {
        double *o = rays6 + r * 6;
        for (int a = 0; a < 3; a++) o[a] = b[a] + (b[3 + a] - b[a]) * uni();
        for (;;) {
            double dx = -1.0 + 2.0 * uni(), dy = -1.0 + 2.0 * uni(), dz = -1.0 + 2.0 * uni();
            double len = std::sqrt(((0.0 + dx * dx) + dy * dy) + dz * dz);
            if (len > 1.0) continue;
            o[3] = dx / len;
            o[4] = dy / len;
            o[5] = dz / len;
            break;
        }
    }
}

// Sphere::fibonacci_sphere + ZenithY::into_normal (geo/sphere.rs:27-32,48-80): direction i of `count` sits at
// zenith (phi * (i + u)) mod 2 pi and height -1 + dydi * (i + u) + dydj * v, with u, v in [0, 1) drawn per
// direction.  The reference draws them from an entropy-seeded generator (renderer.rs:112), so no two of its runs
// agree; here they come from the seeded SplitMix64 stream used by the other generators.
// sin and cos as two separate libm calls (what rustc emits); merged into sincos() they round differently in the
// last place for about one argument in a thousand.
static double (*volatile libm_sin)(double) = ::sin;
static double (*volatile libm_cos)(double) = ::cos;

// x % y of the reference (fmod: the exact remainder) for 0 <= x < 2^52 * y, y > 0.  The remainder x - n * y of the
// right integer n is a double, so one fused multiply-add yields it without error; the quotient's rounding can put
// n one off, which the sign or size of the remainder shows.  Several times faster than the library's bit-by-bit
// loop for quotients in the millions, which is what the directions of a large photon pass have.
static double fmod_positive(double x, double y) {
    double n = std::floor(x / y);
    double r = std::fma(-n, y, x);
    if (r < 0.0) {
        n -= 1.0;
        r = std::fma(-n, y, x);
    } else if (r >= y) {
        n += 1.0;
        r = std::fma(-n, y, x);
    }
    return r;
}

void fibonacci_sphere(uint64_t count, uint64_t seed, double *dirs3) {
    const double pi = 3.14159265358979323846264338327950288;
    const double n = (double)count;
    const double phi = pi * (3.0 - std::sqrt(5.0));
    const double d = 2.0 * pi / phi;
    const double dydi = 2.0 / (n + d);
    const double dydj = dydi * d;
    uint64_t s = seed;
    auto uni = [&s]() { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); };
    for (uint64_t k = 0; k < count; k++) {
        double i = (double)k + uni();
        double j = uni();
        double zenith = std::fmod(phi * i, 2.0 * pi);
        double y = -1.0 + dydi * i + dydj * j;
        double radius = std::sqrt(1.0 - y * y);
        dirs3[k * 3 + 0] = libm_cos(zenith) * radius;
        dirs3[k * 3 + 1] = y;
        dirs3[k * 3 + 2] = libm_sin(zenith) * radius;
    }
}

// The same draws as fibonacci_sphere, left as `ZenithY` (zenith, y) pairs: what the photon pass stores per photon
// and corrects later (src/geo/sphere.rs:52-80, src/render/renderer.rs:116-118).
void fibonacci_zenith_y(uint64_t count, uint64_t seed, double *dirs2) {
    const double pi = 3.14159265358979323846264338327950288;
    const double n = (double)count;
    const double phi = pi * (3.0 - std::sqrt(5.0));
    const double d = 2.0 * pi / phi;
    const double dydi = 2.0 / (n + d);
    const double dydj = dydi * d;
    // direction k takes draws 2k and 2k + 1 of the stream, whose state after m draws is seed + m * gamma: any range
    // of directions can start on its own, so the host's cores share them
    auto range = [=](uint64_t begin, uint64_t end) {
        uint64_t s = seed + 2 * begin * 0x9E3779B97F4A7C15ull;
        auto uni = [&s]() { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); };
        for (uint64_t k = begin; k < end; k++) {
            double i = (double)k + uni();
            double j = uni();
            dirs2[k * 2 + 0] = fmod_positive(phi * i, 2.0 * pi);
            dirs2[k * 2 + 1] = -1.0 + dydi * i + dydj * j;
        }
    };
    unsigned threads = std::thread::hardware_concurrency();
    threads = threads > 16 ? 16 : threads;
    if (threads < 2 || count < (1u << 16)) {
        range(0, count);
        return;
    }
    std::vector<std::thread> pool;
    uint64_t done = 0;
    try {
        pool.reserve(threads - 1);
        for (unsigned t = 0; t + 1 < threads; t++) {
            const uint64_t end = count / threads * (t + 1);
            pool.emplace_back(range, done, end);
            done = end;
        }
    } catch (...) {
        // no more threads to be had: the rest runs here
    }
    range(done, count);
    for (auto &th : pool) th.join();
}

uint64_t ray_sort_key(const double b[6], const double *ray6) {
    uint64_t key = 0;
    uint32_t q[3];
    for (int a = 0; a < 3; a++) {
        double ext = b[3 + a] - b[a];
        double t = ext > 0.0 ? (ray6[a] - b[a]) / ext : 0.0;
        if (!(t > 0.0)) t = 0.0;
        if (t > 1.0) t = 1.0;
        uint32_t v = (uint32_t)(t * 1023.0);
        q[a] = v > 1023u ? 1023u : v;
    }
    for (int bit = 9; bit >= 0; bit--)
        for (int a = 0; a < 3; a++) key = (key << 1) | ((q[a] >> bit) & 1u);
    uint64_t oct = (ray6[3] < 0.0 ? 1u : 0u) | (ray6[4] < 0.0 ? 2u : 0u) | (ray6[5] < 0.0 ? 4u : 0u);
    return (oct << 30) | key;
}

} // namespace rt
