/* default_scene.c -- the reference's default frame (src/lib.rs:68-120: a glass sphere over a checker floor, two
 * lights, photon caustics) through the C ABI of include/rtg.h alone: context, scene, photon map, frame.  Plain C11;
 * this is what the `extern "C"` binding of the Rust host boils down to.
 *
 *   default_scene <width> <height> <directions per light> <time> <out.f64> [<directions.f64>]
 *
 * writes width x height x 3 doubles (the frame's radiosity, row-major) to <out.f64>; an output name ending in .ppm
 * gets an 8-bit picture through the sRGB curve instead.  The emission directions
 * (`ZenithY` pairs, light by light) are read from <directions.f64> when given, otherwise drawn here from a small
 * generator (the reference seeds its own from entropy). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rtg.h"

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static int fail(rtg_ctx *ctx, const char *what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, rtg_last_error(ctx));
    return 1;
}

static uint64_t next_u64(uint64_t *s) { /* SplitMix64 */
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double uniform(uint64_t *s) { return (double)(next_u64(s) >> 11) * (1.0 / 9007199254740992.0); }

/* Sphere::fibonacci_sphere (src/geo/sphere.rs:52-80) */
static void fibonacci_zenith_y(uint64_t count, uint64_t seed, double *dirs2) {
    const double pi = 3.14159265358979323846264338327950288;
    const double phi = pi * (3.0 - sqrt(5.0)), d = 2.0 * pi / phi;
    const double dydi = 2.0 / ((double)count + d), dydj = dydi * d;
    for (uint64_t k = 0; k < count; k++) {
        const double i = (double)k + uniform(&seed), j = uniform(&seed);
        dirs2[k * 2] = fmod(phi * i, 2.0 * pi);
        dirs2[k * 2 + 1] = -1.0 + dydi * i + dydj * j;
    }
}

static unsigned char srgb8(double linear) {
    if (!(linear > 0.0)) return 0; /* negative, zero, NaN */
    if (linear > 1.0) linear = 1.0;
    const double v = linear <= 0.0031308 ? 12.92 * linear : 1.055 * pow(linear, 1.0 / 2.4) - 0.055;
    return (unsigned char)(v * 255.0 + 0.5);
}

int main(int argc, char **argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s <width> <height> <directions per light> <time> <out.f64> [<directions.f64>]\n", argv[0]);
        return 2;
    }
    const uint32_t width = (uint32_t)atoi(argv[1]), height = (uint32_t)atoi(argv[2]);
    const uint64_t count = (uint64_t)atoll(argv[3]);
    const double time = atof(argv[4]);
    const double pi = 3.14159265358979323846264338327950288;

    rtg_ctx *ctx = NULL;
    int rc = rtg_ctx_create(0, &ctx);
    if (rc != RTG_OK) return fail(NULL, "rtg_ctx_create", rc);

    /* SceneBuilder::sphere / plane / lights / view (src/lib.rs:68-102) */
    static const double identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    rtg_object objects[2];
    memset(objects, 0, sizeof(objects));
    for (int i = 0; i < 2; i++)
        if ((rc = rtg_transform_from_forward(identity, &objects[i].transform)) != RTG_OK) return fail(ctx, "rtg_transform_from_forward", rc);
    objects[0].kind = RTG_SPHERE;
    objects[0].sphere_center[1] = -0.2;
    objects[0].sphere_radius = 0.2;
    objects[0].material.has_dielectric = 1;
    objects[0].material.n1 = 1.0;
    objects[0].material.n2 = 1.5;
    objects[1].kind = RTG_PLANE;
    objects[1].plane_position[1] = -0.5;
    objects[1].plane_tan1[2] = -1.0;
    objects[1].plane_tan2[0] = -1.0;
    for (int k = 0; k < 3; k++) {
        objects[1].material.diffuse[k] = 1.0;
        objects[1].material2.diffuse[k] = 0.1;
    }
    const double c = cos(time / 100.0 * pi * 2.0), s = sin(time / 100.0 * pi * 2.0);
    rtg_light lights[2] = {{{c, s, 1.0}, {0.0, 0.0, 100.0}}, {{-c, s, 1.0}, {0.0, 100.0, 0.0}}};
    rtg_view view = {{0.0, 0.0, 1.0}, {tan(pi * 0.5 / 2.0), tan(pi * 0.5 / 2.0)}};

    rtg_scene_desc desc = {objects, 2};
    rtg_scene *scene = NULL;
    if ((rc = rtg_scene_create(ctx, &desc, &scene)) != RTG_OK) return fail(ctx, "rtg_scene_create", rc);

    double *dirs = malloc((size_t)count * 2 * 2 * sizeof(double));
    double *rgb = malloc((size_t)width * height * 3 * sizeof(double));
    if (!dirs || !rgb) return fail(ctx, "malloc", -3);
    if (argc > 6) {
        FILE *f = fopen(argv[6], "rb");
        if (!f || fread(dirs, sizeof(double), (size_t)count * 4, f) != (size_t)count * 4) return fail(ctx, "reading the directions", -1);
        fclose(f);
    } else {
        for (int l = 0; l < 2; l++) fibonacci_zenith_y(count, 1 + (uint64_t)l, dirs + (size_t)l * count * 2);
    }

    /* Renderer::render (src/render/renderer.rs:115-154): photon pass + k-d tree, then the pixel pass */
    double t0 = now_ms();
    rtg_photon_map *map = NULL;
    uint64_t n_photons = 0;
    if ((rc = rtg_photon_map_emit(scene, lights, 2, dirs, count, 4, &map, &n_photons)) != RTG_OK) return fail(ctx, "rtg_photon_map_emit", rc);
    double t1 = now_ms();
    const rtg_rect whole = {0, 0, width, height};
    if ((rc = rtg_render_frame_photons(scene, &view, lights, 2, map, 3, 5, 0.00001, width, height, whole, 4, rgb, NULL)) != RTG_OK)
        return fail(ctx, "rtg_render_frame_photons", rc);
    double t2 = now_ms();
    printf("%s: %llu photons from 2 x %llu directions in %.1f ms, %ux%u frame in %.1f ms\n", rtg_version(), (unsigned long long)n_photons,
           (unsigned long long)count, t1 - t0, width, height, t2 - t1);

    FILE *out = fopen(argv[5], "wb");
    const size_t values = (size_t)width * height * 3, name_len = strlen(argv[5]);
    if (!out) return fail(ctx, "opening the output", -1);
    if (name_len > 4 && strcmp(argv[5] + name_len - 4, ".ppm") == 0) {
        unsigned char *bytes = malloc(values);
        if (!bytes) return fail(ctx, "malloc", -3);
        for (size_t k = 0; k < values; k++) bytes[k] = srgb8(rgb[k]);
        fprintf(out, "P6\n%u %u\n255\n", width, height);
        if (fwrite(bytes, 1, values, out) != values) return fail(ctx, "writing the picture", -1);
        free(bytes);
    } else if (fwrite(rgb, sizeof(double), values, out) != values) {
        return fail(ctx, "writing the frame", -1);
    }
    fclose(out);
    rtg_photon_map_destroy(map);
    rtg_scene_destroy(scene);
    rtg_ctx_destroy(ctx);
    free(dirs);
    free(rgb);
    return 0;
}
