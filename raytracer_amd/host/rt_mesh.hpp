// rt_mesh.hpp -- synthetic mesh and ray generators.
//
// The reference's mesh loaders (src/mesh/: bunny, cow, pinecone, sphere) are not
// part of its published tree, so the build supplies seeded generators instead.
// They use only + - * / and integer hashing, so the same seed gives the same
// bits on every host.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/rtg.h"

namespace rt {

// A terrain of nx * ny quads (two triangles each, wound to face +y / the
// camera), rising away from a camera at (0,0,1) that looks down -z, displaced
// by multi-octave value noise and rotated off the axes so that no triangle has
// a zero-thickness bounding box.  Coordinates are multiplied by `scale`.
void heightfield(uint32_t nx, uint32_t ny, uint64_t seed, double scale, rtg_tri *out);

// The ray distribution of the reference's `bench_mesh` (src/tree/benchmark.rs:15-27):
// origin uniform in the box, direction uniform in the unit ball by rejection,
// then normalised.  Rays are 6 doubles: orig.xyz, dir.xyz.
void bench_mesh_rays(const double lo_hi[6], uint64_t n, uint64_t seed, double *rays6);

// Photon directions: Sphere::fibonacci_sphere + ZenithY::into_normal (src/geo/sphere.rs:27-32,48-80), `count`
// unit vectors (3 doubles each) with the per-direction jitter drawn from a seeded stream.
void fibonacci_sphere(uint64_t count, uint64_t seed, double *dirs3);
void fibonacci_zenith_y(uint64_t count, uint64_t seed, double *dirs2);

// 30-bit Morton code of the origin inside lo_hi, with the direction octant in
// the three top bits: the sort key used to make a ray batch wave-coherent.
uint64_t ray_sort_key(const double lo_hi[6], const double *ray6);

} // namespace rt
