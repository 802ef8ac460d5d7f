// rt_bvh.hpp -- host-side BVH build and flatten (C++ stand-in for the Rust host code).
//
// Mirrors the reference's tree API: BvhForest::new + BvhForest::subdivide
// (src/tree/bvh.rs:48-58, src/tree/bvh_subdivide.rs:8-43) and Bvh::new
// (src/tree/bvh.rs:80-106).  The algorithm and every tie rule are the
// reference's (so the leaf numbering, which is the hit id, is identical); the
// implementation is index-based and multi-threaded instead of moving a vector
// of boxed trees at every level.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/rtg.h"

namespace rt {

struct Bounds {
    double min[3], max[3];
    static Bounds empty();
    Bounds union_(const Bounds &o) const;      // geo/bounds.rs:28-33
    double center(int axis) const;             // geo/bounds.rs:38-40
    double dim(int axis) const;                // geo/bounds.rs:70-72
    double surface_area() const;               // geo/bounds.rs:73-76
};

// f64::minimum / f64::maximum (math/scalar.rs:37-38)
double f64_minimum(double a, double b);
double f64_maximum(double a, double b);

Bounds triangle_bounds(const rtg_tri &t);      // geo/triangle.rs:43-45

// The flat `Bvh` (tree/bvh.rs:20-24).  leaf_source[i] is the index of leaf i in
// the triangle array handed to the builder.
struct Bvh {
    std::vector<rtg_bvh_node> nodes;
    std::vector<rtg_tri> leaves;
    std::vector<double> vertex_normals;   // empty or 9 per leaf
    std::vector<uint32_t> leaf_source;
    uint32_t root = 0;
    uint32_t depth = 0;

    Bounds bounds() const;                     // tree/bvh.rs:134-136
    rtg_bvh view() const;                      // borrow as the C ABI struct
};

// BvhForest::new(tris).subdivide() followed by Bvh::new(&tree).
// `vertex_normals` may be null.  `threads` <= 0 picks the hardware count
// (capped at 16).
Bvh build_subdivide(const rtg_tri *tris, uint64_t n, const double *vertex_normals, int threads);

// BvhForest::new(tris).contract() / .hybrid() followed by Bvh::new (tree/bvh_cluster.rs:42-116,
// tree/bvh_joint.rs:3-8); see rt_bvh_cluster.cpp for the rule applied where the reference's choice among equal
// distances is not restatable.
Bvh build_contract(const rtg_tri *tris, uint64_t n, const double *vertex_normals, int threads);
Bvh build_hybrid(const rtg_tri *tris, uint64_t n, const double *vertex_normals, int threads);

} // namespace rt
