// rt_patch.hpp -- host side of the trispline patch path.
//
// Mirrors the orphan src/trispline.rs of the reference (written against
// glam::DVec3): the NormalHermiteTrispline -> HermiteTrispline ->
// SplineTrispline -> Trispline conversion chain (trispline.rs:68-134) and patch
// evaluation (trispline.rs:161-181), plus what the reference lacks: tessellation
// of patches into the flat facets the BVH is built over, and a seeded terrain
// of patches for the trispline benchmark configuration.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/rtg.h"

namespace rt {

struct D3 {
    double x, y, z;
};

// NormalHermiteTrispline (trispline.rs:24-36)
struct NormalHermiteTrispline {
    D3 p0, p1, p2, p01, p12, p20, n0, n1, n2, control;
};

rtg_trispline trispline_from(const NormalHermiteTrispline &nh);
D3 trispline_position(const rtg_trispline &s, double t, double u);
D3 trispline_ddt(const rtg_trispline &s, double t, double u);
D3 trispline_ddu(const rtg_trispline &s, double t, double u);
D3 trispline_normal(const rtg_trispline &s, double t, double u);

// k*k flat facets per patch (for i, for j: the "up" facet, then the "down" one
// when it exists), with the (t,u) of every corner.
void tessellate(const rtg_trispline *patches, uint64_t n, uint32_t k, rtg_tri *out_tris, rtg_patch_leaf *out_info);

// Terrain of 2*nx*ny patches over the same height function as rt::heightfield,
// vertex normals = normalised sum of the adjacent face normals, control = 0.
std::vector<rtg_trispline> patch_terrain(uint32_t nx, uint32_t ny, uint64_t seed, double scale);

} // namespace rt
