// rt_patch.cpp -- see rt_patch.hpp.
#include "rt_patch.hpp"

#include <cmath>

#include "rt_mesh.hpp"

namespace rt {
namespace {
// glam::DVec3 arithmetic as trispline.rs uses it
inline D3 add(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline D3 sub(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline D3 mul(D3 a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }
inline D3 lmul(double s, D3 a) { return D3{s * a.x, s * a.y, s * a.z}; }
inline D3 neg(D3 a) { return D3{-a.x, -a.y, -a.z}; }
inline D3 cross(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double gdot(D3 a, D3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline double glen(D3 a) { return std::sqrt(gdot(a, a)); }
inline D3 gnormalize(D3 a) { return mul(a, 1.0 / glen(a)); }
inline D3 coef(const rtg_trispline &s, int i) { return D3{s.m[i][0], s.m[i][1], s.m[i][2]}; }
inline void put(rtg_trispline &s, int i, D3 v) {
    s.m[i][0] = v.x;
    s.m[i][1] = v.y;
    s.m[i][2] = v.z;
}
enum { M00 = 0, M01, M02, M03, M10, M11, M12, M20, M21, M30 };

struct Hermite {
    D3 p0, p1, d0, d1;
};
struct Cubic {
    D3 a, b, c, d;
};
void normalize_speed(Hermite &h) {                       // trispline.rs:144-150
    double distance = glen(sub(h.p0, h.p1));
    h.d0 = mul(gnormalize(h.d0), distance);
    h.d1 = mul(gnormalize(h.d1), distance);
}
Cubic to_cubic(const Hermite &h) {                       // trispline.rs:126-134
    Cubic c;
    c.a = add(add(sub(lmul(2.0, h.p0), lmul(2.0, h.p1)), lmul(1.0, h.d0)), lmul(1.0, h.d1));
    c.b = sub(sub(add(lmul(-3.0, h.p0), lmul(3.0, h.p1)), lmul(2.0, h.d0)), lmul(1.0, h.d1));
    c.c = h.d0;
    c.d = h.p0;
    return c;
}
} // namespace

rtg_trispline trispline_from(const NormalHermiteTrispline &nh) {
    // trispline.rs:68-85
    D3 e01 = sub(nh.p2, nh.p01), e20 = sub(nh.p1, nh.p20), e12 = sub(nh.p0, nh.p12);
    Hermite s01{nh.p0, nh.p1, cross(e01, nh.n0), cross(e01, nh.n1)};
    Hermite s02{nh.p0, nh.p2, neg(cross(e20, nh.n0)), neg(cross(e20, nh.n2))};
    Hermite s21{nh.p2, nh.p1, neg(cross(e12, nh.n2)), neg(cross(e12, nh.n1))};
    normalize_speed(s01);
    normalize_speed(s02);
    normalize_speed(s21);
    // trispline.rs:87-124
    Cubic f = to_cubic(s01), g = to_cubic(s02), h = to_cubic(s21);
    D3 s1 = sub(sub(sub(sub(f.c, g.c), lmul(2.0, g.b)), lmul(3.0, g.a)), h.c);
    D3 s3 = sub(sub(f.a, g.a), h.a);
    D3 m11 = nh.control;
    D3 m12 = sub(neg(m11), s1);
    D3 m21 = add(s3, m12);
    rtg_trispline out;
    put(out, M00, f.d);
    put(out, M01, g.c);
    put(out, M02, g.b);
    put(out, M03, g.a);
    put(out, M10, f.c);
    put(out, M11, m11);
    put(out, M12, m12);
    put(out, M20, f.b);
    put(out, M21, m21);
    put(out, M30, f.a);
    return out;
}

D3 trispline_position(const rtg_trispline &s, double t, double u) {      // trispline.rs:162-167
    D3 r = mul(coef(s, M30), t * t * t);
    r = add(r, mul(coef(s, M20), t * t));
    r = add(r, mul(coef(s, M10), t));
    r = add(r, coef(s, M00));
    r = add(r, mul(coef(s, M21), t * t * u));
    r = add(r, mul(coef(s, M11), t * u));
    r = add(r, mul(coef(s, M01), u));
    r = add(r, mul(coef(s, M12), t * u * u));
    r = add(r, mul(coef(s, M02), u * u));
    r = add(r, mul(coef(s, M03), u * u * u));
    return r;
}
D3 trispline_ddt(const rtg_trispline &s, double t, double u) {           // trispline.rs:168-172
    D3 r = mul(coef(s, M30), 3.0 * t * t);
    r = add(r, mul(coef(s, M20), 2.0 * t));
    r = add(r, coef(s, M10));
    r = add(r, mul(coef(s, M21), 2.0 * t * u));
    r = add(r, mul(coef(s, M11), u));
    r = add(r, mul(coef(s, M12), u * u));
    return r;
}
D3 trispline_ddu(const rtg_trispline &s, double t, double u) {           // trispline.rs:173-177
    D3 r = mul(coef(s, M21), t * t);
    r = add(r, mul(coef(s, M11), t));
    r = add(r, coef(s, M01));
    r = add(r, mul(coef(s, M12), t * 2.0 * u));
    r = add(r, mul(coef(s, M02), 2.0 * u));
    r = add(r, mul(coef(s, M03), 3.0 * u * u));
    return r;
}
D3 trispline_normal(const rtg_trispline &s, double t, double u) {        // trispline.rs:178-180
    return gnormalize(cross(trispline_ddt(s, t, u), trispline_ddu(s, t, u)));
}

void tessellate(const rtg_trispline *patches, uint64_t n, uint32_t k, rtg_tri *out_tris, rtg_patch_leaf *out_info) {
    uint64_t w = 0;
    const double kk = (double)k;
    for (uint64_t p = 0; p < n; p++) {
        auto facet = [&](double ta, double ua, double tb, double ub, double tc, double uc) {
            const double tu[3][2] = {{ta, ua}, {tb, ub}, {tc, uc}};
            for (int c = 0; c < 3; c++) {
                D3 q = trispline_position(patches[p], tu[c][0], tu[c][1]);
                out_tris[w].p[c][0] = q.x;
                out_tris[w].p[c][1] = q.y;
                out_tris[w].p[c][2] = q.z;
                out_info[w].tu[c][0] = tu[c][0];
                out_info[w].tu[c][1] = tu[c][1];
            }
            out_info[w].patch = (uint32_t)p;
            out_info[w]._pad = 0;
            w++;
        };
        for (uint32_t i = 0; i < k; i++)
            for (uint32_t j = 0; i + j < k; j++) {
                double t0 = (double)i / kk, t1 = (double)(i + 1) / kk, u0 = (double)j / kk, u1 = (double)(j + 1) / kk;
                facet(t0, u0, t1, u0, t0, u1);
                if (i + j + 1 < k) facet(t1, u0, t1, u1, t0, u1);
            }
    }
}

std::vector<rtg_trispline> patch_terrain(uint32_t nx, uint32_t ny, uint64_t seed, double scale) {
    // vertices of the same terrain as heightfield(), then one patch per triangle
    std::vector<rtg_tri> tris((size_t)2 * nx * ny);
    heightfield(nx, ny, seed, scale, tris.data());
    const size_t vw = nx + 1;
    std::vector<D3> pos(vw * (ny + 1)), nrm(vw * (ny + 1), D3{0, 0, 0});
    auto vid = [&](uint32_t i, uint32_t j) { return (size_t)j * vw + i; };
    // triangle 2q is (P00, P10, P11), triangle 2q+1 is (P00, P11, P01) of quad q = j*nx + i
    auto corners = [&](size_t t, size_t id[3]) {
        size_t q = t / 2;
        uint32_t i = (uint32_t)(q % nx), j = (uint32_t)(q / nx);
        if (t % 2 == 0) {
            id[0] = vid(i, j);
            id[1] = vid(i + 1, j);
            id[2] = vid(i + 1, j + 1);
        } else {
            id[0] = vid(i, j);
            id[1] = vid(i + 1, j + 1);
            id[2] = vid(i, j + 1);
        }
    };
    for (size_t t = 0; t < tris.size(); t++) {
        size_t id[3];
        corners(t, id);
        D3 v[3];
        for (int c = 0; c < 3; c++) {
            v[c] = D3{tris[t].p[c][0], tris[t].p[c][1], tris[t].p[c][2]};
            pos[id[c]] = v[c];
        }
        D3 fn = gnormalize(cross(sub(v[1], v[0]), sub(v[2], v[1])));
        for (int c = 0; c < 3; c++) nrm[id[c]] = add(nrm[id[c]], fn);
    }
    for (D3 &n : nrm) n = gnormalize(n);
    std::vector<rtg_trispline> out(tris.size());
    for (size_t t = 0; t < tris.size(); t++) {
        size_t id[3];
        corners(t, id);
        NormalHermiteTrispline nh;
        nh.p0 = pos[id[0]];
        nh.p1 = pos[id[1]];
        nh.p2 = pos[id[2]];
        // the vertex opposite each edge, mirrored through the edge: (p2 - p01) etc. are then perpendicular to
        // their edge, so a flat triangle with equal normals gives straight boundary curves
        auto perp = [](D3 a, D3 e) { return sub(a, mul(e, gdot(a, e) / gdot(e, e))); };
        nh.p01 = sub(nh.p2, lmul(2.0, perp(sub(nh.p2, nh.p0), sub(nh.p1, nh.p0))));
        nh.p20 = sub(nh.p1, lmul(2.0, perp(sub(nh.p1, nh.p0), sub(nh.p2, nh.p0))));
        nh.p12 = sub(nh.p0, lmul(2.0, perp(sub(nh.p0, nh.p1), sub(nh.p2, nh.p1))));
        nh.n0 = nrm[id[0]];
        nh.n1 = nrm[id[1]];
        nh.n2 = nrm[id[2]];
        nh.control = D3{0.0, 0.0, 0.0};
        out[t] = trispline_from(nh);
    }
    return out;
}

} // namespace rt
