// rt_bvh.cpp -- BvhForest::subdivide + Bvh::new on index arrays (see rt_bvh.hpp).
#include "rt_bvh.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <thread>

namespace rt {

double f64_minimum(double a, double b) {
    if (a != a) return a;
    if (b != b) return b;
    if (a < b) return a;
    if (b < a) return b;
    return std::signbit(a) ? a : b;
}
double f64_maximum(double a, double b) {
    if (a != a) return a;
    if (b != b) return b;
    if (a > b) return a;
    if (b > a) return b;
    return std::signbit(a) ? b : a;
}

Bounds Bounds::empty() {
    const double inf = std::numeric_limits<double>::infinity();
    return Bounds{{inf, inf, inf}, {-inf, -inf, -inf}};
}
Bounds Bounds::union_(const Bounds &o) const {
    Bounds r;
    for (int i = 0; i < 3; i++) {
        r.min[i] = f64_minimum(min[i], o.min[i]);
        r.max[i] = f64_maximum(max[i], o.max[i]);
    }
    return r;
}
double Bounds::center(int axis) const { return (min[axis] + max[axis]) / 2.0; }
double Bounds::dim(int axis) const { return max[axis] - min[axis]; }
double Bounds::surface_area() const {
    double dx = dim(0), dy = dim(1), dz = dim(2);
    return (dx * dy + dx * dz + dy * dz) * 2.0;
}

Bounds triangle_bounds(const rtg_tri &t) {
    Bounds b = Bounds::empty();
    for (int v = 0; v < 3; v++) {
        Bounds p;
        for (int i = 0; i < 3; i++) p.min[i] = p.max[i] = t.p[v][i];
        b = b.union_(p);
    }
    return b;
}

Bounds Bvh::bounds() const {
    Bounds b = Bounds::empty();
    if (nodes.empty()) return b;
    for (int i = 0; i < 3; i++) {
        b.min[i] = nodes[root].lo[i];
        b.max[i] = nodes[root].hi[i];
    }
    return b;
}
rtg_bvh Bvh::view() const {
    rtg_bvh v;
    v.nodes = nodes.data();
    v.n_nodes = nodes.size();
    v.leaves = leaves.data();
    v.n_leaves = leaves.size();
    v.vertex_normals = vertex_normals.empty() ? nullptr : vertex_normals.data();
    v.root = root;
    return v;
}

namespace {

// Binary build tree in pre-order: the subtree over k triangles occupies exactly
// 2k-1 consecutive slots, so slots can be assigned without synchronisation.
struct BuildNode {
    Bounds bounds;
    uint32_t tri;    // source triangle for a single-leaf tree, else UINT32_MAX
    uint32_t left_count;
};

struct Builder {
    const rtg_tri *tris;
    std::vector<Bounds> tb;
    std::vector<double> center[3];
    std::vector<uint32_t> ids;
    std::vector<BuildNode> tree;

    template <class Cmp> void sort_range(uint32_t *b, uint32_t *e, Cmp cmp, int threads) {
        size_t k = (size_t)(e - b);
        if (threads <= 1 || k < (1u << 15)) {
            std::stable_sort(b, e, cmp);
            return;
        }
        uint32_t *mid = b + k / 2;
        int lt = threads / 2;
        std::thread t([&]() { sort_range(b, mid, cmp, lt); });
        sort_range(mid, e, cmp, threads - lt);
        t.join();
        std::inplace_merge(b, mid, e, cmp);   // stable: equal keys keep left-before-right
    }

    void build(size_t b, size_t e, size_t slot, int threads) {
        size_t k = e - b;
        BuildNode &node = tree[slot];
        if (k == 1) {                                             // subdivide: one tree is returned as is
            node.bounds = tb[ids[b]];
            node.tri = ids[b];
            node.left_count = 0;
            return;
        }
        // subdivide_once (bvh_subdivide.rs:21-43)
        Bounds bounds = Bounds::empty();
        for (size_t i = b; i < e; i++) bounds = bounds.union_(tb[ids[i]]);
        int axis = 0;                                             // max_by_key keeps the last maximum
        for (int i = 1; i < 3; i++)
            if (bounds.dim(i) >= bounds.dim(axis)) axis = i;
        const double split_coordinate = bounds.min[axis] + bounds.dim(axis) / 2.0;
        const double *key = center[axis].data();
        sort_range(ids.data() + b, ids.data() + e, [key](uint32_t x, uint32_t y) { return key[x] < key[y]; }, threads);
        // slice::binary_search_by of the std library the reference used (1.52-1.81):
        // probe left + size/2; the first probed equal element is returned.
        size_t size = k, left = 0, right = k, split;
        bool found = false;
        while (left < right) {
            size_t mid = left + size / 2;
            double v = key[ids[b + mid]];
            if (v < split_coordinate)
                left = mid + 1;
            else if (v > split_coordinate)
                right = mid;
            else {
                left = mid;
                found = true;
                break;
            }
            size = right - left;
        }
        (void)found;
        split = left;
        size_t min_split = std::max<size_t>((size_t)((double)k * 0.01), 1);
        size_t max_split = k - min_split;
        if (split < min_split) split = min_split;
        if (split > max_split) split = max_split;
        node.tri = UINT32_MAX;
        node.left_count = (uint32_t)split;
        size_t lslot = slot + 1, rslot = slot + 2 * split;
        if (threads > 1 && k >= (1u << 14)) {
            int lt = std::max(1, (int)((double)threads * (double)split / (double)k + 0.5));
            if (lt >= threads) lt = threads - 1;
            std::thread t([&]() { build(b, b + split, lslot, lt); });
            build(b + split, e, rslot, threads - lt);
            t.join();
        } else {
            build(b, b + split, lslot, 1);
            build(b + split, e, rslot, 1);
        }
        // BvhTree::add_node twice (bvh.rs:72-75)
        node.bounds = Bounds::empty().union_(tree[lslot].bounds).union_(tree[rslot].bounds);
    }
};

rtg_bvh_node make_entry(const Bounds &b, uint32_t cb, uint32_t ce, uint32_t lb, uint32_t le) {
    rtg_bvh_node n;
    for (int i = 0; i < 3; i++) {
        n.lo[i] = b.min[i];
        n.hi[i] = b.max[i];
    }
    n.child_begin = cb;
    n.child_end = ce;
    n.leaf_begin = lb;
    n.leaf_end = le;
    return n;
}

// Bvh::add_bvh_tree (bvh.rs:91-106): own leaves first, then every child's
// subtree, then the children's entries contiguously.
rtg_bvh_node flatten(const Builder &bl, size_t slot, const rtg_tri *tris, const double *normals, Bvh &out,
                     uint32_t level, uint32_t &depth) {
    if (level > depth) depth = level;
    const BuildNode &node = bl.tree[slot];
    uint32_t leaf_start = (uint32_t)out.leaves.size();
    if (node.tri != UINT32_MAX) {
        out.leaves.push_back(tris[node.tri]);
        out.leaf_source.push_back(node.tri);
        if (normals)
            out.vertex_normals.insert(out.vertex_normals.end(), normals + (size_t)node.tri * 9,
                                      normals + (size_t)node.tri * 9 + 9);
    }
    uint32_t leaf_end = (uint32_t)out.leaves.size();
    if (node.tri != UINT32_MAX) {
        uint32_t at = (uint32_t)out.nodes.size();
        return make_entry(node.bounds, at, at, leaf_start, leaf_end);
    }
    rtg_bvh_node l = flatten(bl, slot + 1, tris, normals, out, level + 1, depth);
    rtg_bvh_node r = flatten(bl, slot + 2 * (size_t)node.left_count, tris, normals, out, level + 1, depth);
    uint32_t nodes_start = (uint32_t)out.nodes.size();
    out.nodes.push_back(l);
    out.nodes.push_back(r);
    return make_entry(node.bounds, nodes_start, nodes_start + 2, leaf_start, leaf_end);
}

} // namespace

Bvh build_subdivide(const rtg_tri *tris, uint64_t n, const double *vertex_normals, int threads) {
    Bvh out;
    if (threads <= 0) {
        threads = (int)std::thread::hardware_concurrency();
        if (threads <= 0) threads = 1;
        if (threads > 16) threads = 16;
    }
    if (n == 0) {                                                 // BvhTree::new(): empty bounds, nothing else
        out.nodes.push_back(make_entry(Bounds::empty(), 0, 0, 0, 0));
        out.root = 0;
        out.depth = 1;
        return out;
    }
    Builder bl;
    bl.tris = tris;
    bl.tb.resize((size_t)n);
    for (int a = 0; a < 3; a++) bl.center[a].resize((size_t)n);
    bl.ids.resize((size_t)n);
    for (uint64_t i = 0; i < n; i++) {
        bl.tb[i] = Bounds::empty().union_(triangle_bounds(tris[i]));   // BvhTree::add_leaf (bvh.rs:68-71)
        for (int a = 0; a < 3; a++) bl.center[a][i] = bl.tb[i].center(a);
        bl.ids[i] = (uint32_t)i;
    }
    bl.tree.resize((size_t)(2 * n - 1));
    bl.build(0, (size_t)n, 0, threads);
    out.nodes.reserve((size_t)(2 * n - 1));
    out.leaves.reserve((size_t)n);
    out.leaf_source.reserve((size_t)n);
    if (vertex_normals) out.vertex_normals.reserve((size_t)n * 9);
    uint32_t depth = 0;
    rtg_bvh_node root = flatten(bl, 0, tris, vertex_normals, out, 1, depth);
    out.root = (uint32_t)out.nodes.size();
    out.nodes.push_back(root);
    out.depth = depth;
    return out;
}

} // namespace rt
