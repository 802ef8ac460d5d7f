// rt_bvh_cluster.cpp -- BvhForest::contract_once / contract (src/tree/bvh_cluster.rs:42-116) and
// BvhForest::hybrid (src/tree/bvh_joint.rs:3-8), followed by Bvh::new (src/tree/bvh.rs:80-106).
//
// contract_once pairs the trees of a forest greedily.  Every tree looks at the other trees in order of
// "distance" = surface area of the union of the two boxes (tree/bv_kd_tree.rs:34-36); the tree whose nearest
// remaining neighbour is FARTHEST is taken first (the reference pops the maximum of its priority queue,
// bvh_cluster.rs:67) and joined with that neighbour, `node.add_node(popped); node.add_node(neighbour)`
// (bvh_cluster.rs:103-109); trees that were looking at either of the two move on to their next remaining
// neighbour (bvh_cluster.rs:73-99).  The new forest lists the pairs in the order they were made, then the tree
// left over when the count is odd (bvh_cluster.rs:111-115).
//
// Where two distances are equal the reference's choice depends on the internals of the priority-queue crate
// (1.2.1, not vendored) and of slice::select_nth_unstable, so it cannot be restated; here the lower index wins
// both choices (nearest neighbour = smallest (distance, index); popped tree = largest distance, then lowest
// index).  Inputs without exact ties give the reference's forest.
//
// Neighbour search: a median-split k-d tree over the box centres (tree/kd_tree.rs:79-95) searched depth-first with
// the reference's plane bound (the query box stretched to the splitting plane, bv_kd_tree.rs:37-46); subtrees
// without a remaining tree are skipped.
#include "rt_bvh.hpp"

#include <algorithm>
#include <limits>
#include <queue>
#include <thread>

namespace rt {
namespace {

struct PoolNode {
    Bounds b;
    int32_t left, right;   // children in the pool, -1 for a single-leaf tree
    uint32_t tri;          // source triangle of a single-leaf tree
};

struct Neighbour {
    double distance;
    uint32_t index;
};
constexpr uint32_t kNone = 0xFFFFFFFFu;

inline bool closer(double d, uint32_t i, const Neighbour &n) { return d < n.distance || (d == n.distance && i < n.index); }

struct KdSearch {
    const std::vector<Bounds> &boxes;
    std::vector<uint32_t> order;     // element at every position of the implicit tree (a range's root is its middle)
    std::vector<uint8_t> axis;       // split axis of the range rooted at a position
    std::vector<uint32_t> alive;     // remaining elements in the range rooted at a position
    std::vector<uint32_t> pos_of;

    explicit KdSearch(const std::vector<Bounds> &bx) : boxes(bx) {
        const size_t n = bx.size();
        order.resize(n);
        axis.assign(n, 0);
        alive.assign(n, 0);
        pos_of.resize(n);
        for (size_t i = 0; i < n; i++) order[i] = (uint32_t)i;
        build(0, n);
        for (size_t p = 0; p < n; p++) pos_of[order[p]] = (uint32_t)p;
    }
    void build(size_t lo, size_t hi) {
        if (lo >= hi) return;
        const size_t mid = lo + (hi - lo) / 2;
        Bounds b = Bounds::empty();
        for (size_t p = lo; p < hi; p++) {
            Bounds c;
            for (int k = 0; k < 3; k++) c.min[k] = c.max[k] = boxes[order[p]].center(k);
            b = b.union_(c);
        }
        int a = 0;
        for (int k = 1; k < 3; k++)
            if (b.dim(k) >= b.dim(a)) a = k;
        std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi, [&](uint32_t x, uint32_t y) {
            double cx = boxes[x].center(a), cy = boxes[y].center(a);
            return cx < cy || (cx == cy && x < y);
        });
        axis[mid] = (uint8_t)a;
        alive[mid] = (uint32_t)(hi - lo);
        build(lo, mid);
        build(mid + 1, hi);
    }
    void kill(uint32_t e) {
        size_t lo = 0, hi = order.size();
        const size_t p = pos_of[e];
        for (;;) {
            const size_t mid = lo + (hi - lo) / 2;
            alive[mid]--;
            if (p == mid) return;
            if (p < mid)
                hi = mid;
            else
                lo = mid + 1;
        }
    }
    void search(size_t lo, size_t hi, uint32_t q, const Bounds &qb, const std::vector<uint8_t> &dead,
                Neighbour &best) const {
        if (lo >= hi) return;
        const size_t mid = lo + (hi - lo) / 2;
        if (alive[mid] == 0) return;
        const uint32_t c = order[mid];
        const int a = axis[mid];
        const double plane = boxes[c].center(a);
        const bool left_first = qb.center(a) < plane;                              // kd_tree.rs:174-178
        if (left_first)
            search(lo, mid, q, qb, dead, best);
        else
            search(mid + 1, hi, q, qb, dead, best);
        if (!dead[c] && c != q) {
            double d = qb.union_(boxes[c]).surface_area();                           // bv_kd_tree.rs:34-36
            if (closer(d, c, best)) best = Neighbour{d, c};
        }
        Bounds stretched = qb;                                                       // bv_kd_tree.rs:37-46
        if (plane < stretched.min[a]) stretched.min[a] = plane;
        if (plane > stretched.max[a]) stretched.max[a] = plane;
        if (stretched.surface_area() <= best.distance) {
            if (left_first)
                search(mid + 1, hi, q, qb, dead, best);
            else
                search(lo, mid, q, qb, dead, best);
        }
    }
    Neighbour nearest(uint32_t q, const std::vector<uint8_t> &dead) const {
        Neighbour best{std::numeric_limits<double>::infinity(), kNone};
        search(0, order.size(), q, boxes[q], dead, best);
        return best;
    }
};

struct Forest {
    std::vector<PoolNode> pool;
    std::vector<int32_t> trees;     // roots, in forest order

    int32_t join(int32_t l, int32_t r) {                                             // BvhTree::new + add_node x 2
        PoolNode n;
        n.b = Bounds::empty().union_(pool[l].b).union_(pool[r].b);
        n.left = l;
        n.right = r;
        n.tri = kNone;
        pool.push_back(n);
        return (int32_t)pool.size() - 1;
    }

    void contract_once(int threads) {
        const size_t n = trees.size();
        if (n < 2) return;
        std::vector<Bounds> boxes(n);
        for (size_t i = 0; i < n; i++) boxes[i] = pool[trees[i]].b;
        KdSearch kd(boxes);
        std::vector<uint8_t> dead(n, 0);
        std::vector<Neighbour> head(n);
        {
            int t = std::max(1, std::min<int>(threads, (int)(n / 4096) + 1));
            std::vector<std::thread> pool_threads;
            for (int w = 0; w < t; w++)
                pool_threads.emplace_back([&, w]() {
                    for (size_t i = n * w / t; i < n * (w + 1) / t; i++) head[i] = kd.nearest((uint32_t)i, dead);
                });
            for (auto &th : pool_threads) th.join();
        }
        struct Entry {
            double distance;
            uint32_t index, version;
            bool operator<(const Entry &o) const {                                    // max distance, then lowest index
                return distance < o.distance || (distance == o.distance && index > o.index);
            }
        };
        std::priority_queue<Entry> graph;
        std::vector<uint32_t> version(n, 0);
        std::vector<std::vector<uint32_t>> reverse(n);
        for (size_t i = 0; i < n; i++) {
            reverse[head[i].index].push_back((uint32_t)i);
            graph.push(Entry{head[i].distance, (uint32_t)i, 0});
        }
        std::vector<std::pair<uint32_t, uint32_t>> contractions;
        size_t remaining = n;
        while (remaining >= 2) {
            Entry e = graph.top();
            graph.pop();
            if (dead[e.index] || e.version != version[e.index]) continue;
            const uint32_t n1 = e.index, n2 = head[n1].index;
            for (uint32_t x : {n1, n2}) {
                dead[x] = 1;
                kd.kill(x);
            }
            remaining -= 2;
            contractions.emplace_back(n1, n2);
            for (uint32_t x : {n1, n2}) {
                std::vector<uint32_t> watchers;
                watchers.swap(reverse[x]);
                for (uint32_t w : watchers) {
                    if (dead[w] || head[w].index != x) continue;
                    head[w] = kd.nearest(w, dead);
                    version[w]++;
                    if (head[w].index == kNone) continue;                            // the last tree left
                    reverse[head[w].index].push_back(w);
                    graph.push(Entry{head[w].distance, w, version[w]});
                }
            }
        }
        std::vector<int32_t> next;
        next.reserve(n / 2 + 1);
        for (auto &c : contractions) next.push_back(join(trees[c.first], trees[c.second]));
        for (size_t i = 0; i < n; i++)
            if (!dead[i]) next.push_back(trees[i]);
        trees.swap(next);
    }

    // BvhForest::subdivide over trees of any shape (bvh_subdivide.rs:8-43).
    int32_t subdivide(std::vector<int32_t> &ts, size_t b, size_t e) {
        const size_t k = e - b;
        if (k == 1) return ts[b];
        Bounds bounds = Bounds::empty();
        for (size_t i = b; i < e; i++) bounds = bounds.union_(pool[ts[i]].b);
        int axis = 0;
        for (int i = 1; i < 3; i++)
            if (bounds.dim(i) >= bounds.dim(axis)) axis = i;
        const double want = bounds.min[axis] + bounds.dim(axis) / 2.0;
        std::stable_sort(ts.begin() + b, ts.begin() + e,
                         [&](int32_t x, int32_t y) { return pool[x].b.center(axis) < pool[y].b.center(axis); });
        size_t size = k, left = 0, right = k;
        while (left < right) {
            size_t mid = left + size / 2;
            double v = pool[ts[b + mid]].b.center(axis);
            if (v < want)
                left = mid + 1;
            else if (v > want)
                right = mid;
            else {
                left = mid;
                break;
            }
            size = right - left;
        }
        size_t split = left;
        size_t min_split = std::max<size_t>((size_t)((double)k * 0.01), 1);
        size_t max_split = k - min_split;
        if (split < min_split) split = min_split;
        if (split > max_split) split = max_split;
        int32_t l = subdivide(ts, b, b + split);
        int32_t r = subdivide(ts, b + split, e);
        return join(l, r);
    }
};

rtg_bvh_node entry_of(const Bounds &b, uint32_t cb, uint32_t ce, uint32_t lb, uint32_t le) {
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

// Bvh::add_bvh_tree (bvh.rs:91-106).
rtg_bvh_node flatten(const Forest &f, int32_t id, const rtg_tri *tris, const double *normals, Bvh &out, uint32_t level,
                     uint32_t &depth) {
    if (level > depth) depth = level;
    const PoolNode &node = f.pool[id];
    const uint32_t leaf_start = (uint32_t)out.leaves.size();
    if (node.left < 0) {
        out.leaves.push_back(tris[node.tri]);
        out.leaf_source.push_back(node.tri);
        if (normals)
            out.vertex_normals.insert(out.vertex_normals.end(), normals + (size_t)node.tri * 9,
                                      normals + (size_t)node.tri * 9 + 9);
        const uint32_t at = (uint32_t)out.nodes.size();
        return entry_of(node.b, at, at, leaf_start, leaf_start + 1);
    }
    rtg_bvh_node l = flatten(f, node.left, tris, normals, out, level + 1, depth);
    rtg_bvh_node r = flatten(f, node.right, tris, normals, out, level + 1, depth);
    const uint32_t nodes_start = (uint32_t)out.nodes.size();
    out.nodes.push_back(l);
    out.nodes.push_back(r);
    return entry_of(node.b, nodes_start, nodes_start + 2, leaf_start, leaf_start);
}

Bvh build_with(const rtg_tri *tris, uint64_t n, const double *vertex_normals, int threads, bool full_contract) {
    Bvh out;
    if (threads <= 0) {
        threads = (int)std::thread::hardware_concurrency();
        if (threads <= 0) threads = 1;
        if (threads > 16) threads = 16;
    }
    if (n == 0) {
        out.nodes.push_back(entry_of(Bounds::empty(), 0, 0, 0, 0));
        out.root = 0;
        out.depth = 1;
        return out;
    }
    Forest f;
    f.pool.reserve((size_t)(2 * n));
    f.trees.resize((size_t)n);
    for (uint64_t i = 0; i < n; i++) {                                               // BvhForest::new (bvh.rs:48-58)
        PoolNode leaf;
        leaf.b = Bounds::empty().union_(triangle_bounds(tris[i]));
        leaf.left = leaf.right = -1;
        leaf.tri = (uint32_t)i;
        f.pool.push_back(leaf);
        f.trees[(size_t)i] = (int32_t)i;
    }
    int32_t root;
    if (full_contract) {
        while (f.trees.size() > 1) f.contract_once(threads);                         // bvh_cluster.rs:43-48
        root = f.trees[0];
    } else {
        f.contract_once(threads);                                                    // bvh_joint.rs:4-7
        std::vector<int32_t> ts = f.trees;
        root = f.subdivide(ts, 0, ts.size());
    }
    out.nodes.reserve((size_t)(2 * n - 1));
    out.leaves.reserve((size_t)n);
    out.leaf_source.reserve((size_t)n);
    uint32_t depth = 0;
    rtg_bvh_node r = flatten(f, root, tris, vertex_normals, out, 1, depth);
    out.root = (uint32_t)out.nodes.size();
    out.nodes.push_back(r);
    out.depth = depth;
    return out;
}

} // namespace

Bvh build_contract(const rtg_tri *tris, uint64_t n, const double *vertex_normals, int threads) {
    return build_with(tris, n, vertex_normals, threads, true);
}
Bvh build_hybrid(const rtg_tri *tris, uint64_t n, const double *vertex_normals, int threads) {
    return build_with(tris, n, vertex_normals, threads, false);
}

} // namespace rt
