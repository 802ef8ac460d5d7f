// rtg_build.hip -- BvhForest::new(triangles).subdivide() followed by Bvh::new, on the device.
//
// The tree is the one the CPU builder makes, node for node and bit for bit
// (tree/bvh.rs:48-58 one single-leaf tree per triangle with the triangle's box; tree/bvh_subdivide.rs:8-43 the
// recursive median-position split; tree/bvh.rs:80-106 the flattening): what changes is the order of the work.
// The recursion becomes a loop over tree levels in which every segment of the triangle permutation that still
// holds two or more triangles is split at once:
//
//   * subdivide_once's stable sort of a segment by the box centre on the chosen axis is one device-wide stable
//     radix sort per level, on the key (segment start, rank of the centre on the segment's axis).  Ranks come from
//     three sorts of all centres made once up front; equal centres share a rank, so ties keep their previous order
//     exactly as the stable comparison sort keeps them.
//   * the union of a segment's boxes (bvh_subdivide.rs:22 and BvhTree::add_node, bvh.rs:72-75) is a min/max
//     reduction, exact in any order: every lane walks a short stretch of the permutation, stores the box of a
//     segment that lies wholly inside it, and the rest is combined across the wave and the workgroup before 64-bit
//     integer atomics on an order-preserving image of the doubles (-0.0 below +0.0, as IEEE minimum/maximum
//     order them, geo/bounds.rs:28-33).
//   * the binary search for the split coordinate follows the probe order of the standard library the reference
//     used (first probed equal element wins), then the 1 % / 99 % clamp (bvh_subdivide.rs:29-38).
//
// Build nodes live in pre-order slots (a subtree over k triangles owns 2k-1 consecutive slots); the position of a
// node's entry in the reference's flattened order (children after all descendants, root last) follows from the
// triangle counts.  Positions in the threaded arrays the traversal kernels read (siblings side by side in pair slots,
// root last) come from ranks: first the pairs under the T inner nodes of the top kThreadedTopLevels levels, level by
// level (rank by depth, then slot), then the pairs of all deeper inner nodes line by line as PairSlots
// (rtg_pair_slots.hpp) lays them out, from the numbers of leaders of either kind before a leader in pre-order.  The
// children of pair g sit at 2 g and 2 g + 1.  The last two kernels write the mesh slab, the reference-order node
// array and the leaf -> source-triangle map.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include <string>
#include <vector>

#include "rtg_launch.hpp"

namespace rtg {
namespace {

struct BuildNode {
    uint32_t begin, count;   // segment of the triangle permutation
    uint32_t base;           // entries of the flattened array emitted before this subtree's descendants
    uint32_t ref;            // index of this node's entry in the flattened array
    uint32_t split;          // triangles in the left child (0 while unsplit / for a leaf)
    uint32_t axis;           // bits 0..1 sort axis, bits 8..15 depth + 1 of an inner node of the top levels (else 0),
                             // bit 31 = last of its siblings
    uint32_t pos;            // position in the threaded arrays (set by k_place_*)
    uint32_t chain;          // slot where this node's chain of first children starts below the top levels
};

constexpr uint64_t kSign = 0x8000000000000000ull;
__host__ __device__ inline uint64_t ord_of(double x) {
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    return (u & kSign) ? ~u : (u | kSign);
}
__host__ __device__ inline double ord_back(uint64_t k) {
    uint64_t u = (k & kSign) ? (k ^ kSign) : ~k;
    double x;
    __builtin_memcpy(&x, &u, 8);
    return x;
}
__device__ inline uint64_t shfl_down_u64(uint64_t v, int d) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_down(lo, d, 64);
    hi = __shfl_down(hi, d, 64);
    return ((uint64_t)hi << 32) | lo;
}

// Triangle boxes (geo/triangle.rs:43-45), box centres (geo/bounds.rs:38-40) and the sort image of the centres
// (-0.0 folded onto +0.0: the two compare equal in the reference's sort).  flags bit 0: a vertex is not finite.
__global__ void k_prepare(const double *__restrict__ tris, uint32_t n, double *__restrict__ tb,
                          double *__restrict__ centre, double *__restrict__ sort_key, uint32_t *__restrict__ iota,
                          uint32_t *__restrict__ flags) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *p = tris + (size_t)i * 9;
    bool finite = true;
    for (int a = 0; a < 3; a++) {
        double v0 = p[a], v1 = p[3 + a], v2 = p[6 + a];
        finite = finite && (fabs(v0) < kInf) && (fabs(v1) < kInf) && (fabs(v2) < kInf);
        uint64_t o0 = ord_of(v0), o1 = ord_of(v1), o2 = ord_of(v2);
        uint64_t mn = o0 < o1 ? o0 : o1, mx = o0 > o1 ? o0 : o1;
        mn = mn < o2 ? mn : o2;
        mx = mx > o2 ? mx : o2;
        double lo = ord_back(mn), hi = ord_back(mx);
        tb[(size_t)i * 6 + a] = lo;
        tb[(size_t)i * 6 + 3 + a] = hi;
        double c = (lo + hi) / 2.0;
        centre[(size_t)a * n + i] = c;
        sort_key[(size_t)a * n + i] = c == 0.0 ? 0.0 : c;
    }
    iota[i] = i;
    if (!finite) atomicOr(flags, 1u);
}

__global__ void k_run_heads(const double *__restrict__ sorted, uint32_t n, uint32_t *__restrict__ heads) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    heads[j] = (j == 0 || sorted[j] != sorted[j - 1]) ? j : 0u;
}

__global__ void k_scatter_rank(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ run_start, uint32_t n,
                               uint32_t *__restrict__ rank) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    rank[perm[j]] = run_start[j];
}

__global__ void k_init(BuildNode *nodes, uint64_t *bounds, uint32_t n) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        nodes[0] = BuildNode{0u, n, 0u, 2u * n - 2u, 0u, 0x80000000u, 2u * n - 2u, 0u};
        for (int k = 0; k < 3; k++) {
            bounds[k] = ord_of(kInf);
            bounds[3 + k] = ord_of(-kInf);
        }
    }
}

// Moves every triangle of a segment that was just split (or, with `initial`, every triangle into the root) to its
// child segment and adds its box to the child's.  A lane walks kAssignPerLane consecutive triangles: a child
// segment that begins inside the lane's stretch ends there too or runs on into the next lanes; in the first case
// the lane is the only writer of that child's box and stores it, otherwise the partial boxes are combined across
// the wave (a child's lanes are neighbours) and across the block's four waves before the integer atomics, so the
// few huge segments of the first levels cost a handful of atomics per block instead of one per wave.
constexpr int kAssignPerLane = 16;
constexpr uint32_t kNoChild = 0xFFFFFFFFu;
__global__ __launch_bounds__(256) void k_assign(const BuildNode *__restrict__ nodes, uint32_t *__restrict__ seg,
                                                const uint32_t *__restrict__ ids, const double *__restrict__ tb,
                                                uint32_t n, int initial, unsigned long long *__restrict__ bounds) {
    __shared__ uint32_t s_child[4];
    __shared__ uint64_t s_v[4][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t first = (((uint64_t)blockIdx.x * 4 + wave) * 64 + lane) * kAssignPerLane;
    uint32_t cur = kNoChild, cur_begin = 0, parent = kNoChild;
    BuildNode nd{};
    uint64_t v[6];
    auto reset = [&]() {
        for (int k = 0; k < 3; k++) {
            v[k] = ~0ull;
            v[3 + k] = 0ull;
        }
    };
    auto flush_atomic = [&](uint32_t child, const uint64_t *w) {
        unsigned long long *b = bounds + (size_t)child * 6;
        for (int k = 0; k < 3; k++) {
            atomicMin(b + k, (unsigned long long)w[k]);
            atomicMax(b + 3 + k, (unsigned long long)w[3 + k]);
        }
    };
    reset();
    for (int e = 0; e < kAssignPerLane; e++) {
        const uint64_t i = first + e;
        if (i >= n) break;
        uint32_t child = kNoChild, child_begin = 0;
        if (initial) {
            child = 0;
        } else {
            uint32_t x = seg[i];
            if (x != parent) {
                parent = x;
                nd = nodes[x];
            }
            if (nd.count >= 2) {
                bool left = (uint32_t)i - nd.begin < nd.split;
                child = left ? x + 1 : x + 2 * nd.split;
                child_begin = left ? nd.begin : nd.begin + nd.split;
                seg[i] = child;
            }
        }
        if (child != cur) {
            if (cur != kNoChild) {
                if (cur_begin >= first && !initial) {        // the whole child segment was walked by this lane
                    for (int k = 0; k < 6; k++) bounds[(size_t)cur * 6 + k] = v[k];
                } else {
                    flush_atomic(cur, v);
                }
            }
            cur = child;
            cur_begin = child_begin;
            reset();
        }
        if (child != kNoChild) {
            const double *b = tb + (size_t)ids[i] * 6;
            for (int k = 0; k < 3; k++) {
                uint64_t lo = ord_of(b[k]), hi = ord_of(b[3 + k]);
                v[k] = lo < v[k] ? lo : v[k];
                v[3 + k] = hi > v[3 + k] ? hi : v[3 + k];
            }
        }
    }
    // the lanes whose last child is the same are neighbours: after the steps the first of them holds their union
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t oc = __shfl_down(cur, d, 64);
        bool take = (lane + d < 64) && oc == cur;
        for (int k = 0; k < 3; k++) {
            uint64_t olo = shfl_down_u64(v[k], d), ohi = shfl_down_u64(v[3 + k], d);
            if (take) {
                v[k] = olo < v[k] ? olo : v[k];
                v[3 + k] = ohi > v[3 + k] ? ohi : v[3 + k];
            }
        }
    }
    uint32_t prev = __shfl_up(cur, 1, 64);
    const bool head = (lane == 0) || prev != cur;
    const uint32_t cur0 = __shfl(cur, 0, 64);
    const bool uniform = __all(cur == cur0) != 0;
    if (uniform) {
        if (lane == 0) {
            s_child[wave] = cur;
            for (int k = 0; k < 6; k++) s_v[wave][k] = v[k];
        }
    } else {
        if (lane == 0) s_child[wave] = kNoChild;
        if (head && cur != kNoChild) flush_atomic(cur, v);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = kNoChild;
        uint64_t w[6];
        for (int q = 0; q < 4; q++) {
            if (s_child[q] != c) {
                if (c != kNoChild) flush_atomic(c, w);
                c = s_child[q];
                for (int k = 0; k < 6; k++) w[k] = s_v[q][k];
            } else if (c != kNoChild) {
                for (int k = 0; k < 3; k++) {
                    w[k] = s_v[q][k] < w[k] ? s_v[q][k] : w[k];
                    w[3 + k] = s_v[q][3 + k] > w[3 + k] ? s_v[q][3 + k] : w[3 + k];
                }
            }
        }
        if (c != kNoChild) flush_atomic(c, w);
    }
}

// subdivide_once, first half (bvh_subdivide.rs:22-27): the sort axis (the last of the largest extents) and the
// split coordinate of every active segment.
__global__ void k_plan(BuildNode *__restrict__ nodes, const uint64_t *__restrict__ bounds,
                       const uint32_t *__restrict__ active, uint32_t n_active, double *__restrict__ split_coord) {
    uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_active) return;
    uint32_t x = active[a];
    double lo[3], dim[3];
    for (int k = 0; k < 3; k++) {
        lo[k] = ord_back(bounds[(size_t)x * 6 + k]);
        dim[k] = ord_back(bounds[(size_t)x * 6 + 3 + k]) - lo[k];
    }
    int axis = 0;
    for (int k = 1; k < 3; k++)
        if (dim[k] >= dim[axis]) axis = k;
    split_coord[a] = lo[axis] + dim[axis] / 2.0;
    nodes[x].axis = (nodes[x].axis & 0x8000FF00u) | (uint32_t)axis;
}

__global__ void k_keys(const BuildNode *__restrict__ nodes, const uint32_t *__restrict__ seg,
                       const uint32_t *__restrict__ ids, const uint32_t *__restrict__ rank, uint32_t n,
                       uint32_t rank_bits, uint64_t *__restrict__ keys) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BuildNode nd = nodes[seg[i]];
    uint32_t r = nd.count >= 2 ? rank[(size_t)(nd.axis & 3u) * n + ids[i]] : 0u;
    keys[i] = ((uint64_t)nd.begin << rank_bits) | r;
}

// subdivide_once, second half (bvh_subdivide.rs:29-42), and the two child nodes (bvh_subdivide.rs:16-19).
__global__ void k_split(BuildNode *__restrict__ nodes, uint64_t *__restrict__ bounds,
                        const uint32_t *__restrict__ active, uint32_t n_active, const double *__restrict__ split_coord,
                        const uint32_t *__restrict__ ids, const double *__restrict__ centre, uint32_t n,
                        uint32_t *__restrict__ next_active, uint32_t *__restrict__ next_count, uint32_t level) {
    uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_active) return;
    const uint32_t x = active[a];
    BuildNode nd = nodes[x];
    if (level < kThreadedTopLevels) nodes[x].axis = nd.axis | ((level + 1u) << 8);
    const uint32_t b = nd.begin, k = nd.count;
    const double *key = centre + (size_t)(nd.axis & 3u) * n;
    const double want = split_coord[a];
    uint32_t size = k, left = 0, right = k;
    while (left < right) {
        uint32_t mid = left + size / 2;
        double v = key[ids[b + mid]];
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
    uint32_t split = left;
    uint32_t min_split = (uint32_t)((double)k * 0.01);
    if (min_split < 1) min_split = 1;
    uint32_t max_split = k - min_split;
    if (split < min_split) split = min_split;
    if (split > max_split) split = max_split;
    nodes[x].split = split;
    const uint32_t l = x + 1, r = x + 2 * split;
    nodes[l] = BuildNode{b, split, nd.base, nd.base + 2 * k - 4, 0u, 0u, 0u, level >= kThreadedTopLevels ? nd.chain : l};
    nodes[r] = BuildNode{b + split, k - split, nd.base + 2 * split - 2, nd.base + 2 * k - 3, 0u, 0x80000000u, 0u, r};
    for (int c = 0; c < 3; c++) {
        bounds[(size_t)l * 6 + c] = ord_of(kInf);
        bounds[(size_t)l * 6 + 3 + c] = ord_of(-kInf);
        bounds[(size_t)r * 6 + c] = ord_of(kInf);
        bounds[(size_t)r * 6 + 3 + c] = ord_of(-kInf);
    }
    if (split >= 2) next_active[atomicAdd(next_count, 1u)] = l;
    if (k - split >= 2) next_active[atomicAdd(next_count, 1u)] = r;
}

// Positions in the threaded arrays, top block.  k_top_flag marks the inner nodes of the top levels; after an
// exclusive scan of the marks (`before`), k_top_list packs their (depth, slot) keys in slot order and k_place_top
// ranks every key among all of them (a few thousand at most) and writes the two children's positions.
__global__ void k_top_flag(const BuildNode *__restrict__ nodes, uint32_t n_nodes, uint32_t *__restrict__ flag) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_nodes) flag[s] = (nodes[s].axis & 0xFF00u) ? 1u : 0u;
}
__global__ void k_top_list(const BuildNode *__restrict__ nodes, uint32_t n_nodes, const uint32_t *__restrict__ flag,
                           const uint32_t *__restrict__ before, uint64_t *__restrict__ list) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_nodes && flag[s]) list[before[s]] = ((uint64_t)((nodes[s].axis >> 8) & 0xFFu) << 32) | s;
}
__global__ void k_place_top(BuildNode *__restrict__ nodes, const uint64_t *__restrict__ list, uint32_t n_top) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_top) return;
    const uint64_t key = list[j];
    uint32_t g = 0;
    for (uint32_t i = 0; i < n_top; i++) g += list[i] < key ? 1u : 0u;
    const uint32_t s = (uint32_t)key;
    nodes[s + 1].pos = 2 * g;
    nodes[s + 2 * nodes[s].split].pos = 2 * g + 1;
}
// Leaders below the top levels: the inner nodes at an even distance along their chain of first children.  A leader
// whose first child has children is a double (high word), any other a single (low word).
__device__ inline bool deep_leader(const BuildNode &nd, uint32_t s) {
    return nd.count >= 2 && !(nd.axis & 0xFF00u) && ((s - nd.chain) & 1u) == 0;
}
__global__ void k_leader_flags(const BuildNode *__restrict__ nodes, uint32_t n_nodes, uint64_t *__restrict__ flag) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_nodes) return;
    const BuildNode nd = nodes[s];
    uint64_t f = 0;
    if (deep_leader(nd, s)) f = nd.split >= 2 ? 1ull << 32 : 1ull;      // split = triangles of the first child
    flag[s] = f;
}
// before[s]: doubles (high word) and singles (low word) before slot s.  Doubles, the singles that open a line and
// the singles with a slot of their own are placed here; the singles that join a line in k_place_joiners.
__global__ void k_place_leaders(BuildNode *__restrict__ nodes, uint32_t n_nodes, const uint64_t *__restrict__ before,
                                PairSlots ps, uint32_t *__restrict__ opener) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_nodes) return;
    const BuildNode nd = nodes[s];
    if (!deep_leader(nd, s)) return;
    const uint32_t doubles = (uint32_t)(before[s] >> 32), singles = (uint32_t)before[s];
    uint32_t slot, opens;
    if (nd.split >= 2) {
        slot = ps.double_slot(doubles, singles);
        nodes[s + 2].pos = 2 * (slot + 1);                                   // the follower's children
        nodes[s + 1 + 2 * nodes[s + 1].split].pos = 2 * (slot + 1) + 1;
    } else {
        if (!ps.single_slot(doubles, singles, slot, opens)) return;
        if (opens != kNoOpener) opener[opens] = slot;
    }
    nodes[s + 1].pos = 2 * slot;
    nodes[s + 2 * nd.split].pos = 2 * slot + 1;
}
__global__ void k_place_joiners(BuildNode *__restrict__ nodes, uint32_t n_nodes, const uint64_t *__restrict__ before,
                                PairSlots ps, const uint32_t *__restrict__ opener) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_nodes) return;
    const BuildNode nd = nodes[s];
    if (!deep_leader(nd, s) || nd.split >= 2) return;
    uint32_t slot, opens;
    if (ps.single_slot((uint32_t)(before[s] >> 32), (uint32_t)before[s], slot, opens)) return;
    slot = opener[opens] + 1;
    nodes[s + 1].pos = 2 * slot;
    nodes[s + 2 * nd.split].pos = 2 * slot + 1;
}

// Node outputs: the threaded node and its single-precision companion at the node's threaded position (skip link =
// position of the node in the slot after the subtree, pass link = position of the first child), the reference-order
// node at the flattened index, and the coordinate-range summary the traversal kernels' fast paths depend on.
// summary[0] bit 0: some coordinate outside the reciprocal window, bit 1: outside the single-precision window;
// summary64: bits of the largest |coordinate|.
__global__ void k_emit_nodes(const BuildNode *__restrict__ nodes, const uint64_t *__restrict__ bounds, uint32_t n_nodes,
                             HotNode *__restrict__ hot, DevNode *__restrict__ thr, DevNode *__restrict__ ref,
                             uint32_t *__restrict__ summary, unsigned long long *__restrict__ summary64) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bad = 0;
    double cmax = 0.0;
    if (s < n_nodes) {
        BuildNode nd = nodes[s];
        DevNode t;
        for (int k = 0; k < 3; k++) {
            t.lo[k] = ord_back(bounds[(size_t)s * 6 + k]);
            t.hi[k] = ord_back(bounds[(size_t)s * 6 + 3 + k]);
        }
        const uint32_t k = nd.count;
        const uint32_t after = s + 2 * k - 1;
        t.link = after < n_nodes ? nodes[after].pos : n_nodes;
        t.nchild = k == 1 ? t.link : nodes[s + 1].pos;
        t.leaf_begin = nd.begin;
        t.leaf_count = k == 1 ? 1u : 0u;
        thr[nd.pos] = t;
        HotNode h;
        for (int c = 0; c < 3; c++) {
            h.planes[2 * c] = (float)t.lo[c];
            h.planes[2 * c + 1] = (float)t.hi[c];
            for (int e = 0; e < 2; e++) {
                double v = e ? t.hi[c] : t.lo[c];
                double m = fabs(v);
                if (!(v == 0.0 || (m >= 0x1p-600 && m <= 0x1p400))) bad |= 1u;
                if (!(v == 0.0 || (m >= 0x1p-40 && m <= 0x1p40))) bad |= 2u;
                if (m > cmax) cmax = m;
            }
        }
        h.skip = t.link;
        h.next = hot_next_word(t.link, t.nchild, t.leaf_begin, t.leaf_count);
        hot[nd.pos] = h;
        DevNode r = t;
        r.link = k == 1 ? nd.base : nd.base + 2 * k - 4;
        r.nchild = (k == 1 ? 0u : 2u) | (nd.axis & 0x80000000u);
        ref[nd.ref] = r;
    }
    uint64_t m = ord_of(cmax);          // cmax >= 0: the order image is monotone
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t o = shfl_down_u64(m, d);
        uint32_t ob = __shfl_down(bad, d, 64);
        if ((int)(threadIdx.x & 63) + d < 64) {
            m = o > m ? o : m;
            bad |= ob;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        if (bad) atomicOr(summary, bad);
        atomicMax(summary64, (unsigned long long)m);
    }
}

// Leaves in traversal order (Bvh::add_bvh_tree pushes a tree's leaves before its children, bvh.rs:92-95): leaf p
// is triangle ids[p].
__global__ void k_emit_leaves(const double *__restrict__ tris, const double *__restrict__ normals_in,
                              const uint32_t *__restrict__ ids, uint32_t n, DevTri *__restrict__ out,
                              double *__restrict__ normals_out, uint32_t *__restrict__ leaf_source) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t id = ids[p];
    DevTri t;
    for (int k = 0; k < 9; k++) t.p[k] = tris[(size_t)id * 9 + k];
    t.source = p;
    t.pad = 0;
    out[p] = t;
    if (normals_in)
        for (int k = 0; k < 9; k++) normals_out[(size_t)p * 9 + k] = normals_in[(size_t)id * 9 + k];
    leaf_source[p] = id;
}

__global__ void k_gather_leaf_info(const rtg_patch_leaf *__restrict__ by_facet, const uint32_t *__restrict__ leaf_source,
                                   uint32_t n, rtg_patch_leaf *__restrict__ by_leaf) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) by_leaf[p] = by_facet[leaf_source[p]];
}

struct Buffers {
    std::vector<void *> owned;
    ~Buffers() {
        for (void *p : owned)
            if (p) (void)hipFree(p);
    }
    hipError_t get(size_t bytes, void **out) {
        *out = nullptr;
        hipError_t e = hipMalloc(out, bytes ? bytes : 16);
        if (e == hipSuccess) owned.push_back(*out);
        return e;
    }
    void release(void *p) {           // ownership passes to the caller
        for (void *&q : owned)
            if (q == p) q = nullptr;
    }
};

inline uint32_t grid_for(uint64_t n) { return (uint32_t)((n + 255) / 256); }

} // namespace

#define RTG_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) {                                                                                        \
            (void)hipGetLastError();                                                                                   \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                                                   \
            return e_;                                                                                                 \
        }                                                                                                              \
    } while (0)

hipError_t launch_gather_leaf_info(hipStream_t stream, const rtg_patch_leaf *by_facet, const uint32_t *leaf_source,
                                   uint32_t n, rtg_patch_leaf *by_leaf) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_gather_leaf_info, dim3(grid_for(n)), dim3(256), 0, stream, by_facet, leaf_source, n, by_leaf);
    return hipGetLastError();
}

hipError_t build_subdivide_device(hipStream_t stream, const rtg_tri *d_tris, uint32_t n, const double *d_normals_in,
                                  BuiltMesh *out, std::string &err) {
    *out = BuiltMesh{};
    if (n < 2 || n > (1u << 30)) {
        err = "build_subdivide_device: triangle count out of range";
        return hipErrorInvalidValue;
    }
    const uint32_t n_nodes = 2 * n - 1;
    uint32_t bits = 1;
    while ((1ull << bits) < n) bits++;           // positions and ranks are below n
    const double *tris = reinterpret_cast<const double *>(d_tris);

    Buffers buf;
    double *tb, *centre, *sort_key, *sorted_key, *split_coord;
    uint32_t *iota, *perm, *heads, *run_start, *rank, *ids_a, *ids_b, *seg, *active_a, *active_b, *small;
    uint64_t *keys_a, *keys_b, *bounds;
    BuildNode *nodes;
    RTG_TRY(buf.get((size_t)n * 48, (void **)&tb));
    RTG_TRY(buf.get((size_t)n * 24, (void **)&centre));
    RTG_TRY(buf.get((size_t)n * 24, (void **)&sort_key));
    RTG_TRY(buf.get((size_t)n * 8, (void **)&sorted_key));
    RTG_TRY(buf.get((size_t)n * 4, (void **)&iota));
    RTG_TRY(buf.get((size_t)n * 4, (void **)&perm));
    RTG_TRY(buf.get((size_t)n * 4, (void **)&heads));
    RTG_TRY(buf.get((size_t)n * 4, (void **)&run_start));
    RTG_TRY(buf.get((size_t)n * 12, (void **)&rank));
    RTG_TRY(buf.get((size_t)n * 4, (void **)&ids_a));
    RTG_TRY(buf.get((size_t)n * 4, (void **)&ids_b));
    RTG_TRY(buf.get((size_t)n * 4, (void **)&seg));
    RTG_TRY(buf.get((size_t)n * 8, (void **)&keys_a));
    RTG_TRY(buf.get((size_t)n * 8, (void **)&keys_b));
    RTG_TRY(buf.get(((size_t)n / 2 + 1) * 4, (void **)&active_a));
    RTG_TRY(buf.get(((size_t)n / 2 + 1) * 4, (void **)&active_b));
    RTG_TRY(buf.get(((size_t)n / 2 + 1) * 8, (void **)&split_coord));
    RTG_TRY(buf.get((size_t)n_nodes * sizeof(BuildNode), (void **)&nodes));
    RTG_TRY(buf.get((size_t)n_nodes * 48, (void **)&bounds));
    RTG_TRY(buf.get(64, (void **)&small));       // [0] vertex flags, [1] next-active count, [2] range summary, [4..5] max |c|
    RTG_TRY(hipMemsetAsync(small, 0, 64, stream));

    size_t temp_rank = 0, temp_level = 0, temp_scan = 0, temp_place = 0;
    RTG_TRY(rocprim::radix_sort_pairs(nullptr, temp_rank, sort_key, sorted_key, iota, perm, (size_t)n, 0u, 64u, stream));
    RTG_TRY(rocprim::radix_sort_pairs(nullptr, temp_level, keys_a, keys_b, ids_a, ids_b, (size_t)n, 0u, 2 * bits, stream));
    RTG_TRY(rocprim::inclusive_scan(nullptr, temp_scan, heads, run_start, (size_t)n, rocprim::maximum<uint32_t>(), stream));
    // the placement pass reuses the level sort's key arrays (2n words each) for its marks and their scan
    RTG_TRY(rocprim::exclusive_scan(nullptr, temp_place, (uint32_t *)keys_a, (uint32_t *)keys_b, 0u, (size_t)n_nodes,
                                    rocprim::plus<uint32_t>(), stream));
    size_t temp_bytes = temp_rank > temp_level ? temp_rank : temp_level;
    if (temp_scan > temp_bytes) temp_bytes = temp_scan;
    if (temp_place > temp_bytes) temp_bytes = temp_place;
    void *temp;
    RTG_TRY(buf.get(temp_bytes, &temp));

    const uint32_t g = grid_for(n);
    hipLaunchKernelGGL(k_prepare, dim3(g), dim3(256), 0, stream, tris, n, tb, centre, sort_key, iota, small);
    RTG_TRY(hipGetLastError());
    for (int a = 0; a < 3; a++) {
        size_t tbytes = temp_bytes;
        RTG_TRY(rocprim::radix_sort_pairs(temp, tbytes, sort_key + (size_t)a * n, sorted_key, iota, perm, (size_t)n, 0u,
                                          64u, stream));
        hipLaunchKernelGGL(k_run_heads, dim3(g), dim3(256), 0, stream, sorted_key, n, heads);
        RTG_TRY(hipGetLastError());
        tbytes = temp_bytes;
        RTG_TRY(rocprim::inclusive_scan(temp, tbytes, heads, run_start, (size_t)n, rocprim::maximum<uint32_t>(), stream));
        hipLaunchKernelGGL(k_scatter_rank, dim3(g), dim3(256), 0, stream, perm, run_start, n, rank + (size_t)a * n);
        RTG_TRY(hipGetLastError());
    }
    uint32_t h_small[16];
    RTG_TRY(hipMemcpyAsync(h_small, small, 64, hipMemcpyDeviceToHost, stream));
    RTG_TRY(hipStreamSynchronize(stream));
    if (h_small[0] & 1u) {
        err = "a triangle vertex is not finite";
        return hipErrorInvalidValue;
    }

    RTG_TRY(hipMemcpyAsync(ids_a, iota, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
    RTG_TRY(hipMemsetAsync(seg, 0, (size_t)n * 4, stream));
    hipLaunchKernelGGL(k_init, dim3(1), dim3(64), 0, stream, nodes, bounds, n);
    RTG_TRY(hipGetLastError());
    const uint32_t g_assign = (uint32_t)(((uint64_t)n + 256 * kAssignPerLane - 1) / (256 * kAssignPerLane));
    hipLaunchKernelGGL(k_assign, dim3(g_assign), dim3(256), 0, stream, nodes, seg, ids_a, tb, n, 1,
                       (unsigned long long *)bounds);
    RTG_TRY(hipGetLastError());
    RTG_TRY(hipMemsetAsync(active_a, 0, 4, stream));     // the root is the only active segment
    uint32_t n_active = 1, depth = 1, n_top = 0;
    uint32_t *ids_cur = ids_a, *ids_next = ids_b;
    while (n_active) {
        const uint32_t ga = grid_for(n_active);
        hipLaunchKernelGGL(k_plan, dim3(ga), dim3(256), 0, stream, nodes, bounds, active_a, n_active, split_coord);
        RTG_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_keys, dim3(g), dim3(256), 0, stream, nodes, seg, ids_cur, rank, n, bits, keys_a);
        RTG_TRY(hipGetLastError());
        size_t tbytes = temp_bytes;
        RTG_TRY(rocprim::radix_sort_pairs(temp, tbytes, keys_a, keys_b, ids_cur, ids_next, (size_t)n, 0u, 2 * bits, stream));
        RTG_TRY(hipMemsetAsync(small + 1, 0, 4, stream));
        hipLaunchKernelGGL(k_split, dim3(ga), dim3(256), 0, stream, nodes, bounds, active_a, n_active, split_coord,
                           ids_next, centre, n, active_b, small + 1, depth - 1);
        if (depth - 1 < kThreadedTopLevels) n_top += n_active;
        RTG_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_assign, dim3(g_assign), dim3(256), 0, stream, nodes, seg, ids_next, tb, n, 0,
                           (unsigned long long *)bounds);
        RTG_TRY(hipGetLastError());
        RTG_TRY(hipMemcpyAsync(&n_active, small + 1, 4, hipMemcpyDeviceToHost, stream));
        RTG_TRY(hipStreamSynchronize(stream));
        if (n_active > n / 2) {
            err = "build_subdivide_device: active segment count out of range";
            return hipErrorUnknown;
        }
        std::swap(ids_cur, ids_next);
        std::swap(active_a, active_b);
        depth++;
    }

    // positions in the threaded arrays
    {
        uint32_t *flag = (uint32_t *)keys_a, *before = (uint32_t *)keys_b;
        uint64_t *list;
        RTG_TRY(buf.get(sizeof(uint64_t) << kThreadedTopLevels, (void **)&list));
        if (n_top == 0 || n_top >= (1u << kThreadedTopLevels)) {
            err = "build_subdivide_device: top-level node count out of range";
            return hipErrorUnknown;
        }
        const uint32_t gn = grid_for(n_nodes);
        hipLaunchKernelGGL(k_top_flag, dim3(gn), dim3(256), 0, stream, nodes, n_nodes, flag);
        RTG_TRY(hipGetLastError());
        size_t tbytes = temp_bytes;
        RTG_TRY(rocprim::exclusive_scan(temp, tbytes, flag, before, 0u, (size_t)n_nodes, rocprim::plus<uint32_t>(), stream));
        hipLaunchKernelGGL(k_top_list, dim3(gn), dim3(256), 0, stream, nodes, n_nodes, flag, before, list);
        RTG_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_place_top, dim3(grid_for(n_top)), dim3(256), 0, stream, nodes, list, n_top);
        RTG_TRY(hipGetLastError());
        // below the top block: leaders counted in pre-order, then placed line by line
        uint64_t *lflag, *lbefore;
        RTG_TRY(buf.get((size_t)n_nodes * 8, (void **)&lflag));
        RTG_TRY(buf.get((size_t)n_nodes * 8, (void **)&lbefore));
        hipLaunchKernelGGL(k_leader_flags, dim3(gn), dim3(256), 0, stream, nodes, n_nodes, lflag);
        RTG_TRY(hipGetLastError());
        size_t lbytes = 0;
        RTG_TRY(rocprim::exclusive_scan(nullptr, lbytes, lflag, lbefore, (uint64_t)0, (size_t)n_nodes, rocprim::plus<uint64_t>(), stream));
        void *ltemp = temp;
        if (lbytes > temp_bytes) RTG_TRY(buf.get(lbytes, &ltemp));
        RTG_TRY(rocprim::exclusive_scan(ltemp, lbytes, lflag, lbefore, (uint64_t)0, (size_t)n_nodes, rocprim::plus<uint64_t>(), stream));
        uint64_t last[2];
        RTG_TRY(hipMemcpyAsync(&last[0], lflag + (n_nodes - 1), 8, hipMemcpyDeviceToHost, stream));
        RTG_TRY(hipMemcpyAsync(&last[1], lbefore + (n_nodes - 1), 8, hipMemcpyDeviceToHost, stream));
        RTG_TRY(hipStreamSynchronize(stream));
        PairSlots ps;
        ps.n_top = n_top;
        ps.n_pairs = n - 1;
        ps.n_double = (uint32_t)((last[0] + last[1]) >> 32);
        ps.n_single = (uint32_t)(last[0] + last[1]);
        uint32_t *opener = active_a;                          // n / 2 + 1 words: a line of singles per two singles
        hipLaunchKernelGGL(k_place_leaders, dim3(gn), dim3(256), 0, stream, nodes, n_nodes, (const uint64_t *)lbefore, ps, opener);
        RTG_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_place_joiners, dim3(gn), dim3(256), 0, stream, nodes, n_nodes, (const uint64_t *)lbefore, ps,
                           (const uint32_t *)opener);
        RTG_TRY(hipGetLastError());
    }

    // outputs
    const size_t off_hot = sizeof(MeshSlabHeader), off_tnodes = off_hot + (size_t)n_nodes * sizeof(HotNode),
                 off_tris = off_tnodes + (size_t)n_nodes * sizeof(DevNode), slab_bytes = off_tris + (size_t)n * sizeof(DevTri);
    uint8_t *slab;
    DevNode *ref;
    uint32_t *leaf_source;
    double *normals_out = nullptr;
    RTG_TRY(buf.get(slab_bytes, (void **)&slab));
    RTG_TRY(buf.get((size_t)n_nodes * sizeof(DevNode), (void **)&ref));
    RTG_TRY(buf.get((size_t)n * 4, (void **)&leaf_source));
    if (d_normals_in) RTG_TRY(buf.get((size_t)n * 72, (void **)&normals_out));
    RTG_TRY(hipMemsetAsync(slab, 0, sizeof(MeshSlabHeader), stream));
    hipLaunchKernelGGL(k_emit_nodes, dim3(grid_for(n_nodes)), dim3(256), 0, stream, nodes, bounds, n_nodes,
                       (HotNode *)(slab + off_hot), (DevNode *)(slab + off_tnodes), ref, small + 2,
                       (unsigned long long *)(small + 4));
    RTG_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_emit_leaves, dim3(g), dim3(256), 0, stream, tris, d_normals_in, ids_cur, n,
                       (DevTri *)(slab + off_tris), normals_out, leaf_source);
    RTG_TRY(hipGetLastError());
    uint64_t root_bounds[6];
    RTG_TRY(hipMemcpyAsync(h_small, small, 64, hipMemcpyDeviceToHost, stream));
    RTG_TRY(hipMemcpyAsync(root_bounds, bounds, 48, hipMemcpyDeviceToHost, stream));
    RTG_TRY(hipStreamSynchronize(stream));

    out->d_slab = slab;
    out->d_nodes = ref;
    out->d_leaf_source = leaf_source;
    out->d_normals = normals_out;
    buf.release(slab);
    buf.release(ref);
    buf.release(leaf_source);
    buf.release(normals_out);
    out->n_nodes = n_nodes;
    out->root = n_nodes - 1;
    out->troot = n_nodes - 1;
    out->depth = depth;
    out->fast_div_ok = (h_small[2] & 1u) ? 0u : 1u;
    out->f32_ok = (h_small[2] & 2u) ? 0u : 1u;
    uint64_t cm;
    __builtin_memcpy(&cm, h_small + 4, 8);
    out->coord_max = ord_back(cm);
    for (int k = 0; k < 6; k++) out->bounds[k] = ord_back(root_bounds[k]);
    return hipSuccess;
}

} // namespace rtg
