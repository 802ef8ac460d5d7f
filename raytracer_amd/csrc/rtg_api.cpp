// rtg_api.cpp -- the C ABI of include/rtg.h over the gfx950 kernels.
//
// Host-side duties only: validate and re-lay-out the caller's flattened BVH for
// the GPU, own device memory and the stream, route host/device pointers, and
// turn every failure into an error code + message (nothing throws across the
// ABI).  There is no CPU fallback: without a usable HIP device every entry
// point that needs one fails with RTG_E_NODEVICE / RTG_E_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "rtg_internal.hpp"
#include "rtg_launch.hpp"

using namespace rtg;

namespace {
thread_local std::string g_create_error;
using DeviceBuffer = RtgDeviceBuffer;
} // namespace

namespace {

void set_error(rtg_ctx *ctx, const char *text) {
    if (ctx)
        ctx->err = text;
    else
        g_create_error = text;
}
int fail(rtg_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    set_error(ctx, buf);
    return code;
}

#define HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (void)hipGetLastError();                                                              \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RTG_E_NOMEM : RTG_E_HIP, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                                   \
        }                                                                                         \
    } while (0)

int bind_device(rtg_ctx *ctx) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return RTG_OK;
}

// True when `p` can be dereferenced by a kernel on the ctx's device.
bool is_device_pointer(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

int scratch_get(rtg_ctx *ctx, const char *name, size_t bytes, void **out);

// Tuning of one timed launch: with "time_kernels" on, an event pair for the launch's execution window.  Only a
// launch of the persistent kernel fills the pair (`will_launch`: the scene has a mesh and there is work), so only
// then is one handed out.
int tuning_for_launch(rtg_ctx *ctx, bool will_launch, PhasedTuning &t) {
    t = ctx->tune;
    if (!ctx->time_kernels || !will_launch) return RTG_OK;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    if (!ctx->idle_events.empty()) {
        ev = ctx->idle_events.back();
        ctx->idle_events.pop_back();
    } else {
        HIP_TRY(ctx, hipEventCreate(&ev.first));
        HIP_TRY(ctx, hipEventCreate(&ev.second));
    }
    ctx->timed.push_back(ev);
    t.time_start = ev.first;
    t.time_stop = ev.second;
    return RTG_OK;
}

// Scratch of one persistent launch with `n_out` output slots on `scene`.
int phased_scratch(rtg_ctx *ctx, const DevScene &ds, uint64_t n_out, PhasedScratch &ps) {
    int rc;
    void *v = nullptr;
    if ((rc = scratch_get(ctx, "work_counters", (8 * 32 + 16) * sizeof(uint32_t), &v)) != RTG_OK) return rc;
    ps.work_counters = (uint32_t *)v;
    ps.prim_best = nullptr;
    if (ds.n_mesh_objects != 0 && ds.n_prims != 0) {
        if ((rc = scratch_get(ctx, "prim_best", n_out * sizeof(rtg_hit), &v)) != RTG_OK) return rc;
        ps.prim_best = (rtg_hit *)v;
    }
    return RTG_OK;
}

int scratch_get(rtg_ctx *ctx, const char *name, size_t bytes, void **out) {
    DeviceBuffer &b = ctx->scratch[name];
    if (b.bytes < bytes) {
        if (b.ptr) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipFree(b.ptr));
            b.ptr = nullptr;
            b.bytes = 0;
        }
        size_t want = bytes + bytes / 8 + 256;
        HIP_TRY(ctx, hipMalloc(&b.ptr, want));
        b.bytes = want;
    }
    *out = b.ptr;
    return RTG_OK;
}

// An output the kernel writes: the caller's own device pointer, or a staging
// buffer that is copied back to the caller's host memory after the launch.
struct Out {
    void *user = nullptr;
    void *dev = nullptr;
    size_t bytes = 0;
    bool staged = false;
};
int out_prepare(rtg_ctx *ctx, const char *name, void *user, size_t bytes, Out &o) {
    o.user = user;
    o.bytes = bytes;
    if (!user || bytes == 0) {
        o.dev = nullptr;
        return RTG_OK;
    }
    if (is_device_pointer(user)) {
        if ((uintptr_t)user % 16 != 0) return fail(ctx, RTG_E_INVALID, "device output '%s' must be 16-byte aligned", name);
        o.dev = user;
        return RTG_OK;
    }
    o.staged = true;
    return scratch_get(ctx, name, bytes, &o.dev);
}
int out_finish(rtg_ctx *ctx, Out &o) {
    if (o.staged && o.dev) HIP_TRY(ctx, hipMemcpyAsync(o.user, o.dev, o.bytes, hipMemcpyDeviceToHost, ctx->stream));
    return RTG_OK;
}
// An input the kernel reads: device pointer as is, host pointer uploaded.
int in_prepare(rtg_ctx *ctx, const char *name, const void *user, size_t bytes, const void **dev) {
    if (!user || bytes == 0) {
        *dev = nullptr;
        return RTG_OK;
    }
    if (is_device_pointer(user)) {
        *dev = user;
        return RTG_OK;
    }
    void *d = nullptr;
    int rc = scratch_get(ctx, name, bytes, &d);
    if (rc != RTG_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(d, user, bytes, hipMemcpyHostToDevice, ctx->stream));
    *dev = d;
    return RTG_OK;
}

int upload(rtg_ctx *ctx, const void *host, size_t bytes, void **dev) {
    *dev = nullptr;
    if (bytes == 0) return RTG_OK;
    HIP_TRY(ctx, hipMalloc(dev, bytes));
    hipError_t e = hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(*dev);
        *dev = nullptr;
        return fail(ctx, RTG_E_HIP, "hipMemcpy(H2D, %zu bytes): %s", bytes, hipGetErrorString(e));
    }
    return RTG_OK;
}

// Mat3::inverse / Mat4::inverse in the reference's operation order
// (src/math/mat.rs:80-87,100-154); Transform::from (src/geo/transform.rs:51-58).
void invert3(const double *x, double *r) {
    const double a = x[0], b = x[1], c = x[2], d = x[3], e = x[4], f = x[5], g = x[6], h = x[7], i = x[8];
    const double s = 1.0 / (a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g));
    r[0] = (e * i - f * h) * s;
    r[1] = (c * h - b * i) * s;
    r[2] = (b * f - c * e) * s;
    r[3] = (f * g - d * i) * s;
    r[4] = (a * i - c * g) * s;
    r[5] = (c * d - a * f) * s;
    r[6] = (d * h - e * g) * s;
    r[7] = (b * g - a * h) * s;
    r[8] = (a * e - b * d) * s;
}
void invert4(const double *m, double *n) {
    // 2x2 minors named by (rows)(columns), as the reference names them
    const double m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3], m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
    const double m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11], m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];
    const double a2323 = m22 * m33 - m23 * m32, a1323 = m21 * m33 - m23 * m31, a1223 = m21 * m32 - m22 * m31;
    const double a0323 = m20 * m33 - m23 * m30, a0223 = m20 * m32 - m22 * m30, a0123 = m20 * m31 - m21 * m30;
    const double a2313 = m12 * m33 - m13 * m32, a1313 = m11 * m33 - m13 * m31, a1213 = m11 * m32 - m12 * m31;
    const double a2312 = m12 * m23 - m13 * m22, a1312 = m11 * m23 - m13 * m21, a1212 = m11 * m22 - m12 * m21;
    const double a0313 = m10 * m33 - m13 * m30, a0213 = m10 * m32 - m12 * m30, a0312 = m10 * m23 - m13 * m20;
    const double a0212 = m10 * m22 - m12 * m20, a0113 = m10 * m31 - m11 * m30, a0112 = m10 * m21 - m11 * m20;
    const double det = 1.0 / (m00 * (m11 * a2323 - m12 * a1323 + m13 * a1223) -
                              m01 * (m10 * a2323 - m12 * a0323 + m13 * a0223) +
                              m02 * (m10 * a1323 - m11 * a0323 + m13 * a0123) -
                              m03 * (m10 * a1223 - m11 * a0223 + m12 * a0123));
    n[0] = det * (m11 * a2323 - m12 * a1323 + m13 * a1223);
    n[1] = det * -(m01 * a2323 - m02 * a1323 + m03 * a1223);
    n[2] = det * (m01 * a2313 - m02 * a1313 + m03 * a1213);
    n[3] = det * -(m01 * a2312 - m02 * a1312 + m03 * a1212);
    n[4] = det * -(m10 * a2323 - m12 * a0323 + m13 * a0223);
    n[5] = det * (m00 * a2323 - m02 * a0323 + m03 * a0223);
    n[6] = det * -(m00 * a2313 - m02 * a0313 + m03 * a0213);
    n[7] = det * (m00 * a2312 - m02 * a0312 + m03 * a0212);
    n[8] = det * (m10 * a1323 - m11 * a0323 + m13 * a0123);
    n[9] = det * -(m00 * a1323 - m01 * a0323 + m03 * a0123);
    n[10] = det * (m00 * a1313 - m01 * a0313 + m03 * a0113);
    n[11] = det * -(m00 * a1312 - m01 * a0312 + m03 * a0112);
    n[12] = det * -(m10 * a1223 - m11 * a0223 + m12 * a0123);
    n[13] = det * (m00 * a1223 - m01 * a0223 + m02 * a0123);
    n[14] = det * -(m00 * a1213 - m01 * a0213 + m02 * a0113);
    n[15] = det * (m00 * a1212 - m01 * a0212 + m02 * a0112);
}

// Splits [0, n) over a few host threads (mesh creation touches every node and triangle of trees with tens of
// millions of records).
template <class F> void parallel_ranges(size_t n, F f) { api_parallel_ranges(n, (size_t)1 << 15, f); }

// Walks the caller's tree from the root: checks every range, rejects shared or
// cyclic nodes, and emits (a) the reference-order node array with a
// "last sibling" flag per node and (b) the threaded array with pass and skip
// links (place_threaded below decides where the nodes sit; `troot_out` is the
// root's position).  Iterative, so arbitrarily deep trees do not overflow the
// host stack.
void place_threaded(std::vector<DevNode> &thr_nodes, uint32_t &troot_out);
int layout_tree(rtg_ctx *ctx, const rtg_bvh *bvh, std::vector<DevNode> &ref_nodes, std::vector<DevNode> &thr_nodes,
                uint32_t &depth_out, uint32_t &troot_out) {
    const uint64_t nn = bvh->n_nodes, nl = bvh->n_leaves;
    ref_nodes.assign((size_t)nn, DevNode{});
    thr_nodes.clear();
    thr_nodes.reserve((size_t)nn);
    depth_out = 0;
    troot_out = 0;
    if (nn == 0) return RTG_OK;
    if (bvh->root >= nn) return fail(ctx, RTG_E_INVALID, "bvh root %u out of range (%llu nodes)", bvh->root, (unsigned long long)nn);
    std::atomic<uint64_t> bad_node(UINT64_MAX);
    parallel_ranges((size_t)nn, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; i++) {
            const rtg_bvh_node &s = bvh->nodes[i];
            if (s.child_begin > s.child_end || s.child_end > nn || s.leaf_begin > s.leaf_end || s.leaf_end > nl) {
                uint64_t seen = bad_node.load();
                while (i < seen && !bad_node.compare_exchange_weak(seen, (uint64_t)i)) {
                }
                continue;
            }
            DevNode &d = ref_nodes[i];
            for (int k = 0; k < 3; k++) {
                d.lo[k] = s.lo[k];
                d.hi[k] = s.hi[k];
            }
            d.link = s.child_begin;
            d.nchild = s.child_end - s.child_begin;
            d.leaf_begin = s.leaf_begin;
            d.leaf_count = s.leaf_end - s.leaf_begin;
        }
    });
    if (bad_node.load() != UINT64_MAX) {
        const uint64_t i = bad_node.load();
        const rtg_bvh_node &s = bvh->nodes[i];
        if (s.child_begin > s.child_end || s.child_end > nn)
            return fail(ctx, RTG_E_INVALID, "bvh node %llu: child range [%u,%u) invalid", (unsigned long long)i, s.child_begin, s.child_end);
        return fail(ctx, RTG_E_INVALID, "bvh node %llu: leaf range [%u,%u) invalid", (unsigned long long)i, s.leaf_begin, s.leaf_end);
    }
    std::vector<uint8_t> seen((size_t)nn, 0);
    struct Frame {
        uint32_t node, next_child, thr_index;
    };
    std::vector<Frame> stack;
    auto enter = [&](uint32_t node) -> bool {
        if (seen[node]) return false;
        seen[node] = 1;
        DevNode t = ref_nodes[node];
        t.link = 0;
        t.nchild = 0;
        stack.push_back(Frame{node, bvh->nodes[node].child_begin, (uint32_t)thr_nodes.size()});
        thr_nodes.push_back(t);
        return true;
    };
    ref_nodes[bvh->root].nchild |= 0x80000000u;
    enter(bvh->root);
    while (!stack.empty()) {
        if (stack.size() > depth_out) depth_out = (uint32_t)stack.size();
        Frame &f = stack.back();
        const rtg_bvh_node &s = bvh->nodes[f.node];
        if (f.next_child < s.child_end) {
            uint32_t c = f.next_child++;
            if (c + 1 == s.child_end) ref_nodes[c].nchild |= 0x80000000u;
            if (!enter(c)) return fail(ctx, RTG_E_INVALID, "bvh node %u is reachable twice (shared child or cycle)", c);
        } else {
            thr_nodes[f.thr_index].link = (uint32_t)thr_nodes.size();
            stack.pop_back();
        }
    }
    place_threaded(thr_nodes, troot_out);
    return RTG_OK;
}

// Takes the threaded nodes in pre-order (link = pre-order index of the node after the subtree) and gives every node
// its position, its skip link and its pass link (DevNode::nchild) by position.  A tree in which no node has leaves
// and children at once -- every tree the builders make -- is stored with the children of a node side by side and the
// root last, so that pairs of siblings start at even positions and share a 64-byte half line.  The pairs of the
// parents in the top kThreadedTopLevels levels come first, level by level (what every ray touches sits in one
// compact block).  Below them a binary tree is placed by 128-byte lines (PairSlots in rtg_pair_slots.hpp);
// other trees put the groups of all deeper parents in depth-first order.  A tree with a node that has leaves and
// children stays in pre-order (such a node goes on to its children only after its leaves, which the kernels read
// from the full node).  The visit order is the same in every placement.
void place_threaded(std::vector<DevNode> &thr_nodes, uint32_t &troot_out) {
    const uint32_t n = (uint32_t)thr_nodes.size();
    troot_out = 0;
    if (n == 0) return;
    bool pure = true, binary = true;
    for (uint32_t i = 0; i < n && pure; i++) {
        const bool inner = i + 1 < thr_nodes[i].link;
        pure = !(thr_nodes[i].leaf_count != 0 && inner);
        // two children: the first child's skip link names the second, whose skip link is the parent's
        if (inner && binary) {
            const uint32_t second = thr_nodes[i + 1].link;
            binary = second < thr_nodes[i].link && thr_nodes[second].link == thr_nodes[i].link;
        }
    }
    if (!pure || n == 1) {
        for (uint32_t i = 0; i < n; i++) thr_nodes[i].nchild = i + 1 < thr_nodes[i].link ? i + 1 : thr_nodes[i].link;
        return;
    }
    std::vector<uint32_t> pos(n);
    if (binary) {
        // the device builder's placement, from the same counts (rtg_build.hip): pair slots by rank
        std::vector<uint32_t> depth(n, 0), chain(n, 0);
        std::vector<uint32_t> level_count;
        for (uint32_t s = 0; s < n; s++) {
            if (!(s + 1 < thr_nodes[s].link)) continue;
            const uint32_t l = s + 1, r = thr_nodes[l].link;
            depth[l] = depth[r] = depth[s] + 1;
            chain[r] = r;
            chain[l] = depth[s] >= kThreadedTopLevels ? chain[s] : l;
            if (depth[s] < kThreadedTopLevels) {
                if (level_count.size() <= depth[s]) level_count.resize(depth[s] + 1, 0);
                level_count[depth[s]]++;
            }
        }
        std::vector<uint32_t> level_first(level_count.size() + 1, 0);
        for (size_t d = 0; d < level_count.size(); d++) level_first[d + 1] = level_first[d] + level_count[d];
        PairSlots ps;
        ps.n_top = level_first.back();
        ps.n_pairs = (n - 1) / 2;
        ps.n_double = ps.n_single = 0;
        for (uint32_t s = 0; s < n; s++) {
            if (!(s + 1 < thr_nodes[s].link) || depth[s] < kThreadedTopLevels || ((s - chain[s]) & 1u)) continue;
            if (s + 2 < thr_nodes[s + 1].link) ps.n_double++; else ps.n_single++;
        }
        std::vector<uint32_t> opener(ps.n_single / 2 + 1, 0);
        uint32_t doubles = 0, singles = 0;
        std::vector<uint32_t> level_seen(level_count.size(), 0);
        auto put_pair = [&](uint32_t parent, uint32_t slot) {
            const uint32_t l = parent + 1;
            pos[l] = 2 * slot;
            pos[thr_nodes[l].link] = 2 * slot + 1;
        };
        for (uint32_t s = 0; s < n; s++) {
            if (!(s + 1 < thr_nodes[s].link)) continue;
            if (depth[s] < kThreadedTopLevels) {
                put_pair(s, level_first[depth[s]] + level_seen[depth[s]]++);
                continue;
            }
            if ((s - chain[s]) & 1u) continue;                    // a follower: placed with its leader
            const bool is_double = s + 2 < thr_nodes[s + 1].link;
            uint32_t slot;
            if (is_double) {
                slot = ps.double_slot(doubles, singles);
                put_pair(s + 1, slot + 1);
                doubles++;
            } else {
                uint32_t opens;
                if (ps.single_slot(doubles, singles, slot, opens)) {
                    if (opens != kNoOpener) opener[opens] = slot;
                } else {
                    slot = opener[opens] + 1;
                }
                singles++;
            }
            put_pair(s, slot);
        }
    } else {
        std::vector<uint32_t> order, stack, kids, level, below;
        order.reserve(n);
        auto place_children_of = [&](uint32_t i) {
            kids.clear();
            for (uint32_t c = i + 1; c < thr_nodes[i].link; c = thr_nodes[c].link) kids.push_back(c);
            for (uint32_t c : kids) order.push_back(c);
        };
        level.push_back(0);
        for (uint32_t d = 0; d < kThreadedTopLevels && !level.empty(); d++) {      // the top levels, breadth first
            below.clear();
            for (uint32_t i : level) {
                place_children_of(i);
                for (uint32_t c : kids) below.push_back(c);
            }
            level.swap(below);
        }
        for (size_t k = level.size(); k-- > 0;) stack.push_back(level[k]);         // what hangs below them, depth first
        while (!stack.empty()) {
            const uint32_t i = stack.back();
            stack.pop_back();
            place_children_of(i);
            for (size_t k = kids.size(); k-- > 0;) stack.push_back(kids[k]);
        }
        for (uint32_t j = 0; j + 1 < n; j++) pos[order[j]] = j;
    }
    pos[0] = n - 1;
    std::vector<DevNode> moved(n);
    api_parallel_ranges(n, (size_t)1 << 15, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; i++) {
            DevNode t = thr_nodes[i];
            const uint32_t skip = t.link >= n ? n : pos[t.link];
            t.nchild = i + 1 < t.link ? pos[i + 1] : skip;
            t.link = skip;
            moved[pos[i]] = t;
        }
    });
    thr_nodes.swap(moved);
    troot_out = n - 1;
}

int check_scene_call(rtg_scene *scene) {
    if (!scene || !scene->ctx) return fail(nullptr, RTG_E_INVALID, "null scene");
    return bind_device(scene->ctx);
}

// Per-launch scene view: picks the traversal and sizes the global stack
// overflow slab for `threads` lanes when the deepest tree needs one.
int scene_for_launch(rtg_scene *scene, uint64_t threads, DevScene &out) {
    rtg_ctx *ctx = scene->ctx;
    out = scene->dev;
    out.traversal = ctx->traversal == 0 ? 0u : 1u;
    out.overflow = nullptr;
    out.overflow_levels = 0;
    out.overflow_stride = 1;
    const uint32_t on_chip = kRegLevels + kLdsLevels;
    if (out.traversal == 0 && scene->max_depth > on_chip) {
        uint64_t levels = scene->max_depth - on_chip;
        uint64_t bytes = levels * threads * sizeof(uint32_t);
        if (bytes > (8ull << 30)) {
            out.traversal = 1; // depth-independent
        } else {
            void *p = nullptr;
            int rc = scratch_get(ctx, "stack_overflow", (size_t)bytes, &p);
            if (rc != RTG_OK) return rc;
            out.overflow = (uint32_t *)p;
            out.overflow_levels = (uint32_t)levels;
            out.overflow_stride = (uint32_t)threads;
        }
    }
    return RTG_OK;
}

} // namespace

namespace rtg {
int api_fail(rtg_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    set_error(ctx, buf);
    return code;
}
int api_caught(rtg_ctx *ctx, const char *function) noexcept {
    int code = RTG_E_INTERNAL;
    char text[256];
    try {
        throw;
    } catch (const std::bad_alloc &) {
        code = RTG_E_NOMEM;
        snprintf(text, sizeof(text), "%s: out of host memory", function);
    } catch (const std::exception &e) {
        code = RTG_E_NOMEM;                 // what the standard library throws here is about resources (threads, locks)
        snprintf(text, sizeof(text), "%s: %s", function, e.what());
    } catch (...) {
        snprintf(text, sizeof(text), "%s: unexpected exception", function);
    }
    try {
        set_error(ctx, text);
    } catch (...) {
        // the message is lost, the code is not
    }
    return code;
}
bool api_is_device_pointer(const void *p) { return is_device_pointer(p); }
int api_scratch(rtg_ctx *ctx, const char *name, size_t bytes, void **out) { return scratch_get(ctx, name, bytes, out); }
int api_input(rtg_ctx *ctx, const char *name, const void *user, size_t bytes, const void **dev) {
    return in_prepare(ctx, name, user, bytes, dev);
}
int api_bind_scene(rtg_scene *scene) { return check_scene_call(scene); }
int api_trace_device(rtg_scene *scene, const double *d_rays6, uint64_t n, rtg_hit *d_hits) {
    rtg_ctx *ctx = scene->ctx;
    if (n == 0) return RTG_OK;
    if (n > 0xFFFFFFFFull) return fail(ctx, RTG_E_LIMIT, "more than 2^32-1 rays in one launch (split the call)");
    int rc;
    DevScene ds;
    if ((rc = scene_for_launch(scene, (n + kBlock - 1) / kBlock * kBlock, ds)) != RTG_OK) return rc;
    if (ctx->traversal == 2) {
        PhasedScratch ps;
        if ((rc = phased_scratch(ctx, ds, n, ps)) != RTG_OK) return rc;
        HIP_TRY(ctx, launch_trace_phased(ctx->stream, ds, d_rays6, n, ps, ctx->tune, d_hits, nullptr));
    } else {
        HIP_TRY(ctx, launch_trace(ctx->stream, ds, d_rays6, n, d_hits, nullptr, nullptr));
    }
    return RTG_OK;
}
} // namespace rtg

extern "C" {

const char *rtg_version(void) { return "rtg 0.1 gfx950"; }

int rtg_ctx_create(int hip_device, rtg_ctx **out) try {
    if (!out) return fail(nullptr, RTG_E_INVALID, "rtg_ctx_create: out is null");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        (void)hipGetLastError();
        return fail(nullptr, RTG_E_NODEVICE, "no HIP device (%s)", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    }
    if (hip_device < 0 || hip_device >= count)
        return fail(nullptr, RTG_E_INVALID, "device %d out of range (%d devices)", hip_device, count);
    rtg_ctx *ctx = new (std::nothrow) rtg_ctx();
    if (!ctx) return fail(nullptr, RTG_E_NOMEM, "out of host memory");
    ctx->device = hip_device;
    e = hipSetDevice(hip_device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        delete ctx;
        return fail(nullptr, RTG_E_HIP, "stream create on device %d: %s", hip_device, hipGetErrorString(e));
    }
    e = static_scratch_init(hip_device, &ctx->static_scratch);     // this GPU's address of the fixed scratch (rtg_render_primary_static)
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return fail(nullptr, RTG_E_HIP, "static scratch lookup on device %d: %s", hip_device, hipGetErrorString(e));
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return RTG_OK;
} catch (...) {
    return api_caught(nullptr, "rtg_ctx_create");
}

void rtg_ctx_destroy(rtg_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->scratch)
        if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    if (ctx->side) rtg_ctx_destroy(ctx->side);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    for (auto *list : {&ctx->timed, &ctx->idle_events})
        for (auto &ev : *list) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *rtg_last_error(rtg_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rtg_ctx_set_stream(rtg_ctx *ctx, void *hip_stream) try {
    if (!ctx) return fail(nullptr, RTG_E_INVALID, "null ctx");
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return RTG_OK;
} catch (...) {
    return api_caught(ctx, "rtg_ctx_set_stream");
}
void *rtg_ctx_get_stream(rtg_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int rtg_ctx_synchronize(rtg_ctx *ctx) try {
    if (!ctx) return fail(nullptr, RTG_E_INVALID, "null ctx");
    int rc = bind_device(ctx);
    if (rc != RTG_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RTG_OK;
} catch (...) {
    return api_caught(ctx, "rtg_ctx_synchronize");
}

int rtg_ctx_set_option(rtg_ctx *ctx, const char *key, int64_t value) try {
    if (!ctx || !key) return fail(ctx, RTG_E_INVALID, "null ctx or key");
    if (std::strcmp(key, "traversal") == 0) {
        if (value < 0 || value > 2)
            return fail(ctx, RTG_E_INVALID, "traversal must be 0 (short stack), 1 (threaded) or 2 (phased threaded)");
        ctx->traversal = (int)value;
        return RTG_OK;
    }
    if (std::strcmp(key, "max_blocks") == 0 && value == 0) {     // 0: as many four-wave groups as the GPU holds
        ctx->tune.max_blocks = 0;
        return RTG_OK;
    }
    if (value < 1 || value > (1 << 20)) return fail(ctx, RTG_E_INVALID, "option '%s': value out of range", key);
    if (std::strcmp(key, "tri_min") == 0) {
        ctx->tune.tri_min = (uint32_t)(value > 64 ? 64 : value);
        return RTG_OK;
    }
    if (std::strcmp(key, "patch_tri_min") == 0) {
        ctx->tune.patch_tri_min = (uint32_t)(value > 64 ? 64 : value);
        return RTG_OK;
    }
    if (std::strcmp(key, "obj_min") == 0) {
        ctx->tune.obj_min = (uint32_t)(value > 64 ? 64 : value);
        return RTG_OK;
    }
    if (std::strcmp(key, "max_blocks") == 0) {
        ctx->tune.max_blocks = (uint32_t)value;
        return RTG_OK;
    }
    if (std::strcmp(key, "pretest") == 0) {
        ctx->tune.pretest = value > 1 ? 1u : 0u;   // 2 = on, 1 = off
        return RTG_OK;
    }
    if (std::strcmp(key, "frame_bands") == 0) {
        ctx->frame_bands = value >= 2 ? 2u : 1u;
        return RTG_OK;
    }
    if (std::strcmp(key, "sort_rays") == 0) {
        ctx->sort_rays = value > 1;              // 2 = on, 1 = off
        return RTG_OK;
    }
    if (std::strcmp(key, "time_kernels") == 0) {
        ctx->time_kernels = value > 1;           // 2 = on, 1 = off
        return RTG_OK;
    }
    if (std::strcmp(key, "stats") == 0) {
        ctx->tune.stats = value > 1 ? 1u : 0u;   // 2 = on, 1 = off (0 is rejected above as out of range)
        return RTG_OK;
    }
    if (std::strcmp(key, "photon_trig") == 0) {
        ctx->photon_trig_device = value > 1;     // 2 = device, 1 = host libm
        return RTG_OK;
    }
    if (std::strcmp(key, "photon_batch") == 0) {
        if (value < 12 || value > 24) return fail(ctx, RTG_E_INVALID, "photon_batch must be 12..24 (log2 of the directions per batch)");
        ctx->photon_batch_log2 = (uint32_t)value;
        return RTG_OK;
    }
    if (std::strcmp(key, "kd_finish") == 0) {
        ctx->kd_finish_in_lds = value > 1;       // 2 = on (default), 1 = every level as device-wide passes
        return RTG_OK;
    }
    if (std::strcmp(key, "comm_loopback") == 0) {
        ctx->comm_loopback = value > 1;          // 2 = on, 1 = off
        return RTG_OK;
    }
    if (std::strcmp(key, "keep_state") == 0) {
        ctx->tune.keep_state = value > 1 ? 1u : 0u;   // 2 = on, 1 = off
        return RTG_OK;
    }
    return fail(ctx, RTG_E_INVALID, "unknown option '%s'", key);
} catch (...) {
    return api_caught(ctx, "rtg_ctx_set_option");
}

int rtg_ctx_kernel_time(rtg_ctx *ctx, double *total_ms, uint64_t *launches) try {
    if (!ctx || !total_ms || !launches) return fail(ctx, RTG_E_INVALID, "null argument");
    *total_ms = 0.0;
    *launches = 0;
    int rc = bind_device(ctx);
    if (rc != RTG_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // every pair goes back to the idle list whatever happens, so one bad pair cannot poison later calls
    hipError_t first_error = hipSuccess;
    for (auto &ev : ctx->timed) {
        float ms = 0.0f;
        hipError_t e = hipEventElapsedTime(&ms, ev.first, ev.second);
        if (e == hipSuccess) {
            *total_ms += (double)ms;
            *launches += 1;
        } else {
            (void)hipGetLastError();
            if (first_error == hipSuccess) first_error = e;
        }
    }
    ctx->idle_events.insert(ctx->idle_events.end(), ctx->timed.begin(), ctx->timed.end());
    ctx->timed.clear();
    if (first_error != hipSuccess)
        return fail(ctx, RTG_E_HIP, "rtg_ctx_kernel_time: a launch's events were never filled: %s", hipGetErrorString(first_error));
    return RTG_OK;
} catch (...) {
    return api_caught(ctx, "rtg_ctx_kernel_time");
}

int rtg_ctx_debug_stats(rtg_ctx *ctx, uint32_t out[16]) try {
    if (!ctx || !out) return fail(ctx, RTG_E_INVALID, "null argument");
    int rc = bind_device(ctx);
    if (rc != RTG_OK) return rc;
    auto it = ctx->scratch.find("work_counters");
    if (it == ctx->scratch.end() || !it->second.ptr) return fail(ctx, RTG_E_INVALID, "no phased launch has run on this context");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, (const uint32_t *)it->second.ptr + 8 * 32, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RTG_OK;
} catch (...) {
    return api_caught(ctx, "rtg_ctx_debug_stats");
}

int rtg_selftest_division(rtg_ctx *ctx, uint64_t seed, uint64_t n_pairs, uint32_t family, uint64_t out[3]) try {
    if (!ctx || !out) return fail(ctx, RTG_E_INVALID, "null argument");
    if (family > 3) return fail(ctx, RTG_E_INVALID, "family must be 0..3");
    int rc = bind_device(ctx);
    if (rc != RTG_OK) return rc;
    void *d = nullptr;
    if ((rc = scratch_get(ctx, "selftest", 3 * sizeof(uint64_t), &d)) != RTG_OK) return rc;
    HIP_TRY(ctx, hipMemsetAsync(d, 0, 3 * sizeof(uint64_t), ctx->stream));
    const uint32_t per_thread = 1024, blocks = (uint32_t)((n_pairs + 256ull * per_thread - 1) / (256ull * per_thread));
    if (blocks == 0 || blocks > (1u << 20)) return fail(ctx, RTG_E_LIMIT, "n_pairs out of range");
    HIP_TRY(ctx, launch_selftest_div(ctx->stream, seed, blocks, per_thread, family, (uint64_t *)d));
    HIP_TRY(ctx, hipMemcpyAsync(out, d, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RTG_OK;
} catch (...) {
    return api_caught(ctx, "rtg_selftest_division");
}

int rtg_transform_from_forward(const double forward[16], rtg_transform *out) try {
    if (!forward || !out) return fail(nullptr, RTG_E_INVALID, "null argument");
    std::memcpy(out->forward, forward, sizeof(out->forward));
    invert4(forward, out->reverse);
    double reverse_norm[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) reverse_norm[i * 3 + j] = forward[j * 4 + i];
    invert3(reverse_norm, out->forward_norm);
    return RTG_OK;
} catch (...) {
    return api_caught(nullptr, "rtg_transform_from_forward");
}

int rtg_mesh_create(rtg_ctx *ctx, const rtg_bvh *bvh, rtg_mesh **out) try {
    if (!ctx) return fail(nullptr, RTG_E_INVALID, "null ctx");
    if (!bvh || !out) return fail(ctx, RTG_E_INVALID, "rtg_mesh_create: null argument");
    *out = nullptr;
    if ((bvh->n_nodes && !bvh->nodes) || (bvh->n_leaves && !bvh->leaves))
        return fail(ctx, RTG_E_INVALID, "rtg_mesh_create: null array with non-zero count");
    if (bvh->n_nodes >= 0x7FFFFFFFull || bvh->n_leaves > 0x40000000ull)
        return fail(ctx, RTG_E_LIMIT, "rtg_mesh_create: more than 2^31-1 nodes or 2^30 leaves");
    int rc = bind_device(ctx);
    if (rc != RTG_OK) return rc;

    std::vector<DevNode> ref_nodes, thr_nodes;
    uint32_t depth = 0, troot = 0;
    rc = layout_tree(ctx, bvh, ref_nodes, thr_nodes, depth, troot);
    if (rc != RTG_OK) return rc;
    // Bvh::new (src/tree/bvh.rs:80-106) only ever emits nodes below the root; a tree with stray entries would make
    // the reference-order array (indexed by the caller's ids) and the reachable count disagree
    if (thr_nodes.size() != (size_t)bvh->n_nodes)
        return fail(ctx, RTG_E_INVALID, "bvh has %llu nodes but only %zu are reachable from the root",
                    (unsigned long long)bvh->n_nodes, thr_nodes.size());
    std::vector<DevTri> tris((size_t)bvh->n_leaves);
    parallel_ranges(tris.size(), [&](size_t b, size_t e) {
        for (size_t i = b; i < e; i++) {
            std::memcpy(tris[i].p, bvh->leaves[i].p, sizeof(double) * 9);
            tris[i].source = (uint32_t)i;
            tris[i].pad = 0;
        }
    });

    rtg_mesh *m = new (std::nothrow) rtg_mesh();
    if (!m) return fail(ctx, RTG_E_NOMEM, "out of host memory");
    m->device = ctx->device;
    m->n_leaves = bvh->n_leaves;
    if (bvh->n_nodes) {
        const rtg_bvh_node &r = bvh->nodes[bvh->root];
        for (int k = 0; k < 3; k++) {
            m->bounds[k] = r.lo[k];
            m->bounds[3 + k] = r.hi[k];
        }
    }
    // single-precision companion nodes for the pre-test, and the windows inside which the reciprocal-based and the
    // single-precision box tests are exact (see rtg_phased.hip)
    std::vector<HotNode> hot(thr_nodes.size());
    std::atomic<uint32_t> not_fast(0), not_f32(0);
    std::atomic<uint64_t> cmax_bits(0);          // bits of the largest |coordinate| (non-negative doubles order as integers)
    parallel_ranges(thr_nodes.size(), [&](size_t b, size_t e) {
        uint32_t nf = 0, n32 = 0;
        double local_max = 0.0;
        for (size_t i = b; i < e; i++) {
            const DevNode &n = thr_nodes[i];
            for (int k = 0; k < 3; k++) {
                hot[i].planes[2 * k] = (float)n.lo[k];
                hot[i].planes[2 * k + 1] = (float)n.hi[k];
                for (double c : {n.lo[k], n.hi[k]}) {
                    double a = std::fabs(c);
                    if (!(c == 0.0 || (a >= 0x1p-600 && a <= 0x1p400))) nf = 1;
                    if (!(c == 0.0 || (a >= 0x1p-40 && a <= 0x1p40))) n32 = 1;
                    if (a > local_max) local_max = a;
                }
            }
            hot[i].skip = n.link;
            hot[i].next = hot_next_word(n.link, n.nchild, n.leaf_begin, n.leaf_count);
        }
        if (nf) not_fast.store(1);
        if (n32) not_f32.store(1);
        uint64_t mine, seen = cmax_bits.load();
        std::memcpy(&mine, &local_max, 8);
        while (mine > seen && !cmax_bits.compare_exchange_weak(seen, mine)) {
        }
    });
    const uint32_t fast_ok = not_fast.load() ? 0u : 1u, f32_ok = not_f32.load() ? 0u : 1u;
    double cmax;
    {
        uint64_t bits = cmax_bits.load();
        std::memcpy(&cmax, &bits, 8);
    }

    // one slab: header, hot nodes, threaded nodes, triangles
    const size_t n = thr_nodes.size();
    const size_t off_hot = sizeof(MeshSlabHeader), off_tnodes = off_hot + n * sizeof(HotNode),
                 off_tris = off_tnodes + n * sizeof(DevNode), slab_bytes = off_tris + tris.size() * sizeof(DevTri);
    MeshSlabHeader header{};
    rc = upload(ctx, ref_nodes.data(), ref_nodes.size() * sizeof(DevNode), &m->d_nodes);
    if (rc == RTG_OK) {
        hipError_t e = hipMalloc(&m->d_slab, slab_bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            m->d_slab = nullptr;
            rc = fail(ctx, e == hipErrorOutOfMemory ? RTG_E_NOMEM : RTG_E_HIP, "hipMalloc(%zu): %s", slab_bytes,
                      hipGetErrorString(e));
        }
    }
    auto put = [&](size_t off, const void *src, size_t bytes) {
        if (rc != RTG_OK || bytes == 0) return;
        hipError_t e = hipMemcpy((uint8_t *)m->d_slab + off, src, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            rc = fail(ctx, RTG_E_HIP, "hipMemcpy to the mesh slab: %s", hipGetErrorString(e));
        }
    };
    put(0, &header, sizeof(header));
    put(off_hot, hot.data(), n * sizeof(HotNode));
    put(off_tnodes, thr_nodes.data(), n * sizeof(DevNode));
    put(off_tris, tris.data(), tris.size() * sizeof(DevTri));
    if (rc == RTG_OK && bvh->vertex_normals)
        rc = upload(ctx, bvh->vertex_normals, (size_t)bvh->n_leaves * 9 * sizeof(double), &m->d_normals);
    if (rc != RTG_OK) {
        rtg_mesh_destroy(m);
        return rc;
    }
    m->dev.nodes = (const DevNode *)m->d_nodes;
    m->dev.slab = (const uint8_t *)m->d_slab;
    m->dev.tnodes = (const DevNode *)(m->dev.slab + off_tnodes);
    m->dev.tris = (const DevTri *)(m->dev.slab + off_tris);
    m->dev.normals = (const double *)m->d_normals;
    m->dev.n_nodes = (uint32_t)n;
    m->dev.root = bvh->root;
    m->dev.troot = troot;
    m->dev.depth = depth;
    m->dev.fast_div_ok = fast_ok;
    m->dev.f32_ok = f32_ok && std::isfinite(cmax) ? 1u : 0u;
    // rounded up so the kernel's error bound never under-estimates
    m->dev.coord_max = std::nextafter((float)cmax, std::numeric_limits<float>::infinity());
    *out = m;
    return RTG_OK;
} catch (...) {
    return api_caught(ctx, "rtg_mesh_create");
}

void rtg_mesh_destroy(rtg_mesh *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->d_nodes) (void)hipFree(m->d_nodes);
    if (m->d_slab) (void)hipFree(m->d_slab);
    if (m->d_normals) (void)hipFree(m->d_normals);
    if (m->d_patches) (void)hipFree(m->d_patches);
    if (m->d_leaf_info) (void)hipFree(m->d_leaf_info);
    if (m->d_leaf_source) (void)hipFree(m->d_leaf_source);
    delete m;
}

int rtg_mesh_create_patches(rtg_ctx *ctx, const rtg_bvh *bvh, const rtg_patch_set *ps, rtg_mesh **out) try {
    if (!ctx) return fail(nullptr, RTG_E_INVALID, "null ctx");
    if (!bvh || !ps || !out) return fail(ctx, RTG_E_INVALID, "rtg_mesh_create_patches: null argument");
    *out = nullptr;
    if (ps->n_patches == 0 || !ps->patches) return fail(ctx, RTG_E_INVALID, "rtg_mesh_create_patches: no patches");
    if (bvh->n_leaves && !ps->leaf_info) return fail(ctx, RTG_E_INVALID, "rtg_mesh_create_patches: leaf_info is null");
    if (ps->n_patches >= 0x7FFFFFFFull) return fail(ctx, RTG_E_LIMIT, "rtg_mesh_create_patches: too many patches");
    if (ps->newton_steps > 64) return fail(ctx, RTG_E_LIMIT, "rtg_mesh_create_patches: newton_steps exceeds 64");
    for (uint64_t i = 0; i < bvh->n_leaves; i++)
        if (ps->leaf_info[i].patch >= ps->n_patches)
            return fail(ctx, RTG_E_INVALID, "rtg_mesh_create_patches: leaf %llu names patch %u of %llu",
                        (unsigned long long)i, ps->leaf_info[i].patch, (unsigned long long)ps->n_patches);
    rtg_bvh plain = *bvh;
    plain.vertex_normals = nullptr;
    rtg_mesh *m = nullptr;
    int rc = rtg_mesh_create(ctx, &plain, &m);
    if (rc != RTG_OK) return rc;
    rc = upload(ctx, ps->patches, (size_t)ps->n_patches * sizeof(rtg_trispline), &m->d_patches);
    if (rc == RTG_OK) rc = upload(ctx, ps->leaf_info, (size_t)bvh->n_leaves * sizeof(rtg_patch_leaf), &m->d_leaf_info);
    if (rc != RTG_OK) {
        rtg_mesh_destroy(m);
        return rc;
    }
    MeshSlabHeader header{};
    header.patches = (const rtg_trispline *)m->d_patches;
    header.leaf_info = (const rtg_patch_leaf *)m->d_leaf_info;
    header.newton_steps = ps->newton_steps;
    hipError_t he = hipMemcpy(m->d_slab, &header, sizeof(header), hipMemcpyHostToDevice);
    if (he != hipSuccess) {
        (void)hipGetLastError();
        rtg_mesh_destroy(m);
        return fail(ctx, RTG_E_HIP, "hipMemcpy of the patch header: %s", hipGetErrorString(he));
    }
    m->is_patches = true;
    m->dev.patches = (const rtg_trispline *)m->d_patches;
    m->dev.leaf_info = (const rtg_patch_leaf *)m->d_leaf_info;
    m->dev.newton_steps = ps->newton_steps;
    *out = m;
    return RTG_OK;
} catch (...) {
    return api_caught(ctx, "rtg_mesh_create_patches");
}

int rtg_mesh_build(rtg_ctx *ctx, const rtg_tri *triangles, uint64_t n_triangles, const double *vertex_normals,
                   rtg_mesh **out) try {
    if (!ctx) return fail(nullptr, RTG_E_INVALID, "null ctx");
    if (!out || (n_triangles && !triangles)) return fail(ctx, RTG_E_INVALID, "rtg_mesh_build: null argument");
    *out = nullptr;
    if (n_triangles > (1ull << 30)) return fail(ctx, RTG_E_LIMIT, "rtg_mesh_build: more than 2^30 triangles");
    int rc = bind_device(ctx);
    if (rc != RTG_OK) return rc;
    const bool tris_on_device = is_device_pointer(triangles), normals_on_device = is_device_pointer(vertex_normals);
    if (n_triangles < 2) {
        // BvhTree::new() for no triangles, the triangle's own single-leaf tree for one (bvh_subdivide.rs:10-13):
        // nothing to split, so the flattened tree is written here
        rtg_tri tri{};
        double normals[9];
        rtg_bvh_node node{};
        for (int k = 0; k < 3; k++) {
            node.lo[k] = std::numeric_limits<double>::infinity();
            node.hi[k] = -std::numeric_limits<double>::infinity();
        }
        if (n_triangles == 1) {
            HIP_TRY(ctx, hipMemcpy(&tri, triangles, sizeof(tri), tris_on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
            if (vertex_normals)
                HIP_TRY(ctx, hipMemcpy(normals, vertex_normals, sizeof(normals),
                                       normals_on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
            for (int v = 0; v < 3; v++)
                for (int k = 0; k < 3; k++) {
                    double c = tri.p[v][k];
                    if (!std::isfinite(c)) return fail(ctx, RTG_E_INVALID, "rtg_mesh_build: a triangle vertex is not finite");
                    if (c < node.lo[k] || (c == node.lo[k] && std::signbit(c))) node.lo[k] = c;
                    if (c > node.hi[k] || (c == node.hi[k] && !std::signbit(c))) node.hi[k] = c;
                }
            node.leaf_end = 1;
        }
        rtg_bvh bvh{};
        bvh.nodes = &node;
        bvh.n_nodes = 1;
        bvh.leaves = &tri;
        bvh.n_leaves = n_triangles;
        bvh.vertex_normals = (n_triangles && vertex_normals) ? normals : nullptr;
        bvh.root = 0;
        return rtg_mesh_create(ctx, &bvh, out);
    }

    void *tmp_tris = nullptr, *tmp_normals = nullptr;
    const rtg_tri *d_tris = triangles;
    const double *d_normals = vertex_normals;
    if (!tris_on_device) {
        rc = upload(ctx, triangles, (size_t)n_triangles * sizeof(rtg_tri), &tmp_tris);
        d_tris = (const rtg_tri *)tmp_tris;
    }
    if (rc == RTG_OK && vertex_normals && !normals_on_device) {
        rc = upload(ctx, vertex_normals, (size_t)n_triangles * 72, &tmp_normals);
        d_normals = (const double *)tmp_normals;
    }
    BuiltMesh built;
    if (rc == RTG_OK) {
        std::string err;
        hipError_t e = build_subdivide_device(ctx->stream, d_tris, (uint32_t)n_triangles, d_normals, &built, err);
        if (e != hipSuccess)
            rc = fail(ctx, e == hipErrorInvalidValue ? RTG_E_INVALID : (e == hipErrorOutOfMemory ? RTG_E_NOMEM : RTG_E_HIP),
                      "rtg_mesh_build: %s", err.c_str());
    }
    if (tmp_tris) (void)hipFree(tmp_tris);
    if (tmp_normals) (void)hipFree(tmp_normals);
    if (rc != RTG_OK) return rc;

    rtg_mesh *m = new (std::nothrow) rtg_mesh();
    if (!m) {
        (void)hipFree(built.d_slab);
        (void)hipFree(built.d_nodes);
        (void)hipFree(built.d_leaf_source);
        if (built.d_normals) (void)hipFree(built.d_normals);
        return fail(ctx, RTG_E_NOMEM, "out of host memory");
    }
    m->device = ctx->device;
    m->n_leaves = n_triangles;
    m->d_slab = built.d_slab;
    m->d_nodes = built.d_nodes;
    m->d_normals = built.d_normals;
    m->d_leaf_source = built.d_leaf_source;
    std::memcpy(m->bounds, built.bounds, sizeof(m->bounds));
    const size_t n = built.n_nodes;
    m->dev.nodes = (const DevNode *)m->d_nodes;
    m->dev.slab = (const uint8_t *)m->d_slab;
    m->dev.tnodes = (const DevNode *)(m->dev.slab + sizeof(MeshSlabHeader) + n * sizeof(HotNode));
    m->dev.tris = (const DevTri *)(m->dev.slab + sizeof(MeshSlabHeader) + n * (sizeof(HotNode) + sizeof(DevNode)));
    m->dev.normals = (const double *)m->d_normals;
    m->dev.n_nodes = built.n_nodes;
    m->dev.root = built.root;
    m->dev.troot = built.troot;
    m->dev.depth = built.depth;
    m->dev.fast_div_ok = built.fast_div_ok;
    m->dev.f32_ok = built.f32_ok && std::isfinite(built.coord_max) ? 1u : 0u;
    m->dev.coord_max = std::nextafter((float)built.coord_max, std::numeric_limits<float>::infinity());
    *out = m;
    return RTG_OK;
} catch (...) {
    return api_caught(ctx, "rtg_mesh_build");
}

int rtg_mesh_build_patches(rtg_ctx *ctx, const rtg_tri *facets, uint64_t n_facets, const rtg_patch_set *ps,
                           rtg_mesh **out) try {
    if (!ctx) return fail(nullptr, RTG_E_INVALID, "null ctx");
    if (!facets || !ps || !out) return fail(ctx, RTG_E_INVALID, "rtg_mesh_build_patches: null argument");
    *out = nullptr;
    if (ps->n_patches == 0 || !ps->patches) return fail(ctx, RTG_E_INVALID, "rtg_mesh_build_patches: no patches");
    if (n_facets && !ps->leaf_info) return fail(ctx, RTG_E_INVALID, "rtg_mesh_build_patches: leaf_info is null");
    if (ps->n_patches >= 0x7FFFFFFFull) return fail(ctx, RTG_E_LIMIT, "rtg_mesh_build_patches: too many patches");
    if (ps->newton_steps > 64) return fail(ctx, RTG_E_LIMIT, "rtg_mesh_build_patches: newton_steps exceeds 64");
    for (uint64_t i = 0; i < n_facets; i++)
        if (ps->leaf_info[i].patch >= ps->n_patches)
            return fail(ctx, RTG_E_INVALID, "rtg_mesh_build_patches: facet %llu names patch %u of %llu",
                        (unsigned long long)i, ps->leaf_info[i].patch, (unsigned long long)ps->n_patches);
    rtg_mesh *m = nullptr;
    int rc = rtg_mesh_build(ctx, facets, n_facets, nullptr, &m);
    if (rc != RTG_OK) return rc;
    void *by_facet = nullptr;
    rc = upload(ctx, ps->patches, (size_t)ps->n_patches * sizeof(rtg_trispline), &m->d_patches);
    if (rc == RTG_OK) rc = upload(ctx, ps->leaf_info, (size_t)n_facets * sizeof(rtg_patch_leaf), &by_facet);
    if (rc == RTG_OK && m->d_leaf_source) {              // built on the device: records go from facet to leaf order
        hipError_t e = hipMalloc(&m->d_leaf_info, (size_t)n_facets * sizeof(rtg_patch_leaf));
        if (e == hipSuccess)
            e = launch_gather_leaf_info(ctx->stream, (const rtg_patch_leaf *)by_facet, (const uint32_t *)m->d_leaf_source,
                                        (uint32_t)n_facets, (rtg_patch_leaf *)m->d_leaf_info);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            rc = fail(ctx, RTG_E_HIP, "rtg_mesh_build_patches: %s", hipGetErrorString(e));
        }
        (void)hipFree(by_facet);
        by_facet = nullptr;
    } else if (rc == RTG_OK) {
        m->d_leaf_info = by_facet;                        // fewer than two facets: leaf order is facet order
        by_facet = nullptr;
    }
    if (by_facet) (void)hipFree(by_facet);
    MeshSlabHeader header{};
    header.patches = (const rtg_trispline *)m->d_patches;
    header.leaf_info = (const rtg_patch_leaf *)m->d_leaf_info;
    header.newton_steps = ps->newton_steps;
    if (rc == RTG_OK) {
        hipError_t he = hipMemcpy(m->d_slab, &header, sizeof(header), hipMemcpyHostToDevice);
        if (he != hipSuccess) {
            (void)hipGetLastError();
            rc = fail(ctx, RTG_E_HIP, "hipMemcpy of the patch header: %s", hipGetErrorString(he));
        }
    }
    if (rc != RTG_OK) {
        rtg_mesh_destroy(m);
        return rc;
    }
    m->is_patches = true;
    m->dev.patches = header.patches;
    m->dev.leaf_info = header.leaf_info;
    m->dev.newton_steps = ps->newton_steps;
    *out = m;
    return RTG_OK;
} catch (...) {
    return api_caught(ctx, "rtg_mesh_build_patches");
}

int rtg_mesh_sizes(const rtg_mesh *mesh, uint64_t *n_nodes, uint64_t *n_leaves, uint32_t *root, uint32_t *depth) try {
    if (!mesh) return fail(nullptr, RTG_E_INVALID, "null mesh");
    if (n_nodes) *n_nodes = mesh->dev.n_nodes;
    if (n_leaves) *n_leaves = mesh->n_leaves;
    if (root) *root = mesh->dev.root;
    if (depth) *depth = mesh->dev.depth;
    return RTG_OK;
} catch (...) {
    return api_caught(nullptr, "rtg_mesh_sizes");
}

int rtg_mesh_read_bvh(const rtg_mesh *mesh, rtg_bvh_node *nodes, rtg_tri *leaves, uint32_t *leaf_source) try {
    if (!mesh) return fail(nullptr, RTG_E_INVALID, "null mesh");
    // no context is involved: the mesh is immutable after creation and the copies below are blocking (errors are
    // reported through rtg_last_error(NULL))
    rtg_ctx *ctx = nullptr;
    HIP_TRY(ctx, hipSetDevice(mesh->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    const size_t nn = mesh->dev.n_nodes, nl = (size_t)mesh->n_leaves;
    const size_t chunk = 1u << 16;
    if (nodes && nn) {
        std::vector<DevNode> tmp(std::min(chunk, nn));
        for (size_t at = 0; at < nn; at += chunk) {
            size_t k = std::min(chunk, nn - at);
            HIP_TRY(ctx, hipMemcpy(tmp.data(), mesh->dev.nodes + at, k * sizeof(DevNode), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < k; i++) {
                rtg_bvh_node &d = nodes[at + i];
                std::memcpy(d.lo, tmp[i].lo, 24);
                std::memcpy(d.hi, tmp[i].hi, 24);
                d.child_begin = tmp[i].link;
                d.child_end = tmp[i].link + (tmp[i].nchild & 0x7FFFFFFFu);
                d.leaf_begin = tmp[i].leaf_begin;
                d.leaf_end = tmp[i].leaf_begin + tmp[i].leaf_count;
            }
        }
    }
    if (leaves && nl) {
        std::vector<DevTri> tmp(std::min(chunk, nl));
        for (size_t at = 0; at < nl; at += chunk) {
            size_t k = std::min(chunk, nl - at);
            HIP_TRY(ctx, hipMemcpy(tmp.data(), mesh->dev.tris + at, k * sizeof(DevTri), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < k; i++) std::memcpy(leaves[at + i].p, tmp[i].p, 72);
        }
    }
    if (leaf_source && nl) {
        if (mesh->d_leaf_source)
            HIP_TRY(ctx, hipMemcpy(leaf_source, mesh->d_leaf_source, nl * 4, hipMemcpyDeviceToHost));
        else
            for (size_t i = 0; i < nl; i++) leaf_source[i] = (uint32_t)i;
    }
    return RTG_OK;
} catch (...) {
    return api_caught(nullptr, "rtg_mesh_read_bvh");
}

int rtg_mesh_bounds(const rtg_mesh *mesh, double out[6]) try {
    if (!mesh || !out) return fail(nullptr, RTG_E_INVALID, "null argument");
    std::memcpy(out, mesh->bounds, sizeof(mesh->bounds));
    return RTG_OK;
} catch (...) {
    return api_caught(nullptr, "rtg_mesh_bounds");
}

int rtg_scene_create(rtg_ctx *ctx, const rtg_scene_desc *desc, rtg_scene **out) try {
    if (!ctx) return fail(nullptr, RTG_E_INVALID, "null ctx");
    if (!desc || !out) return fail(ctx, RTG_E_INVALID, "rtg_scene_create: null argument");
    *out = nullptr;
    if (desc->n_objects && !desc->objects) return fail(ctx, RTG_E_INVALID, "rtg_scene_create: null objects");
    if (desc->n_objects >= 0x7FFFFFFFu) return fail(ctx, RTG_E_LIMIT, "too many objects");
    int rc = bind_device(ctx);
    if (rc != RTG_OK) return rc;
    std::vector<DevObject> objs(desc->n_objects);
    uint32_t max_depth = 0;
    for (uint32_t i = 0; i < desc->n_objects; i++) {
        const rtg_object &o = desc->objects[i];
        DevObject &d = objs[i];
        std::memset(&d, 0, sizeof(d));
        d.kind = o.kind;
        d.index = (int32_t)i;
        std::memcpy(d.reverse, o.transform.reverse, sizeof(d.reverse));
        std::memcpy(d.forward_norm, o.transform.forward_norm, sizeof(d.forward_norm));
        auto mat = [](const rtg_material &m) {
            DevMaterial x;
            std::memcpy(x.diffuse, m.diffuse, sizeof(x.diffuse));
            x.n1 = m.n1;
            x.n2 = m.n2;
            x.has_dielectric = m.has_dielectric;
            x.pad = 0;
            return x;
        };
        d.material = mat(o.material);
        d.material2 = mat(o.material2);
        std::memcpy(d.sphere_center, o.sphere_center, sizeof(d.sphere_center));
        d.sphere_radius = o.sphere_radius;
        std::memcpy(d.plane_position, o.plane_position, sizeof(d.plane_position));
        std::memcpy(d.plane_tan1, o.plane_tan1, sizeof(d.plane_tan1));
        std::memcpy(d.plane_tan2, o.plane_tan2, sizeof(d.plane_tan2));
        switch (o.kind) {
        case RTG_SPHERE:
        case RTG_PLANE:
            break;
        case RTG_MESH:
        case RTG_PATCHES:
            if (!o.mesh) return fail(ctx, RTG_E_INVALID, "object %u: mesh handle is null", i);
            // a mesh is immutable once created, so contexts of one GPU may share it (several frames in flight)
            if (o.mesh->device != ctx->device)
                return fail(ctx, RTG_E_INVALID, "object %u: mesh belongs to a context on another GPU", i);
            if ((o.kind == RTG_PATCHES) != o.mesh->is_patches)
                return fail(ctx, RTG_E_INVALID, "object %u: kind %d does not match the mesh handle (%s)", i, o.kind,
                            o.mesh->is_patches ? "patch set" : "triangle mesh");
            d.mesh = o.mesh->dev;
            if (o.mesh->dev.depth > max_depth) max_depth = o.mesh->dev.depth;
            break;
        default:
            return fail(ctx, RTG_E_INVALID, "object %u: unknown kind %d", i, o.kind);
        }
    }
    rtg_scene *s = new (std::nothrow) rtg_scene();
    if (!s) return fail(ctx, RTG_E_NOMEM, "out of host memory");
    s->ctx = ctx;
    s->device = ctx->device;
    s->max_depth = max_depth;
    s->object_leaves.assign(desc->n_objects, 0);
    for (uint32_t i = 0; i < desc->n_objects; i++)
        if (desc->objects[i].kind == RTG_MESH || desc->objects[i].kind == RTG_PATCHES)
            s->object_leaves[i] = desc->objects[i].mesh->n_leaves;
    rc = upload(ctx, objs.data(), objs.size() * sizeof(DevObject), &s->d_objects);
    if (rc != RTG_OK) {
        delete s;
        return rc;
    }
    s->dev.objects = (const DevObject *)s->d_objects;
    s->dev.n_objects = desc->n_objects;
    s->dev.traversal = 0;
    s->dev.overflow = nullptr;
    s->dev.overflow_levels = 0;
    s->dev.overflow_stride = 1;
    for (uint32_t i = 0; i < desc->n_objects; i++)
        if (desc->objects[i].material.has_dielectric || desc->objects[i].material2.has_dielectric) s->any_dielectric = true;
    s->dev.has_patches = 0;
    s->dev.any_f32_ok = 0;
    s->dev.wide_mesh = 0;
    {
        std::vector<DevObject> meshes;
        uint32_t n_prims = 0;
        for (const DevObject &d : objs) {
            if ((d.kind == RTG_MESH || d.kind == RTG_PATCHES) && d.mesh.n_nodes != 0) meshes.push_back(d);
            if (d.kind == RTG_SPHERE || d.kind == RTG_PLANE) n_prims++;
        }
        rc = upload(ctx, meshes.data(), meshes.size() * sizeof(DevObject), &s->d_mesh_objects);
        if (rc != RTG_OK) {
            rtg_scene_destroy(s);
            return rc;
        }
        s->dev.mesh_objects = (const DevObject *)s->d_mesh_objects;
        s->dev.n_mesh_objects = (uint32_t)meshes.size();
        s->dev.n_prims = n_prims;
    }
    for (uint32_t i = 0; i < desc->n_objects; i++) {
        const rtg_object &o = desc->objects[i];
        if (o.kind == RTG_PATCHES) s->dev.has_patches = 1;
        if ((o.kind == RTG_MESH || o.kind == RTG_PATCHES) && o.mesh->dev.f32_ok) s->dev.any_f32_ok = 1;
        if ((o.kind == RTG_MESH || o.kind == RTG_PATCHES) && o.mesh->dev.n_nodes != 0) s->has_mesh = true;
        if ((o.kind == RTG_MESH || o.kind == RTG_PATCHES) && o.mesh->dev.n_nodes >= (1u << 27)) s->dev.wide_mesh = 1;
    }
    *out = s;
    return RTG_OK;
} catch (...) {
    return api_caught(ctx, "rtg_scene_create");
}

void rtg_scene_destroy(rtg_scene *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->d_objects) (void)hipFree(s->d_objects);
    if (s->d_mesh_objects) (void)hipFree(s->d_mesh_objects);
    delete s;
}

int rtg_render_primary(rtg_scene *scene, const rtg_view *view, uint32_t width, uint32_t height, rtg_rect tile,
                       rtg_hit *out_hits, rtg_point *out_points, rtg_ray_counters *out_counters) try {
    int rc = check_scene_call(scene);
    if (rc != RTG_OK) return rc;
    rtg_ctx *ctx = scene->ctx;
    if (!view) return fail(ctx, RTG_E_INVALID, "rtg_render_primary: view is null");
    if (width == 0 || height == 0) return fail(ctx, RTG_E_INVALID, "rtg_render_primary: empty image");
    if ((uint64_t)tile.x0 + tile.w > width || (uint64_t)tile.y0 + tile.h > height)
        return fail(ctx, RTG_E_INVALID, "rtg_render_primary: tile [%u+%u, %u+%u] outside %ux%u", tile.x0, tile.w, tile.y0, tile.h, width, height);
    uint64_t n = (uint64_t)tile.w * tile.h;
    if (n == 0) return RTG_OK;
    if (!out_hits && !out_points && !out_counters) return fail(ctx, RTG_E_INVALID, "rtg_render_primary: no output requested");
    uint64_t tiles = (uint64_t)((tile.w + 15) / 16) * ((tile.h + 15) / 16);
    if (tiles > 0x7FFFFFFFull) return fail(ctx, RTG_E_LIMIT, "rtg_render_primary: tile too large");
    Out oh, op, oc;
    if ((rc = out_prepare(ctx, "out_hits", out_hits, n * sizeof(rtg_hit), oh)) != RTG_OK) return rc;
    if ((rc = out_prepare(ctx, "out_points", out_points, n * sizeof(rtg_point), op)) != RTG_OK) return rc;
    if ((rc = out_prepare(ctx, "out_counters", out_counters, n * sizeof(rtg_ray_counters), oc)) != RTG_OK) return rc;
    DevScene ds;
    if ((rc = scene_for_launch(scene, tiles * kBlock, ds)) != RTG_OK) return rc;
    if (ctx->traversal == 2) {
        void *hits = oh.dev;
        PhasedScratch ps;
        if ((rc = phased_scratch(ctx, ds, n, ps)) != RTG_OK) return rc;
        if (op.dev && !hits && (rc = scratch_get(ctx, "points_hits", n * sizeof(rtg_hit), &hits)) != RTG_OK) return rc;
        PhasedTuning tune;
        if ((rc = tuning_for_launch(ctx, ds.n_mesh_objects != 0, tune)) != RTG_OK) return rc;
        HIP_TRY(ctx, launch_render_primary_phased(ctx->stream, ds, *view, width, height, tile, ps, tune,
                                                  (rtg_hit *)hits, (rtg_ray_counters *)oc.dev));
        if (op.dev)
            HIP_TRY(ctx, launch_points_primary(ctx->stream, ds, *view, width, height, tile, (const rtg_hit *)hits,
                                               (rtg_point *)op.dev));
    } else {
        HIP_TRY(ctx, launch_render_primary(ctx->stream, ds, *view, width, height, tile, (rtg_hit *)oh.dev,
                                           (rtg_point *)op.dev, (rtg_ray_counters *)oc.dev));
    }
    if ((rc = out_finish(ctx, oh)) != RTG_OK) return rc;
    if ((rc = out_finish(ctx, op)) != RTG_OK) return rc;
    if ((rc = out_finish(ctx, oc)) != RTG_OK) return rc;
    if (oh.staged || op.staged || oc.staged) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RTG_OK;
} catch (...) {
    return api_caught(scene ? scene->ctx : nullptr, "rtg_render_primary");
}

int rtg_trace(rtg_scene *scene, const double *rays6, uint64_t n_rays, rtg_hit *out_hits, rtg_point *out_points,
              rtg_ray_counters *out_counters) try {
    int rc = check_scene_call(scene);
    if (rc != RTG_OK) return rc;
    rtg_ctx *ctx = scene->ctx;
    if (n_rays == 0) return RTG_OK;
    if (!rays6) return fail(ctx, RTG_E_INVALID, "rtg_trace: rays6 is null");
    if (!out_hits && !out_points && !out_counters) return fail(ctx, RTG_E_INVALID, "rtg_trace: no output requested");
    if (n_rays > 0xFFFFFFFFull) return fail(ctx, RTG_E_LIMIT, "rtg_trace: more than 2^32-1 rays in one call (split the batch)");
    if ((uintptr_t)rays6 % 16 != 0 && is_device_pointer(rays6))
        return fail(ctx, RTG_E_INVALID, "rtg_trace: device ray buffer must be 16-byte aligned");
    const void *d_rays = nullptr;
    if ((rc = in_prepare(ctx, "in_rays", rays6, n_rays * 6 * sizeof(double), &d_rays)) != RTG_OK) return rc;
    Out oh, op, oc;
    if ((rc = out_prepare(ctx, "out_hits", out_hits, n_rays * sizeof(rtg_hit), oh)) != RTG_OK) return rc;
    if ((rc = out_prepare(ctx, "out_points", out_points, n_rays * sizeof(rtg_point), op)) != RTG_OK) return rc;
    if ((rc = out_prepare(ctx, "out_counters", out_counters, n_rays * sizeof(rtg_ray_counters), oc)) != RTG_OK) return rc;
    DevScene ds;
    uint64_t threads = (n_rays + kBlock - 1) / kBlock * kBlock;
    if ((rc = scene_for_launch(scene, threads, ds)) != RTG_OK) return rc;
    if (ctx->traversal == 2) {
        void *hits = oh.dev;
        PhasedScratch ps;
        if ((rc = phased_scratch(ctx, ds, n_rays, ps)) != RTG_OK) return rc;
        if (op.dev && !hits && (rc = scratch_get(ctx, "points_hits", n_rays * sizeof(rtg_hit), &hits)) != RTG_OK) return rc;
        PhasedTuning tune;
        if ((rc = tuning_for_launch(ctx, ds.n_mesh_objects != 0, tune)) != RTG_OK) return rc;
        if (ctx->sort_rays && ds.n_mesh_objects != 0 && n_rays >= 4096) {
            // trace the batch in coherence order, hand the answers back in the caller's order
            RaySortScratch w;
            void *v = nullptr, *sorted = nullptr, *tmp_hits = nullptr, *tmp_ctr = nullptr;
            if ((rc = scratch_get(ctx, "sort_box", 6 * sizeof(unsigned long long), &v)) != RTG_OK) return rc;
            w.box = (unsigned long long *)v;
            if ((rc = scratch_get(ctx, "sort_keys", n_rays * sizeof(uint64_t), &v)) != RTG_OK) return rc;
            w.keys = (uint64_t *)v;
            if ((rc = scratch_get(ctx, "sort_keys_sorted", n_rays * sizeof(uint64_t), &v)) != RTG_OK) return rc;
            w.keys_sorted = (uint64_t *)v;
            if ((rc = scratch_get(ctx, "sort_index", n_rays * sizeof(uint32_t), &v)) != RTG_OK) return rc;
            w.index = (uint32_t *)v;
            if ((rc = scratch_get(ctx, "sort_perm", n_rays * sizeof(uint32_t), &v)) != RTG_OK) return rc;
            w.perm = (uint32_t *)v;
            w.temp_bytes = ray_sort_temp_bytes(n_rays);
            if ((rc = scratch_get(ctx, "sort_temp", w.temp_bytes, &w.temp)) != RTG_OK) return rc;
            if ((rc = scratch_get(ctx, "sort_rays", n_rays * 6 * sizeof(double), &sorted)) != RTG_OK) return rc;
            if ((rc = scratch_get(ctx, "sort_hits", n_rays * sizeof(rtg_hit), &tmp_hits)) != RTG_OK) return rc;
            if (oc.dev && (rc = scratch_get(ctx, "sort_counters", n_rays * sizeof(rtg_ray_counters), &tmp_ctr)) != RTG_OK) return rc;
            HIP_TRY(ctx, launch_ray_sort(ctx->stream, (const double *)d_rays, n_rays, w, (double *)sorted));
            HIP_TRY(ctx, launch_trace_phased(ctx->stream, ds, (const double *)sorted, n_rays, ps, tune, (rtg_hit *)tmp_hits,
                                             (rtg_ray_counters *)tmp_ctr));
            HIP_TRY(ctx, launch_ray_unsort(ctx->stream, (const rtg_hit *)tmp_hits, (const rtg_ray_counters *)tmp_ctr, w.perm,
                                           n_rays, (rtg_hit *)hits, (rtg_ray_counters *)oc.dev));
        } else {
            HIP_TRY(ctx, launch_trace_phased(ctx->stream, ds, (const double *)d_rays, n_rays, ps, tune,
                                             (rtg_hit *)hits, (rtg_ray_counters *)oc.dev));
        }
        if (op.dev)
            HIP_TRY(ctx, launch_points_batch(ctx->stream, ds, (const double *)d_rays, n_rays, (const rtg_hit *)hits,
                                             (rtg_point *)op.dev));
    } else {
        HIP_TRY(ctx, launch_trace(ctx->stream, ds, (const double *)d_rays, n_rays, (rtg_hit *)oh.dev,
                                  (rtg_point *)op.dev, (rtg_ray_counters *)oc.dev));
    }
    if ((rc = out_finish(ctx, oh)) != RTG_OK) return rc;
    if ((rc = out_finish(ctx, op)) != RTG_OK) return rc;
    if ((rc = out_finish(ctx, oc)) != RTG_OK) return rc;
    if (oh.staged || op.staged || oc.staged) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RTG_OK;
} catch (...) {
    return api_caught(scene ? scene->ctx : nullptr, "rtg_trace");
}

int rtg_trace_paths(rtg_scene *scene, const double *rays6, uint64_t n_rays, uint32_t max_depth, rtg_path *out_paths,
                    uint64_t capacity, uint64_t *n_paths) try {
    int rc = check_scene_call(scene);
    if (rc != RTG_OK) return rc;
    rtg_ctx *ctx = scene->ctx;
    if (!n_paths) return fail(ctx, RTG_E_INVALID, "rtg_trace_paths: n_paths is null");
    *n_paths = 0;
    if (max_depth < 1 || max_depth > 8) return fail(ctx, RTG_E_INVALID, "rtg_trace_paths: max_depth must be 1..8");
    if (n_rays == 0) return RTG_OK;
    if (!rays6) return fail(ctx, RTG_E_INVALID, "rtg_trace_paths: rays6 is null");
    if (capacity && !out_paths) return fail(ctx, RTG_E_INVALID, "rtg_trace_paths: out_paths is null");
    if (n_rays >= (1ull << 31)) return fail(ctx, RTG_E_LIMIT, "rtg_trace_paths: too many rays for one call");
    if ((uintptr_t)rays6 % 16 != 0 && is_device_pointer(rays6))
        return fail(ctx, RTG_E_INVALID, "rtg_trace_paths: device ray buffer must be 16-byte aligned");
    const void *d_rays0 = nullptr;
    if ((rc = in_prepare(ctx, "in_rays", rays6, n_rays * 6 * sizeof(double), &d_rays0)) != RTG_OK) return rc;

    void *v = nullptr;
    if ((rc = scratch_get(ctx, "path_counters", 4 * sizeof(uint32_t), &v)) != RTG_OK) return rc;
    uint32_t *cnt = (uint32_t *)v;            // [0] rays spawned, [1] unused, [2] refracting hits, [3] records of the level
    uint32_t host_cnt[4] = {0, 0, 0, 0};
    auto read_counters = [&]() -> int {
        HIP_TRY(ctx, hipMemcpyAsync(host_cnt, cnt, sizeof(host_cnt), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return RTG_OK;
    };
    const rtg_path *level_records[8];
    const uint64_t *level_keys[8];
    uint32_t level_first[9], level_count[8];
    uint32_t n_levels = 0;
    uint64_t total = 0;
    const double *d_rays = (const double *)d_rays0;
    uint64_t m = n_rays;
    const rtg_view no_view{};
    const rtg_rect no_tile{};
    for (uint32_t depth = 0; depth < max_depth && m != 0; depth++) {
        const std::string tag = std::to_string(depth);
        void *meta = nullptr, *pts = nullptr, *hits = nullptr, *keys = nullptr, *recs = nullptr;
        if (depth == 0) {
            if ((rc = scratch_get(ctx, "path_meta_0", m * sizeof(SpecMeta), &meta)) != RTG_OK) return rc;
            HIP_TRY(ctx, launch_path_roots(ctx->stream, m, (SpecMeta *)meta));
        } else {
            meta = ctx->scratch["path_meta_" + tag].ptr;
        }
        if ((rc = scratch_get(ctx, "path_points", m * sizeof(rtg_point), &pts)) != RTG_OK) return rc;
        if ((rc = scratch_get(ctx, "path_hits", m * sizeof(rtg_hit), &hits)) != RTG_OK) return rc;
        if ((rc = scratch_get(ctx, ("path_keys_" + tag).c_str(), m * sizeof(uint64_t), &keys)) != RTG_OK) return rc;
        if ((rc = scratch_get(ctx, ("path_records_" + tag).c_str(), m * sizeof(rtg_path), &recs)) != RTG_OK) return rc;
        DevScene ds;
        if ((rc = scene_for_launch(scene, (m + kBlock - 1) / kBlock * kBlock, ds)) != RTG_OK) return rc;
        if (ctx->traversal == 2) {
            PhasedScratch ps;
            if ((rc = phased_scratch(ctx, ds, m, ps)) != RTG_OK) return rc;
            HIP_TRY(ctx, launch_trace_phased(ctx->stream, ds, d_rays, m, ps, ctx->tune, (rtg_hit *)hits, nullptr));
            HIP_TRY(ctx, launch_points_batch(ctx->stream, ds, d_rays, m, (const rtg_hit *)hits, (rtg_point *)pts));
        } else {
            HIP_TRY(ctx, launch_trace(ctx->stream, ds, d_rays, m, (rtg_hit *)hits, (rtg_point *)pts, nullptr));
        }
        HIP_TRY(ctx, hipMemsetAsync(cnt, 0, 4 * sizeof(uint32_t), ctx->stream));
        HIP_TRY(ctx, launch_path_emit(ctx->stream, (const SpecMeta *)meta, (const rtg_point *)pts, m, max_depth, cnt + 3,
                                      (uint64_t *)keys, (rtg_path *)recs));
        const bool deeper = depth + 1 < max_depth;
        // the batch's refracting hits are counted to size level 1; below, a level's own size bounds the next
        if (deeper && depth == 0) HIP_TRY(ctx, launch_spec_count(ctx->stream, ds, (const rtg_point *)pts, m, cnt + 2));
        if ((rc = read_counters()) != RTG_OK) return rc;
        level_records[n_levels] = (const rtg_path *)recs;
        level_keys[n_levels] = (const uint64_t *)keys;
        level_first[n_levels] = (uint32_t)total;
        level_count[n_levels] = host_cnt[3];
        n_levels++;
        total += host_cnt[3];
        if (total >= (1ull << 32)) return fail(ctx, RTG_E_LIMIT, "rtg_trace_paths: more than 2^32 paths");
        uint64_t m_next = 0;
        const uint32_t k = depth == 0 ? host_cnt[2] : (uint32_t)m;
        if (deeper && k != 0) {
            const std::string next_tag = std::to_string(depth + 1);
            void *next_rays = nullptr, *next_meta = nullptr;
            if ((rc = scratch_get(ctx, ("path_rays_" + next_tag).c_str(), (size_t)k * 2 * 6 * sizeof(double), &next_rays)) != RTG_OK) return rc;
            if ((rc = scratch_get(ctx, ("path_meta_" + next_tag).c_str(), (size_t)k * 2 * sizeof(SpecMeta), &next_meta)) != RTG_OK) return rc;
            HIP_TRY(ctx, launch_spec_spawn(ctx->stream, ds, no_view, 0, 0, no_tile, d_rays, (const SpecMeta *)meta,
                                           (const rtg_point *)pts, m, depth, max_depth, (double *)next_rays,
                                           (SpecMeta *)next_meta, nullptr, cnt));
            if ((rc = read_counters()) != RTG_OK) return rc;
            m_next = host_cnt[0];
            d_rays = (const double *)next_rays;
        }
        m = m_next;
    }
    *n_paths = total;
    if (total > capacity)
        return fail(ctx, RTG_E_LIMIT, "rtg_trace_paths: %llu paths, room for %llu", (unsigned long long)total,
                    (unsigned long long)capacity);
    if (total == 0) return RTG_OK;

    uint32_t ray_bits = 1;
    while ((1ull << ray_bits) < n_rays) ray_bits++;
    const uint32_t key_bits = 8 + ray_bits;
    void *keys_all = nullptr, *keys_sorted = nullptr, *source = nullptr, *source_sorted = nullptr, *temp = nullptr;
    const size_t temp_bytes = path_sort_temp_bytes((uint32_t)total, key_bits);
    if ((rc = scratch_get(ctx, "path_keys_all", total * sizeof(uint64_t), &keys_all)) != RTG_OK) return rc;
    if ((rc = scratch_get(ctx, "path_keys_sorted", total * sizeof(uint64_t), &keys_sorted)) != RTG_OK) return rc;
    if ((rc = scratch_get(ctx, "path_source", total * sizeof(uint32_t), &source)) != RTG_OK) return rc;
    if ((rc = scratch_get(ctx, "path_source_sorted", total * sizeof(uint32_t), &source_sorted)) != RTG_OK) return rc;
    if ((rc = scratch_get(ctx, "path_sort_temp", temp_bytes, &temp)) != RTG_OK) return rc;
    for (uint32_t l = 0; l < n_levels; l++)
        if (level_count[l])
            HIP_TRY(ctx, hipMemcpyAsync((uint64_t *)keys_all + level_first[l], level_keys[l],
                                        (size_t)level_count[l] * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
    Out op;
    if ((rc = out_prepare(ctx, "out_paths", out_paths, total * sizeof(rtg_path), op)) != RTG_OK) return rc;
    HIP_TRY(ctx, launch_path_sort_gather(ctx->stream, level_records, level_first, n_levels, (uint32_t)total, key_bits,
                                         (const uint64_t *)keys_all, (uint64_t *)keys_sorted, (uint32_t *)source,
                                         (uint32_t *)source_sorted, temp, temp_bytes, (rtg_path *)op.dev));
    if ((rc = out_finish(ctx, op)) != RTG_OK) return rc;
    if (op.staged) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RTG_OK;
} catch (...) {
    return api_caught(scene ? scene->ctx : nullptr, "rtg_trace_paths");
}

namespace {
// compute_direct_irrad for device-resident points through the persistent kernel: one closest-hit launch over all
// (light, point) shadow rays -- and, in the same launch, over `n_rays` further rays whose hits land behind the
// occluder table (*ray_hits) -- then one resolve pass.
int shade_wavefront(rtg_ctx *ctx, const DevScene &ds, const rtg_light *d_lights, uint32_t n_lights,
                    const rtg_point *d_points, uint64_t n_points, double *d_rgb, uint8_t *d_rgba8,
                    const double *d_rays = nullptr, uint64_t n_rays = 0, const rtg_hit **ray_hits = nullptr,
                    const double *d_indirect = nullptr) {
    int rc;
    void *occ = nullptr;
    const uint64_t n_shadow = (uint64_t)n_lights * n_points;
    if (n_shadow + n_rays > 0xFFFFFFFFull)
        return fail(ctx, RTG_E_LIMIT, "more than 2^32-1 shadow and path rays in one launch (split the call)");
    if (n_shadow + n_rays != 0) {
        PhasedScratch ps;
        if ((rc = phased_scratch(ctx, ds, n_shadow + n_rays, ps)) != RTG_OK) return rc;
        if ((rc = scratch_get(ctx, "occluders", (size_t)(n_shadow + n_rays) * sizeof(rtg_hit), &occ)) != RTG_OK) return rc;
        HIP_TRY(ctx, launch_shadow_phased(ctx->stream, ds, d_points, n_points, d_lights, n_lights, d_rays, n_rays, ps,
                                          ctx->tune, (rtg_hit *)occ));
    }
    if (ray_hits) *ray_hits = (const rtg_hit *)occ + n_shadow;
    HIP_TRY(ctx, launch_shade_resolve(ctx->stream, ds, d_lights, n_lights, d_points, n_points, (const rtg_hit *)occ,
                                      d_rgb, d_rgba8, d_indirect));
    return RTG_OK;
}

// Shading of a frame whose primary point records `pts0` are built, with the specular path tree as levels of ray
// batches when `spec`: per level, the rays of the next level are spawned from the current points, one launch traces
// them together with the current level's shadow rays, the resolve pass shades the current level and its radiance is
// parked per path-tree slot; at the end the refracting pixels are re-summed in the reference's depth-first order.
// *done is false when the slot table would exceed the memory budget (the caller then lets the fused kernel redo the
// refracting pixels).
int frame_wavefront(rtg_ctx *ctx, const DevScene &ds, const rtg_view &view, uint32_t width, uint32_t height, rtg_rect tile,
                    const rtg_light *d_lights, uint32_t n_lights, const rtg_point *pts0, uint64_t n, uint32_t max_depth,
                    bool spec, double *frame_rgb, double *d_rgb, uint8_t *d_rgba8, bool *done) {
    int rc;
    *done = true;
    uint32_t *cnt = nullptr;                  // [0] rays spawned, [1] slot groups opened, [2] refracting hits
    uint32_t host_cnt[4] = {0, 0, 0, 0};
    if (spec) {
        void *v = nullptr;
        if ((rc = scratch_get(ctx, "spec_counters", 4 * sizeof(uint32_t), &v)) != RTG_OK) return rc;
        cnt = (uint32_t *)v;
        HIP_TRY(ctx, hipMemsetAsync(cnt, 0, 4 * sizeof(uint32_t), ctx->stream));
    }
    auto read_counters = [&]() -> int {
        HIP_TRY(ctx, hipMemcpyAsync(host_cnt, cnt, sizeof(host_cnt), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return RTG_OK;
    };
    const uint32_t slots_per_group = (1u << max_depth) - 1u;
    const char *ray_names[2] = {"spec_rays_0", "spec_rays_1"}, *meta_names[2] = {"spec_meta_0", "spec_meta_1"};
    void *rays[2] = {nullptr, nullptr}, *meta[2] = {nullptr, nullptr}, *groups = nullptr, *slots = nullptr;
    uint32_t n_groups = 0;
    int cur = 0;                              // rays[cur] / meta[cur] describe the current level when depth >= 1
    const rtg_point *pts = pts0;
    uint64_t m = n;
    for (uint32_t depth = 0;; depth++) {
        // rays of the next level, spawned from this level's points
        uint64_t m_next = 0;
        const int nxt = depth == 0 ? 0 : cur ^ 1;
        if (spec && depth + 1 < max_depth && m != 0) {
            HIP_TRY(ctx, hipMemsetAsync(cnt, 0, sizeof(uint32_t), ctx->stream));
            // How many rays of this level hit a refracting material bounds the next level (two rays each).  Level 0
            // counts them (the slot table is sized by it); below, every ray descends from a refracting hit and the
            // level's own size is bound enough, which saves a pass over the points and a read-back per level.
            uint32_t k = (uint32_t)m;
            if (depth == 0) {
                HIP_TRY(ctx, hipMemsetAsync(cnt + 2, 0, sizeof(uint32_t), ctx->stream));
                HIP_TRY(ctx, launch_spec_count(ctx->stream, ds, pts, m, cnt + 2));
                if ((rc = read_counters()) != RTG_OK) return rc;
                k = host_cnt[2];
            }
            if (k != 0 && depth == 0) {
                n_groups = k;
                const size_t slot_bytes = (size_t)n_groups * slots_per_group * 4 * sizeof(double);
                if (slot_bytes > ((size_t)2 << 30)) {
                    *done = false;
                    spec = false;
                } else {
                    if ((rc = scratch_get(ctx, "spec_groups", (size_t)n_groups * sizeof(uint32_t), &groups)) != RTG_OK) return rc;
                    if ((rc = scratch_get(ctx, "spec_slots", slot_bytes, &slots)) != RTG_OK) return rc;
                    HIP_TRY(ctx, hipMemsetAsync(slots, 0, slot_bytes, ctx->stream));
                }
            }
            if (k != 0 && spec) {
                if ((rc = scratch_get(ctx, ray_names[nxt], (size_t)k * 2 * 6 * sizeof(double), &rays[nxt])) != RTG_OK) return rc;
                if ((rc = scratch_get(ctx, meta_names[nxt], (size_t)k * 2 * sizeof(SpecMeta), &meta[nxt])) != RTG_OK) return rc;
                HIP_TRY(ctx, launch_spec_spawn(ctx->stream, ds, view, width, height, tile,
                                               depth == 0 ? nullptr : (const double *)rays[cur],
                                               depth == 0 ? nullptr : (const SpecMeta *)meta[cur], pts, m, depth, max_depth,
                                               (double *)rays[nxt], (SpecMeta *)meta[nxt], (uint32_t *)groups, cnt));
                if ((rc = read_counters()) != RTG_OK) return rc;
                m_next = host_cnt[0];
            }
        }
        // this level's shadow rays and the next level's rays in one launch, then this level's radiance
        double *rgb = frame_rgb;
        if (depth != 0) {
            void *v = nullptr;
            if ((rc = scratch_get(ctx, "spec_rgb", m * 3 * sizeof(double), &v)) != RTG_OK) return rc;
            rgb = (double *)v;
        }
        const rtg_hit *next_hits = nullptr;
        if ((rc = shade_wavefront(ctx, ds, d_lights, n_lights, pts, m, rgb, depth == 0 ? d_rgba8 : nullptr,
                                  m_next ? (const double *)rays[nxt] : nullptr, m_next, &next_hits)) != RTG_OK)
            return rc;
        if (depth != 0)
            HIP_TRY(ctx, launch_spec_store(ctx->stream, (const SpecMeta *)meta[cur], pts, rgb, m, slots_per_group,
                                           (double *)slots));
        if (m_next == 0) break;
        void *v = nullptr;
        if ((rc = scratch_get(ctx, "spec_points", m_next * sizeof(rtg_point), &v)) != RTG_OK) return rc;
        HIP_TRY(ctx, launch_points_batch(ctx->stream, ds, (const double *)rays[nxt], m_next, next_hits, (rtg_point *)v));
        pts = (const rtg_point *)v;
        m = m_next;
        cur = nxt;
    }
    if (n_groups != 0 && *done)
        HIP_TRY(ctx, launch_spec_sum(ctx->stream, (const uint32_t *)groups, n_groups, slots_per_group, (const double *)slots,
                                     frame_rgb, d_rgb, d_rgba8));
    return RTG_OK;
}
} // namespace

extern "C++" {
namespace rtg {
int api_shade_points(rtg_scene *scene, const rtg_light *d_lights, uint32_t n_lights, const rtg_point *d_points, uint64_t n,
                     const double *d_indirect, double *d_rgb) {
    rtg_ctx *ctx = scene->ctx;
    if (n == 0) return RTG_OK;
    if (n > 0xFFFFFFFFull || (n_lights != 0 && n > 0xFFFFFFFFull / n_lights))
        return fail(ctx, RTG_E_LIMIT, "too many points for one shading launch");
    int rc;
    DevScene ds;
    if ((rc = scene_for_launch(scene, (n + kBlock - 1) / kBlock * kBlock, ds)) != RTG_OK) return rc;
    if (ctx->traversal == 2 && scene->has_mesh)
        return shade_wavefront(ctx, ds, d_lights, n_lights, d_points, n, d_rgb, nullptr, nullptr, 0, nullptr, d_indirect);
    HIP_TRY(ctx, launch_shade_direct(ctx->stream, ds, d_lights, n_lights, d_points, n, d_rgb, nullptr, d_indirect));
    return RTG_OK;
}
} // namespace rtg
} // extern "C++"

int rtg_shade_direct(rtg_scene *scene, const rtg_light *lights, uint32_t n_lights, const rtg_point *points,
                     uint64_t n_points, double *rgb_out, uint8_t *rgba8_out) try {
    int rc = check_scene_call(scene);
    if (rc != RTG_OK) return rc;
    rtg_ctx *ctx = scene->ctx;
    if (n_points == 0) return RTG_OK;
    if (!points) return fail(ctx, RTG_E_INVALID, "rtg_shade_direct: points is null");
    if (n_lights && !lights) return fail(ctx, RTG_E_INVALID, "rtg_shade_direct: lights is null");
    if (!rgb_out && !rgba8_out) return fail(ctx, RTG_E_INVALID, "rtg_shade_direct: no output requested");
    if (n_points > 0xFFFFFFFFull || (n_lights != 0 && n_points > 0xFFFFFFFFull / n_lights))
        return fail(ctx, RTG_E_LIMIT, "rtg_shade_direct: too many points for one launch");
    const void *d_lights = nullptr, *d_points = nullptr;
    if ((rc = in_prepare(ctx, "in_lights", lights, (size_t)n_lights * sizeof(rtg_light), &d_lights)) != RTG_OK) return rc;
    if ((rc = in_prepare(ctx, "in_points", points, n_points * sizeof(rtg_point), &d_points)) != RTG_OK) return rc;
    Out orgb, o8;
    if ((rc = out_prepare(ctx, "out_rgb", rgb_out, n_points * 3 * sizeof(double), orgb)) != RTG_OK) return rc;
    if ((rc = out_prepare(ctx, "out_rgba8", rgba8_out, n_points * 4, o8)) != RTG_OK) return rc;
    DevScene ds;
    uint64_t threads = (n_points + kBlock - 1) / kBlock * kBlock;
    if ((rc = scene_for_launch(scene, threads, ds)) != RTG_OK) return rc;
    if (ctx->traversal == 2 && scene->has_mesh) {
        if ((rc = shade_wavefront(ctx, ds, (const rtg_light *)d_lights, n_lights, (const rtg_point *)d_points, n_points,
                                  (double *)orgb.dev, (uint8_t *)o8.dev)) != RTG_OK)
            return rc;
    } else {
        HIP_TRY(ctx, launch_shade_direct(ctx->stream, ds, (const rtg_light *)d_lights, n_lights,
                                         (const rtg_point *)d_points, n_points, (double *)orgb.dev, (uint8_t *)o8.dev));
    }
    if ((rc = out_finish(ctx, orgb)) != RTG_OK) return rc;
    if ((rc = out_finish(ctx, o8)) != RTG_OK) return rc;
    if (orgb.staged || o8.staged) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RTG_OK;
} catch (...) {
    return api_caught(scene ? scene->ctx : nullptr, "rtg_shade_direct");
}

int rtg_render_frame(rtg_scene *scene, const rtg_view *view, const rtg_light *lights, uint32_t n_lights,
                     uint32_t width, uint32_t height, rtg_rect tile, uint32_t max_depth, double *out_rgb,
                     uint8_t *out_rgba8, rtg_hit *out_hits) try {
    int rc = check_scene_call(scene);
    if (rc != RTG_OK) return rc;
    rtg_ctx *ctx = scene->ctx;
    if (!view) return fail(ctx, RTG_E_INVALID, "rtg_render_frame: view is null");
    if (n_lights && !lights) return fail(ctx, RTG_E_INVALID, "rtg_render_frame: lights is null");
    if (width == 0 || height == 0) return fail(ctx, RTG_E_INVALID, "rtg_render_frame: empty image");
    if ((uint64_t)tile.x0 + tile.w > width || (uint64_t)tile.y0 + tile.h > height)
        return fail(ctx, RTG_E_INVALID, "rtg_render_frame: tile outside the image");
    if (max_depth > (uint32_t)8) return fail(ctx, RTG_E_LIMIT, "rtg_render_frame: max_depth %u exceeds 8", max_depth);
    uint64_t n = (uint64_t)tile.w * tile.h;
    if (n == 0) return RTG_OK;
    if (!out_rgb && !out_rgba8 && !out_hits) return fail(ctx, RTG_E_INVALID, "rtg_render_frame: no output requested");
    uint64_t tiles = (uint64_t)((tile.w + 15) / 16) * ((tile.h + 15) / 16);
    if (tiles > 0x7FFFFFFFull) return fail(ctx, RTG_E_LIMIT, "rtg_render_frame: tile too large");
    const void *d_lights = nullptr;
    if ((rc = in_prepare(ctx, "in_lights", lights, (size_t)n_lights * sizeof(rtg_light), &d_lights)) != RTG_OK) return rc;
    Out orgb, o8, oh;
    if ((rc = out_prepare(ctx, "out_rgb", out_rgb, n * 3 * sizeof(double), orgb)) != RTG_OK) return rc;
    if ((rc = out_prepare(ctx, "out_rgba8", out_rgba8, n * 4, o8)) != RTG_OK) return rc;
    if ((rc = out_prepare(ctx, "out_hits", out_hits, n * sizeof(rtg_hit), oh)) != RTG_OK) return rc;
    DevScene ds;
    if ((rc = scene_for_launch(scene, tiles * kBlock, ds)) != RTG_OK) return rc;
    if (ctx->traversal == 2 && scene->has_mesh && max_depth >= 1 && (uint64_t)n_lights * n <= 0x7FFFFFFFull) {
        // Every pixel is one primary hit plus its shadow rays unless that hit refracts, so the frame is passes of the
        // persistent kernel's building blocks (primary hit ids, point records, shadow rays, resolve); the pixels
        // whose primary hit has a dielectric material -- the only ones with a path tree -- are then redone by the
        // fused kernel.
        const bool spec = scene->any_dielectric && max_depth > 1;
        // An opaque scene's frame has no host read-backs, so it is issued as two row bands on two streams: a
        // persistent launch ends with a tail (its last rays finish on a half-empty GPU) and the other band's
        // launches fill it.  The second band runs on an internal context (own stream and scratch) that is forked
        // from and joined back into the context's stream by events.
        uint32_t n_bands = (!spec && ctx->frame_bands >= 2 && tile.h >= 128) ? 2u : 1u;
        rtg_ctx *lanes[2] = {ctx, nullptr};
        if (n_bands == 2) {
            if (!ctx->side) {
                if (rtg_ctx_create(ctx->device, &ctx->side) != RTG_OK)
                    return fail(ctx, RTG_E_HIP, "rtg_render_frame: internal context: %s", g_create_error.c_str());
                HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
                HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
            }
            lanes[1] = ctx->side;
            lanes[1]->traversal = ctx->traversal;
            lanes[1]->tune = ctx->tune;
            HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));        // inputs uploaded, earlier calls ordered
            HIP_TRY(ctx, hipStreamWaitEvent(lanes[1]->stream, ctx->ev_fork, 0));
        }
        const uint32_t h0 = n_bands == 2 ? ((tile.h / 2 + 7u) / 8u) * 8u : tile.h;
        for (uint32_t band = 0; band < n_bands; band++) {
            rtg_ctx *lane = lanes[band];
            rtg_rect bt = tile;
            bt.y0 = tile.y0 + (band ? h0 : 0u);
            bt.h = band ? tile.h - h0 : h0;
            const uint64_t first = band ? (uint64_t)h0 * tile.w : 0, bn = (uint64_t)bt.w * bt.h;
            auto run = [&]() -> int {
                int r;
                void *pts = nullptr;
                PhasedScratch ps;
                if ((r = phased_scratch(lane, ds, bn, ps)) != RTG_OK) return r;
                if ((r = scratch_get(lane, "frame_points", bn * sizeof(rtg_point), &pts)) != RTG_OK) return r;
                void *hits = oh.dev ? (void *)((rtg_hit *)oh.dev + first) : nullptr;
                if (!hits && (r = scratch_get(lane, "points_hits", bn * sizeof(rtg_hit), &hits)) != RTG_OK) return r;
                HIP_TRY(lane, launch_render_primary_phased(lane->stream, ds, *view, width, height, bt, ps, lane->tune,
                                                           (rtg_hit *)hits, nullptr));
                if (orgb.dev || o8.dev) {
                    double *rgb_out = orgb.dev ? (double *)orgb.dev + first * 3 : nullptr;
                    uint8_t *c8_out = o8.dev ? (uint8_t *)o8.dev + first * 4 : nullptr;
                    void *frame_rgb = rgb_out;   // the path-tree re-sum needs the linear radiance of level 0
                    if (spec && !frame_rgb && (r = scratch_get(lane, "frame_rgb", bn * 3 * sizeof(double), &frame_rgb)) != RTG_OK)
                        return r;
                    HIP_TRY(lane, launch_points_primary(lane->stream, ds, *view, width, height, bt, (const rtg_hit *)hits,
                                                        (rtg_point *)pts));
                    bool done = true;
                    if ((r = frame_wavefront(lane, ds, *view, width, height, bt, (const rtg_light *)d_lights, n_lights,
                                             (const rtg_point *)pts, bn, max_depth, spec, (double *)frame_rgb, rgb_out,
                                             c8_out, &done)) != RTG_OK)
                        return r;
                    if (spec && !done)      // slot table over budget: the fused kernel redoes the refracting pixels
                        HIP_TRY(lane, launch_render_frame(lane->stream, ds, *view, (const rtg_light *)d_lights, n_lights,
                                                          width, height, bt, max_depth, rgb_out, c8_out, nullptr,
                                                          (const rtg_point *)pts));
                }
                return RTG_OK;
            };
            rc = run();
            if (rc != RTG_OK) {
                if (lane != ctx) ctx->err = lane->err;
                if (n_bands == 2) (void)hipStreamSynchronize(lanes[1]->stream);
                return rc;
            }
        }
        if (n_bands == 2) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_join, lanes[1]->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        }
    } else {
        HIP_TRY(ctx, launch_render_frame(ctx->stream, ds, *view, (const rtg_light *)d_lights, n_lights, width, height,
                                         tile, max_depth, (double *)orgb.dev, (uint8_t *)o8.dev, (rtg_hit *)oh.dev));
    }
    if ((rc = out_finish(ctx, orgb)) != RTG_OK) return rc;
    if ((rc = out_finish(ctx, o8)) != RTG_OK) return rc;
    if ((rc = out_finish(ctx, oh)) != RTG_OK) return rc;
    if (orgb.staged || o8.staged || oh.staged) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RTG_OK;
} catch (...) {
    return api_caught(scene ? scene->ctx : nullptr, "rtg_render_frame");
}

} // extern "C"
