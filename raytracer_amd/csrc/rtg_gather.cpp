// rtg_gather.cpp -- gather of finished framebuffer pieces to one GPU, behind the C ABI.
//
// The ray-cast path has no exchange during compute (scene replicated, pixels and frames independent); the only data
// that moves between the GPUs of a node is finished output: one gather to the root per frame.  It is a group of RCCL
// point-to-point transfers over xGMI -- every non-root rank sends its pieces, the root posts the matching receives
// and copies its own pieces device-to-device -- enqueued on a stream, so it overlaps the next frame's traversal.
// Nothing is summed, so there is no ring and no reduction.
//
// RCCL is bound at run time (dlopen of librccl.so.1, first with RTLD_NOLOAD): a process that already carries an
// RCCL -- e.g. through torch.distributed -- shares that one copy, and the library loads on hosts without RCCL, where
// a one-rank communicator (plain device copies) still works.  A communicator can only be adopted from the copy the
// process already carries: handing a foreign ncclComm_t to a second copy of the library would be undefined.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>

#include "rtg_internal.hpp"

using rtg::api_fail;

struct rtg_comm {
    rtg_ctx *ctx = nullptr;
    int device = 0;              // the context's GPU, kept so that destruction does not read the context
    int n_ranks = 1, rank = 0;
    ncclComm_t nccl = nullptr;
    bool owned = false;          // created here (destroyed here) or adopted from the caller
    bool loopback = false;       // one rank, but through RCCL: the root sends its own pieces to itself
};

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    bool resident = false;       // the copy was already part of the process when this library first asked for it
    std::string error;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl &rccl() {
    std::call_once(g_rccl_once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {                    // the copy the process already carries, if any
            g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (g_rccl.handle) {
                g_rccl.resident = true;
                break;
            }
        }
        for (const char *n : names) {
            if (g_rccl.handle) break;
            g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!g_rccl.handle) {
            const char *why = dlerror();
            g_rccl.error = std::string("librccl.so.1 not loadable: ") + (why ? why : "?");
            return;
        }
        bool ok = true;
        auto bind = [&](const char *sym) {
            void *p = dlsym(g_rccl.handle, sym);
            if (!p) {
                ok = false;
                g_rccl.error = std::string("RCCL symbol missing: ") + sym;
            }
            return p;
        };
        g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))bind("ncclGetUniqueId");
        g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))bind("ncclCommInitRank");
        g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))bind("ncclCommDestroy");
        g_rccl.GroupStart = (decltype(g_rccl.GroupStart))bind("ncclGroupStart");
        g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))bind("ncclGroupEnd");
        g_rccl.Send = (decltype(g_rccl.Send))bind("ncclSend");
        g_rccl.Recv = (decltype(g_rccl.Recv))bind("ncclRecv");
        g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))bind("ncclGetErrorString");
        g_rccl.CommCount = (decltype(g_rccl.CommCount))bind("ncclCommCount");
        if (!ok) {
            dlclose(g_rccl.handle);
            g_rccl.handle = nullptr;
        }
    });
    return g_rccl;
}

static_assert(RTG_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id blob is RCCL's unique id");

} // namespace

extern "C" {

int rtg_comm_unique_id(uint8_t id[RTG_COMM_ID_BYTES]) try {
    if (!id) return api_fail(nullptr, RTG_E_INVALID, "rtg_comm_unique_id: id is null");
    const Rccl &r = rccl();
    if (!r.handle) return api_fail(nullptr, RTG_E_UNSUPPORTED, "rtg_comm_unique_id: %s", r.error.c_str());
    ncclUniqueId u;
    ncclResult_t e = r.GetUniqueId(&u);
    if (e != ncclSuccess) return api_fail(nullptr, RTG_E_HIP, "ncclGetUniqueId: %s", r.GetErrorString(e));
    std::memcpy(id, u.internal, RTG_COMM_ID_BYTES);
    return RTG_OK;
} catch (...) {
    return rtg::api_caught(nullptr, "rtg_comm_unique_id");
}

int rtg_comm_create(rtg_ctx *ctx, const uint8_t id[RTG_COMM_ID_BYTES], int n_ranks, int rank, rtg_comm **out) try {
    if (!ctx) return api_fail(nullptr, RTG_E_INVALID, "null ctx");
    if (!out) return api_fail(ctx, RTG_E_INVALID, "rtg_comm_create: out is null");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return api_fail(ctx, RTG_E_INVALID, "rtg_comm_create: rank %d of %d", rank, n_ranks);
    hipError_t he = hipSetDevice(ctx->device);
    if (he != hipSuccess) return api_fail(ctx, RTG_E_HIP, "hipSetDevice(%d): %s", ctx->device, hipGetErrorString(he));
    rtg_comm *c = new (std::nothrow) rtg_comm();
    if (!c) return api_fail(ctx, RTG_E_NOMEM, "out of host memory");
    c->ctx = ctx;
    c->device = ctx->device;
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->loopback = n_ranks == 1 && ctx->comm_loopback;
    if (n_ranks > 1 || c->loopback) {    // a single rank gathers with device copies alone (unless asked otherwise)
        if (!id) {
            delete c;
            return api_fail(ctx, RTG_E_INVALID, "rtg_comm_create: id is null");
        }
        const Rccl &r = rccl();
        if (!r.handle) {
            delete c;
            return api_fail(ctx, RTG_E_UNSUPPORTED, "rtg_comm_create: %s", r.error.c_str());
        }
        ncclUniqueId u;
        std::memcpy(u.internal, id, RTG_COMM_ID_BYTES);
        ncclResult_t e = r.CommInitRank(&c->nccl, n_ranks, u, rank);
        if (e != ncclSuccess) {
            delete c;
            return api_fail(ctx, RTG_E_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, n_ranks, r.GetErrorString(e));
        }
        c->owned = true;
    }
    *out = c;
    return RTG_OK;
} catch (...) {
    return rtg::api_caught(ctx, "rtg_comm_create");
}

int rtg_comm_adopt(rtg_ctx *ctx, void *nccl_comm, int n_ranks, int rank, rtg_comm **out) try {
    if (!ctx) return api_fail(nullptr, RTG_E_INVALID, "null ctx");
    if (!out) return api_fail(ctx, RTG_E_INVALID, "rtg_comm_adopt: out is null");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return api_fail(ctx, RTG_E_INVALID, "rtg_comm_adopt: rank %d of %d", rank, n_ranks);
    if (n_ranks > 1) {
        if (!nccl_comm) return api_fail(ctx, RTG_E_INVALID, "rtg_comm_adopt: communicator is null");
        if (!rccl().handle) return api_fail(ctx, RTG_E_UNSUPPORTED, "rtg_comm_adopt: %s", rccl().error.c_str());
        if (!rccl().resident)
            return api_fail(ctx, RTG_E_UNSUPPORTED,
                            "rtg_comm_adopt: no RCCL was loaded in this process before the library's own copy, so the "
                            "communicator belongs to another copy of RCCL");
        int count = 0;
        ncclResult_t e = rccl().CommCount((ncclComm_t)nccl_comm, &count);
        if (e != ncclSuccess || count != n_ranks)
            return api_fail(ctx, RTG_E_INVALID, "rtg_comm_adopt: the communicator reports %d ranks, the call %d", count, n_ranks);
    }
    rtg_comm *c = new (std::nothrow) rtg_comm();
    if (!c) return api_fail(ctx, RTG_E_NOMEM, "out of host memory");
    c->ctx = ctx;
    c->device = ctx->device;
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->nccl = (ncclComm_t)nccl_comm;
    c->owned = false;
    *out = c;
    return RTG_OK;
} catch (...) {
    return rtg::api_caught(ctx, "rtg_comm_adopt");
}

void rtg_comm_destroy(rtg_comm *comm) {
    if (!comm) return;
    if (comm->owned && comm->nccl && rccl().handle) {
        (void)hipSetDevice(comm->device);
        (void)rccl().CommDestroy(comm->nccl);
    }
    delete comm;
}

int rtg_comm_info(const rtg_comm *comm, int *n_ranks, int *rank, int *rccl_ranks) try {
    if (!comm) return api_fail(nullptr, RTG_E_INVALID, "null communicator");
    if (n_ranks) *n_ranks = comm->n_ranks;
    if (rank) *rank = comm->rank;
    if (rccl_ranks) {
        *rccl_ranks = 0;                         // no RCCL communicator behind it: one rank, device copies
        if (comm->nccl && rccl().handle) {
            ncclResult_t e = rccl().CommCount(comm->nccl, rccl_ranks);
            if (e != ncclSuccess) return api_fail(comm->ctx, RTG_E_HIP, "ncclCommCount: %s", rccl().GetErrorString(e));
        }
    }
    return RTG_OK;
} catch (...) {
    return rtg::api_caught(nullptr, "rtg_comm_info");
}

int rtg_gather_tiles(rtg_comm *comm, int root, const void *send, void *recv, const rtg_piece *pieces,
                     uint32_t n_pieces, void *hip_stream) try {
    if (!comm || !comm->ctx) return api_fail(nullptr, RTG_E_INVALID, "null communicator");
    rtg_ctx *ctx = comm->ctx;
    if (root < 0 || root >= comm->n_ranks) return api_fail(ctx, RTG_E_INVALID, "rtg_gather_tiles: root %d of %d ranks", root, comm->n_ranks);
    if (n_pieces && !pieces) return api_fail(ctx, RTG_E_INVALID, "rtg_gather_tiles: pieces is null");
    const bool is_root = comm->rank == root;
    uint64_t mine = 0, incoming = 0;
    for (uint32_t i = 0; i < n_pieces; i++) {
        if (pieces[i].rank >= (uint32_t)comm->n_ranks)
            return api_fail(ctx, RTG_E_INVALID, "rtg_gather_tiles: piece %u names rank %u of %d", i, pieces[i].rank, comm->n_ranks);
        if (pieces[i].rank == (uint32_t)comm->rank) mine += pieces[i].bytes;
        incoming += pieces[i].bytes;
    }
    if (mine && !send) return api_fail(ctx, RTG_E_INVALID, "rtg_gather_tiles: send is null");
    if (is_root && incoming && !recv) return api_fail(ctx, RTG_E_INVALID, "rtg_gather_tiles: recv is null on the root");
    if ((mine && !rtg::api_is_device_pointer(send)) || (is_root && incoming && !rtg::api_is_device_pointer(recv)))
        return api_fail(ctx, RTG_E_INVALID, "rtg_gather_tiles: send and recv must be device memory");
    hipError_t he = hipSetDevice(ctx->device);
    if (he != hipSuccess) return api_fail(ctx, RTG_E_HIP, "hipSetDevice(%d): %s", ctx->device, hipGetErrorString(he));
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : ctx->stream;

    // the root's own pieces never leave the GPU
    if (is_root && !comm->loopback)
        for (uint32_t i = 0; i < n_pieces; i++) {
            const rtg_piece &p = pieces[i];
            if (p.rank != (uint32_t)root || p.bytes == 0) continue;
            if ((const uint8_t *)send + p.send_offset == (uint8_t *)recv + p.recv_offset) continue;   // rendered in place
            he = hipMemcpyAsync((uint8_t *)recv + p.recv_offset, (const uint8_t *)send + p.send_offset, p.bytes,
                                hipMemcpyDeviceToDevice, stream);
            if (he != hipSuccess) {
                (void)hipGetLastError();
                return api_fail(ctx, RTG_E_HIP, "rtg_gather_tiles: device copy of piece %u: %s", i, hipGetErrorString(he));
            }
        }
    if (comm->n_ranks == 1 && !comm->loopback) return RTG_OK;

    // One group: the sends of a rank and the root's receives from it are matched in list order per peer.
    const Rccl &r = rccl();
    ncclResult_t e = r.GroupStart();
    if (e != ncclSuccess) return api_fail(ctx, RTG_E_HIP, "ncclGroupStart: %s", r.GetErrorString(e));
    ncclResult_t first = ncclSuccess;
    uint32_t bad = 0;
    for (uint32_t i = 0; i < n_pieces && first == ncclSuccess; i++) {
        const rtg_piece &p = pieces[i];
        if (p.bytes == 0) continue;
        if (p.rank == (uint32_t)root) {
            if (!comm->loopback || !is_root) continue;
            if ((const uint8_t *)send + p.send_offset == (uint8_t *)recv + p.recv_offset) continue;   // rendered in place
            first = r.Send((const uint8_t *)send + p.send_offset, (size_t)p.bytes, ncclUint8, root, comm->nccl, stream);
            if (first == ncclSuccess)
                first = r.Recv((uint8_t *)recv + p.recv_offset, (size_t)p.bytes, ncclUint8, root, comm->nccl, stream);
            bad = i;
            continue;
        }
        if (is_root)
            first = r.Recv((uint8_t *)recv + p.recv_offset, (size_t)p.bytes, ncclUint8, (int)p.rank, comm->nccl, stream);
        else if (p.rank == (uint32_t)comm->rank)
            first = r.Send((const uint8_t *)send + p.send_offset, (size_t)p.bytes, ncclUint8, root, comm->nccl, stream);
        bad = i;
    }
    e = r.GroupEnd();
    if (first != ncclSuccess) return api_fail(ctx, RTG_E_HIP, "rtg_gather_tiles: piece %u: %s", bad, r.GetErrorString(first));
    if (e != ncclSuccess) return api_fail(ctx, RTG_E_HIP, "ncclGroupEnd: %s", r.GetErrorString(e));
    return RTG_OK;
} catch (...) {
    return rtg::api_caught(comm ? comm->ctx : nullptr, "rtg_gather_tiles");
}

} // extern "C"
