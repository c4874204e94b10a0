// gsx_comm.hip — the one exchange step of the multi-GPU path: finished RGBA8 pixels (whole frames of a view
// batch, or the tile bands of one frame) are gathered to one rank with RCCL point-to-point calls.
//
// The reference has no counterpart (single process, single GPU).  MI355X peers are fully
// connected by xGMI, so a gather is best done as DIRECT sends: inside one ncclGroupStart / ncclGroupEnd the
// root posts one ncclRecv per peer and every peer one ncclSend; each transfer uses its own link and there is
// no ring.  Everything is queued on the context's HIP stream, i.e. ordered behind the frames already queued on
// it, and nothing here waits on the host.
//
// librccl is loaded with dlopen when the first communicator call is made: single-GPU users of libgsx do not
// load it at all, and a missing RCCL is an error of these calls only.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>

#include "gsx_internal.h"

static_assert(NCCL_UNIQUE_ID_BYTES == GSX_COMM_ID_BYTES, "gsx.h promises the id size");

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

RcclApi* rccl() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return &api;
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (api.handle) break;
    }
    if (!api.handle) { api.error = std::string("cannot load librccl: ") + dlerror(); return &api; }
    bool ok = true;
    auto sym = [&](const char* name) { void* p = dlsym(api.handle, name); if (!p) { ok = false; api.error = std::string("librccl lacks ") + name; } return p; };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { dlclose(api.handle); api.handle = nullptr; }
    return &api;
}

}  // namespace

struct gsx_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    gsx_ctx* ctx = nullptr;
};

#define GSX_RCCL(ctx, api, call)                                                          \
    do {                                                                                  \
        const ncclResult_t r_ = (call);                                                   \
        if (r_ != ncclSuccess) {                                                          \
            (ctx)->err = std::string(#call) + ": " + (api)->GetErrorString(r_);           \
            return GSX_E_HIP;                                                             \
        }                                                                                 \
    } while (0)

extern "C" {

int gsx_comm_unique_id(void* id128_out) {
    if (!id128_out) return GSX_E_INVALID;
    RcclApi* api = rccl();
    if (!api->handle) return GSX_E_UNSUPPORTED;
    ncclUniqueId id;
    if (api->GetUniqueId(&id) != ncclSuccess) return GSX_E_HIP;
    std::memcpy(id128_out, id.internal, GSX_COMM_ID_BYTES);
    return GSX_OK;
}

int gsx_comm_init(gsx_ctx* c, const void* id128, int rank, int nranks, gsx_comm** out) {
    if (!c || !out || nranks < 1 || rank < 0 || rank >= nranks || (!id128 && nranks > 1)) return GSX_E_INVALID;
    *out = nullptr;
    GSX_HIP(c, hipSetDevice(c->device));
    gsx_comm* m = new (std::nothrow) gsx_comm();
    if (!m) return GSX_E_NOMEM;
    m->rank = rank;
    m->nranks = nranks;
    m->ctx = c;
    if (nranks > 1) {     // (a single rank gathers from itself with a device copy and needs no RCCL)
        RcclApi* api = rccl();
        if (!api->handle) { c->err = api->error; delete m; return GSX_E_UNSUPPORTED; }
        ncclUniqueId id;
        std::memcpy(id.internal, id128, GSX_COMM_ID_BYTES);
        const ncclResult_t r = api->CommInitRank(&m->comm, nranks, id, rank);   // collective: returns when every rank has joined
        if (r != ncclSuccess) { c->err = std::string("ncclCommInitRank: ") + api->GetErrorString(r); delete m; return GSX_E_HIP; }
    }
    *out = m;
    return GSX_OK;
}

int gsx_comm_destroy(gsx_comm* m) {
    if (!m) return GSX_E_INVALID;
    if (m->comm) {
        (void)hipSetDevice(m->ctx->device);
        (void)hipStreamSynchronize(m->ctx->stream);
        (void)rccl()->CommDestroy(m->comm);
    }
    delete m;
    return GSX_OK;
}

int gsx_gather(gsx_ctx* c, gsx_comm* m, const void* send_dev, uint64_t send_bytes, void* recv_dev, const uint64_t* recv_offsets,
               const uint64_t* recv_bytes, int root) {
    if (!c || !m || m->ctx != c || root < 0 || root >= m->nranks || (!send_dev && send_bytes)) return GSX_E_INVALID;
    const bool is_root = m->rank == root;
    if (is_root && (!recv_dev || !recv_offsets || !recv_bytes)) return GSX_E_INVALID;
    if (is_root && recv_bytes[root] != send_bytes) { c->err = "gsx_gather: recv_bytes[root] differs from the root's own send_bytes"; return GSX_E_INVALID; }
    GSX_HIP(c, hipSetDevice(c->device));
    // a frame queued by gsx_render may still need on-demand iterations: they must be queued before the
    // exchange that ships its pixels (no-op when the frame was completed already, e.g. by gsx_set_framebuffer_dev)
    { const int frc = gsx_frame_finish(c); if (frc) return frc; }
    if (is_root && send_bytes) {   // the root's own part never touches the network
        unsigned char* dst = static_cast<unsigned char*>(recv_dev) + recv_offsets[root];
        if (dst != send_dev) GSX_HIP(c, hipMemcpyAsync(dst, send_dev, send_bytes, hipMemcpyDeviceToDevice, c->stream));
    }
    if (m->nranks == 1) return GSX_OK;
    RcclApi* api = rccl();
    if (is_root) {
        GSX_RCCL(c, api, api->GroupStart());
        for (int r = 0; r < m->nranks; ++r) {
            if (r == root || recv_bytes[r] == 0) continue;
            const ncclResult_t rr = api->Recv(static_cast<unsigned char*>(recv_dev) + recv_offsets[r], recv_bytes[r], ncclUint8, r, m->comm, c->stream);
            if (rr != ncclSuccess) { (void)api->GroupEnd(); c->err = std::string("ncclRecv: ") + api->GetErrorString(rr); return GSX_E_HIP; }
        }
        GSX_RCCL(c, api, api->GroupEnd());
    } else if (send_bytes) {
        GSX_RCCL(c, api, api->Send(send_dev, send_bytes, ncclUint8, root, m->comm, c->stream));
    }
    return GSX_OK;
}

}  // extern "C"
