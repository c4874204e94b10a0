// gsx_comm.hip — the one exchange step of the multi-GPU path: finished RGBA8 pixels (whole frames of a view
// batch, or the tile bands of one frame) are gathered to one rank with RCCL point-to-point calls.
//
// The reference has no counterpart (single process, single GPU).  MI355X peers are fully
// connected by xGMI, so a gather is best done as DIRECT sends: inside one ncclGroupStart / ncclGroupEnd the
// root posts one ncclRecv per peer and every peer one ncclSend; each transfer uses its own link and there is
// no ring.  The exchange is queued on a stream of its own (gsx_ctx::comm_stream), ordered behind the frames that
// render into the memory it ships and beside every other frame: the next batch is rendered while RCCL moves this one.
// A later frame that targets memory an exchange still uses waits for it on the device (gsx_comm_guard).
//
// librccl is loaded with dlopen when the first communicator call is made: single-GPU users of libgsx do not
// load it at all, and a missing RCCL is an error of these calls only.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
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

static void rccl_load(RcclApi& api);

// Loaded once per process; a function-local static is initialised by exactly one thread (the others wait), so two host
// threads creating communicators at the same time never see a half-filled table.
RcclApi* rccl() {
    static RcclApi api = [] { RcclApi a; rccl_load(a); return a; }();
    return &api;
}

static void rccl_load(RcclApi& api) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (api.handle) break;
    }
    if (!api.handle) { api.error = std::string("cannot load librccl: ") + dlerror(); return; }
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

// st waits for the exchanges that still read or write [lo, hi): a frame about to render there, as a rule.
int gsx_comm_guard(gsx_ctx* c, hipStream_t st, const uint8_t* lo, const uint8_t* hi) {
    for (auto& g : c->gathers) {
        if (!g.pending) continue;
        if (hipEventQuery(g.ev) == hipSuccess) { g.pending = false; continue; }
        bool hit = false;
        for (int k = 0; k < 2; ++k) hit = hit || (g.lo[k] && g.lo[k] < hi && lo < g.hi[k]);
        if (hit) GSX_HIP(c, hipStreamWaitEvent(st, g.ev, 0));
    }
    return GSX_OK;
}

// The main stream waits for every exchange queued so far: whoever synchronizes with it next, or queues behind it, sees them done.
int gsx_comm_join(gsx_ctx* c) {
    for (auto& g : c->gathers) {
        if (!g.pending) continue;
        if (hipEventQuery(g.ev) == hipSuccess) { g.pending = false; continue; }
        GSX_HIP(c, hipStreamWaitEvent(c->stream, g.ev, 0));
    }
    return GSX_OK;
}

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
    // (a single rank gathers from itself with a device copy and needs no RCCL; GSX_COMM_FORCE_RCCL=1 in the environment makes
    // it a real one-rank communicator whose gather is an RCCL send to itself: the one-GPU rehearsal of the exchange)
    static const bool force_rccl = std::getenv("GSX_COMM_FORCE_RCCL") != nullptr;
    if (nranks > 1 || force_rccl) {
        RcclApi* api = rccl();
        if (!api->handle) { c->err = api->error; delete m; return GSX_E_UNSUPPORTED; }
        ncclUniqueId id;
        if (id128) std::memcpy(id.internal, id128, GSX_COMM_ID_BYTES);
        else if (api->GetUniqueId(&id) != ncclSuccess) { c->err = "ncclGetUniqueId failed"; delete m; return GSX_E_HIP; }
        const ncclResult_t r = api->CommInitRank(&m->comm, nranks, id, rank);   // collective: returns when every rank has joined
        if (r != ncclSuccess) { c->err = std::string("ncclCommInitRank: ") + api->GetErrorString(r); delete m; return GSX_E_HIP; }
    }
    // the context keeps track of its communicators so that gsx_destroy can release one that is still alive
    bool filed = false;
    for (auto& slot : c->comms)
        if (!slot) { slot = m; filed = true; break; }
    if (!filed) {
        if (m->comm) (void)rccl()->CommDestroy(m->comm);
        delete m;
        c->err = "gsx_comm_init: more than 8 live communicators on one context";
        return GSX_E_STATE;
    }
    *out = m;
    return GSX_OK;
}

int gsx_comm_destroy(gsx_comm* m) {
    if (!m) return GSX_E_INVALID;
    if (m->ctx) {                  // (null: the context went first and gsx_destroy has already drained and released the communicator)
        gsx_ctx* c = m->ctx;
        if (m->comm) {
            (void)hipSetDevice(c->device);
            if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
            (void)hipStreamSynchronize(c->stream);
            (void)rccl()->CommDestroy(m->comm);
        }
        for (auto& slot : c->comms)
            if (slot == m) slot = nullptr;
    }
    delete m;
    return GSX_OK;
}

// gsx_destroy: communicators that outlive their context are drained and detached here; their handles stay valid for
// gsx_comm_destroy (which then only frees them) and are rejected by gsx_gather (ctx mismatch).
extern "C++" void gsx_comm_detach_all(gsx_ctx* c) {
    for (auto& slot : c->comms) {
        if (!slot) continue;
        if (slot->comm) {
            if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
            (void)rccl()->CommDestroy(slot->comm);
            slot->comm = nullptr;
        }
        slot->ctx = nullptr;
        slot = nullptr;
    }
}

int gsx_gather(gsx_ctx* c, gsx_comm* m, const void* send_dev, uint64_t send_bytes, void* recv_dev, const uint64_t* recv_offsets,
               const uint64_t* recv_bytes, int root) {
    if (!c || !m || m->ctx != c || root < 0 || root >= m->nranks || (!send_dev && send_bytes)) return GSX_E_INVALID;
    const bool is_root = m->rank == root;
    if (is_root && (!recv_dev || !recv_offsets || !recv_bytes)) return GSX_E_INVALID;
    if (is_root && recv_bytes[root] != send_bytes) { c->err = "gsx_gather: recv_bytes[root] differs from the root's own send_bytes"; return GSX_E_INVALID; }
    GSX_HIP(c, hipSetDevice(c->device));
    // the memory this exchange reads (what is sent) and writes (root: everything received)
    const uint8_t* lo[2] = {static_cast<const uint8_t*>(send_dev), nullptr};
    const uint8_t* hi[2] = {lo[0] + send_bytes, nullptr};
    if (is_root) {
        uint64_t a = ~0ull, b = 0;
        for (int r = 0; r < m->nranks; ++r) {
            if (!recv_bytes[r]) continue;
            a = recv_offsets[r] < a ? recv_offsets[r] : a;
            b = recv_offsets[r] + recv_bytes[r] > b ? recv_offsets[r] + recv_bytes[r] : b;
        }
        if (b > a) { lo[1] = static_cast<const uint8_t*>(recv_dev) + a; hi[1] = static_cast<const uint8_t*>(recv_dev) + b; }
    }
    // A frame queued by gsx_render may still need on-demand iterations, which only the host can queue: the
    // frames that render into this memory are completed first.  Frames queued into OTHER memory stay in flight - a caller
    // that posts the exchange of a batch one or two frames late (multigpu.py) never makes the host wait here.
    for (int k = 0; k < 2; ++k) { const int frc = gsx_frame_finish_touching(c, lo[k], hi[k]); if (frc) return frc; }
    if (!c->comm_stream) {
        GSX_HIP(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
        GSX_HIP(c, hipEventCreateWithFlags(&c->ev_comm_pre, hipEventDisableTiming));
        for (auto& g : c->gathers) GSX_HIP(c, hipEventCreateWithFlags(&g.ev, hipEventDisableTiming));
    }
    hipStream_t st = c->comm_stream;
    // The frames that wrote this memory are complete: the host has just waited for them.  A caller that was handed the main
    // stream (gsx_get_stream) may have filled the send buffer there with work of its own, so the exchange is ordered behind
    // what that stream holds now - only then: the wait would also chain the exchange to a frame of the NEXT batch that is in
    // flight on the main stream, and through the guard of the frame after it the two frame slots to each other.
    if (c->stream_handed_out) {
        GSX_HIP(c, hipEventRecord(c->ev_comm_pre, c->stream));
        GSX_HIP(c, hipStreamWaitEvent(st, c->ev_comm_pre, 0));
        c->stream_handed_out = false;
    }
    gsx_ctx::GatherRec& rec = c->gathers[c->gather_pos++ % GSX_GATHER_RING];
    if (rec.pending) GSX_HIP(c, hipEventSynchronize(rec.ev));    // (more than GSX_GATHER_RING exchanges in flight: the oldest is waited for)
    rec.pending = false;
    if (is_root && send_bytes) {   // the root's own part never touches the network
        unsigned char* dst = static_cast<unsigned char*>(recv_dev) + recv_offsets[root];
        if (dst == send_dev) {
            // already in place
        } else if (m->nranks == 1 && m->comm) {      // rehearsal (GSX_COMM_FORCE_RCCL): the same bytes through RCCL's send / recv kernels
            RcclApi* api = rccl();
            GSX_RCCL(c, api, api->GroupStart());
            const ncclResult_t r1 = api->Recv(dst, send_bytes, ncclUint8, 0, m->comm, st);
            const ncclResult_t r2 = r1 == ncclSuccess ? api->Send(send_dev, send_bytes, ncclUint8, 0, m->comm, st) : r1;
            if (r2 != ncclSuccess) { (void)api->GroupEnd(); c->err = std::string("ncclSend/Recv to self: ") + api->GetErrorString(r2); return GSX_E_HIP; }
            GSX_RCCL(c, api, api->GroupEnd());
        } else {
            GSX_HIP(c, hipMemcpyAsync(dst, send_dev, send_bytes, hipMemcpyDeviceToDevice, st));
        }
    }
    if (m->nranks > 1) {
        RcclApi* api = rccl();
        if (is_root) {
            GSX_RCCL(c, api, api->GroupStart());
            for (int r = 0; r < m->nranks; ++r) {
                if (r == root || recv_bytes[r] == 0) continue;
                const ncclResult_t rr = api->Recv(static_cast<unsigned char*>(recv_dev) + recv_offsets[r], recv_bytes[r], ncclUint8, r, m->comm, st);
                if (rr != ncclSuccess) { (void)api->GroupEnd(); c->err = std::string("ncclRecv: ") + api->GetErrorString(rr); return GSX_E_HIP; }
            }
            GSX_RCCL(c, api, api->GroupEnd());
        } else if (send_bytes) {
            GSX_RCCL(c, api, api->Send(send_dev, send_bytes, ncclUint8, root, m->comm, st));
        }
    }
    for (int k = 0; k < 2; ++k) { rec.lo[k] = lo[k]; rec.hi[k] = hi[k]; }
    GSX_HIP(c, hipEventRecord(rec.ev, st));
    rec.pending = true;
    return GSX_OK;
}

}  // extern "C"
