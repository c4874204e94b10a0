// gsx_api.hip — the extern "C" entry points declared in include/gsx.h.
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <new>
#include <unordered_set>
#include <vector>

#include "gsx_internal.h"
#ifndef GSX_RADIX_MAX_BLOCKS
#define GSX_RADIX_MAX_BLOCKS 2048
#endif

uint64_t gsx_sort_workspace_words(uint64_t cap);
uint64_t gsx_sort2_ws_bytes(uint64_t cap);
int gsx_launch_aabb(gsx_ctx* c);
uint32_t gsx_sort_slots(int cus);
uint32_t gsx_pair_slots(int cus);

namespace {

template <typename T>
int dev_alloc(gsx_ctx* c, T** p, uint64_t count) {
    if (*p) { GSX_HIP(c, hipFree(*p)); *p = nullptr; }
    if (count == 0) count = 1;
    GSX_HIP(c, hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    return GSX_OK;
}

template <typename T>
void dev_free(T** p) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
}

int set_device(gsx_ctx* c) {
    GSX_HIP(c, hipSetDevice(c->device));
    return GSX_OK;
}

// Reads the device flag block (synchronises the stream).
int check_flags(gsx_ctx* c) {
    int frc = gsx_frame_finish(c);
    if (frc) return frc;
    GSX_HIP(c, hipMemcpyAsync(c->flags_host, c->flags, sizeof(GsxDevFlags), hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    const bool bad_index = c->flags_host->bad_index != 0, key_bound = c->flags_host->key_bound != 0, lod_fault = c->flags_host->lod_fault != 0;
    if (bad_index || key_bound || lod_fault) {
        // reported once: the words are cleared so that the next (valid) selection / sort on this context succeeds
        static_assert(offsetof(GsxDevFlags, key_bound) == offsetof(GsxDevFlags, bad_index) + 4 && offsetof(GsxDevFlags, lod_fault) == offsetof(GsxDevFlags, bad_index) + 8,
                      "bad_index, key_bound and lod_fault are adjacent");
        GSX_HIP(c, hipMemsetAsync(&c->flags->bad_index, 0, 12, c->stream));
    }
    if (bad_index) { c->err = "selection list holds an id >= n"; return GSX_E_INVALID; }
    if (lod_fault) { c->err = "device: the selection's compaction or the octree walk did not complete (lod_fault)"; return GSX_E_INTERNAL; }
    if (key_bound) { c->err = "device: a depth key left the scene's distance bound (non-finite position?)"; return GSX_E_INTERNAL; }
    return GSX_OK;
}

int alloc_scene(gsx_ctx* c, uint64_t n) {
    int rc;
    if ((rc = dev_alloc(c, &c->pos4, n)) || (rc = dev_alloc(c, &c->geo, n)) || (rc = dev_alloc(c, &c->pos3, n * 3))) return rc;
    if ((rc = dev_alloc(c, &c->sel_idx, n))) return rc;
    c->sel_cap = n;
    for (int b = 0; b < 2; ++b)
        if ((rc = dev_alloc(c, &c->keys[b], n)) || (rc = dev_alloc(c, &c->vals[b], n))) return rc;
    for (int b = 0; b < GSX_PERM_RING; ++b)
        if ((rc = dev_alloc(c, &c->perm[b], n))) return rc;
    c->sort_cap = n;
    c->sort_ws_words = gsx_sort_workspace_words(n);
    if ((rc = dev_alloc(c, &c->sort_ws, c->sort_ws_words))) return rc;
    // two-pass depth sort: (key, index) pairs, their scratch twin, count / prefix matrices
    if ((rc = dev_alloc(c, &c->s2_pairs, n)) || (rc = dev_alloc(c, &c->s2_scratch, n))) return rc;
    c->s2_ws_bytes = gsx_sort2_ws_bytes(n);
    if ((rc = dev_alloc(c, &c->s2_ws, c->s2_ws_bytes))) return rc;
    for (int r = 0; r < 24; ++r) c->s2_backoff[r] = c->s2_backoff_len[r] = 0;
    c->s2_mid_left = 0;
    // LOD / list-selection scratch (flags padded for 16-byte vector loads)
    const uint64_t fl = (n + 15) / 16 * 16 + 16;
    if ((rc = dev_alloc(c, &c->lod_stop, fl)) || (rc = dev_alloc(c, &c->lod_emit, fl))) return rc;
    if ((rc = dev_alloc(c, &c->lod_block_sums, n / 4096 + 2))) return rc;
    if ((rc = dev_alloc(c, &c->lod_agg, 1024))) return rc;      // (>= GSX_CP1_MAXG words)
    GSX_HIP(c, hipMemset(c->lod_agg, 0, 1024 * sizeof(uint64_t)));
    c->lod_seq = 0;
    c->lod_cap = n;
    return GSX_OK;
}

float ev_ms(gsx_ctx* c, int a, int b) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->ev[a], c->ev[b]) != hipSuccess) return 0.0f;
    return ms;
}

}  // namespace

extern "C" {

int gsx_create(int device, gsx_ctx** out) {
    if (!out) return GSX_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return GSX_E_HIP;
    gsx_ctx* c = new (std::nothrow) gsx_ctx();
    if (!c) return GSX_E_NOMEM;
    c->device = device;
    auto fail = [&](int rc) { gsx_destroy(c); return rc; };
    if (hipSetDevice(device) != hipSuccess) return fail(GSX_E_HIP);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(GSX_E_HIP);
    if (hipStreamCreateWithFlags(&c->sort_stream, hipStreamNonBlocking) != hipSuccess) return fail(GSX_E_HIP);
    if (hipStreamCreateWithFlags(&c->parked.fstream, hipStreamNonBlocking) != hipSuccess) return fail(GSX_E_HIP);
    c->fstream = c->stream;              // slot 0 queues its frames on the main stream
    if (hipEventCreateWithFlags(&c->ev_sorted, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_pre, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_s2_stats, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_sample_done, hipEventDisableTiming) != hipSuccess)
        return fail(GSX_E_HIP);
    if (hipHostMalloc(reinterpret_cast<void**>(&c->s2_stats_host), 64, hipHostMallocDefault) != hipSuccess) return fail(GSX_E_NOMEM);
    std::memset(c->s2_stats_host, 0, 64);
    for (int b = 0; b < GSX_PERM_RING; ++b)
        if (hipEventCreateWithFlags(&c->ev_frame_done[b], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_perm[b], hipEventDisableTiming) != hipSuccess)
            return fail(GSX_E_HIP);
    c->sel_stream = c->stream;
    if (gsx_scratch_init(c, device) != GSX_OK) return fail(GSX_E_HIP);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
        c->cus = static_cast<uint32_t>(cus);
        c->sort_slots = gsx_sort_slots(cus);
        c->pair_slots = gsx_pair_slots(cus);
    }
    for (int i = 0; i < EV_COUNT; ++i)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) return fail(GSX_E_HIP);
    for (int i = 0; i < GSX_SAMPLE_RING; ++i) {
        c->samp_pass[i] = -1;
        if (hipEventCreate(&c->samp_ev[i][0]) != hipSuccess || hipEventCreate(&c->samp_ev[i][1]) != hipSuccess) return fail(GSX_E_HIP);
    }
    // per frame slot: control block and aux words in one allocation (aux GSX_AUX_OFFSET bytes in, on cache lines of its own) so
    // that the frame's end needs a single device -> host copy
    static_assert(sizeof(GsxFrameCtl) <= GSX_AUX_OFFSET, "aux words must not share the control block's lines");
    for (GsxFrameSlot* sl : {static_cast<GsxFrameSlot*>(c), &c->parked}) {
        if (hipEventCreateWithFlags(&sl->frame_ev, hipEventDisableTiming) != hipSuccess) return fail(GSX_E_HIP);
        if (hipMalloc(reinterpret_cast<void**>(&sl->ctl), GSX_AUX_OFFSET + 256) != hipSuccess) return fail(GSX_E_NOMEM);
        sl->aux = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(sl->ctl) + GSX_AUX_OFFSET);
        if (hipMalloc(reinterpret_cast<void**>(&sl->live_bits), GSX_MAX_TILES_Y * GSX_LIVE_ROW_WORDS * 4) != hipSuccess) return fail(GSX_E_NOMEM);
        if (hipHostMalloc(reinterpret_cast<void**>(&sl->ctl_host), GSX_AUX_OFFSET + 128, hipHostMallocDefault) != hipSuccess)
            return fail(GSX_E_NOMEM);
        sl->aux_host = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(sl->ctl_host) + GSX_AUX_OFFSET);
    }
    if (hipMalloc(reinterpret_cast<void**>(&c->m_perm), 64) != hipSuccess) return fail(GSX_E_NOMEM);
    if (hipMemset(c->m_perm, 0, 64) != hipSuccess) return fail(GSX_E_HIP);
    if (hipMalloc(reinterpret_cast<void**>(&c->stat), GSX_STAT_BYTES) != hipSuccess) return fail(GSX_E_NOMEM);
    if (hipMalloc(reinterpret_cast<void**>(&c->m_dev), 16) != hipSuccess) return fail(GSX_E_NOMEM);
    if (hipMalloc(reinterpret_cast<void**>(&c->aabb_dev), 32) != hipSuccess) return fail(GSX_E_NOMEM);
    if (hipMalloc(reinterpret_cast<void**>(&c->flags), sizeof(GsxDevFlags)) != hipSuccess) return fail(GSX_E_NOMEM);
    if (hipHostMalloc(reinterpret_cast<void**>(&c->host_word), 64, hipHostMallocDefault) != hipSuccess)
        return fail(GSX_E_NOMEM);
    if (hipHostMalloc(reinterpret_cast<void**>(&c->flags_host), sizeof(GsxDevFlags), hipHostMallocDefault) != hipSuccess)
        return fail(GSX_E_NOMEM);
    if (hipMemset(c->flags, 0, sizeof(GsxDevFlags)) != hipSuccess) return fail(GSX_E_HIP);
    if (hipMemset(c->m_dev, 0, 16) != hipSuccess) return fail(GSX_E_HIP);
    *out = c;
    return GSX_OK;
}

int gsx_destroy(gsx_ctx* c) {
    if (!c) return GSX_E_INVALID;
    (void)hipSetDevice(c->device);
    c->frame_pending = false;
    c->parked.frame_pending = false;
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    gsx_comm_detach_all(c);
    if (c->sort_stream) (void)hipStreamSynchronize(c->sort_stream);
    if (c->parked.fstream) (void)hipStreamSynchronize(c->parked.fstream);
    if (c->fstream) (void)hipStreamSynchronize(c->fstream);
    if (c->active_slot) gsx_slot_switch(c);          // slot 0 (the main stream's) back into the base
    dev_free(&c->pos4); dev_free(&c->pos3); dev_free(&c->geo);
    dev_free(&c->sel_idx); dev_free(&c->m_dev); dev_free(&c->m_perm);
    dev_free(&c->lod_a); dev_free(&c->lod_b); dev_free(&c->lod_parent); dev_free(&c->lod_leaf); dev_free(&c->lod_child); dev_free(&c->lod_queue[0]); dev_free(&c->lod_queue[1]); dev_free(&c->lod_qcount); dev_free(&c->lod_top_anc); dev_free(&c->lod_head_ranges); dev_free(&c->lod_node);
    dev_free(&c->lod_stop); dev_free(&c->lod_emit); dev_free(&c->lod_block_sums); dev_free(&c->lod_agg);
    for (int b = 0; b < 2; ++b) { dev_free(&c->keys[b]); dev_free(&c->vals[b]); }
    for (int b = 0; b < GSX_PERM_RING; ++b) dev_free(&c->perm[b]);
    dev_free(&c->sort_ws); dev_free(&c->aabb_dev); dev_free(&c->flags);
    dev_free(&c->s2_pairs); dev_free(&c->s2_scratch); dev_free(&c->s2_ws);
    if (c->s2_stats_host) { (void)hipHostFree(c->s2_stats_host); c->s2_stats_host = nullptr; }
    if (c->ev_s2_stats) (void)hipEventDestroy(c->ev_s2_stats);
    dev_free(&c->stat);
    for (GsxFrameSlot* sl : {static_cast<GsxFrameSlot*>(c), &c->parked}) {
        for (int b = 0; b < 2; ++b) { dev_free(&sl->pair_key[b]); dev_free(&sl->pair_val[b]); }
        dev_free(&sl->pair_ws); dev_free(&sl->scan_block_sums); dev_free(&sl->fb);
        dev_free(&sl->sub_off); dev_free(&sl->rank_box); dev_free(&sl->rank_rec); dev_free(&sl->ctl); dev_free(&sl->live_bits);
        dev_free(&sl->tile_state); dev_free(&sl->accum);
        dev_free(&sl->tile_depth[0]); dev_free(&sl->tile_depth[1]); dev_free(&sl->tile_band); dev_free(&sl->band_masks);
        if (sl->ctl_host) { (void)hipHostFree(sl->ctl_host); sl->ctl_host = nullptr; }
        if (sl->frame_ev) { (void)hipEventDestroy(sl->frame_ev); sl->frame_ev = nullptr; }
    }
    if (c->parked.fstream) (void)hipStreamDestroy(c->parked.fstream);
    for (int i = 0; i < GSX_SAMPLE_RING; ++i)
        for (int k = 0; k < 2; ++k)
            if (c->samp_ev[i][k]) (void)hipEventDestroy(c->samp_ev[i][k]);
    if (c->host_word) (void)hipHostFree(c->host_word);
    if (c->flags_host) (void)hipHostFree(c->flags_host);
    for (int i = 0; i < EV_COUNT; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (hipEvent_t e : {c->ev_sorted, c->ev_pre, c->ev_sample_done})
        if (e) (void)hipEventDestroy(e);
    for (int b = 0; b < GSX_PERM_RING; ++b) {
        if (c->ev_frame_done[b]) (void)hipEventDestroy(c->ev_frame_done[b]);
        if (c->ev_perm[b]) (void)hipEventDestroy(c->ev_perm[b]);
    }
    for (auto& g : c->gathers)
        if (g.ev) (void)hipEventDestroy(g.ev);
    if (c->ev_comm_pre) (void)hipEventDestroy(c->ev_comm_pre);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    if (c->sort_stream) (void)hipStreamDestroy(c->sort_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return GSX_OK;
}

const char* gsx_last_error(const gsx_ctx* c) { return c ? c->err.c_str() : "null context"; }

int gsx_sync(gsx_ctx* c) {
    if (!c) return GSX_E_INVALID;
    int rc = set_device(c);
    if (rc) return rc;
    return check_flags(c);
}

int gsx_set_option(gsx_ctx* c, const char* name, uint64_t value) {
    if (!c || !name) return GSX_E_INVALID;
    if (!std::strcmp(name, "pair_budget")) {
        if (value > 0x7FFFFFFFull) { c->err = "pair_budget above 2^31-1"; return GSX_E_INVALID; }
        c->opt_pair_budget = value;
    } else if (!std::strcmp(name, "chunk_first")) {
        c->opt_chunk_first = value > 0x7FFFFFFFull ? 0x7FFFFFFFull : value;
    } else if (!std::strcmp(name, "chunk_growth")) {
        if (value == 1 || value > 1024) { c->err = "chunk_growth must be 0 or 2..1024"; return GSX_E_INVALID; }
        c->opt_chunk_growth = value;
    } else if (!std::strcmp(name, "no_history")) {
        c->opt_no_history = value ? 1 : 0;
    } else if (!std::strcmp(name, "lod_walk")) {
        c->opt_lod_walk = value > 2 ? 1 : value;      // 0: head + tail + one-launch compaction; 1: ancestor walk; 2: a launch per level (both with the three-launch compaction)
    } else if (!std::strcmp(name, "tile_caps")) {
        c->opt_tile_caps = value > 2 ? 1 : value;      // (2: caps at 1/8 of the predicted depth - exercises the redo path in tests)
    } else if (!std::strcmp(name, "tile_caps_radius")) {
        if (value > 32) { c->err = "tile_caps_radius above 32"; return GSX_E_INVALID; }
        c->opt_caps_radius = value;
    } else if (!std::strcmp(name, "sort_slots") || !std::strcmp(name, "pair_slots")) {
        // workgroups the depth sort's / the pair sort's scatter kernels keep resident (default: as many as the GPU holds); fewer
        // leave registers and LDS to the kernels of the other streams
        if (value < 8 || value > GSX_RADIX_MAX_BLOCKS) { c->err = "sort_slots / pair_slots must be 8..2048"; return GSX_E_INVALID; }
        int rc = set_device(c);
        if (rc || (rc = gsx_frame_finish(c))) return rc;
        (name[0] == 's' ? c->sort_slots : c->pair_slots) = static_cast<uint32_t>(value);
    } else if (!std::strcmp(name, "sample_whole_every")) {
        if (value < 2 || value > 16) { c->err = "sample_whole_every must be 2..16"; return GSX_E_INVALID; }
        c->opt_sample_whole_every = static_cast<uint32_t>(value);
        c->prof_frame = 0;
    } else if (!std::strcmp(name, "sample_every")) {
        if (value > 1024) { c->err = "sample_every above 1024"; return GSX_E_INVALID; }
        c->opt_sample_every = value ? static_cast<uint32_t>(value) : GSX_SAMPLE_EVERY;
        c->prof_frame = 0;
    } else if (!std::strcmp(name, "sort_algo")) {
        if (value > 2) { c->err = "sort_algo must be 0 (automatic), 1 (LSD passes) or 2 (two-pass whenever its plan allows)"; return GSX_E_INVALID; }
        c->opt_sort_algo = value;
        for (int r = 0; r < 24; ++r) c->s2_backoff[r] = c->s2_backoff_len[r] = 0;
    } else if (!std::strcmp(name, "async_sort")) {
        c->opt_async_sort = value ? 1 : 0;
    } else if (!std::strcmp(name, "sort_ahead")) {
        // 1: gsx_render draws the oldest permutation not drawn yet, so the caller may queue the next frame's sort before this
        // frame (sort k+1, render k, sort k+2, render k+1 ...); setting the option forgets permutations that wait to be drawn
        c->opt_sort_ahead = value ? 1 : 0;
        c->undrawn = 0;
    } else if (!std::strcmp(name, "deferred_readback")) {
        int rc = set_device(c);
        if (rc || (rc = gsx_frame_finish(c))) return rc;     // (frames queued under the old rule are delivered first)
        c->opt_deferred_readback = value ? 1u : 0u;
    } else if (!std::strcmp(name, "frames_in_flight")) {
        if (value < 1 || value > 2) { c->err = "frames_in_flight must be 1 or 2"; return GSX_E_INVALID; }
        int rc = set_device(c);
        if (rc || (rc = gsx_frame_finish(c))) return rc;
        c->opt_frames_in_flight = value;
    } else {
        c->err = std::string("unknown option: ") + name;
        return GSX_E_INVALID;
    }
    return GSX_OK;
}

int gsx_set_profiling(gsx_ctx* c, int enabled) {
    if (!c) return GSX_E_INVALID;
    if (enabled < 0 || enabled > 2) return GSX_E_INVALID;
    c->prof_level = enabled;
    c->profiling = enabled == 1;
    c->prof_frame = 0;                 // (the next sort is a sampled one at level 2)
    return GSX_OK;
}

int gsx_get_scatter_samples(gsx_ctx* c, double sum_ms[4], uint32_t launches[4], int reset) {
    if (!c || !sum_ms || !launches) return GSX_E_INVALID;
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = gsx_frame_finish(c))) return rc;
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    for (uint32_t k = 0; k < GSX_SAMPLE_RING; ++k) gsx_sample_drain(c, (c->samp_pos + k) % GSX_SAMPLE_RING);
    for (int p = 0; p < 4; ++p) {
        sum_ms[p] = c->samp_sum_ms[p];
        launches[p] = c->samp_n[p];
        if (reset) { c->samp_sum_ms[p] = 0.0; c->samp_n[p] = 0; }
    }
    return GSX_OK;
}

int gsx_get_sort_samples(gsx_ctx* c, double* sum_ms, uint32_t* sorts, int reset) {
    if (!c || !sum_ms || !sorts) return GSX_E_INVALID;
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = gsx_frame_finish(c))) return rc;
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    for (uint32_t k = 0; k < GSX_SAMPLE_RING; ++k) gsx_sample_drain(c, (c->samp_pos + k) % GSX_SAMPLE_RING);
    *sum_ms = c->samp_sum_ms[GSX_SAMPLE_WHOLE];
    *sorts = c->samp_n[GSX_SAMPLE_WHOLE];
    if (reset) { c->samp_sum_ms[GSX_SAMPLE_WHOLE] = 0.0; c->samp_n[GSX_SAMPLE_WHOLE] = 0; }
    return GSX_OK;
}

int gsx_get_timings(gsx_ctx* c, gsx_timings* out) {
    if (!c || !out) return GSX_E_INVALID;
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = gsx_frame_finish(c))) return rc;
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    gsx_timings t = c->tim;
    if (c->profiling) {
        if (c->ev_recorded[EV_SEL1]) t.select_ms = ev_ms(c, EV_SEL0, EV_SEL1);
        if (c->ev_recorded[EV_SORT1]) {
            t.keygen_ms = ev_ms(c, EV_KEY0, EV_KEY1);
            t.sort_ms = ev_ms(c, EV_KEY1, EV_SORT1);
            t.scatter_ms = 0.0f;
            for (uint32_t p = 0; p < c->sort_passes && p < 4; ++p) t.scatter_ms += ev_ms(c, EV_X0 + 2 * p, EV_X0 + 2 * p + 1);
        }
        if (c->ev_recorded[EV_RAST1]) {
            t.project_ms = ev_ms(c, EV_PROJ0, EV_PROJ1);
            // the frame loop interleaves binning and rasterizing: sum the two halves of every iteration
            t.bin_ms = 0.0f; t.raster_ms = 0.0f;
            int prev = EV_PROJ1;
            for (uint32_t k = 0; k < c->frame.n_iters && k < GSX_MAX_SCHED; ++k) {
                t.bin_ms += ev_ms(c, prev, EV_IT0 + 2 * k);
                t.raster_ms += ev_ms(c, EV_IT0 + 2 * k, EV_IT0 + 2 * k + 1);
                prev = EV_IT0 + 2 * k + 1;
            }
        }
        t.total_ms = t.select_ms + t.keygen_ms + t.sort_ms + t.project_ms + t.bin_ms + t.raster_ms;
        if (c->ev_recorded[EV_RAST1] && c->stat) {
            std::vector<unsigned long long> st(GSX_STAT_BYTES / 8);
            GSX_HIP(c, hipMemcpy(st.data(), c->stat, GSX_STAT_BYTES, hipMemcpyDeviceToHost));
            t.raster_shades = 0; t.raster_blends = 0;
            t.raster_staged = 0;
            for (int sh = 0; sh < 64; ++sh) { t.raster_shades += st[sh * 16]; t.raster_blends += st[sh * 16 + 1]; t.raster_staged += st[sh * 16 + 2]; }
        }
    }
    t.scatter_sample = 0;
    if (c->prof_level == 2) {
        for (uint32_t k = 0; k < GSX_SAMPLE_RING; ++k) gsx_sample_drain(c, (c->samp_pos + k) % GSX_SAMPLE_RING);   // oldest first
        if (c->sorted_valid && c->samp_last_pass >= 0) {
            t.scatter_ms = c->samp_last_ms;                       // the latest sort's sampled launch
            t.scatter_sample = static_cast<uint32_t>(c->samp_last_pass) + 1u;
        }
    }
    if (c->sorted_valid) {
        t.sort_passes = c->sort_passes;
        t.sort_launches = c->sort_launches;
        t.sort_form = c->sort_two_pass ? 2u : 1u;
        if (c->sort_gated) {                        // the gated top-digit pass: launched (+ its copy), executed only if a key needed it
            uint32_t wide = 0;
            GSX_HIP(c, hipMemcpy(&wide, &c->flags->wide_m, sizeof wide, hipMemcpyDeviceToHost));
            t.sort_launches += 4;                   // its count, scan, scatter and the gated copy
            if (!wide) --t.sort_passes;
        }
    }
    *out = t;
    return GSX_OK;
}

static int upload_common(gsx_ctx* c, const void* aos_dev, uint64_t n) {
    int rc;
    if ((rc = gsx_frame_finish(c))) return rc;
    c->main_dirty = true;
    for (bool& r : c->frame_done_rec) r = false;
    c->hist_valid = false;
    c->parked.hist_valid = false;
    c->sorted_rec = false;
    c->undrawn = 0;
    for (bool& r : c->perm_rec) r = false;
    if ((rc = alloc_scene(c, n))) return rc;
    c->n = n;
    GSX_HIP(c, hipMemsetAsync(c->flags, 0, sizeof(GsxDevFlags), c->stream));
    if ((rc = gsx_launch_upload(c, aos_dev, n))) return rc;
    if ((rc = gsx_launch_aabb(c))) return rc;
    c->lod_kind = 0;
    c->sorted_valid = false;
    if ((rc = gsx_select_all(c))) return rc;
    GSX_HIP(c, hipMemcpyAsync(c->flags_host, c->flags, sizeof(GsxDevFlags), hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    if (c->flags_host->asym_cov) {
        c->err = "splat covariance block (transform 3x3) is not symmetric";
        c->n = 0;
        return GSX_E_UNSUPPORTED;
    }
    return GSX_OK;
}

int gsx_upload_splats(gsx_ctx* c, const void* aos80_host, uint64_t n) {
    if (!c || (!aos80_host && n) || n > 0x7FFFFFF0ull) return GSX_E_INVALID;
    int rc = set_device(c);
    if (rc) return rc;
    void* staging = nullptr;
    if (n) {
        GSX_HIP(c, hipMalloc(&staging, n * 80));
        hipError_t e = hipMemcpyAsync(staging, aos80_host, n * 80, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { (void)hipFree(staging); c->err = hipGetErrorString(e); return GSX_E_HIP; }
    }
    rc = upload_common(c, staging, n);
    (void)hipStreamSynchronize(c->stream);
    if (staging) (void)hipFree(staging);
    return rc;
}

int gsx_upload_splats_dev(gsx_ctx* c, const void* aos80_dev, uint64_t n) {
    if (!c || (!aos80_dev && n) || n > 0x7FFFFFF0ull) return GSX_E_INVALID;
    int rc = set_device(c);
    if (rc) return rc;
    return upload_common(c, aos80_dev, n);
}

int gsx_select_all(gsx_ctx* c) {
    if (!c) return GSX_E_INVALID;
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = gsx_frame_finish(c))) return rc;
    c->main_dirty = true;
    c->sel_iota = true;
    c->m_cap = static_cast<uint32_t>(c->n);
    c->sorted_valid = false;
    const uint32_t m = c->m_cap;
    GSX_HIP(c, hipMemcpyAsync(c->m_dev, &m, 4, hipMemcpyHostToDevice, c->stream));
    GSX_HIP(c, hipStreamSynchronize(c->stream));  // &m is a stack word
    return GSX_OK;
}

int gsx_select_list(gsx_ctx* c, const uint32_t* idx_host, uint32_t m) {
    if (!c || (!idx_host && m)) return GSX_E_INVALID;
    if (m > c->n) { c->err = "selection longer than the scene"; return GSX_E_INVALID; }
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = gsx_frame_finish(c))) return rc;
    c->main_dirty = true;
    if (m) GSX_HIP(c, hipMemcpyAsync(c->vals[0], idx_host, static_cast<uint64_t>(m) * 4, hipMemcpyHostToDevice, c->stream));
    if ((rc = gsx_launch_sort_list(c, m))) return rc;
    c->sel_iota = false;
    c->m_cap = m;
    c->sorted_valid = false;
    return check_flags(c);  // also makes idx_host reusable
}

int gsx_set_lod_octree(gsx_ctx* c, const gsx_octree_flat* t) {
    if (!c || !t || !t->center3 || !t->half3 || !t->parent || !t->is_leaf) return GSX_E_INVALID;
    if (t->n_nodes != c->n) { c->err = "octree: node count must equal splat count (node i owns splat i)"; return GSX_E_INVALID; }
    int rc = set_device(c);
    if (rc) return rc;
    const uint32_t n = t->n_nodes;
    for (uint32_t i = 0; i < n; ++i)
        if (t->parent[i] != 0xFFFFFFFFu && t->parent[i] >= i) { c->err = "octree: parent must precede child (BFS order)"; return GSX_E_INVALID; }
    std::vector<float> a(static_cast<size_t>(n) * 4), b(static_cast<size_t>(n) * 4);
    for (uint32_t i = 0; i < n; ++i) {
        a[4 * i + 0] = t->center3[3 * i + 0]; a[4 * i + 1] = t->center3[3 * i + 1]; a[4 * i + 2] = t->center3[3 * i + 2];
        a[4 * i + 3] = t->half3[3 * i + 0];
        b[4 * i + 0] = t->half3[3 * i + 1]; b[4 * i + 1] = t->half3[3 * i + 2]; b[4 * i + 2] = 0; b[4 * i + 3] = 0;
    }
    if ((rc = dev_alloc(c, &c->lod_a, n)) || (rc = dev_alloc(c, &c->lod_b, n)) || (rc = dev_alloc(c, &c->lod_parent, n)) ||
        (rc = dev_alloc(c, &c->lod_leaf, n)))
        return rc;
    GSX_HIP(c, hipMemcpy(c->lod_a, a.data(), static_cast<uint64_t>(n) * 16, hipMemcpyHostToDevice));
    GSX_HIP(c, hipMemcpy(c->lod_b, b.data(), static_cast<uint64_t>(n) * 16, hipMemcpyHostToDevice));
    GSX_HIP(c, hipMemcpy(c->lod_parent, t->parent, static_cast<uint64_t>(n) * 4, hipMemcpyHostToDevice));
    GSX_HIP(c, hipMemcpy(c->lod_leaf, t->is_leaf, n, hipMemcpyHostToDevice));
    c->lod_kind = 1;
    c->lod_nodes = n;
    // Levels of a BFS-ordered tree are contiguous id ranges and parent[] is non-decreasing: find the ranges.  A tree that is
    // not laid out that way (or deeper than GSX_LOD_MAX_LEVELS) is still cut correctly, by the ancestor-walk kernels.
    c->lod_levels = 0;
    c->lod_top_levels = 0;
    if (n > 0 && t->parent[0] == 0xFFFFFFFFu) {
        uint32_t levels = 1, s = 0, e = 1;
        c->lod_level_start[0] = 0;
        c->lod_level_start[1] = 1;
        bool bfs = true;
        while (e < n && bfs) {
            uint32_t j = e, last = s;
            while (j < n && t->parent[j] != 0xFFFFFFFFu && t->parent[j] >= last && t->parent[j] < e) { last = t->parent[j]; ++j; }
            if (j == e || levels == GSX_LOD_MAX_LEVELS) { bfs = false; break; }
            s = e; e = j;
            c->lod_level_start[++levels] = j;
        }
        // children of a node are contiguous in BFS order: (first child << 4) | count; at most 15 children, 2^28 nodes
        std::vector<uint32_t> cw;
        if (bfs && n < (1u << 28)) {
            cw.assign(n, 0u);
            for (uint32_t i = 1; i < n && bfs; ++i) {
                uint32_t& w = cw[t->parent[i]];
                if ((w & 15u) == 0u) w = i << 4;
                if ((w & 15u) == 15u || (w >> 4) + (w & 15u) != i) bfs = false; else ++w;
            }
            c->lod_max_children = 0;
            for (uint32_t i = 0; i < n; ++i) c->lod_max_children = std::max(c->lod_max_children, cw[i] & 15u);
        } else {
            bfs = false;
        }
        if (bfs) {
            if ((rc = dev_alloc(c, &c->lod_child, n)) || (rc = dev_alloc(c, &c->lod_queue[0], n)) || (rc = dev_alloc(c, &c->lod_queue[1], n)) ||
                (rc = dev_alloc(c, &c->lod_qcount, GSX_LOD_MAX_LEVELS + 2)))
                return rc;
            GSX_HIP(c, hipMemcpy(c->lod_child, cw.data(), static_cast<uint64_t>(n) * 4, hipMemcpyHostToDevice));
            GSX_HIP(c, hipMemset(c->lod_qcount, 0, (GSX_LOD_MAX_LEVELS + 2) * sizeof(uint32_t)));
            c->lod_levels = levels;
            c->lod_top_levels = 1;       // (level 0 is the root alone); at most 8 levels: a top node and its ancestors fill 8 slots
            while (c->lod_top_levels < levels && c->lod_top_levels < 8u && c->lod_level_start[c->lod_top_levels + 1] <= GSX_LOD_TOP_NODES)
                ++c->lod_top_levels;
            const uint32_t top_end = c->lod_level_start[c->lod_top_levels];
            std::vector<uint32_t> anc(static_cast<size_t>(top_end) * 8u, 0xFFFFFFFFu);
            for (uint32_t i = 0; i < top_end; ++i) {
                uint32_t k = 0;
                for (uint32_t p = i; p != 0xFFFFFFFFu && k < 8u; p = t->parent[p]) anc[i * 8u + k++] = p;
            }
            if ((rc = dev_alloc(c, &c->lod_top_anc, static_cast<uint64_t>(top_end) * 8u))) return rc;
            GSX_HIP(c, hipMemcpy(c->lod_top_anc, anc.data(), anc.size() * 4, hipMemcpyHostToDevice));
            {   // the walk kernels' 32-byte node records
                std::vector<uint32_t> rec(static_cast<size_t>(n) * 8u);
                for (uint32_t i = 0; i < n; ++i) {
                    std::memcpy(&rec[8u * i], &a[4u * i], 16);
                    std::memcpy(&rec[8u * i + 4u], &b[4u * i], 8);
                    rec[8u * i + 6u] = cw[i];
                    rec[8u * i + 7u] = t->is_leaf[i] ? 1u : 0u;
                }
                if ((rc = dev_alloc(c, &c->lod_node, static_cast<uint64_t>(n) * 2u))) return rc;
                GSX_HIP(c, hipMemcpy(c->lod_node, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
            }
            c->lod_deepest_leaf = true;
            for (uint32_t i = c->lod_level_start[levels - 1u]; i < n; ++i) c->lod_deepest_leaf = c->lod_deepest_leaf && t->is_leaf[i] != 0;
            // k_octree_head: the descendants one and two levels below each group of 8 top nodes (contiguous ranges in BFS order)
            const uint32_t walk = levels - c->lod_top_levels < 2u ? levels - c->lod_top_levels : 2u;
            const uint32_t last_begin = c->lod_level_start[c->lod_top_levels - 1u], groups = (top_end + 7u) / 8u;
            std::vector<uint32_t> rg(static_cast<size_t>(groups) * 4u, 0u);
            auto children_of = [&](uint32_t lo, uint32_t hi, uint32_t* first, uint32_t* count) {
                uint32_t b = 0, e = 0;
                for (uint32_t i = lo; i < hi; ++i)
                    if (cw[i] & 15u) { if (e == b) b = cw[i] >> 4; e = (cw[i] >> 4) + (cw[i] & 15u); }
                *first = b; *count = e - b;
            };
            for (uint32_t g = 0; g < groups; ++g) {
                const uint32_t lo = std::max(g * 8u, last_begin), hi = std::min(g * 8u + 8u, top_end);
                if (walk >= 1u && lo < hi) children_of(lo, hi, &rg[g * 4u], &rg[g * 4u + 1u]);
                if (walk >= 2u && rg[g * 4u + 1u]) children_of(rg[g * 4u], rg[g * 4u] + rg[g * 4u + 1u], &rg[g * 4u + 2u], &rg[g * 4u + 3u]);
                if (rg[g * 4u + 1u] > 64u || rg[g * 4u + 3u] > 512u) c->lod_max_children = 99u;   // (cannot happen with <= 8 children: the head kernel is not used then)
            }
            if ((rc = dev_alloc(c, &c->lod_head_ranges, groups))) return rc;
            GSX_HIP(c, hipMemcpy(c->lod_head_ranges, rg.data(), rg.size() * 4, hipMemcpyHostToDevice));
        }
    }
    c->lod_stamp = 255;
    return GSX_OK;
}

int gsx_set_lod_hc(gsx_ctx* c, const gsx_hc_flat* f) {
    if (!c || !f || !f->parent || !f->is_leaf || !f->error || !f->weight) return GSX_E_INVALID;
    if (f->n_nodes != c->n) { c->err = "hc: node count must equal splat count (node index == splat index)"; return GSX_E_INVALID; }
    int rc = set_device(c);
    if (rc) return rc;
    const uint32_t n = f->n_nodes;
    for (uint32_t i = 0; i < n; ++i)
        if (f->parent[i] != 0xFFFFFFFFu && f->parent[i] >= n) { c->err = "hc: parent id out of range"; return GSX_E_INVALID; }
    {   // k_emit_flags walks parent chains: a cycle (or parent[i] == i) would never end on the device
        std::vector<uint32_t> seen(n, 0u);           // 0 = unvisited, n + 1 = known to reach a root, else 1 + the walk that found it
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t a = i;
            while (a != 0xFFFFFFFFu && seen[a] == 0u) { seen[a] = i + 1u; a = f->parent[a]; }
            if (a != 0xFFFFFFFFu && seen[a] == i + 1u) { c->err = "hc: parent links form a cycle"; return GSX_E_INVALID; }
            for (a = i; a != 0xFFFFFFFFu && seen[a] == i + 1u; a = f->parent[a]) seen[a] = n + 1u;
        }
    }
    std::vector<float> a(static_cast<size_t>(n) * 4);
    for (uint32_t i = 0; i < n; ++i) { a[4 * i] = f->error[i]; a[4 * i + 1] = f->weight[i]; a[4 * i + 2] = 0; a[4 * i + 3] = 0; }
    if ((rc = dev_alloc(c, &c->lod_a, n)) || (rc = dev_alloc(c, &c->lod_parent, n)) || (rc = dev_alloc(c, &c->lod_leaf, n))) return rc;
    GSX_HIP(c, hipMemcpy(c->lod_a, a.data(), static_cast<uint64_t>(n) * 16, hipMemcpyHostToDevice));
    GSX_HIP(c, hipMemcpy(c->lod_parent, f->parent, static_cast<uint64_t>(n) * 4, hipMemcpyHostToDevice));
    GSX_HIP(c, hipMemcpy(c->lod_leaf, f->is_leaf, n, hipMemcpyHostToDevice));
    c->lod_kind = 2;
    c->lod_nodes = n;
    return GSX_OK;
}

// A camera-dependent cut feeds the next sort and nothing else, so it is queued where that sort runs: on the sort stream,
// beside the frames in flight (they read the size of THEIR selection from a word of their own, gsx_ctx::m_perm).
// Returns the stream in c->sel_stream.  Without the second stream (or with every stage bracketed) the cut waits for the
// frame and goes to the main stream.
static int begin_select(gsx_ctx* c) {
    const bool async = c->opt_async_sort && c->sort_stream && !c->profiling;
    if (!async) {
        c->sel_stream = c->stream;
        const int rc = gsx_frame_finish(c);
        c->main_dirty = true;
        return rc;
    }
    hipStream_t st = c->sel_stream = c->sort_stream;
    if (c->main_dirty) {                 // an upload or an explicit selection queued on the main stream
        GSX_HIP(c, hipEventRecord(c->ev_pre, c->stream));
        c->pre_wait = true;
        c->main_dirty = false;
    }
    if (c->pre_wait) { GSX_HIP(c, hipStreamWaitEvent(st, c->ev_pre, 0)); c->pre_wait = false; }
    return GSX_OK;
}

static int finish_select(gsx_ctx* c, uint32_t* m_out) {
    c->sel_iota = false;
    c->m_cap = c->lod_nodes;
    c->sorted_valid = false;
    c->ev_recorded[EV_SEL1] = c->profiling;
    hipStream_t st = c->sel_stream;
    if (st != c->stream) {               // the main stream joins before it touches the selection (gsx_join_sort)
        GSX_HIP(c, hipEventRecord(c->ev_sorted, st));
        c->sort_join_pending = true;
    }
    c->sel_stream = c->stream;
    if (m_out) {
        GSX_HIP(c, hipMemcpyAsync(c->host_word, c->m_dev, 4, hipMemcpyDeviceToHost, st));
        GSX_HIP(c, hipStreamSynchronize(st));
        *m_out = *c->host_word;
    }
    return GSX_OK;
}

int gsx_select_octree(gsx_ctx* c, const float viewproj[16], float min_screen_area, uint32_t* m_out) {
    if (!c || !viewproj) return GSX_E_INVALID;
    if (c->lod_kind != 1) { c->err = "gsx_select_octree without gsx_set_lod_octree"; return GSX_E_STATE; }
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = begin_select(c))) return rc;
    if ((rc = gsx_launch_select_octree(c, viewproj, min_screen_area))) { c->sel_stream = c->stream; return rc; }
    return finish_select(c, m_out);
}

int gsx_select_hc(gsx_ctx* c, const float cam_pos[3], float threshold, uint32_t* m_out) {
    if (!c || !cam_pos) return GSX_E_INVALID;
    if (c->lod_kind != 2) { c->err = "gsx_select_hc without gsx_set_lod_hc"; return GSX_E_STATE; }
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = begin_select(c))) return rc;
    if ((rc = gsx_launch_select_hc(c, cam_pos, threshold))) { c->sel_stream = c->stream; return rc; }
    return finish_select(c, m_out);
}

int gsx_select_band(gsx_ctx* c, const gsx_uniforms* u, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, uint32_t* m_out) {
    if (!c || !u || width == 0 || height == 0 || y0 >= y1 || y1 > height) return GSX_E_INVALID;
    int rc = set_device(c);
    if (rc) return rc;
    float wt[16];      // view * model, glm's operation order (glm/detail/type_mat4x4.inl:630-648)
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float t = u->viewMatrix[0 * 4 + i] * u->modelMatrix[j * 4 + 0];
            t = t + u->viewMatrix[1 * 4 + i] * u->modelMatrix[j * 4 + 1];
            t = t + u->viewMatrix[2 * 4 + i] * u->modelMatrix[j * 4 + 2];
            t = t + u->viewMatrix[3 * 4 + i] * u->modelMatrix[j * 4 + 3];
            wt[j * 4 + i] = t;
        }
    if ((rc = begin_select(c))) return rc;
    if ((rc = gsx_launch_select_band(c, wt, u->projectionMatrix, u->cutOff * u->splatSize, height, y0, y1))) { c->sel_stream = c->stream; return rc; }
    rc = finish_select(c, m_out);
    c->m_cap = m_out ? *m_out : static_cast<uint32_t>(c->n);      // (finish_select assumes an LOD tree's node count)
    return rc;
}

int gsx_get_selection(gsx_ctx* c, uint32_t* idx_out, uint32_t capacity, uint32_t* m_out) {
    if (!c || !m_out) return GSX_E_INVALID;
    int rc = set_device(c);
    if (rc) return rc;
    gsx_join_sort(c);
    GSX_HIP(c, hipMemcpyAsync(c->host_word, c->m_dev, 4, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    const uint32_t m = *c->host_word;
    *m_out = m;
    if (!idx_out) return GSX_OK;
    if (capacity < m) { c->err = "selection buffer too small"; return GSX_E_INVALID; }
    if (c->sel_iota) {
        for (uint32_t i = 0; i < m; ++i) idx_out[i] = i;
    } else if (m) {
        GSX_HIP(c, hipMemcpy(idx_out, c->sel_idx, static_cast<uint64_t>(m) * 4, hipMemcpyDeviceToHost));
    }
    return GSX_OK;
}

// GSX_TRACE_HOST=1: when the host enters and leaves gsx_sort / gsx_render (CLOCK_MONOTONIC, us - the clock of rocprofv3's
// kernel timestamps; tools/host_vs_gpu.py merges the two)
static const bool g_dbg_host = getenv("GSX_TRACE_HOST") != nullptr;
int gsx_sort(gsx_ctx* c, const float cam_pos[3], uint32_t* idx_out_host) {
    if (!c || !cam_pos) return GSX_E_INVALID;
    struct DbgExit { double t0; ~DbgExit() { if (g_dbg_host) fprintf(stderr, "[host] sort %.1f .. %.1f\n", t0, gsx_dbg_now()); } } dbg_exit{g_dbg_host ? gsx_dbg_now() : 0.0};
    int rc = set_device(c);
    if (rc) return rc;
    // A frame in flight reads perm[perm_cur]; this sort writes the other buffer, so it is queued behind that frame without
    // waiting for it.  Only a frame that still draws from the target buffer (two sorts without a render between) must end first.
    {
        const uint32_t* target = c->perm[(c->perm_cur + 1u) % GSX_PERM_RING];
        if (((c->frame_pending && c->frame.sorted == target) || (c->parked.frame_pending && c->parked.frame.sorted == target)) &&
            (rc = gsx_frame_finish(c)))
            return rc;
    }
    c->cam[0] = cam_pos[0]; c->cam[1] = cam_pos[1]; c->cam[2] = cam_pos[2];
    if ((rc = gsx_launch_sort(c))) return rc;
    c->ev_recorded[EV_SORT1] = c->profiling;
    if (idx_out_host) {
        // the size of THIS permutation: the word the key kernel wrote beside it, read once everything queued has been joined
        // (the selection's own size word may still be written by a cut queued on the sort stream)
        if ((rc = check_flags(c))) return rc;
        GSX_HIP(c, hipMemcpy(c->host_word, c->m_perm + c->perm_cur, 4, hipMemcpyDeviceToHost));
        const uint32_t m = *c->host_word;
        if (m) GSX_HIP(c, hipMemcpy(idx_out_host, c->perm[c->perm_cur], static_cast<uint64_t>(m) * 4, hipMemcpyDeviceToHost));
    }
    return GSX_OK;
}

int gsx_sort_noalloc(gsx_ctx* c, const float cam_pos[3], uint32_t* idx_out_dev, uint32_t capacity) {
    if (!c || !cam_pos || !idx_out_dev) return GSX_E_INVALID;
    if (c->n == 0 && c->m_cap == 0) return GSX_OK;
    if (capacity < c->m_cap) { c->err = "gsx_sort_noalloc: output buffer smaller than the selection"; return GSX_E_INVALID; }
    int rc = set_device(c);
    if (rc) return rc;
    gsx_join_sort(c);
    const float keep[3] = {c->cam[0], c->cam[1], c->cam[2]};   // (the frame loop's schedule looks at the camera of the context's own sorts)
    c->cam[0] = cam_pos[0]; c->cam[1] = cam_pos[1]; c->cam[2] = cam_pos[2];
    rc = gsx_launch_sort_noalloc(c, idx_out_dev);     // queued on the context's stream; no allocation, no readback, no wait
    c->main_dirty = true;                             // (shares the gated-pass counter with the context's own sort: keep them in order)
    c->cam[0] = keep[0]; c->cam[1] = keep[1]; c->cam[2] = keep[2];
    return rc;
}

int gsx_get_sorted_dev(gsx_ctx* c, const uint32_t** idx_dev, uint32_t* m) {
    if (!c || !idx_dev || !m) return GSX_E_INVALID;
    if (!c->sorted_valid) { c->err = "no sorted permutation yet"; return GSX_E_STATE; }
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = check_flags(c))) return rc;
    GSX_HIP(c, hipMemcpy(c->host_word, c->m_perm + c->perm_cur, 4, hipMemcpyDeviceToHost));
    *idx_dev = c->perm[c->perm_cur];
    *m = *c->host_word;
    return GSX_OK;
}

int gsx_get_keys(gsx_ctx* c, const float cam_pos[3], float* dist_out_host, uint32_t capacity) {
    if (!c || !cam_pos || !dist_out_host) return GSX_E_INVALID;
    int rc = set_device(c);
    if (rc) return rc;
    if (capacity < c->m_cap) { c->err = "key buffer too small"; return GSX_E_INVALID; }
    if ((rc = gsx_frame_finish(c))) return rc;
    c->cam[0] = cam_pos[0]; c->cam[1] = cam_pos[1]; c->cam[2] = cam_pos[2];
    c->sorted_valid = false;  // keys[0] doubles as scratch here
    float* dev = reinterpret_cast<float*>(c->keys[0]);
    if ((rc = gsx_launch_keys_only(c, dev))) return rc;
    GSX_HIP(c, hipMemcpyAsync(c->host_word, c->m_dev, 4, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    const uint32_t m = *c->host_word;
    if (m) GSX_HIP(c, hipMemcpy(dist_out_host, dev, static_cast<uint64_t>(m) * 4, hipMemcpyDeviceToHost));
    return GSX_OK;
}

int gsx_render(gsx_ctx* c, const gsx_uniforms* u, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
               void* rgba8_out_host) {
    if (!c || !u || width == 0 || height == 0 || y0 >= y1 || y1 > height) return GSX_E_INVALID;
    if (width > 4096 || height > 4096) { c->err = "frame larger than 4096 x 4096"; return GSX_E_UNSUPPORTED; }
    if (!c->sorted_valid) { c->err = "gsx_render before gsx_sort"; return GSX_E_STATE; }
    int rc = set_device(c);
    if (rc) return rc;
    double dbg_t1 = 0.0;
    struct DbgExit { double t0; double* t1; ~DbgExit() { if (g_dbg_host) fprintf(stderr, "[host] render %.1f wait-done %.1f .. %.1f\n", t0, *t1, gsx_dbg_now()); } } dbg_exit{g_dbg_host ? gsx_dbg_now() : 0.0, &dbg_t1};
    const bool two_slots = c->opt_frames_in_flight >= 2 && !c->profiling && c->parked.fstream;
    if (two_slots) {
        // two frames in flight: this frame takes the slot of the frame before the last one (long finished, as a rule) and is
        // queued beside the last one, which the host does not wait for
        if (c->frame_pending) gsx_slot_switch(c);
        if ((rc = gsx_frame_finish_active(c))) return rc;
    } else if ((rc = gsx_frame_finish(c))) {
        return rc;
    }
    if (g_dbg_host) dbg_t1 = gsx_dbg_now();
    // the permutation this frame draws: the latest one, or - option sort_ahead, two sorts since the last frame - the one before it
    c->perm_draw = (c->opt_sort_ahead && c->undrawn >= 2u) ? (c->perm_cur + GSX_PERM_RING - 1u) % GSX_PERM_RING : c->perm_cur;
    if (c->undrawn) --c->undrawn;
    bool behind_latest = false;          // this frame's stream is behind the LATEST sort (the main stream's pending join is then done)
    if (c->perm_draw != c->perm_cur && c->perm_rec[c->perm_draw]) {
        GSX_HIP(c, hipStreamWaitEvent(c->fstream, c->ev_perm[c->perm_draw], 0));
    } else {
        if (c->sorted_rec) GSX_HIP(c, hipStreamWaitEvent(c->fstream, c->ev_sorted, 0));   // (sorted on another stream, maybe)
        behind_latest = true;
    }
    if ((rc = gsx_behind_sample(c, c->fstream))) return rc;
    if (c->fstream == c->stream && behind_latest) c->sort_join_pending = false;
    if ((rc = gsx_launch_render(c, u, width, height, y0, y1))) return rc;
    c->ev_recorded[EV_RAST1] = c->profiling;
    if (rgba8_out_host && c->opt_deferred_readback && two_slots) {
        // deferred read-back: the frame stays queued; its pixels are copied out when its slot is finished - by the call after
        // the next one, or by gsx_sync - while the frame queued in between runs on the device
        const uint64_t row = static_cast<uint64_t>(width) * 4;
        c->rb_dst = static_cast<uint8_t*>(rgba8_out_host);
        c->rb_off = y0 * row;
        c->rb_bytes = (y1 - y0) * row;
        return GSX_OK;
    }
    if (rgba8_out_host) {
        if ((rc = check_flags(c))) return rc;
        const uint64_t row = static_cast<uint64_t>(width) * 4;
        GSX_HIP(c, hipMemcpy(static_cast<uint8_t*>(rgba8_out_host) + y0 * row, c->fb_cur + y0 * row, (y1 - y0) * row,
                             hipMemcpyDeviceToHost));
    }
    return GSX_OK;
}

#if defined(GSX_DEBUG_COUNT) || defined(GSX_DEBUG_TILETIME)
int gsx_debug_read(gsx_ctx* c, int what, void* out, uint64_t bytes) {
    const void* src = nullptr;
    switch (what) {
        case 2: src = c->pair_val[0]; break;
        case 3: src = c->pair_key[0]; break;
        case 4: src = c->pair_val[1]; break;
        case 5: src = c->pair_key[1]; break;
        case 6: src = c->sub_off; break;
        case 7: src = c->scan_block_sums; break;
        case 8: src = c->ctl; break;
        case 10: src = c->tile_state; break;
        case 11: src = c->live_bits; break;
        case 13: src = c->accum; break;
        default: return GSX_E_INVALID;
    }
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    GSX_HIP(c, hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
    return GSX_OK;
}
int gsx_debug_flags(gsx_ctx* c, uint32_t out[8]) {
    GSX_HIP(c, hipMemcpy(out, c->flags, 32, hipMemcpyDeviceToHost));
    GSX_HIP(c, hipMemset(c->flags, 0, 32));
    return GSX_OK;
}
#endif

int gsx_get_framebuffer_dev(gsx_ctx* c, const void** rgba8_dev, uint32_t* width, uint32_t* height) {
    if (!c || !rgba8_dev) return GSX_E_INVALID;
    if (!c->fb_cur) { c->err = "no frame rendered yet"; return GSX_E_STATE; }
    *rgba8_dev = c->fb_cur;
    if (width) *width = c->fb_w;
    if (height) *height = c->fb_h;
    return GSX_OK;
}

int gsx_set_framebuffer_dev(gsx_ctx* c, void* rgba8_dev, uint64_t capacity_bytes) {
    if (!c || (rgba8_dev && capacity_bytes == 0)) return GSX_E_INVALID;
    int rc = set_device(c);
    if (rc) return rc;
    // (frames already queued keep the target they were started on: every frame slot remembers its own)
    c->fb_ext = static_cast<uint8_t*>(rgba8_dev);
    c->fb_ext_cap = rgba8_dev ? capacity_bytes : 0;
    return GSX_OK;
}

// Page-locked frames for synchronous readers (include/gsx.h).  The set remembers which pointers hipHostMalloc gave out, so a
// block that had to come from malloc is released the right way too.
// (both live until the process ends: a frame released from a static destructor still finds them)
static std::mutex& g_pinned_mu = *new std::mutex;
static std::unordered_set<void*>& g_pinned = *new std::unordered_set<void*>;

void* gsx_host_alloc_pinned(uint64_t bytes) {
    if (bytes == 0) bytes = 1;
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess && p) {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        g_pinned.insert(p);
        return p;
    }
    (void)hipGetLastError();            // (no device, or no page-locked memory left: pageable memory serves)
    return std::malloc(bytes);
}

void gsx_host_free_pinned(void* p) {
    if (!p) return;
    bool pinned;
    {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        pinned = g_pinned.erase(p) != 0;
    }
    if (pinned) (void)hipHostFree(p); else std::free(p);
}

int gsx_get_stream(gsx_ctx* c, void** hip_stream) {
    if (!c || !hip_stream) return GSX_E_INVALID;
    int rc = set_device(c);
    if (rc) return rc;
    // frames may be in flight on the library's other streams: what the caller queues on the main stream from now on runs
    // behind every frame and sort queued so far
    gsx_join_sort(c);
    if ((rc = gsx_comm_join(c))) return rc;
    for (GsxFrameSlot* sl : {static_cast<GsxFrameSlot*>(c), &c->parked})
        if (sl->frame_pending && sl->fstream != c->stream) GSX_HIP(c, hipStreamWaitEvent(c->stream, sl->frame_ev, 0));
    c->stream_handed_out = true;
    *hip_stream = static_cast<void*>(c->stream);
    return GSX_OK;
}

}  // extern "C"
