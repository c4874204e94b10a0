// gsx_internal.h — context layout and kernel launchers shared by the libgsx
// translation units.  gfx950 (MI355X) only; wave = 64 lanes throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/gsx.h"

#define GSX_WAVE 64
#define GSX_SAMPLE_RING 32
#define GSX_PERM_RING 4           // permutation buffers: two frames in flight read two of them, a third may wait to be drawn (option sort_ahead) while the next sort fills the fourth
#define GSX_SAMPLE_WHOLE 4
#define GSX_SAMPLE_EVERY 8u       // profiling level 2 samples every 8th sort by default (gsx_set_option "sample_every")
#define GSX_GATHER_RING 4          // exchanges whose memory ranges are remembered (an older one still running is waited for on the host)
#define GSX_LOD_MAX_LEVELS 48
#define GSX_LOD_TOP_NODES 4096u      // nodes of the leading octree levels one workgroup walks in LDS     // sample slot of "keys + sort as a whole" (slots 0..3: the scatter launch of a radix pass)
#define GSX_STAT_BYTES (64 * 16 * 8)   // rasterizer counters (profiling level 1): 64 shards x one 128-byte line of u64 words
#define GSX_AUX_OFFSET 512   // bytes from the frame control block to the aux words (same allocation)

// ---- sort geometry: one workgroup sorts one tile of NT*IPT keys per pass ----
#ifndef GSX_SORT_NT
#define GSX_SORT_NT 512
#endif
#ifndef GSX_SORT_IPT
#define GSX_SORT_IPT 12
#endif
#ifndef GSX_SORT_WPE
#define GSX_SORT_WPE 4
#endif
#define GSX_SORT_TILE (GSX_SORT_NT * GSX_SORT_IPT)
#define GSX_NOALLOC_MAX_KEYS GSX_SORT_NOALLOC_MAX   // gsx_sort_noalloc: largest selection its 20 MiB scratch array holds
// The binner's pair sort (u16 tile id, u32 payload).  Measured on cfg 2 (A/B in one GPU call, ms per frame): 512 x 12 keys
// per tile with LDS sized for 9-bit digits 0.404; 512 x 8 with registers cut for 5 waves per SIMD 0.394; 512 x 6 at 6 waves 0.400; 512 x 4 at 8 waves 0.396; 256 x 8 0.401.
// (The depth sort's u32 tiles stay at 512 x 12: 512 x 8 takes 48 us per pass against 34.)
#ifndef GSX_PAIR_NT
#define GSX_PAIR_NT 512
#endif
#ifndef GSX_PAIR_IPT
#define GSX_PAIR_IPT 8
#endif
#ifndef GSX_PAIR_MAXD
#define GSX_PAIR_MAXD 512           // digits of a pair-sort pass at most (a frame of up to 512 tiles is sorted in one 9-bit pass)
#endif
#ifndef GSX_PAIR_WPE
#define GSX_PAIR_WPE 5
#endif
#define GSX_PAIR_TILE (GSX_PAIR_NT * GSX_PAIR_IPT)

// ---- raster geometry ----
#define GSX_TILE_W 16
#define GSX_TILE_H 16

#define GSX_MAX_TILES_Y 256          // frame bands up to 4096 rows
#define GSX_LIVE_ROW_WORDS 8         // 256 tiles per row as bits
#define GSX_TBOX_NONE 0x000000FFu    // packed tile box with tx0 > tx1: splat culled
#define GSX_MAX_SCHED 8              // scheduled near -> far iterations per frame
// Per-tile rank caps (gsx_raster.hip): ranks from the near end are cut into bands [R_j, R_{j+1}), R_j a multiple of 1024:
// 0, 1, 2 ... 8 K ranks in steps of 1 K, then four bands per octave.  Band GSX_BANDS - 1 is open ended.
#define GSX_BANDS 64
#define GSX_BAND_INF GSX_BANDS       // cap band of a tile without a cap
#define GSX_CTL_CURSORS (GSX_MAX_SCHED + 1)

// Device-resident control block of the frame loop (gsx_raster.hip).
struct GsxFrameCtl {
    uint32_t cursor[GSX_CTL_CURSORS];  // ranks consumed from the near end before iteration k
    uint32_t n_blocks;     // 1024-rank blocks of the current iteration that fit the pair budget
    uint32_t pairs;        // pairs of the current iteration
    uint32_t live_tiles;   // tiles not yet saturated
    uint32_t m;            // selection size of the frame
    uint32_t total_pairs;  // over all iterations (saturating)
    uint32_t max_pairs;    // largest iteration
    uint32_t iters_used;   // iterations that emitted pairs
    uint32_t stall;        // an iteration whose nearest block alone exceeds the budget
    uint32_t dead_at;      // 1 + cursor at which an iteration first found every tile retired (0 = never)
#ifdef GSX_DEBUG_COUNT
    unsigned long long dbg[4];
#endif
};

// Host-side description of the frame in flight (for gsx_frame_finish).
struct GsxFrameState {
    alignas(16) unsigned char vp_blob[192];
    int mode = 0;
    uint32_t n_tiles = 0;
    uint32_t n_iters = 0;
    uint32_t need_base = 0;   // ranks the schedule was derived from (before the margin)
    const uint32_t* sorted = nullptr;   // the permutation this frame draws (one of gsx_ctx::perm)
    bool continuing = false;
    bool caps_on = false;     // the frame bins every tile only up to its predicted saturation depth (band masks)
    bool redo = false;        // continuation that supplies what the caps withheld from tiles that did not saturate
    uint32_t redos = 0;
};

// Error/diagnostic words written by kernels (never zeroed implicitly).
struct GsxDevFlags {
    uint32_t reserved0;
    uint32_t asym_cov;       // upload saw a non-symmetric covariance block
    uint32_t wide_m;         // key count of the gated top-digit pass of the depth sort: 0, or m once a key needed the top bits
    uint32_t bad_index;      // a selection list named an id >= n
    uint32_t key_bound;      // a depth key fell outside the host-computed [dmin, dmax] bound (NaN position?)
    uint32_t lod_fault;      // 1: the one-launch compaction timed out waiting for a workgroup's count; 2: the octree walk's LDS stack overflowed
    uint32_t pad[2];
};

// Screen-space record of a projected splat: quad corner q in [-1,1]^2 sits at c + q.x*A + q.y*B (pixels);
// m = cutOff * inverse([A B]) maps a pixel offset to the fragment's uv.
struct GsxProj {
    float cx, cy;             // centre in pixels
    float ex, ey;             // half extents of the quad's pixel bbox (|A|+|B| per axis)
    float m00, m01, m10, m11; // u = m00*dx + m01*dy, v = m10*dx + m11*dy
};

// One 64-byte line per splat for everything the projection and the rasterizer read (gathered by sorted rank).
struct GsxGeo {
    float4 pos;     // transform[3]
    float4 covA;    // s00 s01 s02 s11
    float4 covB;    // s12 s22 0 0
    float4 color;   // linear rgb + alpha
};

// Everything one frame in flight owns.  The context holds two: frame k+1 is binned and rasterized on the other slot's stream
// while frame k is still running (the kernels of one frame leave much of the GPU idle: single-workgroup scans, dependent
// launches, the rasterizer's tail), and the host never waits for frame k before it queues frame k+1.  gsx_ctx derives from
// the ACTIVE slot, so the frame loop's code reads c->ctl, c->pair_key ... as before; gsx_slot_switch swaps the two.
#include <ctime>
static inline double gsx_dbg_now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
struct GsxFrameSlot {
    hipStream_t fstream = nullptr;     // the stream this slot's frames are queued on (slot 0: the context's main stream)
    uint64_t frame_seq = 0;            // launch order of the frame in this slot (older frames are finished first)
    uint32_t* scan_block_sums = nullptr;
    uint64_t rank_cap = 0;
    uint16_t* pair_key[2] = {nullptr, nullptr};
    uint32_t* pair_val[2] = {nullptr, nullptr};
    uint64_t pair_cap = 0;
    uint8_t* fb = nullptr;             // RGBA8 width*height (the library's own framebuffer of this slot)
    uint64_t fb_cap = 0;
    uint8_t* fb_cur = nullptr;         // where the slot's queued / last frame was rendered
    uint8_t* rb_dst = nullptr;         // option deferred_readback: host buffer the slot's queued frame is copied into once it is finished
    uint64_t rb_off = 0, rb_bytes = 0;
    uint32_t fb_w = 0, fb_h = 0;
    uint32_t* pair_ws = nullptr;       // radix workspace of the pair sort
    uint64_t pair_ws_words = 0;
    uint64_t pair_alloc = 0;           // allocated pairs per buffer (pair_cap = budget in use <= pair_alloc)
    uint32_t* sub_off = nullptr;       // per count block: 17 in-block pair offsets (64-rank boundaries)
    uint64_t sub_off_cap = 0;
    uint32_t* rank_box = nullptr;      // per rank of the iteration's window: packed tile box (k_chunk_count -> k_chunk_emit)
    float4* rank_rec = nullptr;        // per rank of the window: 48-B record (centre/extents, uv matrix, colour) for k_raster
    uint64_t rank_box_cap = 0, rank_rec_cap = 0;
    GsxFrameCtl* ctl = nullptr;
    GsxFrameCtl* ctl_host = nullptr;   // pinned copy read after the frame
    // aux[0] = cursor of the running iteration (plain data), aux[16] = ranks from the near end consumed by the tiles
    // retired so far (atomicMax), aux[32] = pair count of the running iteration (second 128-byte line: never an atomic).  Kept OFF the control block's cache line: that line is hammered by returnless
    // atomics while tiles retire, and any load from it then queues behind them (measured: raster 3x slower).
    uint32_t* aux = nullptr;
    uint32_t* aux_host = nullptr;      // pinned, 32 words
    uint32_t* live_bits = nullptr;     // GSX_MAX_TILES_Y x GSX_LIVE_ROW_WORDS: tiles still unsaturated
    uint32_t* tile_state = nullptr;    // per tile: 0 untouched, 1 accumulators valid, 2 retired
    uint64_t tile_state_cap = 0;
    float4* accum = nullptr;           // per tile x 256 pixels: (r, g, b, transmittance)
    uint64_t accum_cap = 0;
    GsxFrameState frame;
    // per-tile history: the rank (from the near end) at which each tile retired in the slot's last frame (0 = it did not),
    // written by k_raster into tile_depth[depth_cur]; the next frame of the slot reads it as a prediction
    uint32_t* tile_depth[2] = {nullptr, nullptr};
    uint64_t tile_depth_cap[2] = {0, 0};
    uint32_t depth_cur = 0;
    uint8_t* tile_band = nullptr;      // cap band per tile of the frame in flight (GSX_BAND_INF = none)
    uint64_t tile_band_cap = 0;
    uint32_t* band_masks = nullptr;    // GSX_BANDS x tiles_y x GSX_LIVE_ROW_WORDS: bit = the tile still wants ranks of that band
    uint64_t band_masks_cap = 0;
    bool hist_valid = false;           // tile_depth[depth_cur] describes a finished frame with the geometry below
    uint32_t hist_w = 0, hist_h = 0, hist_y0 = 0, hist_y1 = 0;
    float hist_wt[16] = {}, hist_proj[16] = {};
    float hist_shade[4] = {};          // cutOff, cutOff * splatSize, bayerSize, bayerScale of that frame
    hipEvent_t frame_ev = nullptr;
    bool frame_pending = false;
};

struct gsx_ctx : GsxFrameSlot {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool profiling = false;            // level 1: events around every stage
    int prof_level = 0;                // 0 off, 1 all stages, 2 sampled (one scatter launch, or keys + sort as a whole, in every opt_sample_every-th sort)
    uint32_t prof_frame = 0;           // sorts launched (rotates the sampled pass)
    int sampled_pass = -1;             // level 2: pass whose scatter launch carries events in the last sort
    // level 2 samples: a ring of event pairs, so frames can be queued back to back and the durations collected later
    hipEvent_t samp_ev[GSX_SAMPLE_RING][2] = {};
    int samp_pass[GSX_SAMPLE_RING];    // pass a slot's events bracket, -1 = slot free
    uint32_t samp_pos = 0;
    double samp_sum_ms[GSX_SAMPLE_WHOLE + 1] = {0, 0, 0, 0, 0};
    uint32_t samp_n[GSX_SAMPLE_WHOLE + 1] = {0, 0, 0, 0, 0};
    float samp_last_ms = 0.0f;
    int samp_last_pass = -1;

    // scene (SoA in HBM)
    uint64_t n = 0;
    float4* pos4 = nullptr;    // transform[3] (x, y, z) + a conservative radius in w
    float* pos3 = nullptr;     // x, y, z packed to 12 bytes: what the two-pass sort's key kernel streams
    GsxGeo* geo = nullptr;     // AoS twin for the per-rank gathers of the frame loop (pos4 stays SoA for the key pass)

    // selection
    bool sel_iota = true;
    uint32_t m_cap = 0;        // host upper bound of the selection size
    uint32_t* sel_idx = nullptr;   // ascending ids (capacity sel_cap)
    uint64_t sel_cap = 0;
    uint32_t* m_dev = nullptr;     // device word: selection size

    // LOD structures
    uint32_t lod_kind = 0;     // 0 none, 1 octree, 2 hc
    uint32_t lod_nodes = 0;
    float4* lod_a = nullptr;   // octree: (cx,cy,cz,hx) ; hc: (error, weight, 0, 0)
    float4* lod_b = nullptr;   // octree: (hy,hz,0,0)
    uint32_t* lod_parent = nullptr;
    uint8_t* lod_leaf = nullptr;
    uint8_t* lod_stop = nullptr;   // per-node "emit here if reached" predicate
    uint8_t* lod_emit = nullptr;
    uint32_t* lod_block_sums = nullptr;
    uint64_t lod_cap = 0;
    // octree in BFS order (levels are contiguous id ranges): the cut is evaluated top-down, level by level (gsx_lod.hip)
    uint32_t lod_levels = 0;                            // 0 = not BFS-shaped: the general ancestor-walk kernels are used
    uint32_t lod_top_levels = 0;                        // leading levels small enough for the one-workgroup kernel
    uint32_t lod_level_start[GSX_LOD_MAX_LEVELS + 1] = {};
    uint32_t* lod_child = nullptr;                      // (first child id << 4) | number of children, per node
    uint32_t lod_stamp = 255;                           // frame stamp in the emit bytes (255 = clear them first)
    uint32_t* lod_queue[2] = {nullptr, nullptr};        // per-level worklists (ping-pong), lod_nodes ids each
    uint32_t* lod_qcount = nullptr;                     // queue length per level (zeroed by the compaction's scan kernel after every cut)
    uint4* lod_node = nullptr;                          // 2 x 16 bytes per node: (cx, cy, cz, hx), (hy, hz, child word, leaf)
    uint4* lod_head_ranges = nullptr;                   // per group of 8 top nodes: (first, count) of their descendants one and two levels down
    bool lod_deepest_leaf = false;                      // every node of the deepest level has is_leaf set
    uint32_t lod_max_children = 0;                      // of any node (the in-LDS walk is sized for <= 8)
    uint64_t* lod_agg = nullptr;                        // one-launch compaction: (cut number << 32 | count) per workgroup
    uint32_t lod_seq = 0;                               // cut number of the last compaction
    uint32_t* lod_top_anc = nullptr;                    // 8 ids per top node: itself and its ancestors (UINT32_MAX = none)

    // sort
    uint32_t* keys[2] = {nullptr, nullptr};
    uint32_t* vals[2] = {nullptr, nullptr};
    uint64_t sort_cap = 0;
    uint32_t* sort_ws = nullptr;       // radix count matrix + digit totals
    uint64_t sort_ws_words = 0;
    uint32_t* host_word = nullptr;     // pinned staging word(s) for small device->host reads
    bool sorted_valid = false;
    // The sort's last kernel writes into perm[(perm_cur + 1) % GSX_PERM_RING] and perm_cur then moves on: the frames still in
    // flight (and any continuation iterations gsx_frame_finish queues for them later) keep reading their own buffers, so the
    // next frame's sort can be queued without waiting for them.
    uint32_t* perm[GSX_PERM_RING] = {};
    uint32_t perm_cur = 0;             // perm[perm_cur] holds the latest permutation; the next sort writes perm[(perm_cur + 1) % GSX_PERM_RING]
    uint32_t* m_perm = nullptr;        // GSX_PERM_RING device words: size of the selection each permutation was sorted from
    // two-pass depth sort (gsx_sort2.hpp): pair array, its scratch twin (oversized buckets only), count / prefix matrices
    uint2* s2_pairs = nullptr;
    uint2* s2_scratch = nullptr;
    unsigned char* s2_ws = nullptr;
    uint64_t s2_ws_bytes = 0;
    struct Gsx2Stats* s2_stats_host = nullptr;   // pinned: bucket statistics of the latest two-pass sort (copied behind it)
    hipEvent_t ev_s2_stats = nullptr;
    bool s2_stats_rec = false;
    uint32_t s2_stats_tick = 0;
    // sorts left on the LSD passes after a sort that sent many keys through k2_big / length of the next such stretch - kept per
    // bucket width r0 of the plan: a pose whose key range forces coarse buckets says nothing about the poses that get fine ones
    // (cfg 4 deals 64 different views in turn)
    uint32_t s2_backoff[24] = {}, s2_backoff_len[24] = {};
    uint32_t s2_stats_r0 = 0;          // r0 of the sort whose statistics are on their way
    uint32_t s2_mid_left = 0;          // > 0: the sorts launch k2_local's third instance (buckets of 9217 .. 16384 pairs in LDS); re-armed by the statistics
    bool s2_stats_mid = false;         // whether the sort whose statistics are on their way ran with it
    uint32_t s2_hint_scale = 1;
    uint32_t s2_m_hint = 0, s2_hint_cap = 0;   // selection size seen by a recent two-pass sort (x 1.5) and the capacity it was seen under: sizes the bucket count
    uint64_t opt_sort_algo = 0;                    // 0 automatic, 1 LSD passes only, 2 two-pass whenever its plan allows (any size)
    bool sort_two_pass = false;                    // the latest depth sort ran the two-pass form
    uint32_t sort_launches = 0;
    uint32_t sort_passes = 0;
    bool sort_gated = false;   // the latest depth sort queued a gated top-digit pass (counted in sort_passes)
    uint32_t cus = 256;                              // compute units of the device
    uint32_t sort_slots = 1024, pair_slots = 1024;   // resident workgroups of the scatter kernels (CUs x blocks/CU)
    float cam[3] = {0, 0, 0};
    float aabb_min[3] = {0, 0, 0}, aabb_max[3] = {0, 0, 0};  // of the uploaded positions
    bool aabb_valid = false;
    float* aabb_dev = nullptr;         // 6 floats (ordered-int encoded during the reduction)

    GsxDevFlags* flags = nullptr;
    GsxDevFlags* flags_host = nullptr; // pinned

    // raster: the per-frame state lives in the GsxFrameSlot base (the ACTIVE slot); `parked` is the other slot
    GsxFrameSlot parked;
    uint32_t active_slot = 0;          // which physical slot the base members currently hold
    uint64_t frame_counter = 0;
    uint8_t* fb_ext = nullptr;         // caller-owned render target of the frames launched from now on (gsx_set_framebuffer_dev), nullptr = the slot's own
    uint64_t fb_ext_cap = 0;
    uint64_t pair_grown = 0;           // budget a frame had to grow to (kept for later frames)
    uint32_t caps_off_frames = 0, caps_backoff = 0;   // rank caps: frames left without them after a failed prediction / length of the next such pause
    unsigned long long* stat = nullptr;   // GSX_STAT_BYTES of sharded counters: [shard*16 + 0] (wave, splat) shades, [+1] pixels blended
    // ranks (from the near end) earlier frames needed before every tile was retired; ~0u = the whole selection
    uint32_t need_hist[8] = {};
    uint32_t need_pos = 0;             // frames recorded so far (ring position = need_pos % 8)
    uint64_t rest_pairs = ~0ull;       // (tile, splat) pairs the last frame emitted behind its first window (~0 = unknown)
    float last_cam[3] = {0, 0, 0};     // camera of the last rendered frame (history is only trusted for a similar pose)
    bool have_last_cam = false;
    bool rest_merged = false;          // the last schedule took everything behind the first window in one iteration
    float density0 = 0.0f;             // (tile, splat) pairs per rank in the last frame's first window (0 = unknown)
    // tuning (gsx_set_option); 0 = automatic
    uint64_t opt_pair_budget = 0, opt_chunk_first = 0, opt_chunk_growth = 0, opt_no_history = 0, opt_lod_walk = 0, opt_tile_caps = 1, opt_caps_radius = 0;
    // The depth sort of frame k+1 runs on a stream of its own, beside the binning / rasterizing of frame k (it writes the
    // OTHER permutation buffer and its own key / value arrays).  Ordering is by events:
    //   sort stream waits for   ev_pre         (main stream, recorded when a selection / upload was queued since the last sort)
    //                           ev_frame_done[b] (main stream: the last frame that read permutation buffer b is complete)
    //   main stream waits for   ev_sorted      (sort stream: end of the latest sort) before anything that reads the permutation
    //                                           or rewrites what the sort reads (gsx_frame_finish does the join)
    uint64_t opt_async_sort = 1;
    hipStream_t sort_stream = nullptr;
    hipStream_t sel_stream = nullptr;     // where the selection kernels of the current call are queued (main or sort stream)
    hipEvent_t ev_sorted = nullptr, ev_pre = nullptr, ev_frame_done[GSX_PERM_RING] = {};
    bool frame_done_rec[GSX_PERM_RING] = {};
    bool sorted_rec = false;              // ev_sorted has been recorded (behind the latest sort, on whichever stream it ran)
    // Option sort_ahead: the caller queues the sort of frame k+1 BEFORE it queues frame k, so that the sort is on the device while
    // the host is still launching frame k's kernels (in call order sort k, render k, sort k+1 the launches of render k - some
    // thirty of them - stand between the two sorts, and the device waits for the host).  gsx_render then draws the OLDEST
    // permutation not drawn yet (at most two wait), each behind the event of its own sort.
    uint64_t opt_sort_ahead = 0;
    hipEvent_t ev_sample_done = nullptr;  // end of the latest sampled sort (profiling level 2): it is timed alone on the device
    bool sample_pending = false;
    uint32_t undrawn = 0;                 // permutations sorted since the last gsx_render (<= 1 without sort_ahead, <= 2 with it)
    uint32_t perm_draw = 0;               // ring slot the frame being queued draws from
    hipEvent_t ev_perm[GSX_PERM_RING] = {};   // end of the sort that filled ring slot b (recorded with sort_ahead)
    bool perm_rec[GSX_PERM_RING] = {};
    uint64_t opt_frames_in_flight = 2;
    uint32_t opt_deferred_readback = 0;             // gsx_render(..., host buffer) queues the frame and returns; the pixels arrive when the slot is finished
    uint32_t opt_sample_whole_every = 2;            // ... and every how many SAMPLES one brackets keys + sort as a whole instead of its launches
    uint32_t opt_sample_every = GSX_SAMPLE_EVERY;   // profiling level 2: every how many sorts one is a sampled (synchronous) one
    bool sort_join_pending = false;      // a sort was queued on sort_stream and the main stream has not waited for it yet
    bool main_dirty = true;              // the main stream holds work the next sort must see (selection, upload): record ev_pre
    bool pre_wait = false;               // ev_pre is recorded and the sort stream has not waited for it yet
    // The exchange (gsx_gather) runs on a stream of its own, beside the frames queued after it.  Every exchange leaves a record
    // of the memory it reads and writes; a frame that is about to render into such memory makes its stream wait for the record's
    // event (gsx_comm_guard), and everything that joins the context's streams (gsx_sync, gsx_get_stream ...) joins this one too.
    hipStream_t comm_stream = nullptr;   // created by the first gsx_gather
    hipEvent_t ev_comm_pre = nullptr;
    bool stream_handed_out = false;      // gsx_get_stream was called since the last exchange: the caller may have queued work of its own there
    struct GatherRec {
        const uint8_t* lo[2] = {nullptr, nullptr};    // [0] what is sent, [1] (root) what is received into
        const uint8_t* hi[2] = {nullptr, nullptr};
        hipEvent_t ev = nullptr;
        bool pending = false;            // recorded and not yet known to have completed
    } gathers[GSX_GATHER_RING];
    uint32_t gather_pos = 0;
    struct gsx_comm* comms[8] = {};       // live communicators of this context (gsx_comm_init files them, gsx_destroy detaches them)

    // timing
    hipEvent_t ev[40] = {};
    bool ev_recorded[40] = {};
    gsx_timings tim = {};
};

enum GsxEvent {
    EV_SEL0 = 0, EV_SEL1, EV_KEY0, EV_KEY1, EV_SORT1, EV_PROJ0, EV_PROJ1, EV_BIN1, EV_RAST1,
    EV_X0, /* 4 passes x (start, stop) */ EV_X_END = EV_X0 + 8,
    EV_IT0 = EV_X_END, /* per scheduled iteration: (binned, rastered) */ EV_IT_END = EV_IT0 + 2 * GSX_MAX_SCHED, EV_COUNT = EV_IT_END
};

#define GSX_HIP(ctx, call)                                                        \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) {                                                   \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);       \
            return (e_ == hipErrorOutOfMemory) ? GSX_E_NOMEM : GSX_E_HIP;         \
        }                                                                         \
    } while (0)

// ---- exactly rounded f32 arithmetic for the bit-exact paths (keys, LOD predicates) ----
// HIP's __fmul_rn/__fadd_rn/__fsqrt_rn are NOT usable for this: unless OCML_BASIC_ROUNDED_OPERATIONS
// is defined they expand to plain operators (open to FMA contraction) and to the approximate
// native sqrt.  These helpers switch contraction off per function; sqrtf and '/' are correctly
// rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt.  gsx_sort.hip and
// gsx_lod.hip are additionally compiled with -ffp-contract=off.
#if defined(__HIPCC__)
__device__ __forceinline__ float gsx_mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float gsx_add(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float gsx_sub(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ float gsx_div(float a, float b) { return a / b; }
__device__ __forceinline__ float gsx_sqrt(float a) { return sqrtf(a); }
// glm::distance(position, cameraPos) = sqrt((dx*dx + dy*dy) + dz*dz), d = cameraPos - position (glm/detail/func_geometric.inl:8-23,47-55)
__device__ __forceinline__ float gsx_distance(float4 p, float cx, float cy, float cz) {
    const float dx = gsx_sub(cx, p.x), dy = gsx_sub(cy, p.y), dz = gsx_sub(cz, p.z);
    const float d2 = gsx_add(gsx_add(gsx_mul(dx, dx), gsx_mul(dy, dy)), gsx_mul(dz, dz));
    return gsx_sqrt(d2);
}
__device__ __forceinline__ uint32_t gsx_key_from_distance(float d) { return (~__float_as_uint(d)) & 0x7FFFFFFFu; }
#endif

static inline uint32_t gsx_div_up(uint64_t a, uint64_t b) { return static_cast<uint32_t>((a + b - 1) / b); }

// ---- launchers (defined in the .hip files) ----
int gsx_launch_upload(gsx_ctx* c, const void* aos80_dev, uint64_t n);
int gsx_launch_sort(gsx_ctx* c);                 // keygen + plan + radix passes on the current selection
int gsx_scratch_init(gsx_ctx* c, int device);    // looks up the per-device address of the no-alloc scratch array (gsx_create)
int gsx_launch_sort_noalloc(gsx_ctx* c, uint32_t* idx_out_dev);   // the same sort inside the file-scope scratch array
int gsx_launch_keys_only(gsx_ctx* c, float* dist_dev);
int gsx_launch_select_octree(gsx_ctx* c, const float vp[16], float thr);
int gsx_launch_select_hc(gsx_ctx* c, const float cam[3], float thr);
int gsx_launch_select_band(gsx_ctx* c, const float wt[16], const float proj[16], float cut_s, uint32_t height, uint32_t y0, uint32_t y1);
int gsx_launch_sort_list(gsx_ctx* c, uint32_t m); // ascending re-order of an explicit index list
int gsx_launch_render(gsx_ctx* c, const gsx_uniforms* u, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1);
void gsx_sample_drain(gsx_ctx* c, uint32_t slot);   // folds a finished level-2 sample into samp_sum_ms / samp_n
int gsx_behind_sample(gsx_ctx* c, hipStream_t st);                 // work queued on st next starts behind a sampled sort that is still running
void gsx_join_sort(gsx_ctx* c);                  // main stream waits for the sort stream's latest sort (no host wait)
int gsx_frame_finish(gsx_ctx* c);                // waits for every queued frame (oldest first); runs extra iterations if ranks are left
int gsx_frame_finish_active(gsx_ctx* c);         // the same for the active slot's frame only
int gsx_frame_finish_touching(gsx_ctx* c, const uint8_t* lo, const uint8_t* hi);   // only the queued frames that render into [lo, hi)
int gsx_comm_guard(gsx_ctx* c, hipStream_t st, const uint8_t* lo, const uint8_t* hi);   // st waits for the exchanges that still use [lo, hi)
void gsx_comm_detach_all(gsx_ctx* c);           // gsx_destroy: drains and releases the communicators that are still alive
int gsx_comm_join(gsx_ctx* c);                   // main stream waits for every exchange queued so far (no host wait)
void gsx_slot_switch(gsx_ctx* c);                // the other frame slot becomes the active one
