/*
 * gsx.h — C ABI of libgsx.so: the MI355X (gfx950) implementation of the
 * per-frame Gaussian-splat hot path of AllIAskOfYou/gaussian-splatting
 *   LOD select -> depth keys -> view-ordered index sort -> projection ->
 *   tile-binned alpha-blended rasterization.
 *
 * This is the drop-in boundary.  The reference has no FFI for this path: it is
 * a set of C++ classes (SplatMesh / SplatMeshOctree / SplatMeshHC /
 * SplatMeshGridHC, renderer.hpp).  The host-side C++ facade under
 * gaussian-splatting_amd/host/ keeps those class interfaces and calls ONLY the
 * entry points declared here; each declaration cites the reference code it
 * replaces (paths relative to the reference checkout).
 *
 * Conventions: every function returns GSX_OK (0) or a negative gsx_status; no
 * exception crosses the ABI; the caller owns every host pointer it passes; the
 * context owns all device memory; one context per GPU; calls on a context are
 * serialized on its HIP stream (a context is thread-compatible, not
 * thread-safe).  Pointers named *_host are host memory, *_dev are device
 * memory on the context's GPU.  Matrices are column-major float[16] (glm).
 */
#ifndef GSX_H
#define GSX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct gsx_ctx gsx_ctx;

typedef enum gsx_status {
    GSX_OK = 0,
    GSX_E_INVALID = -1,     /* bad argument */
    GSX_E_HIP = -2,         /* HIP runtime error (see gsx_last_error) */
    GSX_E_NOMEM = -3,       /* device allocation failed */
    GSX_E_STATE = -4,       /* call out of order (e.g. sort before upload) */
    GSX_E_UNSUPPORTED = -5, /* input outside the supported domain */
    GSX_E_INTERNAL = -6     /* device-side invariant violated (bounded spin expired, overflow) */
} gsx_status;

/* main.cpp:48-58 / shader_quads_ordered.wgsl:1-10 — the 224-byte uniform block. */
typedef struct gsx_uniforms {
    float projectionMatrix[16];
    float viewMatrix[16];
    float modelMatrix[16];
    float splatSize;
    float cutOff;
    float fov;
    uint32_t bayerSize;
    float bayerScale;
    float padding[3];
} gsx_uniforms;

/* Octree (Octree.hpp:57-173) flattened in the BFS order Octree::generate walks
 * (Octree.cpp:149-187): node i owns splat i.  center3/half3 = bb.center(),
 * bb.size()/2 (BB.hpp:25-33); parent[0] = UINT32_MAX. */
typedef struct gsx_octree_flat {
    uint32_t n_nodes;
    const float* center3;     /* n_nodes x 3 */
    const float* half3;       /* n_nodes x 3 */
    const uint32_t* parent;   /* n_nodes */
    const uint8_t* is_leaf;   /* n_nodes, Node::is_leaf() (Octree.hpp:67-69) */
} gsx_octree_flat;

/* HC forest (HC.hpp:28-46) flattened by node->index (== position in HC::splats,
 * or in GridHC::splats with the per-cell offset of GridHC.cpp:79-88 applied).
 * parent = UINT32_MAX for the roots left in HC::nodes. weight = Splat::weight()
 * (Splat.h:24-28), error = Node::error. */
typedef struct gsx_hc_flat {
    uint32_t n_nodes;
    const uint32_t* parent;
    const uint8_t* is_leaf;
    const float* error;
    const float* weight;
} gsx_hc_flat;

/* Device-side timings of the last frame, milliseconds (HIP events on the
 * context's stream; valid after gsx_sync). */
typedef struct gsx_timings {
    float select_ms;   /* LOD cut kernels */
    float keygen_ms;   /* depth keys + digit histograms */
    float sort_ms;     /* radix passes only (count + scan + scatter kernels and the gaps between them) */
    float scatter_ms;  /* inside sort_ms: the k_radix_scatter launches alone (sort_passes launches) */
    float project_ms;
    float bin_ms;
    float raster_ms;
    float total_ms;    /* first to last recorded event of the frame */
    uint32_t sort_passes;   /* radix passes actually executed (digits the frame's key range does not reach are never
                               planned; the gated top-digit pass counts only when a key switched it on) */
    uint32_t sort_launches; /* kernel launches inside sort_ms */
    uint64_t tile_pairs;    /* (tile, splat) pairs the binner emitted (after occlusion culling) */
    uint32_t render_iters;  /* near -> far iterations of the frame that emitted pairs */
    uint32_t scatter_sample; /* profiling level 2: 1 + index of the radix pass whose scatter launch scatter_ms times; else 0 */
    uint64_t raster_shades;  /* profiling level 1: (wave, splat) visits of the rasterizer's shading loop in the last frame */
    uint64_t raster_blends;  /* profiling level 1: pixels blended (a visit blends up to 128 pixels) */
    uint64_t raster_staged;  /* profiling level 1: (tile, splat) pairs the rasterizer staged before their tile was saturated (of tile_pairs binned) */
    uint32_t render_capped;  /* 1 if the last frame binned every tile only up to its predicted saturation depth (option "tile_caps") */
    uint32_t render_redos;   /* continuations that supplied pairs a cap had withheld from a tile that did not saturate (0 as a rule) */
    uint32_t sort_form;      /* 1 = stable LSD radix passes (count, scan, scatter per digit); 2 = the two-pass form: keys + top-digit counts,
                                column scan, partition, in-LDS bucket sort (sort_passes = 2, scatter_ms = partition + bucket sort) */
    uint32_t reserved0;
} gsx_timings;

/* ---- lifecycle (replaces Renderer::init device bring-up, renderer.cpp:3-100) ---- */
int gsx_create(int device, gsx_ctx** out);
int gsx_destroy(gsx_ctx* ctx);
const char* gsx_last_error(const gsx_ctx* ctx);
int gsx_sync(gsx_ctx* ctx);
/* 0 = no events (default).  1 = HIP events around every stage and every radix scatter launch
 * (gsx_get_timings fills all *_ms fields; the ~20 events cost tens of microseconds per frame).
 * 2 = sampled: every 8th sort (option "sample_every") carries events, in turn (alternately; option "sample_whole_every") (a) one pair around every HBM pass of the
 * sort - the k_radix_scatter launches of the LSD form, or the partition kernel and the bucket-sort kernels of the two-pass form
 * (gsx_timings.sort_form == 2: "pass" 0 and 1) - and (b) one pair around keys + sort as a whole (scatter_ms = the latest
 * sampled launch, scatter_sample says which pass).  A sampled sort runs on the main stream behind every queued frame, and what
 * is queued on the context's other streams while it runs starts behind it, so the brackets time the kernels themselves, not
 * their share of a GPU they split with the rasterizer.  The samples go through a ring of event pairs, so frames can be queued
 * back to back without reading anything in between; gsx_get_scatter_samples / gsx_get_sort_samples return what has accumulated. */
int gsx_set_profiling(gsx_ctx* ctx, int enabled);
int gsx_get_timings(gsx_ctx* ctx, gsx_timings* out);
/* Profiling level 2 (no counterpart in the reference): per radix pass p < 4 of the depth sort, the summed
 * duration (ms) and the number of sampled k_radix_scatter launches since the last reset.  Waits for the
 * queued work like gsx_get_timings; reset != 0 clears the sums. */
int gsx_get_scatter_samples(gsx_ctx* ctx, double sum_ms[4], uint32_t launches[4], int reset);
/* Profiling level 2 also brackets keys + sort AS A WHOLE (from before the key kernel to behind the last pass) in
 * every second SAMPLED sort: summed duration (ms) and number of such sorts since the last reset. */
int gsx_get_sort_samples(gsx_ctx* ctx, double* sum_ms, uint32_t* sorts, int reset);
/* Tuning knobs of the frame loop (no counterpart in the reference; 0 restores the default):
 *   "pair_budget"   (tile, splat) pairs one near -> far iteration may emit (sizes the pair buffers,
 *                   12 bytes per pair); the device cuts iterations to fit and a frame that needs
 *                   more simply runs more iterations
 *   "chunk_first"   ranks of the first (nearest) iteration;  "chunk_growth"  window growth factor (setting it also
 *                   switches off the merged schedule: with a steady camera and few pairs behind the first window in the
 *                   last frame, the default takes everything behind that window in one iteration)
 *   "no_history"    1 = always pre-queue iterations for the whole selection (default: only as many ranks as
 *                   the last 8 frames needed before every tile was saturated, plus 12.5 % and a quarter of that depth again
 *                   (2 K .. 32 K ranks); the rest runs on demand)
 *   "sort_algo"     form of the depth sort: 0 = automatic (default: two passes over HBM -- a partition on the top <= 13 key bits, then
 *                   every bucket finished inside LDS -- for selections of 2^17 .. ~2.4e7 splats whose key range is wide enough;
 *                   the three 9-bit LSD radix passes otherwise, and for some frames after a sort that found most keys in
 *                   buckets too dense for LDS), 1 = LSD passes only, 2 = two-pass whenever its plan allows, at any size.
 *                   Same permutation either way (distance descending, index ascending).
 *   "async_sort"    0 = queue LOD cuts and depth sorts on the main stream, behind the frame in flight (default 1: on a
 *                   stream of their own, beside it; same results)
 *   "sort_ahead"    1 = gsx_render draws the OLDEST permutation that has not been drawn yet (at most two wait; a third sort
 *                   before any frame drops the oldest), so a viewer can call sort(k+1), render(k), sort(k+2), render(k+1) ...:
 *                   the next frame's sort reaches the device before the host has worked through this frame's ~30 launches, and
 *                   every frame still draws the permutation sorted for its own camera.  Default 0: gsx_render draws the latest
 *                   permutation (the reference's sortSplats-then-draw order, SplatMesh.h:84-114 + main.cpp:359-368).  Setting
 *                   the option, to either value, forgets permutations that wait to be drawn (the latest one stays current).
 *   "frames_in_flight"  1 = wait for frame k before queuing frame k + 1 (default 2: frame k + 1 is binned and rasterized on a
 *                   second frame slot while frame k is still running; same pixels)
 *   "deferred_readback"  1 = gsx_render with a host buffer queues the frame and returns at once; the pixels of call k are in its
 *                   buffer when call k + 2 returns (its slot is taken over then), or after gsx_sync, and the buffer must stay
 *                   valid until then (page-locked memory, gsx_host_alloc_pinned, keeps the copy a plain DMA).  A viewer that
 *                   can show a frame two calls late keeps both frame slots busy although every frame is read back.  Default 0:
 *                   gsx_render returns with the pixels.  Needs frames_in_flight 2 and profiling off or sampled; otherwise the
 *                   call is the synchronous one.  Setting the option delivers what is queued first.
 *   "tile_caps"     0 = bin every live tile through the whole window (default 1: a tile is binned only up to 1.5 x the depth
 *                   at which it saturated in the frame slot's last frame, when the picture has moved by at most a few tiles
 *                   since - exactly up to that depth when view and shading uniforms are unchanged; a tile that does not saturate after
 *                   all gets the withheld pairs in a continuation; same pixels)
 *   "tile_caps_radius"  largest neighbourhood radius (in tiles, default 4, at most 32) up to which a moving view still uses the
 *                   rank caps; the radius needed follows from how far the scene moved on screen since the slot's last frame
 *   "sample_every"  profiling level 2: every how many sorts one is a sampled one (default 8; a sampled sort runs on the main
 *                   stream behind every queued frame, and what is queued on the other streams while it runs starts behind it,
 *                   so that its events time the kernels themselves - which costs the pipeline one bubble per sample,
 *                   about one frame time: bench.py reports its frame time with and without them)
 *   "sample_whole_every"  profiling level 2: every how many SAMPLED sorts one brackets keys + sort as a whole instead of
 *                   its launches one by one (2..16, default 2 = alternately).  Setting either option restarts the count: the
 *                   next sort is a sampled one.
 *   "sort_slots", "pair_slots"  workgroups the scatter kernels of the depth sort / of the binner's pair sort keep resident
 *                   (8 .. 2048; default: as many as the GPU holds, 512 on MI355X); fewer leave registers and LDS to the
 *                   kernels of the context's other streams (measured: no gain)
 *   "lod_walk"      how an octree laid out in BFS order is cut.  0 (default): top-down in three launches (leading levels,
 *                   everything below from per-workgroup stacks in LDS, one-launch compaction).  2: top-down with one launch
 *                   per level and a three-launch compaction (round 2's form).  1: the general kernels (every node tests
 *                   itself, then walks its ancestor chain), which are also the path for trees in any other order.  Same
 *                   result under all three; 1 and 2 are kept for A/B measurements.  (1 and 2 also send HC cuts, list and
 *                   band selections through the three-launch compaction.)
 * Unknown name -> GSX_E_INVALID. */
int gsx_set_option(gsx_ctx* ctx, const char* name, uint64_t value);

/* ---- scene upload: SplatMesh::initializeSplatBuffer (SplatMesh.h:139-147).
 * aos80 = n reference `Splat` records (Splat.h:19-29: mat4 transform, vec4
 * color).  Repacked once into SoA float4 arrays in HBM.  The 3x3 covariance
 * block must be symmetric (it is for split_to_splat/merge outputs), else
 * GSX_E_UNSUPPORTED. ---- */
int gsx_upload_splats(gsx_ctx* ctx, const void* aos80_host, uint64_t n);
int gsx_upload_splats_dev(gsx_ctx* ctx, const void* aos80_dev, uint64_t n);

/* ---- LOD selection (the `indices` a SplatMesh* subclass sorts) ---- */
/* Base class: indices = iota (SplatMesh.h:68-69). */
int gsx_select_all(gsx_ctx* ctx);
/* Explicit list, e.g. HC::get_indices_depth / GridHC::get_indices_error results
 * (HC.cpp:134-168, GridHC.cpp:76-92), which are view independent and cached by
 * the host facade.  The list is re-ordered ascending on device so that ties in
 * the sort break by index. */
int gsx_select_list(gsx_ctx* ctx, const uint32_t* idx_host, uint32_t m);
int gsx_set_lod_octree(gsx_ctx* ctx, const gsx_octree_flat* tree);
int gsx_set_lod_hc(gsx_ctx* ctx, const gsx_hc_flat* forest);
/* Octree::get_indices(camera, min_screen_area) (Octree.cpp:189-216) with
 * viewproj = proj * view (BB.hpp:60-61). m_out may be NULL (no host sync). */
int gsx_select_octree(gsx_ctx* ctx, const float viewproj[16], float min_screen_area, uint32_t* m_out);
/* HC::get_indices / GridHC::get_indices(camera, threshold, w) (HC.cpp:100-132,
 * GridHC.cpp:94-111). */
int gsx_select_hc(gsx_ctx* ctx, const float cam_pos[3], float threshold, uint32_t* m_out);
/* Tile-band sharding (no counterpart in the reference, which draws one frame on one GPU): selects the splats
 * whose quad can reach rows [y0, y1) of a width x height frame drawn with these uniforms -- a conservative test on the
 * splat's position and a radius kept beside it, so nothing the rasterizer would draw into the band is dropped -- for a rank
 * that renders only that band (gsx_render's y0 / y1).  gsx_sort then orders just those: its permutation is the full
 * permutation with the culled entries removed, and the band's pixels are the same bytes.  Selects from ALL uploaded splats
 * (not from an LOD cut).  m_out may be NULL (no host sync). */
int gsx_select_band(gsx_ctx* ctx, const gsx_uniforms* u, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, uint32_t* m_out);
/* Copies the current selection (ascending ids) to the host. */
int gsx_get_selection(gsx_ctx* ctx, uint32_t* idx_out_host, uint32_t capacity, uint32_t* m_out);

/* ---- SplatMesh::sortSplats(indices, cameraPos) (SplatMesh.h:84-114): keys =
 * glm::distance(position, cameraPos), order = distance descending, ties by
 * ascending index.  Sorts the current selection; the permutation stays in HBM
 * (replaces the queue.writeBuffer(sortIndexBuffer) upload, SplatMesh.h:56-57).
 * idx_out_host (nullable) receives the m sorted indices.
 * With idx_out_host == NULL the call only queues the sort; it may be issued while
 * frames queued by gsx_render are still running: the result goes to another
 * permutation buffer (there are three), so those frames keep drawing their own
 * order, and the sort runs on a stream of its own BESIDE them (as does a
 * camera-dependent LOD cut queued before it).  gsx_render / gsx_sync / any call
 * that reads or replaces the selection is ordered behind it by events. ---- */
int gsx_sort(gsx_ctx* ctx, const float cam_pos[3], uint32_t* idx_out_host);
int gsx_get_sorted_dev(gsx_ctx* ctx, const uint32_t** idx_dev, uint32_t* m);
/* gsx_sort for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference.  Same keys, same order (distance descending, ties by
 * ascending index) over the current selection, but the sort's scratch (key / value ping-pong arrays, radix count
 * matrix) lives in a file-scope device array of the library instead of the context's buffers, and the permutation is
 * written to caller-owned DEVICE memory idx_out_dev (capacity >= selection size, else GSX_E_INVALID).  The call only
 * queues kernels on the context's stream: it calls no hipMalloc / hipFree, reads nothing back and does not wait; the
 * context's own permutation (gsx_render, gsx_get_sorted_dev) is left untouched, and device-side error words are
 * reported by the next gsx_sync as usual.
 * Limit: selections of at most GSX_SORT_NOALLOC_MAX splats (the scratch array is 20 MiB per GPU, held until the process
 * ends); larger -> GSX_E_UNSUPPORTED, use gsx_sort.  All contexts on one GPU share that GPU's array: calls that overlap
 * on one GPU (two contexts / two host threads) race -- serialize them. */
#define GSX_SORT_NOALLOC_MAX (1u << 20)
int gsx_sort_noalloc(gsx_ctx* ctx, const float cam_pos[3], uint32_t* idx_out_dev, uint32_t capacity);
/* Parity aid: the f32 distance key (SplatMesh.h:93) of every selected splat, in
 * selection order, computed by the same device function the sort uses. */
int gsx_get_keys(gsx_ctx* ctx, const float cam_pos[3], float* dist_out_host, uint32_t capacity);

/* ---- drawIndexed(6, m) through vs_main/fs_main with the pipeline's blend state
 * (SplatMesh.h:58; shader_quads_ordered.wgsl:42-73,120-212; main.cpp:338-344;
 * clear colour renderer.cpp:162-164).  Renders rows [y0,y1) of a width x height
 * RGBA8 frame from the last gsx_sort result.  rgba8_out_host (nullable)
 * receives width*height*4 bytes (rows outside the band untouched).
 * Frame bands up to 4096 x 4096 pixels; larger -> GSX_E_UNSUPPORTED.
 * With rgba8_out_host == NULL the call only QUEUES the frame (near -> far
 * iterations).  The frame is complete, and the device
 * framebuffer valid, after the next gsx_sync / gsx_get_timings / download;
 * that call also runs the on-demand iterations a heavy frame may still need.
 * Up to two queued frames are in flight at a time, on two streams (the call
 * waits on the host only for the frame BEFORE the last one, whose slot it
 * takes).  With the option "deferred_readback" a call WITH a host buffer is queued the same way and the buffer is filled when
 * the frame's slot is taken over (by the call after the next one) or at gsx_sync.  A frame is a pure function of (scene, permutation, uniforms):
 * neither the iteration schedule nor what else runs beside it changes a byte. ---- */
int gsx_render(gsx_ctx* ctx, const gsx_uniforms* u, uint32_t width, uint32_t height,
               uint32_t y0, uint32_t y1, void* rgba8_out_host);
/* Device pointer of the RGBA8 frame queued LAST (contents complete after gsx_sync, see gsx_render).  The library keeps one
 * framebuffer per frame slot, so consecutive frames land in alternating buffers: ask again after every gsx_render. */
int gsx_get_framebuffer_dev(gsx_ctx* ctx, const void** rgba8_dev, uint32_t* width, uint32_t* height);
/* Render target in caller-owned device memory (the counterpart of the swap-chain texture view the reference's
 * render pass is given, renderer.cpp:132-176 / main.cpp:243-252).  Directs every FOLLOWING gsx_render to rgba8_dev, which
 * must hold width*height*4 bytes of each frame rendered into it (capacity_bytes is checked per frame ->
 * GSX_E_INVALID); frames already queued keep the target they were started on, on-demand iterations included, and the
 * call does not wait for them.  NULL returns to the library's own framebuffers.  The caller keeps the memory alive until the
 * target is changed again or the context is destroyed; rows outside a rendered band are left untouched.
 * Used by the multi-GPU view batches: each rank renders straight into its slot of the batch that RCCL gathers. */
int gsx_set_framebuffer_dev(gsx_ctx* ctx, void* rgba8_dev, uint64_t capacity_bytes);
/* The context's main HIP stream (hipStream_t), for callers that order their own device work or collectives against the
 * frames (e.g. torch.cuda.ExternalStream + RCCL).  The library queues sorts and every other frame on streams of its own;
 * this call makes the main stream wait (on the device) for every sort and frame queued SO FAR, so work the caller queues
 * on the returned stream afterwards runs behind them: call it again whenever new frames have been queued.  No reference
 * counterpart: the reference's queue is the wgpu::Queue handed to SplatMesh::initialize (SplatMesh.h:72). */
int gsx_get_stream(gsx_ctx* ctx, void** hip_stream);

/* ---- multi-GPU: the gather of finished pixels to one rank.  No counterpart in the reference,
 * which is one process on one GPU.  One process and one context per GPU; the splat set is replicated; ranks render
 * different camera views or different tile bands of one frame (gsx_render's y0/y1) and nothing is exchanged until
 * the pixels are finished.  The exchange is RCCL point-to-point over xGMI: inside one group the root posts a receive
 * per peer and every peer one send, so each transfer has its own link.  librccl is loaded on first use. ---- */
typedef struct gsx_comm gsx_comm;
#define GSX_COMM_ID_BYTES 128
/* Rank 0 creates the id (ncclGetUniqueId) and hands it to the other ranks by whatever the host has (MPI, a file, a
 * torch.distributed broadcast). */
int gsx_comm_unique_id(void* id128_out);
/* Collective over the nranks processes: returns when all have joined.  nranks == 1 needs no id and no RCCL. */
int gsx_comm_init(gsx_ctx* ctx, const void* id128, int rank, int nranks, gsx_comm** out);
/* Destroy a communicator before its context.  A context destroyed first drains and releases the communicators it still
 * has (their handles stay valid for gsx_comm_destroy, which then only frees them; gsx_gather rejects them). */
int gsx_comm_destroy(gsx_comm* comm);
/* Every rank contributes send_bytes bytes at send_dev; on `root`, rank r's bytes land at recv_dev + recv_offsets[r]
 * (recv_bytes[r] of them; recv_bytes[r] must equal what rank r sends, recv_bytes[root] the root's own send_bytes).
 * recv_* are read on the root only.  With offsets = y0 * width * 4 the bands of one frame are gathered straight into
 * place; with offsets = r * frame_bytes * batch a batch of whole frames per rank.
 * The exchange runs on a stream of the context's own, beside the frames queued after it.  It is ordered behind (a) the
 * queued frames that render into the memory it ships or fills — those are completed first, which waits for them on the
 * host; frames queued into other memory stay in flight, so a caller that posts the exchange of a batch a frame or two late
 * never waits — and (b), if gsx_get_stream has been called since the last exchange, everything on that stream at the time of the call.  A frame rendered later into memory
 * an exchange still uses waits for it on the device.  The exchange is complete after gsx_sync, or behind any later
 * gsx_get_stream; exchanges of one context run one after another in call order (every rank must post them in the same
 * order).  The memory must stay valid until then.
 * EVERY rank must post the SAME NUMBER of exchanges: the root's grouped receive for a peer that never sends (or a peer's send
 * the root never receives) never completes, and the next gsx_sync hangs with it.  When views do not divide evenly over the
 * ranks, the short ranks post their last buffer again (multigpu.GsxFrameGather.finish takes the collective batch count). */
int gsx_gather(gsx_ctx* ctx, gsx_comm* comm, const void* send_dev, uint64_t send_bytes, void* recv_dev,
               const uint64_t* recv_offsets, const uint64_t* recv_bytes, int root);

/* ---- host-only helpers (no GPU touched): the load-time half of ResourceManager and
 * the scene-graph math that produces the path's inputs. ---- */
/* raw_to_split + optional mean-centering + split_to_splat over n 32-byte SplatRaw
 * records (Splat.h:60-91, ResourceManager.cpp:52-62) -> n 80-byte Splat records. */
int gsx_host_convert_raw(const void* raw32_host, uint64_t n, int center, void* aos80_out_host);
/* ResourceManager::loadSplatsRaw(path, center) (ResourceManager.cpp:36-65) + split_to_splat.
 * A file that cannot be opened yields n = 0 and GSX_OK, as the reference returns an
 * empty vector.  *aos80_out is malloc'ed; release with gsx_host_free. */
int gsx_host_load_splat_file(const char* path, int center, void** aos80_out, uint64_t* n_out);
void gsx_host_free(void* p);
/* Page-locked host memory for frames read back by gsx_render(..., rgba8_out_host): the copy then goes by DMA straight into
 * the caller's buffer (6 M splats, 1080p, one synchronous frame: 1.23 -> 0.91 ms).  No counterpart in the reference, whose
 * frames never leave the GPU (the swap chain of renderer.cpp:73-95); the facade's Surface keeps its frame in it.  Falls back
 * to malloc when the runtime refuses (the memory is then pageable, nothing else changes); NULL only when that fails too.
 * Release with gsx_host_free_pinned, at any time before or after the contexts are destroyed. */
void* gsx_host_alloc_pinned(uint64_t bytes);
void gsx_host_free_pinned(void* p);
/* 3D-Gaussian-splatting PLY input (INRIA training output).  No counterpart in the reference
 * (its configs name .ply files but ResourceManager.cpp:36-65 reads only 32-byte .splat records): the PLY is
 * converted to exactly those records (host/gsx_ply.hpp states the conversion), after which the reference's own
 * raw_to_split / split_to_splat path applies.  *_raw yields n 32-byte SplatRaw records, *_file the 80-byte Splat
 * records ready for gsx_upload_splats.  Missing file -> n = 0 and GSX_OK (as loadSplatsRaw); ASCII / big-endian /
 * list properties -> GSX_E_UNSUPPORTED; broken header, missing property or truncated data -> GSX_E_INVALID.
 * Outputs are malloc'ed; release with gsx_host_free. */
int gsx_host_load_ply_raw(const char* path, void** raw32_out, uint64_t* n_out);
int gsx_host_load_ply_file(const char* path, int center, void** aos80_out, uint64_t* n_out);
/* Camera::getViewMatrix / getProjectionMatrix (Camera.h:27-34) and proj*view (BB.hpp:60-61).
 * Any output pointer may be NULL. */
int gsx_host_camera(const float world16[16], float fov_deg, float aspect, float znear, float zfar,
                    float view16[16], float proj16[16], float viewproj16[16]);
/* World matrix of OrbitCamera(distance)'s camera after n_rot successive
 * Node::rotate(axis_i, angle_i) calls on the orbit node (OrbitCamera.h:24-30,
 * Node.cpp:10-27, main.cpp:469-486). */
int gsx_host_orbit_pose(float distance, const float* axes3, const float* angles, uint32_t n_rot, float world16[16]);
/* The same camera driven by window events as the reference's Application forwards them (main.cpp:139-155,235-236 ->
 * OrbitCamera.h:32-74): events4 holds n_events rows of 4 doubles, {0, xpos, ypos, deltaTime} = cursor moved,
 * {1, button, action, -} = mouse button (button 0, action 1 = pressed), {2, yoffset, -, deltaTime} = scrolled;
 * orbit_rate / scroll_rate are GUI::Parameters' (gui.hpp:22-23).  The optional viewer front end (host/gsx_viewer.hpp) runs
 * on the same handlers. */
int gsx_host_orbit_events(float distance, float orbit_rate, float scroll_rate, const double* events4, uint32_t n_events,
                          float world16[16]);

/* ---- host-side LOD structures: Octree (Octree.cpp:33-187), HC (HC.cpp:4-98), GridHC
 * (GridHC.cpp:10-74) built on the CPU exactly as the reference builds them at load time
 * (SplatMeshOctree.hpp:57-79, SplatMeshHC.hpp:58-91, SplatMeshGridHC.cpp:26-52), then
 * flattened for gsx_set_lod_octree / gsx_set_lod_hc. ---- */
typedef struct gsx_lod gsx_lod;
int gsx_host_octree_build(const void* raw32_host, uint64_t n, int center, gsx_lod** out);
int gsx_host_hc_build(const void* aos80_host, uint64_t n, gsx_lod** out);
int gsx_host_gridhc_build(const void* aos80_host, uint64_t n, uint32_t sx, uint32_t sy, uint32_t sz, gsx_lod** out);
void gsx_host_lod_free(gsx_lod* lod);
uint64_t gsx_host_lod_size(const gsx_lod* lod);    /* number of splats (= nodes) the structure generated */
uint64_t gsx_host_lod_roots(const gsx_lod* lod);   /* HC / GridHC: roots left in HC::nodes */
uint64_t gsx_host_lod_cells(const gsx_lod* lod);   /* GridHC: non-empty cells */
int gsx_host_lod_splats(const gsx_lod* lod, void* aos80_out_host);   /* Octree::splats / HC::splats / GridHC::splats */
/* Any output pointer may be NULL. */
int gsx_host_lod_flat_octree(const gsx_lod* lod, float* center3, float* half3, uint32_t* parent, uint8_t* is_leaf,
                             uint32_t* first_child, uint32_t* n_children);
int gsx_host_lod_flat_hc(const gsx_lod* lod, uint32_t* parent, uint8_t* is_leaf, float* error, float* weight,
                         uint32_t* child2, uint32_t* roots, uint32_t* cell_root_start);
/* View-independent cuts, in the reference's output order: Octree::get_indices_depth
 * (Octree.hpp:131-162), HC::get_indices_depth (HC.cpp:134-168), GridHC::get_indices_error
 * (GridHC.cpp:76-92).  Feed the result to gsx_select_list. */
int gsx_host_lod_cut_depth(gsx_lod* lod, uint32_t depth, uint32_t* idx_out_host, uint64_t capacity, uint64_t* m_out);
/* CPU evaluation of the camera-dependent cuts (Octree::get_indices, HC/GridHC::get_indices), kept
 * for hosts without a GPU context at hand; the per-frame path is gsx_select_octree / gsx_select_hc. */
int gsx_host_lod_cut_camera(gsx_lod* lod, const float world16[16], float fov_deg, float aspect, float znear, float zfar,
                            float threshold, uint32_t* idx_out_host, uint64_t capacity, uint64_t* m_out);

#ifdef __cplusplus
}
#endif
#endif /* GSX_H */
