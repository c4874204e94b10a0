// gsx_raster.hip — projection (K3), tile binning (K4) and the tile rasterizer
// (K5).
//
// Reference path replaced: vs_main / fs_main of resources/shader_quads_ordered.wgsl
// (wgsl:28-73,120-212) executed by drawIndexed(6, M) (SplatMesh.h:58) with the
// pipeline's SrcAlpha/OneMinusSrcAlpha colour blend, dst-alpha-preserving alpha
// blend, no depth buffer (main.cpp:338-344,359) and the (0.05,0.05,0.05,1) clear
// (renderer.cpp:162-164).
//
// Design: the frame is drawn in a few near -> far iterations over the depth-sorted
// permutation.  Per iteration: project the window's splats (vs_main ONCE per splat, the reference runs it
// 4x, once per quad vertex; gathered by rank from 64-byte scene records) and count their tiles, skipping
// splats whose tile box holds no unsaturated tile; scan; emit (tile, splat) pairs in near -> far order;
// stable 2-pass radix sort by tile id (the depth sort's kernels), which keeps depth order inside each tile;
// composite.  Because every quad is a view-plane-parallel parallelogram at constant clip w, uv is an affine
// function of the pixel position, so coverage + uv come from one 2x2 inverse instead of triangle setup.
// The rasterizer composites each 16x16 tile front to back with a transmittance accumulator
//     C = sum_i c_i a_i prod_{j nearer than i} (1 - a_j)  +  T_final * clear
// which is algebraically the reference's far-to-near "over" blend and lets a tile retire as soon as every
// pixel's transmittance is below 2^-11 (bounded error < 0.13/255); retired tiles are never binned, sorted
// or shaded again, and splats behind the last needed rank are never even projected.
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "gsx_internal.h"
#include "gsx_radix.hpp"

struct GsxViewParams {
    float wt[16];      // view * model
    float proj[16];
    float cut_s;       // cutOff * splatSize
    float cut;         // cutOff
    float width, height;
    uint32_t y0, y1;   // band rows [y0, y1)
    uint32_t tiles_x, tiles_y, tile_y0;  // tile grid of the band (tile rows start at tile_y0)
    uint32_t bayer_size;
    float bayer_scale;
};

// Inclusive tile bbox of a projected splat inside the band's tile grid.
__device__ __forceinline__ bool gsx_tile_bbox(float4 c, const GsxViewParams& vp, uint32_t& tx0, uint32_t& tx1, uint32_t& ty0,
                                              uint32_t& ty1) {
    if (!(c.z >= 0.0f)) return false;
    const float x0 = fmaxf(ceilf(c.x - c.z - 0.5f), 0.0f), x1 = fminf(floorf(c.x + c.z - 0.5f), vp.width - 1.0f);
    const float y0 = fmaxf(ceilf(c.y - c.w - 0.5f), static_cast<float>(vp.y0));
    const float y1 = fminf(floorf(c.y + c.w - 0.5f), static_cast<float>(vp.y1) - 1.0f);
    if (x1 < x0 || y1 < y0) return false;
    tx0 = static_cast<uint32_t>(x0) / GSX_TILE_W; tx1 = static_cast<uint32_t>(x1) / GSX_TILE_W;
    ty0 = static_cast<uint32_t>(y0) / GSX_TILE_H - vp.tile_y0; ty1 = static_cast<uint32_t>(y1) / GSX_TILE_H - vp.tile_y0;
    return true;
}

// ---------------------------------------------------------------------------
// K3: per-splat projection = vs_main (wgsl:42-73) evaluated once per splat.  Device function: it runs inside
// k_chunk_count on the ranks an iteration actually reaches (splats hidden behind retired tiles are never
// projected).  Returns false for a culled splat; a = (cx, cy, ex, ey), mm = (m00, m01, m10, m11).
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool gsx_project(const float4 p, const float4 ca, const float4 cb, const GsxViewParams& vp,
                                            float4& a, float4& mm) {
    const float* W = vp.wt;  // column-major 4x4; W[c*4+r]
    // s_center = wt * sm[3]  (wgsl:53)
    float ctr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ctr[r] = (W[0 + r] * p.x + W[4 + r] * p.y) + (W[8 + r] * p.z + W[12 + r] * p.w);
    // S symmetric: [s00 s01 s02; s01 s11 s12; s02 s12 s22]
    const float s00 = ca.x, s01 = ca.y, s02 = ca.z, s11 = ca.w, s12 = cb.x, s22 = cb.y;
    // WS = W3 * S (rows 0,1 only are needed for the 2x2 block of W S W^T)
    float ws[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float w0 = W[0 + r], w1 = W[4 + r], w2 = W[8 + r];
        ws[r][0] = w0 * s00 + w1 * s01 + w2 * s02;
        ws[r][1] = w0 * s01 + w1 * s11 + w2 * s12;
        ws[r][2] = w0 * s02 + w1 * s12 + w2 * s22;
    }
    // V[c][r] = sum_k WS[r][k] * W[k-th column][c-th row]
    float v[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 2; ++r) v[c][r] = ws[r][0] * W[0 + c] + ws[r][1] * W[4 + c] + ws[r][2] * W[8 + c];
    // eigenvectors (wgsl:28-40), same expression order
    const float a11 = v[0][0], a22 = v[1][1], a01 = v[0][1], a10 = v[1][0];
    const float a1221 = a01 * a10;
    const float det = sqrtf(a11 * a11 + 4.0f * a1221 - 2.0f * a11 * a22 + a22 * a22);
    const float lam1 = 0.5f * (a11 + a22 + det), lam2 = 0.5f * (a11 + a22 - det);
    const float phi = 0.5f * atan2f(2.0f * a01, a11 - a22);
    float sn, cs;
    sincosf(phi, &sn, &cs);
    const float s1 = sqrtf(lam1), s2 = sqrtf(lam2);
    // view-space quad basis scaled by cutOff*splatSize (wgsl:63)
    const float e0x = cs * s1 * vp.cut_s, e0y = sn * s1 * vp.cut_s;
    const float e1x = -sn * s2 * vp.cut_s, e1y = cs * s2 * vp.cut_s;
    const float* P = vp.proj;
    // clip = P * (x, y, z, w); host guarantees P[0][2]=P[1][2]=P[0][3]=P[1][3]=0,
    // so clip.z / clip.w are constant over the quad (wgsl:65-68)
    const float clip_z = P[10] * ctr[2] + P[14] * ctr[3];
    const float clip_w = P[11] * ctr[2] + P[15] * ctr[3];
    const float ccx = (P[0] * ctr[0] + P[4] * ctr[1]) + (P[8] * ctr[2] + P[12] * ctr[3]);
    const float ccy = (P[1] * ctr[0] + P[5] * ctr[1]) + (P[9] * ctr[2] + P[13] * ctr[3]);
    const float iw = 1.0f / clip_w;
    const float hw = 0.5f * vp.width, hh = 0.5f * vp.height;
    // screen = ((ndc.x*0.5+0.5)*W, (0.5-0.5*ndc.y)*H)
    const float cx = (ccx * iw * 0.5f + 0.5f) * vp.width;
    const float cy = (0.5f - ccy * iw * 0.5f) * vp.height;
    const float ax = (P[0] * e0x + P[4] * e0y) * iw * hw, ay = -(P[1] * e0x + P[5] * e0y) * iw * hh;
    const float bx = (P[0] * e1x + P[4] * e1y) * iw * hw, by = -(P[1] * e1x + P[5] * e1y) * iw * hh;
    const float dd = ax * by - bx * ay;
    GsxProj o;
    o.cx = cx; o.cy = cy;
    o.ex = fabsf(ax) + fabsf(bx);
    o.ey = fabsf(ay) + fabsf(by);
    const float idd = vp.cut / dd;
    o.m00 = by * idd; o.m01 = -bx * idd;
    o.m10 = -ay * idd; o.m11 = ax * idd;
    // clip volume 0 <= z <= w (WebGPU), w > 0; non-finite vertices drop the primitive
    bool ok = (clip_w > 0.0f) && (clip_z >= 0.0f) && (clip_z <= clip_w);
    const float chk = cx + cy + o.ex + o.ey + o.m00 + o.m01 + o.m10 + o.m11;
    ok = ok && (fabsf(chk) < 3.0e38f) && (dd != 0.0f);  // false for NaN/inf
    // pixel-centre bbox: i + 0.5 in [c - e, c + e]
    if (ok) {
        const float x0 = ceilf(cx - o.ex - 0.5f), x1 = floorf(cx + o.ex - 0.5f);
        const float yy0 = ceilf(cy - o.ey - 0.5f), yy1 = floorf(cy + o.ey - 0.5f);
        ok = (x1 >= x0) && (yy1 >= yy0) && (x1 >= 0.0f) && (x0 <= vp.width - 1.0f) && (yy1 >= static_cast<float>(vp.y0)) &&
             (yy0 <= static_cast<float>(vp.y1) - 1.0f);
    }
    a = make_float4(o.cx, o.cy, o.ex, o.ey);
    mm = make_float4(o.m00, o.m01, o.m10, o.m11);
    return ok;
}

// ---------------------------------------------------------------------------
// K4: occlusion-aware binning, near -> far in a few iterations.
//
// A frame is drawn in iterations that walk the depth-sorted permutation from its NEAR end.
// Iteration k takes the next window of ranks, expands the splats whose tile box still holds
// a live (unsaturated) tile into (tile, splat) pairs in near -> far order, sorts the pairs by
// tile id (stable, so depth order survives), and composites them onto per-pixel accumulators
// (K5).  Tiles whose 256 pixels are all saturated are retired in a bit mask, and everything
// behind them is never expanded, sorted or shaded again: measured on the 6M-splat scenes the
// rasterizer consumes 3-4 % of the pairs a one-shot binner emits.
//
// Nothing in the loop returns to the host.  The iteration's size is decided on the device:
// k_chunk_scan keeps the longest prefix of the window's 1024-rank blocks whose pairs fit the
// pair buffers and advances the frame cursor by exactly that much, so a heavy window simply
// spills into the next iteration.  The host reads the control block once, after the frame,
// and runs extra iterations only if ranks are left (gsx_frame_finish).
// ---------------------------------------------------------------------------
#define GSX_BIN_NT 256
#define GSX_BIN_IPT 4
#define GSX_BIN_TILE (GSX_BIN_NT * GSX_BIN_IPT)
#define GSX_BIN_SUBS 16          // 64-rank sub-blocks per count block (one emit workgroup each)

// ---- per-tile rank caps ----------------------------------------------------------------------------------------------
// Measured on cfg 2: of 10.3 M (tile, splat) pairs an occlusion-aware frame still bins and sorts, the rasterizer stages 2.3 M;
// the rest lies behind the point where its tile saturated INSIDE an iteration's window.  Where a tile saturates barely moves
// from one frame to the next, so every tile gets a cap: the rank (from the near end) at which it retired in the slot's last
// frame, taken over a neighbourhood as wide as the picture can have moved, times 1.5, rounded up to a band boundary.  Ranks
// are cut into GSX_BANDS bands whose bounds are multiples of 1024 (a count block of 1024 ranks never straddles one), and per
// band a bit mask says which tiles still want ranks of that band; k_chunk_count ANDs the mask of its block's band into the
// live-tile mask, nothing else changes.  A tile that is still live at the end of the frame although its cap withheld ranks
// (scene or view changed more than predicted) is counted by the rasterizer; the host then runs a REDO continuation from the
// smallest such cap with the complementary masks, which supplies exactly the withheld pairs, in order.  Pixels freeze at
// saturation (k_raster), so the image is the same with and without caps, predicted well or badly.
__host__ __device__ __forceinline__ uint32_t gsx_band_of(uint32_t q /* rank / 1024 */) {
    if (q < 8u) return q;
    uint32_t o = 31u - static_cast<uint32_t>(
#ifdef __HIP_DEVICE_COMPILE__
        __clz(q)
#else
        __builtin_clz(q)
#endif
    );
    const uint32_t j = 8u + (o - 3u) * 4u + ((q >> (o - 2u)) & 3u);
    return j < GSX_BANDS - 1u ? j : GSX_BANDS - 1u;
}
// lower bound of band j, in units of 1024 ranks
__host__ __device__ __forceinline__ uint32_t gsx_band_lo(uint32_t j) {
    if (j < 8u) return j;
    const uint32_t o = 3u + (j - 8u) / 4u, sub = (j - 8u) & 3u;
    return (4u + sub) << (o - 2u);
}

// One launch per frame: clears the control block, the live-tile mask and the per-tile state, and - with rank caps - gives every
// tile its cap band and builds the band masks.  A thread owns one tile POSITION of a row padded to 256 columns
// (GSX_LIVE_ROW_WORDS x 32), so a wave holds exactly two mask words of one tile row and a band's mask is one ballot:
// band_masks[j][ty][w] bit = tile (ty, 32 w + bit) wants ranks of band j, i.e. j < its cap band.
__global__ __launch_bounds__(256) void k_frame_init(GsxFrameCtl* __restrict__ ctl, const uint32_t* __restrict__ m_ptr,
                                                    uint32_t* __restrict__ live_bits, uint32_t* __restrict__ tile_state,
                                                    uint32_t tiles_x, uint32_t tiles_y, uint32_t* __restrict__ aux,
                                                    const uint32_t* __restrict__ depth_prev, uint32_t* __restrict__ depth_cur,
                                                    uint8_t* __restrict__ tile_band, uint32_t* __restrict__ band_masks,
                                                    int radius /* < 0: no caps */,
                                                    uint32_t cap_scale_q8 /* cap = depth * this / 256 */, uint32_t cap_margin) {
    static_assert(GSX_LIVE_ROW_WORDS * 32 == 256 && GSX_BANDS == 64, "one 256-thread workgroup per tile row, one lane per band");
    const uint32_t lw = (tiles_x + 31u) / 32u, n_tiles = tiles_x * tiles_y;
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    if (gtid < 64) aux[gtid] = 0u;
    if (gtid < sizeof(GsxFrameCtl) / 4) {
        uint32_t v = 0;
        if (gtid == offsetof(GsxFrameCtl, m) / 4) v = *m_ptr;
        if (gtid == offsetof(GsxFrameCtl, live_tiles) / 4) v = n_tiles;
        reinterpret_cast<uint32_t*>(ctl)[gtid] = v;
    }
    for (uint32_t i = gtid; i < lw * tiles_y; i += gsz) {
        const uint32_t w = i % lw, left = tiles_x - w * 32u;
        live_bits[i] = left >= 32u ? 0xFFFFFFFFu : ((1u << left) - 1u);
    }
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t ty = blockIdx.x; ty < tiles_y; ty += gridDim.x) {      // (uniform per workgroup: the ballots below see whole waves)
        const uint32_t tx = threadIdx.x;
        const bool in_row = tx < tiles_x;
        const uint32_t i = ty * tiles_x + tx;
        uint32_t band = 0u;                                              // (positions past the row want nothing)
        if (in_row) {
            tile_state[i] = 0u;
            depth_cur[i] = 0u;
        }
        if (radius < 0) continue;
        if (in_row) {
            // cap band of the tile: deepest retirement in the neighbourhood last frame; a neighbour that never retired lifts the cap
            uint32_t deepest = 0;
            bool open_end = false;
            for (int y = static_cast<int>(ty) - radius; y <= static_cast<int>(ty) + radius; ++y)
                for (int x = static_cast<int>(tx) - radius; x <= static_cast<int>(tx) + radius; ++x) {
                    if (x < 0 || y < 0 || x >= static_cast<int>(tiles_x) || y >= static_cast<int>(tiles_y)) continue;
                    const uint32_t d = depth_prev[static_cast<uint32_t>(y) * tiles_x + static_cast<uint32_t>(x)];
                    open_end |= d == 0u;
                    deepest = d > deepest ? d : deepest;
                }
            band = GSX_BAND_INF;
            if (!open_end) {
                const uint64_t cap = static_cast<uint64_t>(deepest) * cap_scale_q8 / 256u + cap_margin;   // x 1.5 + 2 K ranks by default
                const uint64_t q = (cap + 1023u) >> 10;
                if (q < (1ull << 31)) {
                    band = gsx_band_of(static_cast<uint32_t>(q));
                    if (gsx_band_lo(band) < q) ++band;           // round UP to a band boundary: ranks below R_band are wanted
                    if (band >= GSX_BANDS) band = GSX_BAND_INF;
                }
            }
            tile_band[i] = static_cast<uint8_t>(band);
        }
        // lane j keeps band j's ballot; the wave's 64 positions are mask words 2 * wave and 2 * wave + 1 of the row
        uint64_t mine = 0;
        for (uint32_t j = 0; j < GSX_BANDS; ++j) {
            const uint64_t bal = __builtin_amdgcn_ballot_w64(j < band);
            if (lane == j) mine = bal;
        }
        uint32_t* dst = band_masks + (static_cast<uint64_t>(lane) * tiles_y + ty) * GSX_LIVE_ROW_WORDS + (threadIdx.x >> 6) * 2u;
        dst[0] = static_cast<uint32_t>(mine);
        dst[1] = static_cast<uint32_t>(mine >> 32);
    }
}

// true if any tile of the inclusive box is still live.  s_live = one row of GSX_LIVE_ROW_WORDS (8) words per tile
// row in LDS (16-byte aligned).  Four rows are fetched per step with 16-byte reads and tested against per-word column
// masks, so a splat covering a thousand tiles of a mostly retired frame costs ~35 dependent steps, not ~1000.
__device__ __forceinline__ bool gsx_box_live(const uint32_t* s_live, uint32_t tx0, uint32_t tx1, uint32_t ty0, uint32_t ty1) {
    const uint32_t w0 = tx0 >> 5, w1 = tx1 >> 5;
    const uint32_t first = 0xFFFFFFFFu << (tx0 & 31u), last = 0xFFFFFFFFu >> (31u - (tx1 & 31u));
    if (w1 - w0 <= 1u) {
        // nearly every box: at most two mask words per tile row (the general form below builds eight column masks with
        // compares and selects - ~100 instructions a rank, a fifth of k_chunk_count, for boxes up to 256 tiles wide)
        const uint32_t m0 = w1 == w0 ? (first & last) : first, m1 = w1 == w0 ? 0u : last;
        const uint32_t i1 = w1;                               // (== w0 with m1 == 0 when the box lies in one word)
        uint32_t acc = 0u;
        for (uint32_t ty = ty0; ty <= ty1; ++ty) acc |= (s_live[ty * GSX_LIVE_ROW_WORDS + w0] & m0) | (s_live[ty * GSX_LIVE_ROW_WORDS + i1] & m1);
        return acc != 0u;
    }
    uint32_t cm[GSX_LIVE_ROW_WORDS];
#pragma unroll
    for (uint32_t w = 0; w < GSX_LIVE_ROW_WORDS; ++w) {
        uint32_t m = (w >= w0 && w <= w1) ? 0xFFFFFFFFu : 0u;
        if (w == w0) m &= first;
        if (w == w1) m &= last;
        cm[w] = m;
    }
    const uint4* rows = reinterpret_cast<const uint4*>(s_live);
    for (uint32_t ty = ty0; ty <= ty1; ty += 4u) {
        uint32_t acc = 0u;
#pragma unroll
        for (uint32_t r = 0; r < 4u; ++r) {
            const uint32_t y = (ty + r <= ty1) ? ty + r : ty1;      // (repeats the last row instead of branching)
            const uint4 lo = rows[y * 2u], hi = rows[y * 2u + 1u];
            acc |= (lo.x & cm[0]) | (lo.y & cm[1]) | (lo.z & cm[2]) | (lo.w & cm[3]) | (hi.x & cm[4]) | (hi.y & cm[5]) | (hi.z & cm[6]) |
                   (hi.w & cm[7]);
        }
        if (acc) return true;
    }
    return false;
}

// Projected rank -> pair count; box = tx0 | ty0<<8 | (w-1)<<16 | (h-1)<<24 of its tile box (GSX_RBOX_NONE when the
// splat is culled or every tile of the box is retired).
#define GSX_RBOX_NONE 0xFFFFFFFFu   // not a valid box: tx0 + (w-1) <= 255 always
__device__ __forceinline__ uint32_t gsx_rank_pairs(const float4 p, const float4 ca, const float4 cb, const GsxViewParams& vp,
                                                   const uint32_t* s_live, uint32_t& box, float4& a, float4& mm) {
    if (!gsx_project(p, ca, cb, vp, a, mm)) return 0u;
    uint32_t tx0, tx1, ty0, ty1;
    if (!gsx_tile_bbox(a, vp, tx0, tx1, ty0, ty1) || !gsx_box_live(s_live, tx0, tx1, ty0, ty1)) return 0u;
    box = tx0 | (ty0 << 8) | ((tx1 - tx0) << 16) | ((ty1 - ty0) << 24);
    return (tx1 - tx0 + 1u) * (ty1 - ty0 + 1u);
}

// K4a.  Block b of iteration `iter` owns the 1024 ranks hi-1-b*1024 ... hi-(b+1)*1024 (hi = m - cursor),
// nearest first: pair count of the block -> block_sums[b], and the exclusive in-block pair offset at every
// 64-rank boundary -> sub_off[b*17 + s] (s = 16 holds the block total).
#ifndef GSX_CNT_NT
#define GSX_CNT_NT 512           // two ranks per thread: the projection is a long dependent chain behind a 64-B gather, so the
#endif                           // kernel wants many waves per CU, not many ranks per thread (256 x 4 ranks: +35 us per frame)
#define GSX_CNT_IPT (GSX_BIN_TILE / GSX_CNT_NT)
__global__ __launch_bounds__(GSX_CNT_NT) void k_chunk_count(const GsxFrameCtl* __restrict__ ctl, uint32_t iter, uint32_t window,
                                                            const uint32_t* __restrict__ sorted, const GsxGeo* __restrict__ geo,
                                                            const uint32_t* __restrict__ live_bits, GsxViewParams vp,
                                                            uint32_t* __restrict__ block_sums, uint32_t* __restrict__ sub_off,
                                                            uint32_t* __restrict__ rank_box, float4* __restrict__ rank_rec,
                                                            const uint32_t* __restrict__ band_masks /* null: no caps */, uint32_t redo) {
    static_assert(GSX_CNT_NT * GSX_CNT_IPT == GSX_BIN_TILE && GSX_CNT_NT % (GSX_BIN_SUBS * 1) == 0, "a block owns 1024 ranks");
    const uint32_t tiles_x = vp.tiles_x, tiles_y = vp.tiles_y;
    __shared__ uint32_t s_tmp[GSX_CNT_NT / 64];
    __shared__ __attribute__((aligned(16))) uint32_t s_live[GSX_MAX_TILES_Y * GSX_LIVE_ROW_WORDS];
    const uint32_t b = blockIdx.x;
    const uint32_t cur = ctl->cursor[iter], m = ctl->m;
    const uint32_t remaining = m - cur, win = window < remaining ? window : remaining;
    if (ctl->live_tiles == 0u || b * GSX_BIN_TILE >= win) {
        if (threadIdx.x == 0) block_sums[b] = 0u;
        if (threadIdx.x <= GSX_BIN_SUBS) sub_off[b * (GSX_BIN_SUBS + 1) + threadIdx.x] = 0u;
        return;
    }
    const uint32_t local0 = b * GSX_BIN_TILE + threadIdx.x * GSX_CNT_IPT;
    // the gathers of the thread's ranks are issued first (one 64-B line per splat), the live-tile mask is staged behind them
    float4 gp[GSX_CNT_IPT], ga[GSX_CNT_IPT], gb[GSX_CNT_IPT], gc[GSX_CNT_IPT];
    const uint32_t hi = m - cur;
#pragma unroll
    for (int j = 0; j < GSX_CNT_IPT; ++j) {
        const uint32_t local = local0 + j;
        const uint32_t idx = sorted[hi - 1u - (local < win ? local : win - 1u)];
        const float4* g = reinterpret_cast<const float4*>(geo + idx);
        gp[j] = g[0]; ga[j] = g[1]; gb[j] = g[2]; gc[j] = g[3];
    }
    const uint32_t lw = (tiles_x + 31u) / 32u;
    // the block's 1024 ranks lie in ONE band (cursors and band bounds are multiples of 1024): tiles whose cap is below it do
    // not want them - or, in a redo continuation, are the only ones that do
    const uint32_t* bm = band_masks ? band_masks + static_cast<uint64_t>(gsx_band_of((cur >> 10) + b)) * (GSX_LIVE_ROW_WORDS * tiles_y) : nullptr;
    for (uint32_t i = threadIdx.x; i < GSX_LIVE_ROW_WORDS * tiles_y; i += GSX_CNT_NT) {      // rows padded to 8 words
        const uint32_t ty = i / GSX_LIVE_ROW_WORDS, w = i % GSX_LIVE_ROW_WORDS;
        uint32_t v = w < lw ? live_bits[ty * lw + w] : 0u;
        if (bm) v &= redo ? ~bm[i] : bm[i];
        s_live[i] = v;
    }
    __syncthreads();
    uint32_t tot = 0;
    for (int j = 0; j < GSX_CNT_IPT; ++j) {      // (not unrolled: the projection is long)
        uint32_t rb = GSX_RBOX_NONE;
        float4 a, mm;
        const float4 col = gc[j];
        const uint32_t cnt = (local0 + j < win) ? gsx_rank_pairs(gp[j], ga[j], gb[j], vp, s_live, rb, a, mm) : 0u;
        if (cnt) {
            // the 48-B screen-space record the rasterizer stages, in window-rank order
            float4* rec = rank_rec + static_cast<uint64_t>(local0 + j) * 3;
            rec[0] = a; rec[1] = mm; rec[2] = col;
        }
        rank_box[local0 + j] = rb;               // what k_chunk_emit needs per rank, in rank order
        tot += cnt;
    }
    uint32_t block_total;
    const uint32_t excl = gsx_block_excl_scan<GSX_CNT_NT>(tot, s_tmp, &block_total);
    constexpr uint32_t TPS = GSX_CNT_NT / GSX_BIN_SUBS;          // threads per 64-rank sub-block
    if (threadIdx.x % TPS == 0) sub_off[b * (GSX_BIN_SUBS + 1) + threadIdx.x / TPS] = excl;
    if (threadIdx.x == 0) {
        block_sums[b] = block_total;
        sub_off[b * (GSX_BIN_SUBS + 1) + GSX_BIN_SUBS] = block_total;
    }
}

// K4b.  Pair emission, parallel over PAIR SLOTS: wave W writes slots [W*S, (W+1)*S) of the iteration's pair array
// (S = GSX_EMIT_SLICE), whatever ranks they belong to, so a window of few huge splats (thousands of tiles each)
// spreads over the chip like a window of many small ones.  The wave finds its first 1024-rank block with a
// 64-ary search over the scanned block offsets, then walks 64-rank sub-blocks: it reads the tile boxes
// k_chunk_count left per rank, scans the 64 counts, and writes its part of the sub-block's pairs (near -> far)
// with coalesced stores; every slot finds its rank by binary search in the 65 offsets held in the wave's own
// LDS slice.  No workgroup barrier anywhere.
#ifndef GSX_EMIT_SLICE
#define GSX_EMIT_SLICE 512u
#endif
__global__ __launch_bounds__(GSX_BIN_NT) void k_chunk_emit(const GsxFrameCtl* __restrict__ ctl, uint32_t tiles_x,
                                                           const uint32_t* __restrict__ block_sums, const uint32_t* __restrict__ sub_off,
                                                           const uint32_t* __restrict__ rank_box,
                                                           uint16_t* __restrict__ pair_key, uint32_t* __restrict__ pair_val
#ifdef GSX_DEBUG_TILETIME
                                                           , unsigned long long* __restrict__ dbg_slices
#endif
                                                           ) {
    constexpr uint32_t SUB = GSX_BIN_TILE / GSX_BIN_SUBS;        // 64 ranks
    constexpr uint32_t NW = GSX_BIN_NT / 64;
    static_assert(SUB == 64 && GSX_BIN_SUBS == 16, "one rank per lane, 17 sub offsets per block");
    __shared__ uint32_t s_off[NW][SUB + 1], s_box[NW][SUB];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t total = ctl->pairs, nbk = ctl->n_blocks;
    uint32_t* off = s_off[wave];
    uint32_t* sbx = s_box[wave];
    // Slices are taken from the far end, and the farthest eighth of the pairs is cut into slices of GSX_EMIT_SLICE / 8: behind
    // the rank caps the last slices cross a dozen nearly empty blocks, two dependent loads each, and took 25-35 us against 8 for
    // the rest (tools/tile_times.py) - started first and spread over eight times as many waves they end with the bulk.
    constexpr uint32_t SMALL = GSX_EMIT_SLICE / 8u;
    const uint32_t n_big = (total - total / 8u) / GSX_EMIT_SLICE;                 // slices of GSX_EMIT_SLICE from slot 0
    const uint32_t big_end = n_big * GSX_EMIT_SLICE;
    const uint32_t n_small = (total - big_end + SMALL - 1u) / SMALL;              // the rest, in slices of SMALL
    const uint64_t n_slices = static_cast<uint64_t>(n_big) + n_small;
    for (uint64_t wi = blockIdx.x * NW + wave; wi < n_slices; wi += static_cast<uint64_t>(gridDim.x) * NW) {
        uint32_t s0, s1;
        if (wi < n_small) {                                                       // far end first
            s0 = big_end + (n_small - 1u - static_cast<uint32_t>(wi)) * SMALL;
            s1 = (total - s0 < SMALL) ? total : s0 + SMALL;
        } else {
            s0 = (n_big - 1u - static_cast<uint32_t>(wi - n_small)) * GSX_EMIT_SLICE;
            s1 = s0 + GSX_EMIT_SLICE;
        }
        const uint64_t w = wi;      // (debug record index)
        (void)w;
#ifdef GSX_DEBUG_TILETIME
        const unsigned long long dbg_t0 = wall_clock64();
        uint32_t dbg_visits = 0, dbg_blocks = 0;
#endif
        // largest b with block_sums[b] <= s0 (block_sums[0] == 0; empty blocks share their successor's offset)
        uint32_t lo = 0, hi = nbk;
        while (hi - lo > 1u) {
            const uint32_t step = (hi - lo + 63u) / 64u;
            const uint32_t idx = lo + lane * step;
            const uint32_t v = idx < hi ? block_sums[idx] : 0xFFFFFFFFu;
            const uint32_t c = static_cast<uint32_t>(__popcll(__builtin_amdgcn_ballot_w64(v <= s0)));   // >= 1
            lo += (c - 1u) * step;
            hi = (lo + step < hi) ? lo + step : hi;
        }
        uint32_t cur = s0;
        for (uint32_t b = lo; b < nbk && cur < s1; ++b) {
#ifdef GSX_DEBUG_TILETIME
            ++dbg_blocks;
#endif
            const uint32_t bbase = block_sums[b];
            const uint32_t so = lane <= GSX_BIN_SUBS ? sub_off[b * (GSX_BIN_SUBS + 1) + lane] : 0xFFFFFFFFu;
            const uint32_t blk_total = __shfl(so, GSX_BIN_SUBS);
            if (bbase + blk_total <= cur) continue;               // an empty block (wave-uniform)
            if (bbase >= s1) break;
            // first sub-block that ends behind `cur`: number of inner boundaries so[1..16] at or before it
            uint32_t sb = static_cast<uint32_t>(__popcll(__builtin_amdgcn_ballot_w64(lane >= 1u && lane <= GSX_BIN_SUBS && bbase + so <= cur)));
            for (; sb < GSX_BIN_SUBS && cur < s1; ++sb) {
                const uint32_t sub_base = bbase + __shfl(so, sb), sub_end = bbase + __shfl(so, sb + 1u);
                if (sub_end <= cur) continue;                     // empty sub-block
                if (sub_base >= s1) { cur = s1; break; }
#ifdef GSX_DEBUG_TILETIME
                ++dbg_visits;
#endif
                const uint32_t local = b * GSX_BIN_TILE + sb * SUB + lane;
                const uint32_t box = rank_box[local];
                const uint32_t cnt = box == GSX_RBOX_NONE ? 0u : (((box >> 16) & 0xFFu) + 1u) * ((box >> 24) + 1u);
                uint32_t incl = cnt;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t t = __shfl_up(incl, o);
                    if (lane >= static_cast<uint32_t>(o)) incl += t;
                }
                __builtin_amdgcn_wave_barrier();                 // previous sub-block's LDS reads are done (same wave, in order)
                off[lane] = incl - cnt;
                sbx[lane] = box;
                if (lane == 63) off[64] = incl;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t from = (cur > sub_base ? cur : sub_base) - sub_base;
                const uint32_t to = (s1 < sub_end ? s1 : sub_end) - sub_base;
                for (uint32_t slot = from + lane; slot < to; slot += 64) {
                    uint32_t q = 0;                               // last q with off[q] <= slot (skips empty ranks)
#pragma unroll
                    for (uint32_t step = 32; step >= 1; step >>= 1)
                        if (off[q + step] <= slot) q += step;
                    const uint32_t rel = slot - off[q], bx = sbx[q];
                    const uint32_t wx = ((bx >> 16) & 0xFFu) + 1u;
                    const uint32_t dy = rel / wx, dx = rel - dy * wx;
                    const uint32_t tx = (bx & 0xFFu) + dx, ty = ((bx >> 8) & 0xFFu) + dy;
                    pair_key[sub_base + slot] = static_cast<uint16_t>(ty * tiles_x + tx);
                    pair_val[sub_base + slot] = b * GSX_BIN_TILE + sb * SUB + q;   // the splat's rank inside the window
                }
                cur = sub_base + to;
            }
        }
#ifdef GSX_DEBUG_TILETIME
        if (lane == 0 && w < 65536u) {      // (tools/tile_times.py: start, end, sub-block visits | blocks << 32 of every slice)
            dbg_slices[w * 4 + 0] = dbg_t0;
            dbg_slices[w * 4 + 1] = wall_clock64();
            dbg_slices[w * 4 + 2] = dbg_visits | (static_cast<unsigned long long>(dbg_blocks) << 32);
            dbg_slices[w * 4 + 3] = 1;
        }
#endif
    }
}

// 64-bit exclusive scan of one value per thread over 1024 threads
__device__ __forceinline__ unsigned long long gsx_scan1024_u64(unsigned long long v, unsigned long long* s_w, unsigned long long* total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long t = __shfl_up(incl, off);
        if (lane >= static_cast<uint32_t>(off)) incl += t;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    unsigned long long wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const unsigned long long t = s_w[w];
        if (static_cast<uint32_t>(w) < wave) wave_off += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return wave_off + incl - v;
}

// K4s.  One workgroup: exclusive scan of the iteration's block sums (in place), the number of leading
// blocks whose pairs fit `budget`, and the cursor of the next iteration.
__global__ __launch_bounds__(1024) void k_chunk_scan(GsxFrameCtl* __restrict__ ctl, uint32_t iter, uint32_t window,
                                                     uint32_t nb, uint32_t* __restrict__ sums, uint32_t budget,
                                                     uint32_t* __restrict__ aux) {
    constexpr int PER = 8;                           // consecutive block sums per thread
    __shared__ unsigned long long s_w[16];
    __shared__ unsigned long long s_carry;
    __shared__ uint32_t s_fit, s_pairs;
    if (threadIdx.x == 0) { s_carry = 0ull; s_fit = 0u; s_pairs = 0u; }
    __syncthreads();
    const bool dead = ctl->live_tiles == 0u;         // every count block wrote 0: the window is consumed whole
    // (read here, beside the block sums, not in the tail)
    const uint32_t cur = ctl->cursor[iter], m = ctl->m, tp = ctl->total_pairs, mp = ctl->max_pairs, used = ctl->iters_used,
                   dead_at = ctl->dead_at;
    if (dead && threadIdx.x == 0) s_fit = nb;
    for (uint32_t b0 = 0; b0 < nb && !dead; b0 += 1024 * PER) {
        const unsigned long long carry = s_carry;
        if (carry > budget) break;                   // uniform: nothing behind this point fits
        uint32_t v[PER];
        unsigned long long sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t i = b0 + threadIdx.x * PER + j;
            v[j] = (i < nb) ? sums[i] : 0u;
            sum += v[j];
        }
        unsigned long long total;
        unsigned long long run = carry + gsx_scan1024_u64(sum, s_w, &total);
        uint32_t fit = 0, pairs = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t i = b0 + threadIdx.x * PER + j;
            if (i < nb) {
                sums[i] = static_cast<uint32_t>(run < 0xFFFFFFFFull ? run : 0xFFFFFFFFull);
                if (run + v[j] <= budget) { ++fit; pairs = static_cast<uint32_t>(run + v[j]); }   // fitting blocks form a prefix
            }
            run += v[j];
        }
        if (fit) { atomicAdd(&s_fit, fit); atomicMax(&s_pairs, pairs); }
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + total;
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // (every word of *ctl the tail needs was read at the top: behind a store to *ctl the compiler must assume the next load
        //  aliases it, and five dependent round trips to L2 were a third of this kernel)
        const uint32_t fit = s_fit, pairs = s_pairs;
        const uint32_t remaining = m - cur, win = window < remaining ? window : remaining;
        const unsigned long long take64 = static_cast<unsigned long long>(fit) * GSX_BIN_TILE;
        const uint32_t take = take64 < win ? static_cast<uint32_t>(take64) : win;
        ctl->n_blocks = fit;
        ctl->pairs = pairs;
        aux[0] = cur;                                   // for k_raster, off the control block's line
        aux[1] = m;
        aux[32] = pairs;                                // the iteration's pair count, cursor and m on a line of their own that no
        aux[33] = cur;                                  // atomic ever touches: k_raster reads them through a const pointer, i.e.
        aux[34] = m;                                    // with scalar loads that do not queue behind the retiring tiles' atomics
        aux[17] = 0u;                                   // tiles this iteration's rasterizer finds live although capped ...
        aux[18] = 0xFFFFFFFFu;                          // ... and the smallest of their caps (in 1024 ranks)
        if (iter == 0u && cur == 0u) { aux[2] = pairs; aux[3] = take; }   // density of the nearest window, for the host's schedule
        ctl->cursor[iter + 1] = cur + take;
        ctl->total_pairs = (tp + pairs < tp) ? 0xFFFFFFFFu : tp + pairs;
        if (pairs > mp) ctl->max_pairs = pairs;
        if (pairs) ctl->iters_used = used + 1u;
        if (take == 0u && win != 0u) ctl->stall = 1u;   // the nearest block alone exceeds the budget
        if (dead && dead_at == 0u) ctl->dead_at = cur + 1u;
    }
}

// K4c (stable radix sort of the pairs by 16-bit tile id) uses gsx_radix.hpp.

// lower_bound(sorted pair keys, target) by one wave: 64-ary search, one probe per lane and round (four dependent
// loads for 8 M pairs).  Every lane returns the result.
__device__ __forceinline__ uint32_t gsx_wave_lower_bound(const uint16_t* __restrict__ keys, uint32_t n, uint32_t target, uint32_t lane) {
    uint32_t lo = 0, len = n;                           // the answer lies in [lo, lo + len]
    while (len > 0u) {
        const uint32_t end = lo + len;
        const uint32_t step = (len + 63u) / 64u;
        const uint32_t idx = lo + (lane + 1u) * step - 1u;                    // last key of this lane's segment
        const bool lt = idx < end && static_cast<uint32_t>(keys[idx]) < target;
        const uint32_t cnt = static_cast<uint32_t>(__popcll(__builtin_amdgcn_ballot_w64(lt)));   // leading segments entirely below
        lo += cnt * step;
        if (lo >= end) return end;
        len = min(step - 1u, end - lo);                 // (the segment's last key is >= target, or the segment is cut by `end`)
    }
    return lo;
}

// ---------------------------------------------------------------------------
// K5: tile rasterizer.  One 128-thread workgroup per 16x16 tile, each wave a
// 16x8 half (two pixels per lane); the tile's pair list of this iteration is staged through
// LDS in batches (a 32-B record and a 16-B colour gather per thread) and every thread walks the batch,
// near -> far, on top of the accumulators the previous iterations left.
// ---------------------------------------------------------------------------
#define GSX_RASTER_NT 128            // threads per tile: 2 waves x 64 lanes x 2 pixels
#define GSX_RASTER_BATCH 128
#ifndef GSX_T_EPS
#define GSX_T_EPS (1.0f / 2048.0f)
#endif

// wgsl:75-117
__device__ __forceinline__ uint32_t gsx_bayer(uint32_t x, uint32_t y, uint32_t bsize, uint32_t state) {
    uint32_t s = state % 24u, fac[3], arr[4] = {0, 1, 2, 3};
    fac[0] = s / 6u; s -= fac[0] * 6u;
    fac[1] = s / 2u; s -= fac[1] * 2u;
    fac[2] = s;
#pragma unroll
    for (uint32_t i = 0; i < 3; ++i) {
        const uint32_t j = fac[i] + i;
        uint32_t ai = 0, aj = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) { if (k == i) ai = arr[k]; if (k == j) aj = arr[k]; }
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) { if (k == i) arr[k] = aj; else if (k == j) arr[k] = ai; }
    }
    uint32_t value = 0, cur = bsize, cx = x % bsize, cy = y % bsize, k = 1;
    while (cur > 1u) {
        cur >>= 1;
        const uint32_t cellx = cx / cur, celly = cy / cur;
        cx %= cur; cy %= cur;
        const uint32_t sel = (cellx * 2u + celly) & 3u;
        value += (sel == 0 ? arr[0] : sel == 1 ? arr[1] : sel == 2 ? arr[2] : arr[3]) * k;
        k *= 4u;
    }
    return value;
}

typedef float gsx_f2 __attribute__((ext_vector_type(2)));   // two pixels of one lane: arithmetic lowers to v_pk_*_f32

__device__ __forceinline__ uint32_t gsx_pack_rgba8(float r, float g, float b) {
    const uint32_t r8 = static_cast<uint32_t>(fminf(fmaxf(r, 0.0f), 1.0f) * 255.0f + 0.5f);
    const uint32_t g8 = static_cast<uint32_t>(fminf(fmaxf(g, 0.0f), 1.0f) * 255.0f + 0.5f);
    const uint32_t b8 = static_cast<uint32_t>(fminf(fmaxf(b, 0.0f), 1.0f) * 255.0f + 0.5f);
    return r8 | (g8 << 8) | (b8 << 16) | 0xFF000000u;
}

// One 128-thread workgroup per 16x16 tile.  Wave w owns the 16x8 half (rows 8w .. 8w+7); a lane owns the two
// horizontally adjacent pixels (2*(lane&7), 2*(lane&7)+1) of row lane>>3, so that every arithmetic step of the
// shading loop is one packed-f32 instruction for both pixels (measured: the loop is VALU-bound).
template <int MODE /*0 blend, 1 threshold (bayerSize 10), 2 bayer dither*/, bool COUNT /*profiling level 1: count shades and blends*/>
__global__ __launch_bounds__(GSX_RASTER_NT) void k_raster(const uint16_t* __restrict__ pair_key, const uint32_t* __restrict__ pair_val,
                                                          const float4* __restrict__ rank_rec, uint32_t iter, GsxViewParams vp, uint32_t first,
                                                          uint32_t* __restrict__ tile_state, uint32_t* __restrict__ live_bits,
                                                          float4* __restrict__ accum, GsxFrameCtl* __restrict__ ctl,
                                                          uint32_t* __restrict__ aux, const uint32_t* __restrict__ info /* aux + 32: pairs, cursor, m */,
                                                          uint8_t* __restrict__ fb,
                                                          unsigned long long* __restrict__ stat, uint32_t* __restrict__ tile_depth,
                                                          const uint8_t* __restrict__ tile_band /* null: no caps */, uint32_t redo) {
    constexpr uint32_t NW = GSX_RASTER_NT / 64;                   // 2 waves = 2 halves
    unsigned long long n_shades = 0, n_blends = 0, n_staged = 0;  // COUNT: (wave, splat) visits / pixels blended / pairs staged (wave 0)
    static_assert(GSX_RASTER_BATCH == GSX_RASTER_NT, "one staged splat per thread");
    __shared__ float4 s_a[GSX_RASTER_BATCH];  // cx cy ex ey
    __shared__ float4 s_m[GSX_RASTER_BATCH];  // m00 m10 m01 m11
    __shared__ float4 s_c[GSX_RASTER_BATCH];  // r g b a
    __shared__ uint32_t s_rank[MODE == 2 ? GSX_RASTER_BATCH : 1];
    __shared__ uint64_t s_mask[NW][NW];        // [half][staging wave]: splats of the batch whose quad touches the half
    __shared__ int s_live[2];
    __shared__ uint32_t s_range[2];
    const uint32_t tile = blockIdx.x;
    const uint32_t state = tile_state[tile];           // 0 untouched, 1 accumulators valid, 2 retired
    if (state == 2u) return;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // The tile's rank cap.  The binner decides per tile BOX, so a list may hold pairs from the wrong side of its tile's cap;
    // what is composited is decided here, per pair: ranks below the cap in the frame's normal iterations, ranks from the cap
    // on in a redo continuation - every pair exactly once.
    unsigned long long cap = ~0ull;
    if (tile_band) {
        const uint32_t band = tile_band[tile];
        if (band < GSX_BAND_INF) cap = static_cast<unsigned long long>(gsx_band_lo(band)) << 10;
        if (redo && band >= GSX_BAND_INF) return;      // nothing was withheld from this tile
    }
    // this tile's slice of the iteration's sorted pair list: wave 0 finds its begin, wave 1 its end (the pair count comes
    // from the aux words, on a cache line no atomic touches)
    {
        const uint32_t at = gsx_wave_lower_bound(pair_key, info[0], tile + wave, lane);
        if (lane == 0) s_range[wave] = at;
    }
    __syncthreads();
    const uint32_t begin = s_range[0], end = s_range[1];
    if (begin == end && !first) {                      // nothing new for this tile in this iteration
        // (still live: if its cap withheld ranks the frame reaches, the host has to know - see the tail of the kernel)
        if (tid == 0 && !redo && cap < info[2]) { atomicAdd(&aux[17], 1u); atomicMin(&aux[18], static_cast<uint32_t>(cap >> 10)); }
        return;
    }
    const uint32_t ttx = tile % vp.tiles_x, tty = tile / vp.tiles_x;
    const uint32_t tx = ttx, ty = tty + vp.tile_y0;
    const uint32_t px = tx * GSX_TILE_W + (lane & 7) * 2, py = ty * GSX_TILE_H + wave * 8 + (lane >> 3);
    const gsx_f2 fx = {static_cast<float>(px) + 0.5f, static_cast<float>(px) + 1.5f};
    const float fy = static_cast<float>(py) + 0.5f;
    const bool row_in = (py >= vp.y0) && (py < vp.y1);
    const bool in0 = row_in && (px < static_cast<uint32_t>(vp.width)), in1 = row_in && (px + 1 < static_cast<uint32_t>(vp.width));
#ifdef GSX_DEBUG_COUNT
    unsigned long long dbg_shaded = 0, dbg_hits = 0, dbg_useful = 0;
#endif
#ifdef GSX_DEBUG_TILETIME     // (tools/tile_times.py: start / end of every retired tile on the 100 MHz wall clock, parked in its unused accumulators)
    const unsigned long long dbg_t0 = wall_clock64();
#endif
    gsx_f2 T = {in0 ? 1.0f : 0.0f, in1 ? 1.0f : 0.0f}, cr = {0.0f, 0.0f}, cg = {0.0f, 0.0f}, cb = {0.0f, 0.0f};
    float4* acc = accum + static_cast<uint64_t>(tile) * (GSX_TILE_W * GSX_TILE_H) + tid * 2;
    if (state == 1u) {
        const float4 a0 = acc[0], a1 = acc[1];
        cr = gsx_f2{a0.x, a1.x}; cg = gsx_f2{a0.y, a1.y}; cb = gsx_f2{a0.z, a1.z}; T = gsx_f2{a0.w, a1.w};
    }
    const float cut = vp.cut;
    float cut_v = cut;                               // (kept in a vector register: the loop's select would re-copy the scalar every visit)
    asm volatile("" : "+v"(cut_v));
    gsx_f2 cutT = {T.x >= GSX_T_EPS ? cut_v : -1.0f, T.y >= GSX_T_EPS ? cut_v : -1.0f};   // per-pixel hit threshold (see the shading loop)
    const uint32_t rank_hi = MODE == 2 ? info[2] - info[1] : 0u;   // m - cursor
    // pixel-centre rectangle of half 0; half w is offset by 8 rows
    const float tx0f = static_cast<float>(tx * GSX_TILE_W) + 0.5f, ty0f = static_cast<float>(ty * GSX_TILE_H) + 0.5f;
    // near -> far: this iteration's list of the tile is already in that order
    uint32_t lo_done = begin;                          // end of the last batch this tile consumed
    for (uint32_t lo = begin; lo < end;) {
        const uint32_t nb = min(static_cast<uint32_t>(GSX_RASTER_BATCH), end - lo);
        lo_done = lo + nb;
        if (COUNT && wave == 0) n_staged += nb;
        __syncthreads();  // previous batch fully consumed
        if (tid == 0) s_live[0] = 0;
        // stage: thread j gathers splat j of the batch and tests its quad against the two 16x8 halves
        // (separating axes: the pixel axes and the quad's u / v axes, with a rounding margin so that no
        // pixel the per-pixel test below would accept is ever dropped)
        bool valid = tid < nb;
        float4 a = make_float4(0.0f, 0.0f, -1.0f, -1.0f), m = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const uint32_t pv = valid ? pair_val[lo + tid] : 0u;     // rank inside the iteration's window (0 = nearest)
        if (tile_band) {                                         // (uniform) which side of the tile's cap this pair is on
            const unsigned long long ar = static_cast<unsigned long long>(info[1]) + pv;
            valid = valid && (redo ? ar >= cap : ar < cap);
        }
        if (valid) {
            if (MODE == 2) s_rank[tid] = rank_hi - 1u - pv;      // the draw's instance index = rank in the sorted order
            const float4* src = rank_rec + static_cast<uint64_t>(pv) * 3;
            a = src[0]; m = src[1];
            float4 col = src[2];
            col.w = __builtin_amdgcn_logf(col.w);                // log2(opacity): folded into the exponent of the falloff (log2(0) = -inf -> alpha 0)
            s_a[tid] = a; s_m[tid] = make_float4(m.x, m.z, m.y, m.w); s_c[tid] = col;    // (m00 m10 | m01 m11: the dy terms of u and v are one packed multiply)
        }
        // (computed by every lane and masked with `valid` explicitly: lanes past the batch must
        // contribute 0 bits to the ballots below)
        bool hit[NW];
        {
            const float amx = fabsf(m.x), amy = fabsf(m.y), amz = fabsf(m.z), amw = fabsf(m.w);
            const float dxc = (tx0f + 7.5f) - a.x, adx = fabsf(dxc) + 7.5f;
            const bool boxx = !(a.x + a.z < tx0f || a.x - a.z > tx0f + 15.0f);
#pragma unroll
            for (uint32_t q = 0; q < NW; ++q) {
                const float y0 = ty0f + static_cast<float>(q * 8);
                const bool box = boxx && !(a.y + a.w < y0 || a.y - a.w > y0 + 7.0f);
                const float dyc = (y0 + 3.5f) - a.y, ady = fabsf(dyc) + 3.5f;
                const float uc = m.x * dxc + m.y * dyc, vc = m.z * dxc + m.w * dyc;
                const float ur = amx * 7.5f + amy * 3.5f + 1.0e-5f * (amx * adx + amy * ady) + 1.0e-6f;
                const float vr = amz * 7.5f + amw * 3.5f + 1.0e-5f * (amz * adx + amw * ady) + 1.0e-6f;
                hit[q] = valid && box && (fabsf(uc) - ur <= cut) && (fabsf(vc) - vr <= cut);
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < NW; ++q) {
            const uint64_t bal = __builtin_amdgcn_ballot_w64(hit[q]);
            if (lane == 0) s_mask[q][wave] = bal;
        }
        __syncthreads();
        bool wave_live = __builtin_amdgcn_ballot_w64(T.x >= GSX_T_EPS || T.y >= GSX_T_EPS) != 0;
        for (uint32_t w = 0; w < NW && wave_live; ++w) {
            const uint64_t mk = s_mask[wave][w];
            // (readfirstlane returns int: convert each half to u32 BEFORE widening, or bit 31 sign-extends)
            const uint32_t mk_lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(mk))));
            const uint32_t mk_hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(mk >> 32))));
            // The mask is walked in four 16-bit parts; before each part that holds a splat the wave asks whether all its 128
            // pixels are saturated.  The walk is scalar code: the shading loop issues about as many scalar as vector
            // instructions, and the CU's one scalar unit serves a SIMD once in four cycles - as often as its vector unit.
#pragma unroll 1
            for (uint32_t part = 0; part < 4u && wave_live; ++part) {
                uint32_t bits = ((part & 2u ? mk_hi : mk_lo) >> ((part & 1u) * 16u)) & 0xFFFFu;
                if (bits == 0u) continue;
                if (__builtin_amdgcn_ballot_w64(T.x >= GSX_T_EPS || T.y >= GSX_T_EPS) == 0) { wave_live = false; break; }   // all 128 pixels saturated
                const uint32_t jbase = w * 64u + part * 16u;
                while (bits) {
                    const uint32_t j = jbase + static_cast<uint32_t>(__builtin_ctz(bits));
                    bits &= bits - 1u;
                    const float4 sa = s_a[j];
                    const float4 sm = s_m[j];
                    const gsx_f2 dx = fx - sa.x;
                    const float dy = fy - sa.y;
                    const gsx_f2 tdy = gsx_f2{sm.z, sm.w} * dy;
                    const gsx_f2 u = sm.x * dx + tdy.x, v = sm.y * dx + tdy.y;
                    // A pixel is FROZEN at the first splat that leaves its transmittance below GSX_T_EPS: nothing behind
                    // that splat contributes, whenever the wave / tile happens to stop.  This makes the frame a pure
                    // function of (scene, permutation, uniforms): iteration windows, the history-driven schedule, the
                    // pair budget and the band a tile is rendered in only decide WHEN work is skipped, never what a
                    // pixel holds (skipped contribution < GSX_T_EPS per channel = 0.13/255).
                    // cutT = the quad's cutOff while the pixel is live, -1 once it is frozen: inside the quad AND live is
                    // one max and one compare per pixel.  (Records are finite: gsx_project culls what is not.)
                    const bool h0 = fmaxf(fabsf(u.x), fabsf(v.x)) <= cutT.x, h1 = fmaxf(fabsf(u.y), fabsf(v.y)) <= cutT.y;
#ifdef GSX_DEBUG_COUNT
                    ++dbg_shaded;
                    dbg_hits += __popcll(__builtin_amdgcn_ballot_w64(h0)) + __popcll(__builtin_amdgcn_ballot_w64(h1));
                    dbg_useful += __popcll(__builtin_amdgcn_ballot_w64(h0 && T.x >= GSX_T_EPS)) + __popcll(__builtin_amdgcn_ballot_w64(h1 && T.y >= GSX_T_EPS));
#endif
                    if (COUNT) {
                        ++n_shades;
                        n_blends += __popcll(__builtin_amdgcn_ballot_w64(h0)) + __popcll(__builtin_amdgcn_ballot_w64(h1));
                    }
                    // (no branch around the blend: nearly every visit hits some lane, and a branch costs the loop three scalar
                    //  instructions and a fetch bubble - measured 0.173 -> 0.152 ms for the launch on its own)
                    {
                        const float4 c = s_c[j];      // c.w = log2(opacity)
                        // opacity * exp(-r2/2) = 2^(r2 * -log2(e)/2 + log2(opacity)): one packed FMA for both pixels, then v_exp_f32
                        const gsx_f2 e2 = (u * u + v * v) * -0.72134752044448170368f + c.w;
                        gsx_f2 alpha = {h0 ? __builtin_amdgcn_exp2f(e2.x) : 0.0f, h1 ? __builtin_amdgcn_exp2f(e2.y) : 0.0f};
                        if (MODE == 1) { alpha.x = (alpha.x > vp.bayer_scale) ? 1.0f : 0.0f; alpha.y = (alpha.y > vp.bayer_scale) ? 1.0f : 0.0f; }
                        if (MODE == 2) {
                            const float lim = static_cast<float>(vp.bayer_size * vp.bayer_size + 1u);
                            const float bv0 = static_cast<float>(gsx_bayer(px, py, vp.bayer_size, s_rank[j]));
                            const float bv1 = static_cast<float>(gsx_bayer(px + 1, py, vp.bayer_size, s_rank[j]));
                            alpha.x = (!h0 || (alpha.x * vp.bayer_scale) * lim < bv0 + 1.0f) ? 0.0f : 1.0f;
                            alpha.y = (!h1 || (alpha.y * vp.bayer_scale) * lim < bv1 + 1.0f) ? 0.0f : 1.0f;
                        }
                        const gsx_f2 wgt = T * alpha;
                        cr += wgt * c.x; cg += wgt * c.y; cb += wgt * c.z;
                        T -= wgt;  // T *= (1 - alpha)
                        cutT = gsx_f2{T.x >= GSX_T_EPS ? cut_v : -1.0f, T.y >= GSX_T_EPS ? cut_v : -1.0f};
                    }
                }
            }
        }
        wave_live = __builtin_amdgcn_ballot_w64(T.x >= GSX_T_EPS || T.y >= GSX_T_EPS) != 0;
        if (lane == 0 && wave_live) s_live[0] = 1;
        lo += nb;
        __syncthreads();
        if (!s_live[0]) break;  // every pixel of the tile is saturated
    }
#ifdef GSX_DEBUG_COUNT
    if (lane == 0) {
        atomicAdd(&ctl->dbg[0], dbg_shaded); atomicAdd(&ctl->dbg[1], dbg_hits); atomicAdd(&ctl->dbg[2], dbg_useful);
        atomicAdd(&ctl->dbg[3], (end - begin));
    }
#endif
    if (COUNT && lane == 0) {   // sharded over 64 cache lines: thousands of waves adding to ONE word would serialize for ~100 us
        unsigned long long* shard = stat + (tile & 63u) * 16u;
        atomicAdd(shard, n_shades);
        atomicAdd(shard + 1, n_blends);
        if (wave == 0) atomicAdd(shard + 2, n_staged);
    }
    // retire the tile or park its accumulators for the next iteration
    __syncthreads();
    if (tid == 0) s_live[1] = 0;
    __syncthreads();
    const bool any_live = __builtin_amdgcn_ballot_w64(T.x >= GSX_T_EPS || T.y >= GSX_T_EPS) != 0;   // every lane votes: keep it outside the lane-0 branch
    if (lane == 0 && any_live) s_live[1] = 1;
    __syncthreads();
    if (s_live[1]) {
        acc[0] = make_float4(cr.x, cg.x, cb.x, T.x);
        acc[1] = make_float4(cr.y, cg.y, cb.y, T.y);
        if (tid == 0) {
            tile_state[tile] = 1u;
            // live and capped below the frame's depth: possibly starved
            if (!redo && cap < info[2]) { atomicAdd(&aux[17], 1u); atomicMin(&aux[18], static_cast<uint32_t>(cap >> 10)); }
        }
    } else if (tid == 0) {
#ifdef GSX_DEBUG_TILETIME
        reinterpret_cast<unsigned long long*>(accum + static_cast<uint64_t>(tile) * (GSX_TILE_W * GSX_TILE_H))[0] = dbg_t0;
        reinterpret_cast<unsigned long long*>(accum + static_cast<uint64_t>(tile) * (GSX_TILE_W * GSX_TILE_H))[1] = wall_clock64();
#endif
        tile_state[tile] = 2u;
        atomicAnd(&live_bits[tty * ((vp.tiles_x + 31u) / 32u) + (ttx >> 5)], ~(1u << (ttx & 31u)));
        atomicSub(&ctl->live_tiles, 1u);
        // how deep (in ranks from the near end) this tile had to look: next frame's schedule is sized from the maximum
        // (aux lives on its own cache lines, see gsx_internal.h)
        if (lo_done > begin) {
            const uint32_t depth = info[1] + pair_val[lo_done - 1] + 1u;
            atomicMax(&aux[16], depth);
            tile_depth[tile] = depth;                  // next frame's cap for this tile (0 = never retired)
        }
    }
    // the framebuffer always holds the composite of everything drawn so far over the clear colour
    uint32_t* row = reinterpret_cast<uint32_t*>(fb) + static_cast<uint64_t>(py) * static_cast<uint32_t>(vp.width);
    if (in0) row[px] = gsx_pack_rgba8(cr.x + T.x * 0.05f, cg.x + T.x * 0.05f, cb.x + T.x * 0.05f);
    if (in1) row[px + 1] = gsx_pack_rgba8(cr.y + T.y * 0.05f, cg.y + T.y * 0.05f, cb.y + T.y * 0.05f);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static void mul44_glm(const float* a, const float* b, float* out) {
    // glm/detail/type_mat4x4.inl:630-648
    float r[16];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float t = a[0 * 4 + i] * b[j * 4 + 0];
            t = t + a[1 * 4 + i] * b[j * 4 + 1];
            t = t + a[2 * 4 + i] * b[j * 4 + 2];
            t = t + a[3 * 4 + i] * b[j * 4 + 3];
            r[j * 4 + i] = t;
        }
    for (int k = 0; k < 16; ++k) out[k] = r[k];
}

template <typename T>
static int ensure(gsx_ctx* c, T** p, uint64_t* cap, uint64_t need, bool exact = false) {
    if (need <= *cap && *p) return GSX_OK;
    if (std::getenv("GSX_TRACE")) std::fprintf(stderr, "[gsx] %.1f ensure: %llu -> %llu elements of %zu bytes\n", gsx_dbg_now(), (unsigned long long)*cap, (unsigned long long)need, sizeof(T));
    if (*p) GSX_HIP(c, hipFree(*p));
    *p = nullptr; *cap = 0;
    const uint64_t want = exact ? need : need + need / 4 + 1024;
    GSX_HIP(c, hipMalloc(reinterpret_cast<void**>(p), want * sizeof(T)));
    *cap = want;
    return GSX_OK;
}

// Pair buffers for a budget of `budget` pairs per iteration (2 x u16 keys + 2 x u32 payloads).
static int ensure_pairs(gsx_ctx* c, uint64_t budget) {
    if (budget > c->pair_alloc || !c->pair_key[0]) {
        for (int b = 0; b < 2; ++b) {
            if (c->pair_key[b]) { GSX_HIP(c, hipFree(c->pair_key[b])); c->pair_key[b] = nullptr; }
            if (c->pair_val[b]) { GSX_HIP(c, hipFree(c->pair_val[b])); c->pair_val[b] = nullptr; }
        }
        c->pair_alloc = 0;
        for (int b = 0; b < 2; ++b) {
            GSX_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->pair_key[b]), (budget + 16) * sizeof(uint16_t)));
            GSX_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->pair_val[b]), (budget + 16) * sizeof(uint32_t)));
        }
        c->pair_alloc = budget;
    }
    c->pair_cap = budget;
    return ensure(c, &c->pair_ws, &c->pair_ws_words, gsx_radix_workspace_words(GSX_RADIX_MAX_BLOCKS));
}

// One iteration of the frame loop: window of `window` ranks behind cursor[iter].
static int launch_iteration(gsx_ctx* c, uint32_t iter, uint32_t window, bool first) {
    const GsxFrameState& f = c->frame;
    const GsxViewParams& vp = *reinterpret_cast<const GsxViewParams*>(f.vp_blob);
    const uint32_t* sorted = f.sorted;
    const uint32_t nb = gsx_div_up(window, GSX_BIN_TILE);
    const uint32_t budget = static_cast<uint32_t>(c->pair_cap);
    if (nb > 0) {
        hipLaunchKernelGGL(k_chunk_count, dim3(nb), dim3(GSX_CNT_NT), 0, c->fstream, c->ctl, iter, window, sorted, c->geo,
                           c->live_bits, vp, c->scan_block_sums, c->sub_off, c->rank_box, c->rank_rec,
                           f.caps_on ? c->band_masks : nullptr, f.redo ? 1u : 0u);
        hipLaunchKernelGGL(k_chunk_scan, dim3(1), dim3(1024), 0, c->fstream, c->ctl, iter, window, nb, c->scan_block_sums, budget, c->aux);
        // slot-parallel: one wave per GSX_EMIT_SLICE pairs.  The pair count is only known on the device (a window of a few
        // thousand huge splats can fill the whole budget), so the grid is sized for the budget; surplus waves exit at once.
        uint32_t emit_grid = gsx_div_up(gsx_div_up(c->pair_cap, GSX_EMIT_SLICE), GSX_BIN_NT / 64);
        if (emit_grid > 4096u) emit_grid = 4096u;
        if (emit_grid < 1u) emit_grid = 1u;
        hipLaunchKernelGGL(k_chunk_emit, dim3(emit_grid), dim3(GSX_BIN_NT), 0, c->fstream, c->ctl, vp.tiles_x, c->scan_block_sums,
                           c->sub_off, c->rank_box, c->pair_key[0], c->pair_val[0]
#ifdef GSX_DEBUG_TILETIME
                           , reinterpret_cast<unsigned long long*>(c->accum) + (static_cast<uint64_t>(1) << 20)   // (8 MB into the accumulators: tiles 0 .. 2047 park there only while live)
#endif
                           );
    }
    // stable sort of the iteration's pairs by tile id; the pair count stays on the device
    const uint32_t* count_ptr = &c->ctl->pairs;
    uint32_t tile_bits = 1;
    while ((1u << tile_bits) < f.n_tiles) ++tile_bits;
    const GsxRadixPlan plan = gsx_radix_plan(tile_bits);
    for (int p = 0; p < plan.n_passes; ++p)
        if ((1u << plan.bits[p]) > static_cast<uint32_t>(GSX_PAIR_MAXD)) { c->err = "pair sort digit wider than GSX_PAIR_MAXD"; return GSX_E_INTERNAL; }
    // grid: one round of resident workgroups (sized for the budget: the pair count is device-side)
    const uint32_t g = gsx_radix_blocks(c->pair_cap, GSX_PAIR_TILE, c->pair_slots);
    const uint32_t cols = gsx_radix_columns(c->pair_slots);
    uint32_t* cnt = c->pair_ws;
    uint32_t* totals = cnt + static_cast<uint64_t>(GSX_RADIX_MAX_DIGITS) * cols;
    int cur = 0;
    for (int p = 0; p < plan.n_passes; ++p) {
        hipLaunchKernelGGL((k_radix_hist<uint16_t, GSX_PAIR_NT>), dim3(g), dim3(GSX_PAIR_NT), 0, c->fstream, c->pair_key[cur],
                           count_ptr, plan.shift[p], plan.bits[p], cnt, static_cast<uint32_t>(GSX_PAIR_TILE), cols);
        hipLaunchKernelGGL((k_radix_scan<0>), dim3(1u << plan.bits[p]), dim3(256), 0, c->fstream, cnt, count_ptr,
                           static_cast<uint32_t>(GSX_PAIR_TILE), cols, totals);
        hipLaunchKernelGGL((k_radix_scatter<uint16_t, GSX_PAIR_NT, GSX_PAIR_IPT, false, GSX_PAIR_MAXD, GSX_PAIR_WPE>), dim3(g), dim3(GSX_PAIR_NT), 0, c->fstream,
                           c->pair_key[cur], c->pair_val[cur], c->pair_key[cur ^ 1], c->pair_val[cur ^ 1], count_ptr, plan.shift[p],
                           plan.bits[p], cnt, totals, cols);
        cur ^= 1;
    }
    if (c->profiling && iter < GSX_MAX_SCHED && !f.continuing) GSX_HIP(c, hipEventRecord(c->ev[EV_IT0 + 2 * iter], c->fstream));
    const uint32_t fi = first ? 1u : 0u;
    const bool count = c->profiling && c->stat != nullptr;
#define GSX_LAUNCH_RASTER(M, CNT)                                                                                                       \
    hipLaunchKernelGGL((k_raster<M, CNT>), dim3(f.n_tiles), dim3(GSX_RASTER_NT), 0, c->fstream, c->pair_key[cur], c->pair_val[cur],    \
                       c->rank_rec, iter, vp, fi, c->tile_state, c->live_bits, c->accum, c->ctl, c->aux, c->aux + 32, c->fb_cur, c->stat,                 \
                       c->tile_depth[c->depth_cur], f.caps_on ? c->tile_band : nullptr, f.redo ? 1u : 0u)
    if (f.mode == 0) { if (count) GSX_LAUNCH_RASTER(0, true); else GSX_LAUNCH_RASTER(0, false); }
    else if (f.mode == 1) { if (count) GSX_LAUNCH_RASTER(1, true); else GSX_LAUNCH_RASTER(1, false); }
    else { if (count) GSX_LAUNCH_RASTER(2, true); else GSX_LAUNCH_RASTER(2, false); }
#undef GSX_LAUNCH_RASTER
    if (c->profiling && iter < GSX_MAX_SCHED && !f.continuing) GSX_HIP(c, hipEventRecord(c->ev[EV_IT0 + 2 * iter + 1], c->fstream));
    return GSX_OK;
}

// Everything queued so far for the frame in flight reads permutation buffer b: the sort that overwrites b next waits for this.
static void gsx_mark_frame_done(gsx_ctx* c) {
    for (uint32_t b = 0; b < GSX_PERM_RING; ++b)
        if (c->frame.sorted == c->perm[b] && c->ev_frame_done[b] && hipEventRecord(c->ev_frame_done[b], c->fstream) == hipSuccess)
            c->frame_done_rec[b] = true;
}

int gsx_launch_render(gsx_ctx* c, const gsx_uniforms* u, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1) {
    static_assert(sizeof(GsxViewParams) <= sizeof(c->frame.vp_blob), "vp_blob too small");
    const float* P = u->projectionMatrix;
    if (P[2] != 0.0f || P[3] != 0.0f || P[6] != 0.0f || P[7] != 0.0f) {
        c->err = "projection matrix couples clip z/w to x/y (column 0/1, rows 2/3 must be 0)";
        return GSX_E_UNSUPPORTED;
    }
    GsxViewParams vp;
    mul44_glm(u->viewMatrix, u->modelMatrix, vp.wt);  // wgsl:48
    for (int k = 0; k < 16; ++k) vp.proj[k] = P[k];
    vp.cut = u->cutOff;
    vp.cut_s = u->cutOff * u->splatSize;
    vp.width = static_cast<float>(w);
    vp.height = static_cast<float>(h);
    vp.y0 = y0; vp.y1 = y1;
    vp.tiles_x = gsx_div_up(w, GSX_TILE_W);
    vp.tile_y0 = y0 / GSX_TILE_H;
    vp.tiles_y = gsx_div_up(y1, GSX_TILE_H) - vp.tile_y0;
    vp.bayer_size = u->bayerSize;
    vp.bayer_scale = u->bayerScale;
    const uint32_t n_tiles = vp.tiles_x * vp.tiles_y;
    if (n_tiles > 65535u || vp.tiles_x > 256u || vp.tiles_y > GSX_MAX_TILES_Y) {
        c->err = "frame band larger than 4096 x 4096 pixels / 65535 tiles";
        return GSX_E_UNSUPPORTED;
    }
    const int mode = (u->bayerSize == 1u) ? 0 : (u->bayerSize == 10u ? 1 : 2);
    if (mode == 2 && !(u->bayerSize == 2u || u->bayerSize == 4u || u->bayerSize == 8u || u->bayerSize == 16u)) {
        c->err = "bayerSize must be 1, 10 or a power of two (the shader divides by zero otherwise)";
        return GSX_E_UNSUPPORTED;
    }
    const uint32_t m_cap = c->m_cap;

    int rc;
    if (c->fb_ext) {
        if (c->fb_ext_cap < static_cast<uint64_t>(w) * h * 4) {
            c->err = "render target set with gsx_set_framebuffer_dev is smaller than width*height*4 bytes";
            return GSX_E_INVALID;
        }
        c->fb_cur = c->fb_ext;
        c->fb_w = w; c->fb_h = h;
        // an exchange that still ships (or fills) this memory runs on the exchange stream: the frame waits for it on the device
        if ((rc = gsx_comm_guard(c, c->fstream, c->fb_cur, c->fb_cur + static_cast<uint64_t>(w) * h * 4)) != GSX_OK) return rc;
    } else {
        if ((rc = ensure(c, &c->fb, &c->fb_cap, static_cast<uint64_t>(w) * h * 4)) != GSX_OK) return rc;
        if ((rc = gsx_comm_guard(c, c->fstream, c->fb, c->fb + static_cast<uint64_t>(w) * h * 4)) != GSX_OK) return rc;
        if (c->fb_w != w || c->fb_h != h || c->fb_cur != c->fb) {
            GSX_HIP(c, hipMemsetAsync(c->fb, 0, static_cast<uint64_t>(w) * h * 4, c->fstream));
            c->fb_w = w; c->fb_h = h;
        }
        c->fb_cur = c->fb;
    }
    if ((rc = ensure(c, &c->tile_state, &c->tile_state_cap, static_cast<uint64_t>(n_tiles) + 2)) != GSX_OK) return rc;
    if ((rc = ensure(c, &c->accum, &c->accum_cap, static_cast<uint64_t>(n_tiles) * GSX_TILE_W * GSX_TILE_H)) != GSX_OK) return rc;
    // per-tile rank caps: usable when the slot's last frame had this geometry and the picture has moved by a few tiles at most
    int cap_radius = -1;
    bool cap_exact = false;
    {
        const bool same = c->hist_valid && c->hist_w == w && c->hist_h == h && c->hist_y0 == y0 && c->hist_y1 == y1 &&
                          c->tile_depth[c->depth_cur] && c->tile_depth_cap[c->depth_cur] >= n_tiles;
        if (c->caps_off_frames) --c->caps_off_frames;                      // (a redo continuation switched the caps off for a while)
        else if (same && c->opt_tile_caps && !c->opt_no_history && c->opt_chunk_first % 1024u == 0u && c->aabb_valid) {   // (cursors must stay multiples of 1024)
            // how far the scene moves on screen between the two frames: the worst corner (in pixels) of the INNER half of the
            // scene's bounding box - the box itself is stretched by outliers, and a camera inside it is the rule for them.
            // With a corner behind a camera (the camera stands inside the inner box) nearby splats move by more than any corner,
            // so only an unchanged view counts as steady then.  The estimate may be too small for splats close to the camera:
            // a prediction that fails costs a redo continuation and switches the caps off for a growing number of frames.
            float moved = 0.0f, dmat = 0.0f;
            bool ok = true, behind = false;
            for (int k = 0; k < 16; ++k) {
                dmat = fmaxf(dmat, fabsf(vp.wt[k] - c->hist_wt[k]));
                dmat = fmaxf(dmat, fabsf(vp.proj[k] - c->hist_proj[k]));
            }
            for (int k = 0; k < 8 && ok; ++k) {
                float p[3];
                for (int ax = 0; ax < 3; ++ax) {
                    const float mid = 0.5f * (c->aabb_min[ax] + c->aabb_max[ax]), half = 0.25f * (c->aabb_max[ax] - c->aabb_min[ax]);
                    p[ax] = ((k >> ax) & 1) ? mid + half : mid - half;
                }
                float scr[2][2];
                bool front = true;
                for (int which = 0; which < 2; ++which) {
                    const float* W = which ? vp.wt : c->hist_wt;
                    const float* Pm = which ? vp.proj : c->hist_proj;
                    float v[4], q[4];
                    for (int r = 0; r < 4; ++r) v[r] = W[r] * p[0] + W[4 + r] * p[1] + W[8 + r] * p[2] + W[12 + r];
                    for (int r = 0; r < 4; ++r) q[r] = Pm[r] * v[0] + Pm[4 + r] * v[1] + Pm[8 + r] * v[2] + Pm[12 + r] * v[3];
                    if (!(q[3] > 1.0e-3f)) { front = false; break; }
                    scr[which][0] = (q[0] / q[3] * 0.5f + 0.5f) * vp.width;
                    scr[which][1] = (0.5f - q[1] / q[3] * 0.5f) * vp.height;
                }
                if (!front) { behind = true; continue; }
                const float dx = scr[1][0] - scr[0][0], dy = scr[1][1] - scr[0][1];
                const float d = sqrtf(dx * dx + dy * dy);
                if (!(d < 1.0e9f)) { ok = false; break; }
                moved = d > moved ? d : moved;
            }
            if (behind && !(dmat <= 1.0e-6f)) ok = false;
            if (ok) {
                const int radius = 1 + static_cast<int>(ceilf(1.5f * moved / static_cast<float>(GSX_TILE_W)));
                if (radius <= static_cast<int>(c->opt_caps_radius ? c->opt_caps_radius : 4u)) cap_radius = radius;
                // Nothing the frame depends on has changed since the slot's last frame (a viewer at rest): every tile saturates at
                // exactly the rank it did then, so the cap is that rank itself - no neighbourhood, no factor, no margin.  (If the
                // caller did change something the matrices do not show, the redo continuation still completes the frame.)
                const float shade[4] = {vp.cut, vp.cut_s, static_cast<float>(vp.bayer_size), vp.bayer_scale};
                cap_exact = dmat == 0.0f && std::memcmp(shade, c->hist_shade, sizeof shade) == 0 && c->opt_tile_caps == 1;
            }
        }
    }
    const uint32_t depth_prev = c->depth_cur;
    c->depth_cur ^= 1u;
    if ((rc = ensure(c, &c->tile_depth[c->depth_cur], &c->tile_depth_cap[c->depth_cur], static_cast<uint64_t>(n_tiles) + 2)) != GSX_OK) return rc;
    if ((rc = ensure(c, &c->tile_band, &c->tile_band_cap, static_cast<uint64_t>(n_tiles) + 16)) != GSX_OK) return rc;
    if (cap_radius >= 0 &&
        (rc = ensure(c, &c->band_masks, &c->band_masks_cap, static_cast<uint64_t>(GSX_BANDS) * vp.tiles_y * GSX_LIVE_ROW_WORDS, true)) != GSX_OK)
        return rc;
    c->hist_valid = false;               // (set again when this frame has finished)
    {
        const uint64_t nb_max = static_cast<uint64_t>(gsx_div_up(m_cap, GSX_BIN_TILE)) + 1;
        if ((rc = ensure(c, &c->scan_block_sums, &c->rank_cap, nb_max)) != GSX_OK) return rc;
        if ((rc = ensure(c, &c->sub_off, &c->sub_off_cap, nb_max * (GSX_BIN_SUBS + 1))) != GSX_OK) return rc;
        if ((rc = ensure(c, &c->rank_box, &c->rank_box_cap, nb_max * GSX_BIN_TILE)) != GSX_OK) return rc;
        if ((rc = ensure(c, &c->rank_rec, &c->rank_rec_cap, nb_max * GSX_BIN_TILE * 3)) != GSX_OK) return rc;
    }
    // pair budget: what one iteration may emit (the device cuts iterations to fit it)
    uint64_t budget = c->opt_pair_budget;
    if (budget == 0) {
        budget = static_cast<uint64_t>(m_cap) * 16u;
        if (budget < (1u << 20)) budget = 1u << 20;
        if (budget > (32u << 20)) budget = 32u << 20;
        if (budget < c->pair_grown) budget = c->pair_grown;   // keep what an earlier frame had to grow to
    }
    if ((rc = ensure_pairs(c, budget)) != GSX_OK) return rc;

    GsxFrameState& f = c->frame;
    std::memcpy(f.vp_blob, &vp, sizeof(vp));
    f.mode = mode;
    f.sorted = c->perm[c->perm_draw];
    f.n_tiles = n_tiles;
    f.continuing = false;
    f.caps_on = cap_radius >= 0;
    f.redo = false;
    f.redos = 0;

    // (K3, the projection, runs inside k_chunk_count on the ranks each iteration reaches)
    if (c->profiling && c->stat) GSX_HIP(c, hipMemsetAsync(c->stat, 0, GSX_STAT_BYTES, c->fstream));
    if (c->profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_PROJ0], c->fstream));
    hipLaunchKernelGGL(k_frame_init, dim3(vp.tiles_y), dim3(256), 0, c->fstream, c->ctl, c->m_perm + c->perm_draw, c->live_bits,
                       c->tile_state, vp.tiles_x, vp.tiles_y, c->aux, c->tile_depth[depth_prev], c->tile_depth[c->depth_cur], c->tile_band,
                       c->band_masks, cap_exact ? 0 : cap_radius, cap_exact ? 256u : (c->opt_tile_caps == 2 ? 32u : 384u),
                       (cap_exact || c->opt_tile_caps == 2) ? 0u : 2048u);
    if (c->profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_PROJ1], c->fstream));

    // K4 + K5: near -> far iterations with geometrically growing rank windows.  Only as many ranks are scheduled
    // as recent frames needed before every tile was retired (plus a margin): what lies behind is hidden anyway, and if this
    // frame does need more, gsx_frame_finish runs the rest (correct, just not pre-queued).
    uint64_t need = c->need_pos ? 0 : ~0u;
    for (uint32_t k = 0; k < 8 && k < c->need_pos; ++k) need = c->need_hist[k] > need ? c->need_hist[k] : need;
    if (c->need_pos >= 2) {
        // a camera moving into the scene needs 20-30 % more ranks from one frame to the next, and the latest depth known here is
        // two frames old (two frames in flight): continue the trend of the last two, or the frame ends with live tiles and the
        // host has to run a continuation over everything that is left (5 of the 120 orbit frames of cfg 2 did)
        const uint64_t last = c->need_hist[(c->need_pos - 1u) % 8u], prev = c->need_hist[(c->need_pos - 2u) % 8u];
        if (last != ~0u && prev != ~0u && last > prev) {
            const uint64_t ahead = last + 2u * (last - prev);
            need = ahead > need ? ahead : need;
        }
    }
    if (need > m_cap) need = m_cap;
    c->frame.need_base = static_cast<uint32_t>(need);
    {   // margin: 12 % plus a quarter of the depth again (at least 2 K, at most 32 K ranks) — a fixed 32 K margin
        // would queue a whole dead iteration behind a dense foreground that saturates within ~15 K ranks
        uint64_t pad = need / 4;
        if (pad < 2048u) pad = 2048u;
        if (pad > 32768u) pad = 32768u;
        need = need + need / 8 + pad;
    }
    need = (need + 1023u) / 1024u * 1024u;   // iteration cursors stay multiples of 1024 until the end: a count block never straddles a cap band
    if (need > m_cap || c->opt_no_history) need = m_cap;
    uint32_t n_iters = 0;
    uint64_t covered = 0;
    // first window: 256 K ranks, fewer when the nearest splats are large on screen (history: pairs per rank of the last
    // frame's first window), so that the first iteration stays near 8 M pairs
    uint64_t win = c->opt_chunk_first ? c->opt_chunk_first : (256u << 10);
    if (!c->opt_chunk_first && !c->opt_no_history && c->density0 > 0.0f) {
        uint64_t wd = static_cast<uint64_t>(8.0e6f / c->density0);
        wd = (wd + 1023u) / 1024u * 1024u;
        if (wd < 2048u) wd = 2048u;
        if (wd < win) win = wd;
    }
    // With per-tile rank caps the windows no longer have to do the occlusion culling (a tile's list ends at its own predicted
    // depth, wherever the window ends), so the frame is ONE iteration: measured on cfg 2, 0.425 ms against 0.475 in two.
    if (f.caps_on && !c->opt_chunk_first && !c->opt_chunk_growth && need > win) win = need;
    const uint64_t growth = c->opt_chunk_growth ? c->opt_chunk_growth : 4u;
    // Behind a dense foreground the first window is small and the growth rule would queue two more iterations for the few
    // tiles it leaves live; when the last frame emitted only a few pairs behind its first window, one iteration takes all
    // the rest (an iteration's fixed cost is ~60 us; the live-tile mask keeps its pair count small whatever its length).
    // Hysteresis: the merged schedule emits somewhat more of those pairs than the split one it replaces.  Only for a steady
    // camera (moved less than a tenth of its distance to the scene centre since the last frame, as in an orbit): after a
    // jump to another pose the last frame says nothing about this one, and a mispredicted merge costs more than it saves
    // (measured on the 64-pose set of cfg 4: 1.35 -> 1.47 ms per view without this test).
    bool steady = false;
    if (c->have_last_cam && c->aabb_valid) {
        float d2 = 0.0f, r2 = 0.0f;
        for (int k = 0; k < 3; ++k) {
            const float d = c->cam[k] - c->last_cam[k], r = c->cam[k] - 0.5f * (c->aabb_min[k] + c->aabb_max[k]);
            d2 += d * d;
            r2 += r * r;
        }
        steady = d2 <= 0.01f * r2;
    }
    for (int k = 0; k < 3; ++k) c->last_cam[k] = c->cam[k];
    c->have_last_cam = true;
    const bool merge_rest = steady && !c->opt_chunk_growth && !c->opt_no_history && c->rest_pairs != ~0ull &&
                            c->rest_pairs <= (c->rest_merged ? (4ull << 20) : (2ull << 20)) && c->rest_pairs * 4u < c->pair_cap;
    c->rest_merged = merge_rest;
    do {
        uint64_t wnd = win;
        if (n_iters + 1 == GSX_MAX_SCHED || covered + wnd + wnd / 2 >= need || (merge_rest && n_iters >= 1)) wnd = need - covered;   // last one takes the rest
        if ((rc = launch_iteration(c, n_iters, static_cast<uint32_t>(wnd), n_iters == 0)) != GSX_OK) return rc;
        covered += wnd;
        win *= growth;
        ++n_iters;
    } while (covered < need);
    f.n_iters = n_iters;
    GSX_HIP(c, hipMemcpyAsync(c->ctl_host, c->ctl, GSX_AUX_OFFSET + 128, hipMemcpyDeviceToHost, c->fstream));   // control block + aux
    GSX_HIP(c, hipEventRecord(c->frame_ev, c->fstream));
    gsx_mark_frame_done(c);
    c->frame_seq = ++c->frame_counter;
    c->frame_pending = true;
    if (c->profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_RAST1], c->fstream));
    GSX_HIP(c, hipGetLastError());
    return GSX_OK;
}

// Waits for the frame queued by gsx_launch_render and, if the device could not fit every rank into
// the scheduled iterations (pair budget hit), keeps running iterations until the cursor reaches m.
void gsx_slot_switch(gsx_ctx* c) {
    GsxFrameSlot& a = *c;
    std::swap(a, c->parked);
    c->active_slot ^= 1u;
}

// Every queued frame, oldest first; the active slot is the same afterwards as before.
int gsx_frame_finish(gsx_ctx* c) {
    gsx_join_sort(c);      // whatever the caller queues next may read the permutation or rewrite the sort's inputs
    int rc = gsx_comm_join(c);   // ... or the pixels an exchange is still moving
    if (rc) return rc;
    if (c->parked.frame_pending && (!c->frame_pending || c->parked.frame_seq < c->frame_seq)) {
        gsx_slot_switch(c);
        rc = gsx_frame_finish_active(c);
        gsx_slot_switch(c);
        if (rc) return rc;
    }
    if ((rc = gsx_frame_finish_active(c))) return rc;
    if (c->parked.frame_pending) {
        gsx_slot_switch(c);
        rc = gsx_frame_finish_active(c);
        gsx_slot_switch(c);
    }
    return rc;
}

// Only the queued frames that render into [lo, hi), oldest first; the others stay in flight.
int gsx_frame_finish_touching(gsx_ctx* c, const uint8_t* lo, const uint8_t* hi) {
    if (!lo || hi <= lo) return GSX_OK;
    auto touches = [&](const GsxFrameSlot& s) {
        if (!s.frame_pending || !s.fb_cur) return false;
        const uint8_t* a = s.fb_cur;
        const uint8_t* b = a + static_cast<uint64_t>(s.fb_w) * s.fb_h * 4;
        return a < hi && lo < b;
    };
    const bool act = touches(*c);
    bool par = touches(c->parked);
    int rc = GSX_OK;
    if (par && (!act || c->parked.frame_seq < c->frame_seq)) {
        gsx_slot_switch(c);
        rc = gsx_frame_finish_active(c);
        gsx_slot_switch(c);
        if (rc) return rc;
        par = false;
    }
    if (act && (rc = gsx_frame_finish_active(c))) return rc;
    if (par) {
        gsx_slot_switch(c);
        rc = gsx_frame_finish_active(c);
        gsx_slot_switch(c);
    }
    return rc;
}

static int frame_finish_active_impl(gsx_ctx* c);

// Completes the active slot's queued frame (more iterations if it needs them), then - option deferred_readback - copies it into
// the host buffer its gsx_render call named.  The copy runs on the slot's own stream, which is idle by then, so the other
// slot's frame keeps running beside it.
int gsx_frame_finish_active(gsx_ctx* c) {
    const bool was_pending = c->frame_pending;
    const int rc = frame_finish_active_impl(c);
    uint8_t* dst = c->rb_dst;
    c->rb_dst = nullptr;
    if (rc != GSX_OK || !was_pending || !dst || !c->fb_cur) return rc;
    GSX_HIP(c, hipMemcpyAsync(dst + c->rb_off, c->fb_cur + c->rb_off, c->rb_bytes, hipMemcpyDeviceToHost, c->fstream));
    GSX_HIP(c, hipStreamSynchronize(c->fstream));
    return GSX_OK;
}

static int frame_finish_active_impl(gsx_ctx* c) {
    if (!c->frame_pending) return GSX_OK;
    GSX_HIP(c, hipEventSynchronize(c->frame_ev));
    GsxFrameState& f = c->frame;
    uint32_t last = f.n_iters;
    c->density0 = c->aux_host[3] >= 1024u ? static_cast<float>(c->aux_host[2]) / static_cast<float>(c->aux_host[3]) : 0.0f;
    static const bool trace = std::getenv("GSX_TRACE") != nullptr;
    for (int guard = 0; ; ++guard) {
        const GsxFrameCtl& h = *c->ctl_host;
        const uint32_t cur = h.cursor[last];
        if (trace)
            std::fprintf(stderr, "[gsx] %.1f slot-stream %p main %d frame: iters=%u last=%u cursor=%u m=%u live=%u pairs=%u total=%u max=%u used=%u stall=%u budget=%llu opt=%llu grown=%llu alloc=%llu ctx=%p\n",
                         gsx_dbg_now(), (void*)c->fstream, (int)(c->fstream == c->stream), f.n_iters, last, cur, h.m, h.live_tiles, h.pairs, h.total_pairs, h.max_pairs, h.iters_used, h.stall,
                         static_cast<unsigned long long>(c->pair_cap), static_cast<unsigned long long>(c->opt_pair_budget),
                         static_cast<unsigned long long>(c->pair_grown), static_cast<unsigned long long>(c->pair_alloc), static_cast<void*>(c));
        if (cur >= h.m || h.live_tiles == 0u) {
            // the frame has reached its end.  Tiles that are still live although their rank cap withheld pairs get them now:
            // one continuation from the smallest such cap, with the complementary band masks (k_chunk_count, redo = 1)
            if (h.live_tiles != 0u && f.caps_on && !f.redo && c->aux_host[17] != 0u) {
                const uint64_t start64 = static_cast<uint64_t>(c->aux_host[18]) << 10;
                if (start64 < h.m) {
                    const uint32_t start = static_cast<uint32_t>(start64), zero = 0;
                    f.redo = true;
                    f.continuing = true;
                    ++f.redos;
                    c->caps_backoff = c->caps_backoff ? (c->caps_backoff < 64u ? c->caps_backoff * 2u : 64u) : 4u;
                    c->caps_off_frames = c->caps_backoff;
                    int rc;
                    if ((rc = gsx_behind_sample(c, c->fstream)) != GSX_OK) { c->frame_pending = false; return rc; }   // (a sampled sort is timed alone)
                    GSX_HIP(c, hipMemcpyAsync(&c->ctl->cursor[0], &start, 4, hipMemcpyHostToDevice, c->fstream));
                    GSX_HIP(c, hipMemcpyAsync(&c->ctl->stall, &zero, 4, hipMemcpyHostToDevice, c->fstream));
                    if ((rc = launch_iteration(c, 0, h.m - start, false)) != GSX_OK) { c->frame_pending = false; return rc; }
                    GSX_HIP(c, hipMemcpyAsync(c->ctl_host, c->ctl, GSX_AUX_OFFSET + 128, hipMemcpyDeviceToHost, c->fstream));
                    gsx_mark_frame_done(c);
                    GSX_HIP(c, hipStreamSynchronize(c->fstream));
                    last = 1;
                    continue;
                }
            }
            break;
        }
        if (guard > 100000) { c->err = "frame did not complete"; c->frame_pending = false; return GSX_E_INTERNAL; }
        f.continuing = true;
        int rc;
        if (h.stall) {
            // the nearest 1024 ranks alone exceed the budget: double the pair buffers
            const uint64_t bigger = c->pair_cap * 2u;
            if (bigger > 0x7FFFFFFFull) { c->err = "tile pair budget exhausted"; c->frame_pending = false; return GSX_E_NOMEM; }
            if ((rc = ensure_pairs(c, bigger)) != GSX_OK) { c->frame_pending = false; return rc; }
            if (c->opt_pair_budget) c->opt_pair_budget = bigger; else c->pair_grown = bigger;
        }
        // restart the cursor ring at slot 0 and run one more iteration over everything that is left
        const uint32_t zero = 0;
        if ((rc = gsx_behind_sample(c, c->fstream)) != GSX_OK) { c->frame_pending = false; return rc; }   // (a sampled sort is timed alone)
        GSX_HIP(c, hipMemcpyAsync(&c->ctl->cursor[0], &c->ctl->cursor[last], 4, hipMemcpyDeviceToDevice, c->fstream));
        GSX_HIP(c, hipMemcpyAsync(&c->ctl->stall, &zero, 4, hipMemcpyHostToDevice, c->fstream));
        // (half as many ranks again as the frame has consumed, at least 256 K: the few tiles that are still live rarely need
        //  more, and a window over everything that is left projects millions of hidden ranks; the loop comes round if they do)
        uint32_t cont = h.m - cur;
        {
            const uint32_t want = cur / 2u > (256u << 10) ? cur / 2u : (256u << 10);
            if (cont > want + want / 2u) cont = (want + 1023u) / 1024u * 1024u;
        }
        if ((rc = launch_iteration(c, 0, cont, false)) != GSX_OK) { c->frame_pending = false; return rc; }
        GSX_HIP(c, hipMemcpyAsync(c->ctl_host, c->ctl, GSX_AUX_OFFSET + 128, hipMemcpyDeviceToHost, c->fstream));
        gsx_mark_frame_done(c);
        GSX_HIP(c, hipStreamSynchronize(c->fstream));
        last = 1;
    }
    {
        // how deep this frame had to go: the cursor at which an iteration found every tile retired, else (all retired
        // by the end of the queued iterations) what the schedule was based on, else everything
        const GsxFrameCtl& h = *c->ctl_host;
        uint32_t rec = ~0u;
        if (h.live_tiles == 0u) rec = c->aux_host[16];    // every tile retired: the deepest rank any of them consumed
        else if (h.dead_at) rec = h.dead_at - 1u;
        c->need_hist[c->need_pos++ % 8u] = rec;
    }
    {
        const uint64_t total = c->ctl_host->total_pairs, first = c->aux_host[2];
        c->rest_pairs = total > first ? total - first : 0u;
    }
    c->tim.tile_pairs = c->ctl_host->total_pairs;
    c->tim.render_iters = c->ctl_host->iters_used;
    if (f.caps_on && !f.redos && c->caps_backoff) --c->caps_backoff;      // predictions that hold earn the trust back
    c->tim.render_capped = f.caps_on ? 1u : 0u;
    c->tim.render_redos = f.redos;
    {   // what the slot's next frame may predict from: this frame's per-tile depths, valid for this geometry and view
        const GsxViewParams& vp = *reinterpret_cast<const GsxViewParams*>(f.vp_blob);
        c->hist_w = static_cast<uint32_t>(vp.width); c->hist_h = static_cast<uint32_t>(vp.height);
        c->hist_y0 = vp.y0; c->hist_y1 = vp.y1;
        std::memcpy(c->hist_wt, vp.wt, sizeof c->hist_wt);
        std::memcpy(c->hist_proj, vp.proj, sizeof c->hist_proj);
        c->hist_shade[0] = vp.cut; c->hist_shade[1] = vp.cut_s;
        c->hist_shade[2] = static_cast<float>(vp.bayer_size); c->hist_shade[3] = vp.bayer_scale;
        c->hist_valid = true;
    }
    c->frame_pending = false;
    return GSX_OK;
}

uint32_t gsx_pair_slots(int cus) {
    int per = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_radix_scatter<uint16_t, GSX_PAIR_NT, GSX_PAIR_IPT, false, GSX_PAIR_MAXD, GSX_PAIR_WPE>, GSX_PAIR_NT, 0) != hipSuccess || per < 1) per = 2;
    return static_cast<uint32_t>(cus) * static_cast<uint32_t>(per);
}
