// gsx_raster.hip — projection (K3), tile binning (K4) and the tile rasterizer
// (K5).
//
// Reference path replaced: vs_main / fs_main of resources/shader_quads_ordered.wgsl
// (wgsl:28-73,120-212) executed by drawIndexed(6, M) (SplatMesh.h:58) with the
// pipeline's SrcAlpha/OneMinusSrcAlpha colour blend, dst-alpha-preserving alpha
// blend, no depth buffer (main.cpp:338-344,359) and the (0.05,0.05,0.05,1) clear
// (renderer.cpp:162-164).
//
// Design: vs_main runs ONCE per splat (the reference runs it 4x, once per quad
// vertex) in original splat order with coalesced SoA loads and leaves a 48-B
// screen-space record.  Because every quad is a view-plane-parallel
// parallelogram at constant clip w, uv is an affine function of the pixel
// position, so coverage + uv come from one 2x2 inverse instead of triangle
// setup.  The binner expands the depth-sorted permutation into (tile, splat)
// pairs in rank order and a stable 2-pass radix sort by tile id (the same radix
// passes as the depth sort) yields per-tile lists that preserve depth order.
// The rasterizer composites each 16x16 tile in sorted order, walking the list
// from the NEAR end with a transmittance accumulator
//     C = sum_i c_i a_i prod_{j nearer than i} (1 - a_j)  +  T_final * clear
// which is algebraically the reference's far-to-near "over" blend and lets a
// tile stop as soon as every pixel's transmittance is below 2^-11 (bounded
// error < 0.13/255).
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

// ---------------------------------------------------------------------------
// K3: per-splat projection.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_project(const float4* __restrict__ pos4, const float4* __restrict__ covA,
                                                 const float4* __restrict__ covB, const float4* __restrict__ color4,
                                                 uint32_t n, GsxViewParams vp, GsxProj* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pos4[i], ca = covA[i], cb = covB[i], col = color4[i];
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
    o.r = col.x; o.g = col.y; o.b = col.z; o.a = col.w;
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
    if (!ok) o.ex = -1.0f;
    // 48-B record as three 16-B stores
    float4* dst = reinterpret_cast<float4*>(out + i);
    dst[0] = make_float4(o.cx, o.cy, o.ex, o.ey);
    dst[1] = make_float4(o.m00, o.m01, o.m10, o.m11);
    dst[2] = make_float4(o.r, o.g, o.b, o.a);
}

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
// K4a/K4b: per-rank tile counts -> block sums; then emit pairs in rank order.
// ---------------------------------------------------------------------------
#define GSX_BIN_NT 256
#define GSX_BIN_IPT 4
#define GSX_BIN_TILE (GSX_BIN_NT * GSX_BIN_IPT)

template <bool EMIT, bool RANK_PAYLOAD>
__global__ __launch_bounds__(GSX_BIN_NT) void k_bin(const uint32_t* __restrict__ m_ptr, const uint32_t* __restrict__ sorted,
                                                    const GsxProj* __restrict__ proj,
                                                    GsxViewParams vp, uint32_t* __restrict__ block_sums,
                                                    uint16_t* __restrict__ pair_key, uint32_t* __restrict__ pair_val,
                                                    uint64_t pair_cap, GsxDevFlags* __restrict__ flags) {
    __shared__ uint32_t s_tmp[GSX_BIN_NT / 64];
    const uint32_t m = *m_ptr;
    uint32_t cnt[GSX_BIN_IPT], idx[GSX_BIN_IPT], bx0[GSX_BIN_IPT], bx1[GSX_BIN_IPT], by0[GSX_BIN_IPT], by1[GSX_BIN_IPT];
    uint32_t tot = 0;
    // thread t owns ranks base + t*IPT .. +IPT-1 so that pairs come out in rank order
    const uint32_t base = blockIdx.x * GSX_BIN_TILE + threadIdx.x * GSX_BIN_IPT;
#pragma unroll
    for (int j = 0; j < GSX_BIN_IPT; ++j) {
        const uint32_t r = base + j;
        cnt[j] = 0;
        if (r < m) {
            idx[j] = sorted[r];
            const float4 c = *reinterpret_cast<const float4*>(proj + idx[j]);
            if (gsx_tile_bbox(c, vp, bx0[j], bx1[j], by0[j], by1[j])) cnt[j] = (bx1[j] - bx0[j] + 1) * (by1[j] - by0[j] + 1);
        }
        tot += cnt[j];
    }
    uint32_t block_total;
    const uint32_t excl = gsx_block_excl_scan<GSX_BIN_NT>(tot, s_tmp, &block_total);
    if (!EMIT) {
        if (threadIdx.x == 0) block_sums[blockIdx.x] = block_total;
        return;
    }
    // Emit cooperatively: the block's pairs occupy [block_base, block_base + block_total) in rank
    // order; every thread fills slots tid, tid + NT, ... (coalesced stores) and finds the owning
    // rank by binary search in the per-rank offsets held in LDS.
    __shared__ uint32_t s_off[GSX_BIN_TILE + 1];     // exclusive offsets per rank of this block
    __shared__ uint32_t s_box[GSX_BIN_TILE];         // tx0 | ty0 << 8 | width << 16
    __shared__ uint32_t s_pay[GSX_BIN_TILE];
    {
        uint32_t run = excl;
#pragma unroll
        for (int j = 0; j < GSX_BIN_IPT; ++j) {
            const uint32_t r = threadIdx.x * GSX_BIN_IPT + j;
            s_off[r] = run;
            run += cnt[j];
            if (cnt[j]) {
                s_box[r] = bx0[j] | (by0[j] << 8) | ((bx1[j] - bx0[j] + 1) << 16);
                s_pay[r] = RANK_PAYLOAD ? (base + j) : idx[j];
            }
        }
        if (threadIdx.x == GSX_BIN_NT - 1) s_off[GSX_BIN_TILE] = run;
    }
    __syncthreads();
    const uint64_t block_base = block_sums[blockIdx.x];
    if (block_base + block_total > pair_cap) {
        if (threadIdx.x == 0) atomicOr(&flags->pair_overflow, 1u);
        return;
    }
    for (uint32_t slot = threadIdx.x; slot < block_total; slot += GSX_BIN_NT) {
        uint32_t lo = 0, hi = GSX_BIN_TILE;          // last r with s_off[r] <= slot (skips empty ranks)
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_off[mid] <= slot) lo = mid; else hi = mid;
        }
        const uint32_t local = slot - s_off[lo], box = s_box[lo];
        const uint32_t wx = box >> 16;
        const uint32_t dy = local / wx, dx = local - dy * wx;
        const uint32_t tx = (box & 0xFFu) + dx, ty = ((box >> 8) & 0xFFu) + dy;
        pair_key[block_base + slot] = static_cast<uint16_t>(ty * vp.tiles_x + tx);
        pair_val[block_base + slot] = s_pay[lo];
    }
}

// single block: in-place exclusive scan of block sums, total -> *total_out
__global__ __launch_bounds__(1024) void k_scan_sums(uint32_t* __restrict__ sums, uint32_t nb, uint32_t* __restrict__ total_out) {
    __shared__ uint32_t s_tmp[1024 / 64];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nb; b0 += 1024) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = (i < nb) ? sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = gsx_block_excl_scan<1024>(v, s_tmp, &total);
        const uint32_t carry = s_carry;
        if (i < nb) sums[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = s_carry;
}

// K4c (stable radix sort of the pairs by 16-bit tile id) uses gsx_radix.hpp.

// tile_start[t] = lower_bound(sorted pair keys, t), t in [0, n_tiles]
__global__ __launch_bounds__(256) void k_tile_ranges(const uint16_t* __restrict__ keys, uint32_t count, uint32_t n_tiles,
                                                     uint32_t* __restrict__ tile_start) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    uint32_t lo = 0, hi = count;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < t) lo = mid + 1; else hi = mid;
    }
    tile_start[t] = lo;
}

// ---------------------------------------------------------------------------
// K5: tile rasterizer.  One 256-thread workgroup per 16x16 tile, each wave an
// 8x8 pixel quadrant; the tile's splat list is staged through LDS in batches of
// 256 records (one 48-B gather per thread) and every thread walks the batch.
// ---------------------------------------------------------------------------
#define GSX_RASTER_BATCH 256
#define GSX_T_EPS (1.0f / 2048.0f)

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

template <int MODE /*0 blend, 1 threshold (bayerSize 10), 2 bayer dither*/>
__global__ __launch_bounds__(256) void k_raster(const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ pair_val,
                                                const GsxProj* __restrict__ proj, const uint32_t* __restrict__ sorted_all,
                                                GsxViewParams vp, uint8_t* __restrict__ fb, GsxDevFlags* __restrict__ dbg) {
    __shared__ float4 s_a[GSX_RASTER_BATCH];  // cx cy ex ey
    __shared__ float4 s_m[GSX_RASTER_BATCH];  // m00 m01 m10 m11
    __shared__ float4 s_c[GSX_RASTER_BATCH];  // r g b a
    __shared__ uint32_t s_rank[MODE == 2 ? GSX_RASTER_BATCH : 1];
    __shared__ int s_live;
    const uint32_t tile = blockIdx.x;
    const uint32_t tx = tile % vp.tiles_x, ty = tile / vp.tiles_x + vp.tile_y0;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // wave = 8x8 quadrant, lane = (lx, ly) inside it
    const uint32_t qx = (wave & 1) * 8, qy = (wave >> 1) * 8;
    const uint32_t px = tx * GSX_TILE_W + qx + (lane & 7), py = ty * GSX_TILE_H + qy + (lane >> 3);
    const float fx = static_cast<float>(px) + 0.5f, fy = static_cast<float>(py) + 0.5f;
    // quadrant rectangle in pixel-centre coordinates for wave-level reject
    const float wx0 = static_cast<float>(tx * GSX_TILE_W + qx) + 0.5f, wx1 = wx0 + 7.0f;
    const float wy0 = static_cast<float>(ty * GSX_TILE_H + qy) + 0.5f, wy1 = wy0 + 7.0f;
    const bool inside = (px < static_cast<uint32_t>(vp.width)) && (py >= vp.y0) && (py < vp.y1);
    const uint32_t* sorted = sorted_all;   // only read in MODE 2 (pairs then carry ranks)

    float T = inside ? 1.0f : 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
    const uint32_t begin = tile_start[tile], end = tile_start[tile + 1];
    // near -> far: the per-tile list is in far -> near (rank ascending) order
    for (uint32_t hi = end; hi > begin;) {
        const uint32_t nb = min(static_cast<uint32_t>(GSX_RASTER_BATCH), hi - begin);
        __syncthreads();  // previous batch fully consumed
        if (tid == 0) s_live = 0;
        if (tid < nb) {
            uint32_t pv = pair_val[hi - 1 - tid];
            if (MODE == 2) { s_rank[tid] = pv; pv = sorted[pv]; }
            const float4* src = reinterpret_cast<const float4*>(proj + pv);
            s_a[tid] = src[0]; s_m[tid] = src[1]; s_c[tid] = src[2];
        }
        __syncthreads();
        if (__builtin_amdgcn_ballot_w64(T >= GSX_T_EPS) != 0) {
            for (uint32_t j = 0; j < nb; ++j) {
                const float4 a = s_a[j];
                // wave-uniform reject: quad bbox vs this wave's 8x8 quadrant
                if (a.x + a.z < wx0 || a.x - a.z > wx1 || a.y + a.w < wy0 || a.y - a.w > wy1) continue;
                const float4 m = s_m[j];
                const float dx = fx - a.x, dy = fy - a.y;
                const float u = m.x * dx + m.y * dy, v = m.z * dx + m.w * dy;
                if (fabsf(u) <= vp.cut && fabsf(v) <= vp.cut) {
                    const float4 c = s_c[j];
                    float alpha = c.w * __expf(-0.5f * (u * u + v * v));
                    if (MODE == 1) alpha = (alpha > vp.bayer_scale) ? 1.0f : 0.0f;
                    if (MODE == 2) {
                        const float bv = static_cast<float>(gsx_bayer(px, py, vp.bayer_size, s_rank[j]));
                        alpha = ((alpha * vp.bayer_scale) * static_cast<float>(vp.bayer_size * vp.bayer_size + 1u) < bv + 1.0f) ? 0.0f : 1.0f;
                    }
                    const float w = T * alpha;
                    cr += w * c.x; cg += w * c.y; cb += w * c.z;
                    T -= w;  // T *= (1 - alpha)
                }
            }
            const bool wave_live = __builtin_amdgcn_ballot_w64(T >= GSX_T_EPS) != 0;
            if (lane == 0 && wave_live) s_live = 1;
        }
        hi -= nb;
#ifdef GSX_DEBUG_COUNT
        if (tid == 0) atomicAdd(&dbg->pad[0], nb);
#endif
        __syncthreads();
        if (!s_live) break;  // every pixel of the tile is saturated
    }
    if (inside) {
        cr += T * 0.05f; cg += T * 0.05f; cb += T * 0.05f;
        const uint32_t r8 = static_cast<uint32_t>(fminf(fmaxf(cr, 0.0f), 1.0f) * 255.0f + 0.5f);
        const uint32_t g8 = static_cast<uint32_t>(fminf(fmaxf(cg, 0.0f), 1.0f) * 255.0f + 0.5f);
        const uint32_t b8 = static_cast<uint32_t>(fminf(fmaxf(cb, 0.0f), 1.0f) * 255.0f + 0.5f);
        reinterpret_cast<uint32_t*>(fb)[static_cast<uint64_t>(py) * static_cast<uint32_t>(vp.width) + px] =
            r8 | (g8 << 8) | (b8 << 16) | 0xFF000000u;
    }
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
static int ensure(gsx_ctx* c, T** p, uint64_t* cap, uint64_t need) {
    if (need <= *cap && *p) return GSX_OK;
    if (*p) GSX_HIP(c, hipFree(*p));
    *p = nullptr; *cap = 0;
    const uint64_t want = need + need / 4 + 1024;
    GSX_HIP(c, hipMalloc(reinterpret_cast<void**>(p), want * sizeof(T)));
    *cap = want;
    return GSX_OK;
}

int gsx_launch_render(gsx_ctx* c, const gsx_uniforms* u, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1) {
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
    if (n_tiles > 65535u || vp.tiles_x > 256u || vp.tiles_y > 256u) {
        c->err = "frame band larger than 4096 x 4096 pixels / 65535 tiles";
        return GSX_E_UNSUPPORTED;
    }
    const int mode = (u->bayerSize == 1u) ? 0 : (u->bayerSize == 10u ? 1 : 2);
    if (mode == 2 && !(u->bayerSize == 2u || u->bayerSize == 4u || u->bayerSize == 8u || u->bayerSize == 16u)) {
        c->err = "bayerSize must be 1, 10 or a power of two (the shader divides by zero otherwise)";
        return GSX_E_UNSUPPORTED;
    }
    const uint32_t n = static_cast<uint32_t>(c->n);
    const uint32_t m_cap = c->m_cap;

    int rc;
    if ((rc = ensure(c, &c->fb, &c->fb_cap, static_cast<uint64_t>(w) * h * 4)) != GSX_OK) return rc;
    if (c->fb_w != w || c->fb_h != h) {
        GSX_HIP(c, hipMemsetAsync(c->fb, 0, static_cast<uint64_t>(w) * h * 4, c->stream));
        c->fb_w = w; c->fb_h = h;
    }
    if ((rc = ensure(c, &c->tile_start, &c->tile_cap, static_cast<uint64_t>(n_tiles) + 2)) != GSX_OK) return rc;
    const uint32_t nb_bin = gsx_div_up(m_cap, GSX_BIN_TILE);
    if ((rc = ensure(c, &c->scan_block_sums, &c->rank_cap, static_cast<uint64_t>(nb_bin) + 1)) != GSX_OK) return rc;

    const uint32_t* sorted = c->vals[c->sorted_buf];

    // K3
    if (c->profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_PROJ0], c->stream));
    if (n > 0)
        hipLaunchKernelGGL(k_project, dim3(gsx_div_up(n, 256)), dim3(256), 0, c->stream, c->pos4, c->covA, c->covB, c->color4,
                           n, vp, c->proj);
    if (c->profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_PROJ1], c->stream));

    // K4a: count
    uint32_t total_pairs = 0;
    if (nb_bin > 0) {
        hipLaunchKernelGGL((k_bin<false, false>), dim3(nb_bin), dim3(GSX_BIN_NT), 0, c->stream, c->m_dev, sorted, c->proj, vp,
                           c->scan_block_sums, nullptr, nullptr, 0ull, c->flags);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, c->stream, c->scan_block_sums, nb_bin, c->pair_total_dev);
        GSX_HIP(c, hipMemcpyAsync(c->host_word, c->pair_total_dev, 4, hipMemcpyDeviceToHost, c->stream));
        GSX_HIP(c, hipStreamSynchronize(c->stream));
        total_pairs = *c->host_word;
    }
    c->tim.tile_pairs = total_pairs;
    const uint16_t* sorted_keys = nullptr;
    const uint32_t* sorted_vals = nullptr;
    if (total_pairs > 0) {
        uint64_t cap0 = c->pair_cap, cap1 = c->pair_cap, cap2 = c->pair_cap, cap3 = c->pair_cap;
        if ((rc = ensure(c, &c->pair_key[0], &cap0, total_pairs)) != GSX_OK) return rc;
        if ((rc = ensure(c, &c->pair_key[1], &cap1, total_pairs)) != GSX_OK) return rc;
        if ((rc = ensure(c, &c->pair_val[0], &cap2, total_pairs)) != GSX_OK) return rc;
        if ((rc = ensure(c, &c->pair_val[1], &cap3, total_pairs)) != GSX_OK) return rc;
        c->pair_cap = cap0 < cap1 ? cap0 : cap1;
        if (cap2 < c->pair_cap) c->pair_cap = cap2;
        if (cap3 < c->pair_cap) c->pair_cap = cap3;
        // K4b: emit
        if (mode == 2)
            hipLaunchKernelGGL((k_bin<true, true>), dim3(nb_bin), dim3(GSX_BIN_NT), 0, c->stream, c->m_dev, sorted, c->proj, vp,
                               c->scan_block_sums, c->pair_key[0], c->pair_val[0], c->pair_cap, c->flags);
        else
            hipLaunchKernelGGL((k_bin<true, false>), dim3(nb_bin), dim3(GSX_BIN_NT), 0, c->stream, c->m_dev, sorted, c->proj, vp,
                               c->scan_block_sums, c->pair_key[0], c->pair_val[0], c->pair_cap, c->flags);
        // K4c: stable sort of the pairs by tile id
        uint32_t tile_bits = 1;
        while ((1u << tile_bits) < n_tiles) ++tile_bits;
        const GsxRadixPlan plan = gsx_radix_plan(tile_bits);
        const uint32_t g = gsx_radix_blocks(total_pairs, GSX_PAIR_TILE, c->pair_slots);
        if ((rc = ensure(c, &c->pair_ws, &c->pair_ws_words, gsx_radix_workspace_words(g))) != GSX_OK) return rc;
        GSX_HIP(c, hipMemcpyAsync(c->pair_total_dev, c->host_word, 4, hipMemcpyHostToDevice, c->stream));  // (already there; keeps count_ptr explicit)
        uint32_t* cnt = c->pair_ws;
        uint32_t* totals = cnt + static_cast<uint64_t>(GSX_RADIX_MAX_DIGITS) * g;
        int cur = 0;
        for (int p = 0; p < plan.n_passes; ++p) {
            hipLaunchKernelGGL((k_radix_hist<uint16_t, GSX_PAIR_NT>), dim3(g), dim3(GSX_PAIR_NT), 0, c->stream,
                               c->pair_key[cur], c->pair_total_dev, plan.shift[p], plan.bits[p], cnt);
            hipLaunchKernelGGL((k_radix_scan<0>), dim3(1u << plan.bits[p]), dim3(256), 0, c->stream, cnt, g, totals);
            hipLaunchKernelGGL((k_radix_scatter<uint16_t, GSX_PAIR_NT, GSX_PAIR_IPT, false>), dim3(g), dim3(GSX_PAIR_NT), 0, c->stream,
                               c->pair_key[cur], c->pair_val[cur], c->pair_key[cur ^ 1], c->pair_val[cur ^ 1], c->pair_total_dev,
                               plan.shift[p], plan.bits[p], cnt, totals);
            cur ^= 1;
        }
        sorted_keys = c->pair_key[cur];
        sorted_vals = c->pair_val[cur];
        hipLaunchKernelGGL(k_tile_ranges, dim3(gsx_div_up(n_tiles + 1, 256)), dim3(256), 0, c->stream, sorted_keys, total_pairs,
                           n_tiles, c->tile_start);
    } else {
        GSX_HIP(c, hipMemsetAsync(c->tile_start, 0, (static_cast<uint64_t>(n_tiles) + 1) * 4, c->stream));
    }
    if (c->profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_BIN1], c->stream));

    // K5
    if (n_tiles > 0) {
        if (mode == 0)
            hipLaunchKernelGGL(k_raster<0>, dim3(n_tiles), dim3(256), 0, c->stream, c->tile_start, sorted_vals, c->proj, sorted,
                               vp, c->fb, c->flags);
        else if (mode == 1)
            hipLaunchKernelGGL(k_raster<1>, dim3(n_tiles), dim3(256), 0, c->stream, c->tile_start, sorted_vals, c->proj, sorted,
                               vp, c->fb, c->flags);
        else
            hipLaunchKernelGGL(k_raster<2>, dim3(n_tiles), dim3(256), 0, c->stream, c->tile_start, sorted_vals, c->proj, sorted,
                               vp, c->fb, c->flags);
    }
    if (c->profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_RAST1], c->stream));
    GSX_HIP(c, hipGetLastError());
    return GSX_OK;
}

uint32_t gsx_pair_slots(int cus) {
    int per = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_radix_scatter<uint16_t, GSX_PAIR_NT, GSX_PAIR_IPT, false>, GSX_PAIR_NT, 0) != hipSuccess || per < 1) per = 2;
    return static_cast<uint32_t>(cus) * static_cast<uint32_t>(per);
}
