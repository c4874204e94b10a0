// gsx_sort2.hpp — the depth sort in TWO passes over HBM (gfx950): one wide most-significant-digit partition, then one
// kernel that finishes every bucket inside LDS.
//
//   A  k2_keygen_hist   depth keys (SplatMesh.h:90-94) + per-column counts of the top digit (up to 2^13 buckets;
//                       a column = one workgroup's contiguous key range; LDS histogram, 2-byte counts out)
//   B  k2_scan          per bucket: exclusive prefix of the counts over the columns, and the bucket's total
//   C  k2_partition     every workgroup turns its column of the prefix matrix into absolute output cursors in LDS
//                       (bucket base + prefix) and streams its keys: ONE returning LDS add per key yields the
//                       destination of the (key, index) pair.  No ranking, no staging, no barrier in the loop.
//   D  k2_local         one workgroup per bucket (<= GSX2_CAP pairs): counting sort on the next q key bits in LDS
//                       (packed 16-bit bins), then every element finds its place inside its bin by counting the
//                       smaller (remaining key bits, index) composites of the bin -- bins hold ~1 element, so this
//                       is where the tie rule "equal distance -> ascending index" is enforced, whatever order the
//                       LDS atomics of C and D handed out.  Writes the permutation (and the sorted keys if asked).
//   E  k2_big           buckets that do not fit LDS (a cluster denser than the bucket width resolves): one workgroup
//                       per such bucket runs a stable 8-bit LSD radix over (remaining key bits, index) through a
//                       scratch copy in HBM.  Correct for any input; slow, so the host falls back to the three-pass
//                       LSD sort of gsx_radix.hpp for later frames when a frame sent many keys this way.
//
// Replaces std::sort(indices, distances[a] > distances[b]) (SplatMesh.h:97-108) like gsx_radix.hpp does, with
// 44 + ~8 B of HBM traffic per splat instead of 72 and 4 launches instead of 10.  The result is the same total order
// (key ascending = distance descending, then index ascending): the order of equal keys never depends on the order in
// which atomics were served, because k2_local / k2_big order (key, index) composites explicitly.
#pragma once
#include "gsx_internal.h"
#include "gsx_radix.hpp"

#define GSX2_NT 512                      // threads of k2_keygen_hist, k2_local, k2_big
#define GSX2_PNT 1024                    // threads of k2_partition (one workgroup per CU: its column of pairs fills the LDS)
#define GSX2_MAX_T 13                    // bits of the top digit at most
#define GSX2_MAX_D (1 << GSX2_MAX_T)     // buckets at most
#define GSX2_MIN_COLS 512                // columns (= workgroups of A and C) at least: two rounds of CUs
#define GSX2_MAX_COLS 2048
#define GSX2_COLMAX 12288                // keys of a column at most (k2_partition stages them in LDS: 96 KB)
#define GSX2_KPT (GSX2_COLMAX / GSX2_PNT)
#define GSX2_CAP 9216                    // pairs of the largest bucket k2_local sorts in LDS
#define GSX2_IPT (GSX2_CAP / GSX2_NT)
#define GSX2_IPT2 32                     // the third instance of k2_local, launched only for scenes that need it ...
#define GSX2_CAP2 (GSX2_IPT2 * GSX2_NT)  // ... sorts buckets of up to 16384 pairs in LDS (80 KB: one or two workgroups per CU)
#define GSX2_MAX_Q 13                    // bits of k2_local's bins at most (2^13 packed 16-bit bins = 16 KB)

struct Gsx2Stats {
    uint32_t big_buckets;    // buckets larger than what that sort's k2_local instances took in LDS (GSX2_CAP, or GSX2_CAP2 with the third instance)
    uint32_t big_keys;       // keys in them
    uint32_t max_bucket;
    uint32_t keys;           // selection size of that sort (a cut's size is known only on the device: the next plan sizes its buckets by it)
};

// Bucket of a (relative) key.  Keys are inverted float bit patterns, so 2^23 consecutive key values span a factor of two in
// distance.  The far 2^25 key values (where nearly all splats of a scene lie) are cut into buckets of 2^r0 values; behind
// them every further window of 2^23 values (distance halves, the splat count drops ~8 x) gets half as many buckets as the one
// before, down to one bucket per window.  With a camera inside the scene's bounding box the host can bound the nearest
// distance only by 0 and the key range is the full 31 bits: the ~5 K buckets still sit where the splats are.
struct Gsx2Map {
    uint32_t r0;      // low key bits inside a bucket of the fine region
    uint32_t nf;      // log2 of twice the buckets of the first coarse window: 24 - rc
    uint32_t base;    // buckets of the fine region: flen >> r0
    uint32_t flen;    // key values of the fine region: 4 windows of 2^23 (3 for the 13-bit plan of a wide key range, which must leave table entries for the coarse windows)
    uint32_t rc;      // low key bits inside a bucket of the first coarse window: r0 + 1 (r0 + 2 for that plan)
};
__host__ __device__ __forceinline__ uint32_t gsx2_bucket(uint32_t rel, const Gsx2Map mp) {
    if (rel < mp.flen) return rel >> mp.r0;
    const uint32_t e = (rel - mp.flen) >> 23;
    if (e < mp.nf) return mp.base + (1u << mp.nf) - (1u << (mp.nf - e)) + ((rel & 0x7FFFFFu) >> (mp.rc + e));
    return mp.base + (1u << mp.nf) - 1u + (e - mp.nf);
}
// low key bits that vary inside bucket d
__host__ __device__ __forceinline__ uint32_t gsx2_bucket_bits(uint32_t d, const Gsx2Map mp) {
    if (d < mp.base) return mp.r0;
    const uint32_t x = d - mp.base;
    if (x + 1u >= (1u << mp.nf)) return 23u;
    const uint32_t y = (1u << mp.nf) - 1u - x;                       // >= 1
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lg = 31u - static_cast<uint32_t>(__clz(static_cast<int>(y)));
#else
    const uint32_t lg = 31u - static_cast<uint32_t>(__builtin_clz(y));
#endif
    return mp.rc + (mp.nf - 1u - lg);
}

// ---- A: keys + per-column counts of the top digit ------------------------------------------------------------
// cnt[col * nd + d], 2-byte counts (a column holds at most GSX2_COLMAX keys).  Grid = number of columns.
struct Gsx2Pos { float x, y, z; };       // 12-byte position record of the key pass (a wave's loads cover 768 contiguous bytes)
template <bool IOTA>
__global__ __launch_bounds__(GSX2_NT) void k2_keygen_hist(const Gsx2Pos* __restrict__ pos3, const uint32_t* __restrict__ sel,
                                                          const uint32_t* __restrict__ m_ptr, float cx, float cy, float cz, uint32_t key_lo,
                                                          uint32_t key_span, Gsx2Map mp, uint32_t nd,
                                                          uint32_t* __restrict__ keys, uint16_t* __restrict__ cnt, GsxDevFlags* __restrict__ flags,
                                                          uint32_t* __restrict__ m_copy, Gsx2Stats* __restrict__ stats) {
    __shared__ uint32_t s_hist[GSX2_MAX_D];
    const uint32_t m = *m_ptr;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (m_copy) *m_copy = m;
    }
    for (uint32_t t = threadIdx.x; t < nd; t += GSX2_NT) s_hist[t] = 0;
    __syncthreads();
    const uint32_t col = gsx_logical_block();
    uint32_t begin, end;
    gsx_block_range(m, gridDim.x, col, begin, end);
    constexpr int UNROLL = 4;
    for (uint32_t base = begin + threadIdx.x; base < end; base += GSX2_NT * UNROLL) {
        Gsx2Pos p[UNROLL];
        bool ok[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const uint32_t k = base + j * GSX2_NT;
            ok[j] = k < end;
            if (ok[j]) p[j] = pos3[IOTA ? k : sel[k]];
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            if (!ok[j]) continue;
            const uint32_t k = base + j * GSX2_NT;
            const uint32_t key = gsx_key_from_distance(gsx_distance(make_float4(p[j].x, p[j].y, p[j].z, 0.0f), cx, cy, cz));
            uint32_t rel = key - key_lo;
            if (key < key_lo || rel > key_span) {   // outside the bound: NaN/inf position
                atomicOr(&flags->key_bound, 1u);
                rel = (key < key_lo) ? 0u : key_span;
            }
            keys[k] = rel;
            atomicAdd(&s_hist[gsx2_bucket(rel, mp)], 1u);
        }
    }
    __syncthreads();
    // two 2-byte counts per lane: a wave writes 256 contiguous bytes (nd is a multiple of 128)
    uint32_t* out = reinterpret_cast<uint32_t*>(cnt + static_cast<uint64_t>(col) * nd);
    for (uint32_t t = threadIdx.x; t < nd / 2; t += GSX2_NT) out[t] = s_hist[2 * t] | (s_hist[2 * t + 1] << 16);
}

// ---- B: per bucket, exclusive prefix over the columns -----------------------------------------------------------
// Workgroup = 16 buckets (8 lanes x 2 packed counts) x 64 chunks of ncols / 64 columns; two sweeps over the counts
// (the second one hits L2), loads in batches of 8.  Grid = nd / 16; ncols is a multiple of 64.
__global__ __launch_bounds__(512) void k2_scan(const uint16_t* __restrict__ cnt, uint32_t* __restrict__ pre, uint32_t* __restrict__ totals,
                                               uint32_t nd, uint32_t ncols) {
    __shared__ uint2 s_part[64][8];
    const uint32_t pr = threadIdx.x & 7u, ch = threadIdx.x >> 3;
    const uint32_t d0 = blockIdx.x * 16u + pr * 2u;
    const uint32_t cl = ncols / 64u, c0 = ch * cl;
    const uint32_t* cnt32 = reinterpret_cast<const uint32_t*>(cnt);
    const uint64_t row = nd >> 1;                                    // words per column
    const uint64_t at0 = static_cast<uint64_t>(c0) * row + (d0 >> 1);
    uint32_t s0 = 0, s1 = 0, keep[8];
    for (uint32_t j0 = 0; j0 < cl; j0 += 8) {
        uint32_t w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = (j0 + j < cl) ? cnt32[at0 + (j0 + j) * row] : 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) { s0 += w[j] & 0xFFFFu; s1 += w[j] >> 16; keep[j] = w[j]; }
    }
    s_part[ch][pr] = make_uint2(s0, s1);
    __syncthreads();
    uint32_t r0 = 0, r1 = 0;
    for (uint32_t k = 0; k < ch; ++k) {
        const uint2 t = s_part[k][pr];
        r0 += t.x;
        r1 += t.y;
    }
    uint2* pre2 = reinterpret_cast<uint2*>(pre);
    for (uint32_t j0 = 0; j0 < cl; j0 += 8) {
        uint32_t w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = (cl <= 8u) ? keep[j] : ((j0 + j < cl) ? cnt32[at0 + (j0 + j) * row] : 0u);   // (second sweep: L2)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j0 + j < cl) pre2[at0 + (j0 + j) * row] = make_uint2(r0, r1);
            r0 += w[j] & 0xFFFFu;
            r1 += w[j] >> 16;
        }
    }
    if (ch == 63) { totals[d0] = r0; totals[d0 + 1] = r1; }
}

// ---- C: partition ------------------------------------------------------------------------------------------------
// One workgroup per column.  The column's keys are counting-sorted by bucket INSIDE LDS first (start of a bucket =
// prefix of the column's own counts; one returning LDS add per key hands out the slot), then written out in slot
// order, so that the pairs a column contributes to one bucket leave as one contiguous run: writing every pair
// straight from its thread (6 M isolated 8-byte stores) took 74 us at 6 M keys, bound by requests, not bytes.
template <bool IOTA>
__global__ __launch_bounds__(GSX2_PNT) void k2_partition(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ sel,
                                                         const uint32_t* __restrict__ m_ptr, Gsx2Map mp, uint32_t nd,
                                                         const uint16_t* __restrict__ cnt, const uint32_t* __restrict__ pre,
                                                         const uint32_t* __restrict__ totals, uint32_t* __restrict__ base_out,
                                                         uint2* __restrict__ pairs, Gsx2Stats* __restrict__ stats, GsxDevFlags* __restrict__ flags,
                                                         uint32_t cap_lds /* largest bucket this sort's k2_local launches take in LDS */) {
    __shared__ uint2 s_pair[GSX2_COLMAX];
    __shared__ uint32_t s_cur[GSX2_MAX_D / 2];     // packed 16-bit slot cursors (start of the bucket inside the column, then running)
    __shared__ uint32_t s_adj[GSX2_MAX_D];         // output index of slot p of bucket d = s_adj[d] + p
    __shared__ uint32_t s_tmp[GSX2_PNT / 64];
    __shared__ uint32_t s_big[3];
    constexpr int PER = GSX2_MAX_D / GSX2_PNT;     // 8 consecutive buckets per thread
    const uint32_t tid = threadIdx.x;
    const uint32_t m = *m_ptr;
    const uint32_t col = gsx_logical_block();
    uint32_t begin, end;
    gsx_block_range(m, gridDim.x, col, begin, end);
    const uint32_t csize = end - begin;
    if (csize > GSX2_COLMAX) {                      // (the host sizes the grid so that this cannot happen)
        if (tid == 0) atomicOr(&flags->key_bound, 4u);
        return;
    }
    uint32_t key[GSX2_KPT], idx[GSX2_KPT];
#pragma unroll
    for (int j = 0; j < GSX2_KPT; ++j) {
        const uint32_t k = begin + j * GSX2_PNT + tid;
        if (k < end) {
            key[j] = keys[k];
            idx[j] = IOTA ? k : sel[k];
        }
    }
    if (tid < 3) s_big[tid] = 0;
    uint32_t tot[PER], own[PER], sum_t = 0, sum_o = 0;
    const uint32_t d_first = tid * PER;
    {
        uint4 c4 = make_uint4(0, 0, 0, 0);
        if (d_first < nd) c4 = *reinterpret_cast<const uint4*>(cnt + static_cast<uint64_t>(col) * nd + d_first);   // nd is a multiple of 128
        const uint32_t cw[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t d = d_first + j;
            tot[j] = d < nd ? totals[d] : 0u;
            own[j] = (cw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
            sum_t += tot[j];
            sum_o += own[j];
        }
    }
    uint32_t acc = gsx_block_excl_scan<GSX2_PNT>(sum_t, s_tmp, nullptr);
    uint32_t st = gsx_block_excl_scan<GSX2_PNT>(sum_o, s_tmp, nullptr);
    uint32_t nbig = 0, kbig = 0, mx = 0;
#pragma unroll
    for (int j = 0; j < PER; j += 2) {
        uint32_t packed = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t d = d_first + j + h;
            if (d < nd) {
                s_adj[d] = acc + pre[static_cast<uint64_t>(col) * nd + d] - st;
                if (col == 0) {
                    base_out[d] = acc;
                    if (d + 1 == nd) base_out[nd] = acc + tot[j + h];
                    if (tot[j + h] > cap_lds) { ++nbig; kbig += tot[j + h]; }
                    mx = tot[j + h] > mx ? tot[j + h] : mx;
                }
            }
            packed |= st << (h * 16);
            acc += tot[j + h];
            st += own[j + h];
        }
        s_cur[(d_first + j) >> 1] = packed;
    }
    if (col == 0) {     // (workgroup-uniform)
        if (nbig) { atomicAdd(&s_big[0], nbig); atomicAdd(&s_big[1], kbig); }
        atomicMax(&s_big[2], mx);
    }
    __syncthreads();
    if (col == 0 && tid == 0) { stats->big_buckets = s_big[0]; stats->big_keys = s_big[1]; stats->max_bucket = s_big[2]; stats->keys = m; }
#pragma unroll
    for (int j = 0; j < GSX2_KPT; ++j) {
        const uint32_t k = begin + j * GSX2_PNT + tid;
        if (k < end) {
            const uint32_t d = gsx2_bucket(key[j], mp);
            const uint32_t sh = (d & 1u) * 16u;
            const uint32_t old = atomicAdd(&s_cur[d >> 1], 1u << sh);
            const uint32_t slot = (old >> sh) & 0xFFFFu;
            if (slot < GSX2_COLMAX) s_pair[slot] = make_uint2(key[j], idx[j]);
        }
    }
    __syncthreads();
    for (uint32_t p = tid; p < csize; p += GSX2_PNT) {
        const uint2 pr = s_pair[p];
        const uint32_t dst = s_adj[gsx2_bucket(pr.x, mp)] + p;
        if (dst < m) pairs[dst] = pr;   // (always true for a consistent count matrix)
    }
}

// ---- D: one bucket per workgroup, finished in LDS -----------------------------------------------------------------
// Element word in LDS: (key bits below the bin << idx_bits) | index.  Two instances share the buckets by size, so that
// the many ordinary buckets do not pay for the LDS of the largest: IPT = 8 (up to 4096 pairs, 32 KB: four workgroups per
// CU) and IPT = GSX2_IPT = 18 (up to GSX2_CAP = 9216, 52 KB: still three per CU), whose workgroups also take what nothing sorts
// in LDS -- a larger bucket, or a coarse bucket near the camera whose key bits do not fit the word -- through
// gsx2_big_bucket (defined below).  (A third instance for 8192 .. 12288 pairs at 64 KB cost every sort a launch that
// found nothing to do: 5 us.  9216 is what three workgroups per CU leave room for; the bench orbit's largest bucket is 8819.)
// That third instance exists all the same, IPT = GSX2_IPT2 = 32 (9217 .. 16384 pairs, 80 KB), and is launched only while the
// statistics of recent sorts show buckets over GSX2_CAP (the host passes cap_lds = GSX2_CAP2 to all instances then): at
// 20 M clustered splats and 5 120 buckets, 559 buckets hold 9 .. 12 K pairs, which one workgroup each sorted through
// global memory in 291 us (tools/sort2bench.hip).
__device__ __forceinline__ void gsx2_big_bucket(uint2* pairs, uint2* scratch, uint32_t b0, uint32_t size, uint32_t r, uint32_t idx_bits,
                                                uint32_t* result, uint32_t* lds);

template <int IPT>
__global__ __launch_bounds__(GSX2_NT) __attribute__((amdgpu_waves_per_eu(IPT == 8 ? 8 : (IPT == GSX2_IPT ? 6 : 2), IPT == 8 ? 8 : (IPT == GSX2_IPT ? 6 : 2))))
void k2_local(uint2* pairs, uint2* scratch, const uint32_t* __restrict__ base, uint32_t nd, Gsx2Map mp, uint32_t idx_bits,
              uint32_t* __restrict__ result, uint32_t cap_lds /* GSX2_CAP, or GSX2_CAP2 when the IPT2 instance is launched too */) {
    constexpr bool REST = IPT == GSX2_IPT;           // the instance that also takes what does not fit LDS
    constexpr int QMAX = GSX2_MAX_Q;
    constexpr uint32_t LO = IPT == 8 ? 0u : (IPT == GSX2_IPT ? 8u * GSX2_NT : static_cast<uint32_t>(GSX2_CAP)), HI = IPT * GSX2_NT;
    __shared__ uint32_t s_bins[(1 << QMAX) / 2];    // packed 16-bit: counts, then exclusive starts
    __shared__ uint32_t s_comp[IPT * GSX2_NT];
    __shared__ uint32_t s_tmp[GSX2_NT / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t d = blockIdx.x;
    const uint32_t b0 = base[d], size = base[d + 1] - b0;
    if (size == 0) return;
    const uint32_t r = gsx2_bucket_bits(d, mp);
    const uint32_t rmask = (1u << r) - 1u;
    const uint32_t idx_mask = idx_bits >= 32 ? 0xFFFFFFFFu : ((1u << idx_bits) - 1u);
    // bins: at most 2^QMAX and never finer than a key value; the key bits below the bin share a word with the index
    const uint32_t q_max = r < static_cast<uint32_t>(QMAX) ? r : static_cast<uint32_t>(QMAX);
    const uint32_t q_min = r + idx_bits > 32u ? r + idx_bits - 32u : 0u;
    const bool unpackable = size > 64u && q_min > (r < static_cast<uint32_t>(GSX2_MAX_Q) ? r : static_cast<uint32_t>(GSX2_MAX_Q));
    if (size > cap_lds || unpackable) {
        if (REST) gsx2_big_bucket(pairs, scratch, b0, size, r, idx_bits, result, s_comp);
        return;
    }
    if (size <= LO || size > HI) return;            // another instance's bucket
    if (size <= 64) {          // one wave, rank by counting over the whole bucket
        if (tid >= 64) return;
        uint2 p = make_uint2(0, 0);
        if (lane < size) p = pairs[b0 + lane];
        const uint32_t low = p.x & rmask;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < size; ++j) {
            const uint32_t lj = __shfl(low, j), ij = __shfl(p.y, j);
            rank += (lj < low || (lj == low && ij < p.y)) ? 1u : 0u;
        }
        if (lane < size) {
            result[b0 + rank] = p.y;
        }
        return;
    }
    uint32_t q = 32u - static_cast<uint32_t>(__builtin_clz(size - 1u));        // one to two bins per element
    q = q < 6u ? 6u : q;
    q = q < q_min ? q_min : q;
    q = q > q_max ? q_max : q;
    const uint32_t nbw = (1u << q) / 2u;          // words of bins in use
    const uint32_t sub = r - q;                     // key bits below the bin
    const uint32_t submask = (1u << sub) - 1u;
    const uint32_t ni = (size + GSX2_NT - 1u) / GSX2_NT;
    for (uint32_t t = tid; t < nbw; t += GSX2_NT) s_bins[t] = 0;
    uint2 pin[IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        if (static_cast<uint32_t>(i) < ni) {       // (workgroup-uniform)
            const uint32_t e = i * GSX2_NT + tid;
            pin[i] = pairs[b0 + (e < size ? e : size - 1u)];
        }
    }
#if defined(GSX2_ABL) && GSX2_ABL == 1      // (GSX2_ABL: ablation switches for tools/sort2bench.hip, never defined in the library build)
#pragma unroll
    for (int i = 0; i < IPT; ++i)
        if (static_cast<uint32_t>(i) < ni && i * GSX2_NT + tid < size) result[b0 + i * GSX2_NT + tid] = pin[i].y;
    return;
#endif
    __syncthreads();
    uint32_t comp[IPT], ba[IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        if (static_cast<uint32_t>(i) < ni) {
            const uint32_t e = i * GSX2_NT + tid;
            const uint32_t low = pin[i].x & rmask;
            const uint32_t bin = low >> sub;
            comp[i] = ((low & submask) << idx_bits) | pin[i].y;
            ba[i] = bin;
            if (e < size) {
                const uint32_t sh = (bin & 1u) * 16u;
                const uint32_t old = atomicAdd(&s_bins[bin >> 1], 1u << sh);
                ba[i] |= ((old >> sh) & 0xFFFFu) << 16;
            }
        }
    }
    __syncthreads();
    {   // exclusive starts: thread t owns words [t * PW, (t + 1) * PW)
        constexpr int PW = (1 << QMAX) / 2 / GSX2_NT;     // 4 or 8 words
        uint32_t w[PW], sum = 0;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const uint32_t t = tid * PW + j;
            w[j] = t < nbw ? s_bins[t] : 0u;
            sum += (w[j] & 0xFFFFu) + (w[j] >> 16);
        }
        uint32_t run = gsx_block_excl_scan<GSX2_NT>(sum, s_tmp, nullptr);
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const uint32_t t = tid * PW + j;
            const uint32_t c0 = w[j] & 0xFFFFu, c1 = w[j] >> 16;
            if (t < nbw) s_bins[t] = run | ((run + c0) << 16);
            run += c0 + c1;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        if (static_cast<uint32_t>(i) < ni) {
            const uint32_t e = i * GSX2_NT + tid;
            if (e < size) {
                const uint32_t bin = ba[i] & 0xFFFFu, arr = ba[i] >> 16;
                const uint32_t start = (s_bins[bin >> 1] >> ((bin & 1u) * 16u)) & 0xFFFFu;
                s_comp[start + arr] = comp[i];
            }
        }
    }
    __syncthreads();
    // Place inside the bin: rank = number of smaller words among the bin's elements.  All items of a thread walk their
    // bins together (one LDS read per item and step in flight instead of one dependent read after another).
    const uint32_t nbins = 1u << q;
    uint32_t bl[IPT], cmax = 0;        // bl = start of the element's bin | elements in it << 16; ba becomes arrival | rank << 16
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        bl[i] = 0;
        if (static_cast<uint32_t>(i) < ni) {
            const uint32_t e = i * GSX2_NT + tid;
            if (e < size) {
                const uint32_t bin = ba[i] & 0xFFFFu;
                const uint32_t start = (s_bins[bin >> 1] >> ((bin & 1u) * 16u)) & 0xFFFFu;
                const uint32_t nb = bin + 1u;
                const uint32_t stop = nb < nbins ? ((s_bins[nb >> 1] >> ((nb & 1u) * 16u)) & 0xFFFFu) : size;
                uint32_t len = stop - start;
#if defined(GSX2_ABL) && GSX2_ABL == 2
                len = 1;
#endif
                bl[i] = start | (len << 16);
                if (len > 1u) cmax = len > cmax ? len : cmax;
            }
            ba[i] >>= 16;
        }
    }
    if (cmax > 1u) {                    // (per thread: most threads hold no element that shares its bin)
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            if (static_cast<uint32_t>(i) < ni) {
                const uint32_t len = bl[i] >> 16, start = bl[i] & 0xFFFFu;
                if (len == 2u) {                                   // the usual collision: one partner
                    const uint32_t v = s_comp[start + 1u - (ba[i] & 1u)];
                    ba[i] += v < comp[i] ? 0x10000u : 0u;
                } else if (len > 2u) {
                    uint32_t rk = 0;
                    for (uint32_t k = 0; k < len; k += 2u) {
                        const uint32_t v0 = s_comp[start + k];
                        const uint32_t v1 = k + 1u < len ? s_comp[start + k + 1u] : 0xFFFFFFFFu;
                        rk += (v0 < comp[i] ? 1u : 0u) + (v1 < comp[i] ? 1u : 0u);
                    }
                    ba[i] += rk << 16;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        if (static_cast<uint32_t>(i) < ni) {
            const uint32_t arr = ba[i] & 0xFFFFu;
            const uint32_t rk = (bl[i] >> 16) > 1u ? (ba[i] >> 16) : arr;
            ba[i] = ((bl[i] & 0xFFFFu) + rk) | ((rk != arr ? 1u : 0u) << 31);   // final slot; bit 31: the element moves
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        if (static_cast<uint32_t>(i) < ni) {
            const uint32_t e = i * GSX2_NT + tid;
            if (e < size) {
                if (ba[i] >> 31) s_comp[ba[i] & 0x7FFFFFFFu] = comp[i];
            }
        }
    }
    __syncthreads();
    for (uint32_t e = tid; e < size; e += GSX2_NT) result[b0 + e] = s_comp[e] & idx_mask;
}

// ---- E: buckets beyond LDS ------------------------------------------------------------------------------------------
// One workgroup sorts one such bucket: stable LSD radix, 8 bits a pass, over the 64-bit composite (key bits below the top
// digit << idx_bits | index), ping-pong between the pair array and a scratch copy of it; the last pass writes the
// permutation.  Tiles of 512 x 4 pairs, ranked by wave ballots like gsx_radix.hpp.  `lds`: 10 KB + 16 words.
__device__ __forceinline__ void gsx2_big_bucket(uint2* pairs, uint2* scratch, uint32_t b0, uint32_t size, uint32_t r, uint32_t idx_bits,
                                                uint32_t* result, uint32_t* lds) {
    constexpr int NW = GSX2_NT / 64, IT = 4;
    uint32_t* s_hist = lds;
    uint32_t* s_run = lds + 256;
    uint32_t (*s_w)[256] = reinterpret_cast<uint32_t (*)[256]>(lds + 512);
    uint32_t* s_tmp = lds + 512 + NW * 256;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t rmask = (1u << r) - 1u;
    const uint32_t total_bits = r + idx_bits;
    const uint32_t n_pass = (total_bits + 7u) / 8u > 0u ? (total_bits + 7u) / 8u : 1u;
    uint2* src = pairs + b0;
    uint2* dst = scratch + b0;
    for (uint32_t pass = 0; pass < n_pass; ++pass) {
        const uint32_t shift = pass * 8u;
        const bool last = pass + 1 == n_pass;
        auto digit_of = [&](uint2 p) -> uint32_t {
            const unsigned long long comp = (static_cast<unsigned long long>(p.x & rmask) << idx_bits) | p.y;
            return static_cast<uint32_t>(comp >> shift) & 255u;
        };
        __syncthreads();
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        for (uint32_t e = tid; e < size; e += GSX2_NT) atomicAdd(&s_hist[digit_of(src[e])], 1u);
        __syncthreads();
        {
            const uint32_t v = tid < 256 ? s_hist[tid] : 0u;
            const uint32_t ex = gsx_block_excl_scan<GSX2_NT>(v, s_tmp, nullptr);
            if (tid < 256) s_run[tid] = ex;
        }
        __syncthreads();
        for (uint32_t tb = 0; tb < size; tb += GSX2_NT * IT) {
            for (uint32_t t = tid; t < NW * 256; t += GSX2_NT) (&s_w[0][0])[t] = 0;
            __syncthreads();
            uint2 p[IT];
            uint32_t dg[IT], rk[IT];
            bool ok[IT];
#pragma unroll
            for (int i = 0; i < IT; ++i) {      // wave-striped so that (wave, item, lane) order is element order
                const uint32_t e = tb + wave * (64 * IT) + i * 64 + lane;
                ok[i] = e < size;
                p[i] = ok[i] ? src[e] : make_uint2(0, 0);
                dg[i] = ok[i] ? digit_of(p[i]) : 256u;      // 256: matches no real digit
                const uint64_t act = __builtin_amdgcn_ballot_w64(ok[i]);
                uint64_t peers = gsx_match_digit(dg[i] & 255u, 8) & (ok[i] ? act : ~act);
                const uint32_t lower = gsx_mbcnt(peers);
                uint32_t prev = 0;
                if (ok[i] && lower == 0) prev = atomicAdd(&s_w[wave][dg[i]], static_cast<uint32_t>(__popcll(peers)));
                rk[i] = static_cast<uint32_t>(__shfl(prev, static_cast<int>(__builtin_ctzll(peers)))) + lower;
            }
            __syncthreads();
            if (tid < 256) {
                uint32_t acc = s_run[tid];
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const uint32_t c = s_w[w][tid];
                    s_w[w][tid] = acc;
                    acc += c;
                }
                s_run[tid] = acc;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < IT; ++i) {
                if (ok[i]) {
                    const uint32_t o = s_w[wave][dg[i]] + rk[i];
                    if (last) result[b0 + o] = p[i].y;
                    else dst[o] = p[i];
                }
            }
            __syncthreads();
        }
        uint2* t = src; src = dst; dst = t;
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct Gsx2Plan {
    bool ok = false;
    Gsx2Map map = {0, 0, 0, 0, 0};
    uint32_t nd = 0, idx_bits = 0, ncols = 0;
};

// Columns for up to m_cap keys: at least GSX2_MIN_COLS, a multiple of 64, none longer than GSX2_COLMAX keys; 0 = too many keys.
static inline uint32_t gsx2_cols_for(uint64_t m_cap) {
    uint32_t ncols = GSX2_MIN_COLS;
    while (ncols < GSX2_MAX_COLS && ((m_cap + ncols - 1) / ncols + 255) / 256 * 256 > GSX2_COLMAX) ncols += 64;
    return ((m_cap + ncols - 1) / ncols + 255) / 256 * 256 > GSX2_COLMAX ? 0u : ncols;
}

// key_span: largest relative key of the frame (host bound); m_cap: host bound of the selection size; ids are < n_ids.
// m_hint (0 = none): what the selection is expected to hold - the number of BUCKETS follows it (a cut of 1.1 M nodes out of a
// tree of 8.5 M wants ~1 K buckets, not 8 K: every workgroup of the partition initialises tables of that length); the number of
// COLUMNS always follows m_cap, because a column must hold its share of whatever the selection turns out to be.
static inline Gsx2Plan gsx2_plan(uint32_t key_span, uint32_t m_cap, uint64_t n_ids, uint32_t t_force = 0, bool any_size = false,
                                 uint32_t m_hint = 0) {
    Gsx2Plan p;
    if (key_span > 0x7FFFFFFFu) return p;
    const uint32_t fine_span = key_span < (1u << 25) ? key_span : (1u << 25) - 1u;
    uint32_t bits = 0;
    while ((fine_span >> bits) != 0) ++bits;
    uint32_t t = 10;
    const uint32_t m_for_t = (m_hint && m_hint < m_cap) ? m_hint : m_cap;
    while (t < GSX2_MAX_T && (1536ull << t) < m_for_t) ++t;      // ~1.5 K keys per bucket if they were spread evenly over the fine region
    if (t_force) t = t_force;
    // 13 bits behind a key range wider than the fine region: 8192 fine buckets would leave no table entries for the coarse
    // windows, so that plan cuts three windows finely instead of four and starts the coarse ones a halving later (7.4 K buckets)
    const bool wide13 = t == GSX2_MAX_T && key_span >= (1u << 25);
    if (bits < t + 5) return p;                                   // narrow key range: the LSD passes are as good
    if (m_cap < (1u << 17) && !any_size) return p;                // small sorts are launch-bound either way
    if (n_ids == 0 || m_cap == 0) return p;
    const uint32_t ncols = gsx2_cols_for(m_cap);
    if (!ncols) return p;
    p.ncols = ncols;
    p.map.r0 = bits - t;
    if (p.map.r0 > 22u) return p;
    p.map.flen = wide13 ? (3u << 23) : (1u << 25);
    p.map.rc = p.map.r0 + (wide13 ? 2u : 1u);
    p.map.nf = 24u - p.map.rc;
    p.map.base = p.map.flen >> p.map.r0;
    p.nd = (gsx2_bucket(key_span, p.map) + 1u + 127u) / 128u * 128u;
    if (p.nd > GSX2_MAX_D) return p;
    uint32_t ib = 1;
    while (ib < 32 && (n_ids - 1) >> ib) ++ib;
    p.idx_bits = ib;
    // k2_local packs the key bits below its bin and the index into one word: the fine buckets must allow that
    // (coarser buckets that cannot are sorted by k2_big when they hold more than a wave's worth)
    const uint32_t q_min = p.map.r0 + ib > 32 ? p.map.r0 + ib - 32 : 0u;
    if (q_min > 12u || q_min > p.map.r0) return p;
    p.ok = true;
    return p;
}

// Workspace beside the pair arrays: count matrix (2 B per column x bucket), prefix matrix (4 B), totals, bases, list, stats.
struct Gsx2Workspace {
    uint16_t* cnt;
    uint32_t *pre, *totals, *base;
    Gsx2Stats* stats;
};
static inline uint64_t gsx2_ws_bytes(uint64_t cap) {
    uint32_t ncols = gsx2_cols_for(cap);
    if (!ncols) ncols = GSX2_MIN_COLS;      // (such a scene never takes this path; keep the layout valid)
    return static_cast<uint64_t>(ncols) * GSX2_MAX_D * (2 + 4) + (GSX2_MAX_D * 2 + 64) * 4ull + 256;
}
static inline Gsx2Workspace gsx2_ws_carve(unsigned char* ws, uint64_t cap) {
    uint32_t ncols = gsx2_cols_for(cap);
    if (!ncols) ncols = GSX2_MIN_COLS;
    Gsx2Workspace w;
    w.pre = reinterpret_cast<uint32_t*>(ws);
    w.cnt = reinterpret_cast<uint16_t*>(ws + static_cast<uint64_t>(ncols) * GSX2_MAX_D * 4);
    w.totals = reinterpret_cast<uint32_t*>(ws + static_cast<uint64_t>(ncols) * GSX2_MAX_D * 6);
    w.base = w.totals + GSX2_MAX_D;
    w.stats = reinterpret_cast<Gsx2Stats*>(w.base + GSX2_MAX_D + 64);
    return w;
}
