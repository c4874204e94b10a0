// gsx_radix.hpp — stable LSD radix sort of (key, u32 value) pairs for gfx950,
// as three kernels per digit pass:
//
//   H  k_radix_hist     digit counts of every column range (contiguous keys, one per tile
//                       while that fits two rounds of workgroups); block b takes columns
//                       b, b + G, ... (replicated LDS atomics, 16-byte loads)
//   S  k_radix_scan     one block per digit: exclusive scan of that digit's counts
//                       over columns (+ the digit's total)
//   X  k_radix_scatter  the same column walk: ranks a tile's keys with wave64 ballots,
//                       stages them through LDS in digit order and writes them out;
//                       per-digit running offsets live in registers of the digit's owner
//
// Replaces the reference's std::sort(indices, distances[a] > distances[b])
// (SplatMesh.h:97-108); the same passes order the binner's (tile, splat) pairs.
// Stability turns "input in ascending index order" into the contract's tie rule
// (distance descending, then index ascending).
//
// Why not a chained-scan ("onesweep") pass: on MI355X the look-back's status words
// cross eight non-coherent XCD L2s, so every hop is a fabric round trip; measured
// on 6M keys the look-back cost 25 us of a 66 us pass, and the whole pass is
// latency- not bandwidth-bound.  Here no workgroup ever waits on another one: the
// only cross-workgroup data is the small count matrix handed over at kernel
// boundaries, the workgroups of an XCD write neighbouring columns at the same time
// (their partial lines merge in that XCD's L2), and nothing can spin or hang.
//
// Ranking uses no LDS read-modify-write chain: per item 8-9 ballots give the
// peer mask, the lowest peer issues ONE returning ds_add, and results are
// broadcast afterwards, so the 12 items of a thread are in flight together.
#pragma once
#include "gsx_internal.h"

#define GSX_RADIX_MAX_BITS 9
#define GSX_RADIX_MAX_DIGITS (1 << GSX_RADIX_MAX_BITS)
#define GSX_RADIX_MAX_BLOCKS 2048

__device__ __forceinline__ uint32_t gsx_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ uint32_t gsx_mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

// Exclusive scan of one value per thread over a block of NT threads.
template <int NT>
__device__ __forceinline__ uint32_t gsx_block_excl_scan(uint32_t v, uint32_t* s_tmp, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if (lane >= static_cast<uint32_t>(off)) incl += t;
    }
    if (lane == 63) s_tmp[wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        const uint32_t t = s_tmp[w];
        if (static_cast<uint32_t>(w) < wave) wave_off += t;
        tot += t;
    }
    __syncthreads();
    if (total) *total = tot;
    return wave_off + incl - v;
}

// Lanes of the wave holding the same digit as this lane (all 64 lanes must call).  Per digit bit: one ballot,
// then every lane ORs in the lanes whose bit differs from its own (ballot ^ -mybit); kept in 32-bit halves so
// that a bit costs ~6 VALU instructions (a 64-bit select-and-AND formulation compiled to 9).
__device__ __forceinline__ uint64_t gsx_match_digit(uint32_t d, uint32_t bits) {
    uint32_t mis_lo = 0u, mis_hi = 0u;
#pragma unroll
    for (int b = 0; b < GSX_RADIX_MAX_BITS; ++b) {
        if (static_cast<uint32_t>(b) < bits) {   // wave-uniform
            const uint32_t bit = (d >> b) & 1u;
            const uint64_t bal = __builtin_amdgcn_ballot_w64(bit != 0u);
            const uint32_t mine = 0u - bit;                         // 0 or 0xFFFFFFFF
            mis_lo |= static_cast<uint32_t>(bal) ^ mine;
            mis_hi |= static_cast<uint32_t>(bal >> 32) ^ mine;
        }
    }
    return ~((static_cast<uint64_t>(mis_hi) << 32) | mis_lo);
}

// Logical range index of this workgroup.  Workgroups are dealt round-robin over the 8
// XCDs (blockIdx % 8 labels the XCD group), each with a private L2; giving every XCD
// group one contiguous eighth of the key space makes neighbouring ranges -- whose
// output runs are adjacent for every digit -- share an L2, so partial-line writes
// combine before they leave the chip.  Bijective for any grid size.  Speed only:
// correctness never depends on placement.
__device__ __forceinline__ uint32_t gsx_logical_block() {
#ifdef GSX_NO_XCD_REMAP
    return blockIdx.x;
#else
    const uint32_t g = gridDim.x, orig = blockIdx.x;
    const uint32_t q = g / 8, r = g % 8, xcd = orig % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
#endif
}

// Keys [begin, end) owned by logical block `b` when `count` keys are dealt to `g` blocks in
// equal contiguous ranges (range length rounded up to 256 keys so every range starts on
// a 1-KiB boundary).  The grid is sized to the machine (one round of resident
// workgroups), so all blocks carry the same load and there is no tail round.
__device__ __forceinline__ void gsx_block_range(uint32_t count, uint32_t g, uint32_t b, uint32_t& begin, uint32_t& end) {
    const uint64_t per = ((static_cast<uint64_t>(count) + g - 1) / g + 255) / 256 * 256;
    const uint64_t lo = per * b;
    begin = static_cast<uint32_t>(lo < count ? lo : count);
    end = static_cast<uint32_t>(lo + per < count ? lo + per : count);
}

// Number of column ranges a pass over `count` keys uses: one per tile while that is at most `max_cols` (= two rounds
// of resident workgroups), else max_cols / 2 longer ranges (one per resident workgroup).  Every kernel of a pass
// derives it from the same device-side count, so the count matrix stride always agrees.
__host__ __device__ __forceinline__ uint32_t gsx_n_cols(uint32_t count, uint32_t tile, uint32_t max_cols) {
    const uint32_t t = (count + tile - 1u) / tile;
    return t < 1u ? 1u : (t > max_cols ? (max_cols / 2u > 0u ? max_cols / 2u : 1u) : t);
}

// Contention-tolerant LDS digit histogram: GSX_HREP interleaved copies, lane l uses copy
// l % GSX_HREP, so even a constant digit is only an 8-way conflict.
#define GSX_HREP 8
__device__ __forceinline__ void gsx_hist_zero(uint32_t* s_hist, int nt) {
    for (uint32_t t = threadIdx.x; t < GSX_RADIX_MAX_DIGITS * GSX_HREP; t += nt) s_hist[t] = 0;
    __syncthreads();
}
__device__ __forceinline__ void gsx_hist_add(uint32_t* s_hist, uint32_t d) {
    atomicAdd(&s_hist[d * GSX_HREP + (threadIdx.x & (GSX_HREP - 1))], 1u);
}
__device__ __forceinline__ void gsx_hist_store(const uint32_t* s_hist, uint32_t nd, int nt, uint32_t* __restrict__ cnt, uint32_t n_cols,
                                               uint32_t col) {
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < nd; d += nt) {
        uint32_t acc = 0;
#pragma unroll
        for (int r = 0; r < GSX_HREP; ++r) acc += s_hist[d * GSX_HREP + r];
        cnt[static_cast<uint64_t>(d) * n_cols + col] = acc;
    }
}

// ---- H: digit counts of each column range -> cnt[d * n_cols + col] (block b: columns b, b + G, ...) -------
template <typename KeyT, int NT>
__global__ __launch_bounds__(NT) void k_radix_hist(const KeyT* __restrict__ keys, const uint32_t* __restrict__ count_ptr,
                                                   uint32_t shift, uint32_t bits, uint32_t* __restrict__ cnt, uint32_t tile,
                                                   uint32_t max_cols) {
    __shared__ uint32_t s_hist[GSX_RADIX_MAX_DIGITS * GSX_HREP];
    const uint32_t nd = 1u << bits, mask = nd - 1u;
    const uint32_t count = *count_ptr;
    if (count == 0u) return;                 // (nothing to sort, or a gated pass that is switched off: the scatter returns too)
    const uint32_t n_cols = gsx_n_cols(count, tile, max_cols);
    for (uint32_t col = gsx_logical_block(); col < n_cols; col += gridDim.x) {
        gsx_hist_zero(s_hist, NT);
        uint32_t begin, end;
        gsx_block_range(count, n_cols, col, begin, end);
        constexpr int VEC = 16 / sizeof(KeyT);          // keys per 16-byte load; ranges start 16-byte aligned
        using VecT = uint4;
        const uint32_t n_vec = (end - begin) / VEC;
        const VecT* kv = reinterpret_cast<const VecT*>(keys + begin);
        for (uint32_t i = threadIdx.x; i < n_vec; i += NT) {
            const VecT v = kv[i];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (sizeof(KeyT) == 4) {
                    gsx_hist_add(s_hist, (w[j] >> shift) & mask);
                } else {
                    gsx_hist_add(s_hist, ((w[j] & 0xFFFFu) >> shift) & mask);
                    gsx_hist_add(s_hist, ((w[j] >> 16) >> shift) & mask);
                }
            }
        }
        for (uint32_t g = begin + n_vec * VEC + threadIdx.x; g < end; g += NT)
            gsx_hist_add(s_hist, (static_cast<uint32_t>(keys[g]) >> shift) & mask);
        gsx_hist_store(s_hist, nd, NT, cnt, n_cols, col);
        __syncthreads();
    }
}

// ---- S: per digit, exclusive scan over blocks (in place) + digit total --------
template <int DUMMY>
__global__ __launch_bounds__(256) void k_radix_scan(uint32_t* __restrict__ cnt, const uint32_t* __restrict__ count_ptr, uint32_t tile,
                                                   uint32_t max_cols, uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_tmp[4];
    if (*count_ptr == 0u) return;
    const uint32_t g = gsx_n_cols(*count_ptr, tile, max_cols);
    uint32_t* row = cnt + static_cast<uint64_t>(blockIdx.x) * g;
    constexpr int PER = GSX_RADIX_MAX_BLOCKS / 256;   // 8 consecutive entries per thread
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t i = threadIdx.x * PER + j;
        v[j] = (i < g) ? row[i] : 0u;
        sum += v[j];
    }
    uint32_t total;
    uint32_t run = gsx_block_excl_scan<256>(sum, s_tmp, &total);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t i = threadIdx.x * PER + j;
        if (i < g) row[i] = run;
        run += v[j];
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

// ---- X: rank + scatter ---------------------------------------------------------
// The keys are dealt to `n_cols` contiguous column ranges (one count column each); block b works on columns
// b, b + G, b + 2G, ... (b = logical, XCD-contiguous index).  At any moment the workgroups of one XCD therefore
// write neighbouring columns, whose runs for a digit are adjacent in memory, and that XCD's L2 merges them into
// whole lines before they leave the chip (measured: 100 MB -> 52 MB written per 6 M-key pass against one
// two-tile range per block).  The next tile's keys are requested before the current tile is written out.
// MAXD: digits the pass can have at most (LDS is sized for it: the pair sort's 6..8-bit digits leave room for a third
// workgroup per CU); WPE: waves per SIMD the register budget is cut for.
template <typename KeyT, int NT, int IPT, bool IOTA_VALS, int MAXD = GSX_RADIX_MAX_DIGITS, int WPE = 4>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_radix_scatter(const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                      KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                      const uint32_t* __restrict__ count_ptr, uint32_t shift, uint32_t bits,
                                                      const uint32_t* __restrict__ cnt_scanned, const uint32_t* __restrict__ totals,
                                                      uint32_t max_cols) {
    constexpr int NW = NT / 64;
    constexpr int TILE = NT * IPT;
    __shared__ KeyT s_keys[TILE];
    __shared__ uint32_t s_vals[TILE];
    __shared__ uint32_t s_whist[NW][MAXD];   // per-wave digit counts, then wave prefixes + bin base
    __shared__ uint32_t s_gofs[MAXD];        // global index = s_gofs[d] + LDS slot
    __shared__ uint32_t s_tmp[NW];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nd = 1u << bits, mask = nd - 1u;
    const uint32_t count = *count_ptr;
    const uint32_t n_cols = gsx_n_cols(count, NT * IPT, max_cols);
    uint32_t col = gsx_logical_block();
    uint32_t rbegin, rend;
    if (col >= n_cols) return;
    gsx_block_range(count, n_cols, col, rbegin, rend);
    if (rbegin >= rend) return;
    uint32_t tile_base = rbegin;

    // Thread t owns the PER consecutive digits t*PER .. t*PER+PER-1: their bases (exclusive scan of the digit
    // totals), the column's scanned counts and the running output offsets stay in its registers.
    constexpr int PER = MAXD / NT > 0 ? MAXD / NT : 1;   // (threads past the last digit own none)
    static_assert(PER * NT >= MAXD, "every digit needs an owner");
    KeyT key[IPT];
    uint32_t val[IPT];
    uint32_t col_cnt[PER], base[PER], run[PER];
    auto request = [&](uint32_t tb, uint32_t re) {     // wave-striped: item i of lane l = wave_base + i*64 + l
        const uint32_t wave_base = tb + wave * (64 * IPT);
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const uint32_t g = wave_base + i * 64 + lane;
            key[i] = (g < re) ? keys_in[g] : static_cast<KeyT>(~static_cast<KeyT>(0));
            val[i] = IOTA_VALS ? g : ((g < re) ? vals_in[g] : 0u);
        }
    };
    auto request_col = [&](uint32_t c) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t d = tid * PER + j;
            col_cnt[j] = (d < nd) ? cnt_scanned[static_cast<uint64_t>(d) * n_cols + c] : 0u;
        }
    };
    request(tile_base, rend);
    request_col(col);
    {
        uint32_t tot[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t d = tid * PER + j;
            tot[j] = (d < nd) ? totals[d] : 0u;
            sum += tot[j];
        }
        uint32_t acc = gsx_block_excl_scan<NT>(sum, s_tmp, nullptr);
#pragma unroll
        for (int j = 0; j < PER; ++j) { base[j] = acc; acc += tot[j]; }
    }
    bool new_col = true;

    for (;;) {
        for (uint32_t t = tid; t < NW * MAXD; t += NT) (&s_whist[0][0])[t] = 0;
        __syncthreads();   // also: previous tile's write-out has finished reading s_keys/s_vals/s_gofs

        // ---- rank inside the wave: ballots + one returning LDS add per (item, digit) ----
        uint32_t* wh = s_whist[wave];
        uint32_t lead_prev[IPT], pk[IPT];   // pk = lanes below with the same digit | leader lane << 16
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const uint32_t d = (static_cast<uint32_t>(key[i]) >> shift) & mask;
#ifdef GSX_ABL_NORANK   // (GSX_ABL_*: ablation switches for tools/sortbench.hip, never defined in the library build)
            const uint64_t peers = 1ull << lane;
#else
            const uint64_t peers = gsx_match_digit(d, bits);
#endif
            const uint32_t lower = gsx_mbcnt(peers);
            pk[i] = lower | (static_cast<uint32_t>(__builtin_ctzll(peers)) << 16);
            lead_prev[i] = 0;
            if (lower == 0) lead_prev[i] = atomicAdd(&wh[d], static_cast<uint32_t>(__popcll(peers)));
        }
        uint32_t rank[IPT];
#pragma unroll
        for (int i = 0; i < IPT; ++i) rank[i] = static_cast<uint32_t>(__shfl(lead_prev[i], pk[i] >> 16)) + (pk[i] & 0xFFFFu);
        __syncthreads();

        // ---- per digit: prefix over waves, bin base, global offset ----
        {
            uint32_t acc[PER], sum = 0;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t d = tid * PER + j;
                acc[j] = 0;
                if (d < nd) {
#pragma unroll
                    for (int w = 0; w < NW; ++w) acc[j] += s_whist[w][d];
                }
                sum += acc[j];
                if (new_col) run[j] = base[j] + col_cnt[j];
            }
            uint32_t bin_base = gsx_block_excl_scan<NT>(sum, s_tmp, nullptr);
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t d = tid * PER + j;
                if (d < nd) {
                    uint32_t pre = bin_base;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        const uint32_t c = s_whist[w][d];
                        s_whist[w][d] = pre;       // bin base + counts of the lower waves
                        pre += c;
                    }
                    s_gofs[d] = run[j] - bin_base;
                    run[j] += acc[j];
                }
                bin_base += acc[j];
            }
        }
        __syncthreads();

        // ---- scatter into LDS in digit order ----
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const uint32_t d = (static_cast<uint32_t>(key[i]) >> shift) & mask;
            const uint32_t pos = s_whist[wave][d] + rank[i];
            s_keys[pos] = key[i];
            s_vals[pos] = val[i];
        }
        const uint32_t valid = (rend - tile_base < static_cast<uint32_t>(TILE)) ? (rend - tile_base) : TILE;

        // ---- next tile (of this column, else the block's next column): request its keys now ----
        bool more = true;
        tile_base += TILE;
        new_col = false;
        if (tile_base >= rend) {
            col += gridDim.x;
            new_col = true;
            more = col < n_cols;
            if (more) {
                gsx_block_range(count, n_cols, col, rbegin, rend);
                tile_base = rbegin;
                more = rbegin < rend;
            }
        }
        if (more) {
            request(tile_base, rend);
            if (new_col) request_col(col);
        }
        __syncthreads();

        // ---- write-out: consecutive LDS slots of one digit are consecutive in HBM ----
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const uint32_t p = i * NT + tid;
            if (p < valid) {
                const KeyT k = s_keys[p];
                const uint32_t d = (static_cast<uint32_t>(k) >> shift) & mask;
#ifdef GSX_ABL_LINWRITE
                const uint32_t dst = tile_base - TILE + p + (s_gofs[d] == 0xFFFFFFFFu);
#else
                const uint32_t dst = s_gofs[d] + p;
#endif
#ifdef GSX_ABL_NOWRITE
                if (k == static_cast<KeyT>(0x12345677u) && dst == 0x7FFFFFFFu)
#endif
                if (dst < count) {   // (always true for a consistent count matrix; a corrupted one must not become a stray write)
                if (keys_out) keys_out[dst] = k;     // (null: the last pass of a sort whose keys nobody reads afterwards)
                vals_out[dst] = s_vals[p];
                }
            }
        }
        if (!more) break;
    }
}

// ---- host side -------------------------------------------------------------------
struct GsxRadixPlan {
    int n_passes = 0;
    uint32_t shift[4] = {0, 0, 0, 0};
    uint32_t bits[4] = {0, 0, 0, 0};
};

// Split `total_bits` key bits into the fewest passes of at most GSX_RADIX_MAX_BITS bits, balanced.
static inline GsxRadixPlan gsx_radix_plan(uint32_t total_bits) {
    GsxRadixPlan p;
    if (total_bits == 0) total_bits = 1;
    p.n_passes = static_cast<int>((total_bits + GSX_RADIX_MAX_BITS - 1) / GSX_RADIX_MAX_BITS);
    uint32_t left = total_bits, shift = 0;
    for (int i = 0; i < p.n_passes; ++i) {
        const uint32_t b = (left + (p.n_passes - i) - 1) / (p.n_passes - i);
        p.shift[i] = shift;
        p.bits[i] = b;
        shift += b;
        left -= b;
    }
    return p;
}

// Upper bound of the column ranges of a sort (gsx_n_cols): two rounds of `slots` resident workgroups.
static inline uint32_t gsx_radix_columns(uint32_t slots) {
    uint32_t t = 2u * slots;
    if (t < 2u) t = 2u;
    if (t > GSX_RADIX_MAX_BLOCKS) t = GSX_RADIX_MAX_BLOCKS;
    return t;
}
// Grid of a sort of up to `cap` keys: one round of resident workgroups (`slots`), but never
// more blocks than tiles.
static inline uint32_t gsx_radix_blocks(uint64_t cap, uint32_t tile, uint32_t slots) {
    uint64_t t = (cap + tile - 1) / tile;
    if (t < 1) t = 1;
    if (t > slots) t = slots;
    if (t > GSX_RADIX_MAX_BLOCKS) t = GSX_RADIX_MAX_BLOCKS;
    return static_cast<uint32_t>(t);
}
// words of the count matrix + totals for one sort with up to `blocks` column ranges
static inline uint64_t gsx_radix_workspace_words(uint32_t blocks) {
    return static_cast<uint64_t>(GSX_RADIX_MAX_DIGITS) * blocks + GSX_RADIX_MAX_DIGITS;
}
