// tools/gsx_radix2_experiment.hpp — EXPERIMENT, not part of libgsx (round 2): a second-generation scatter pass with
// packed 8-byte (key, index) pairs, LDS-mask digit matching and hand-batched LDS round trips.  Bit-exact, half the
// instructions of k_radix_scatter -- and no faster ("what bounds the scatter"); kept with
// tools/sortbench2.hip as the record of that measurement.
// What changed against k_radix_scatter (gsx_radix.hpp), and why (rocprofv3 SQ counters of round 1: 27 % of wave
// cycles issuing, 47 % parked on s_waitcnt / barriers; HBM traffic 1.04 x algorithmic, so bytes were not the problem):
//
//   * pairs: a key and its splat id are one uint2 in LDS and in HBM.  Half the LDS instructions (ds_write_b64 /
//     ds_read_b64 instead of two b32 each), half the global store instructions, and a digit's run of ~12 items in
//     a tile is one 96-byte piece of one array instead of two 48-byte pieces of two arrays.  The first pass still
//     reads bare u32 keys (the key kernel's output; the index is the position or sel[position]) and the last pass
//     writes bare u32 indices (the permutation), so the packed form costs no HBM bytes: 12 + 16 + 12 B per key over
//     three passes instead of 12 + 16 + 16.
//   * matching through LDS.  Finding the lanes of a wave that hold the same digit cost 5 VALU + 2 s_nop per digit
//     bit and item with ballots (63 issue slots per item of a 9-bit pass, half the kernel's vector instructions;
//     rocprofv3: VALU issue of the 4 waves on a SIMD covers ~2/3 of the kernel's duration).  Now the low 7 bits go
//     through a per-wave table of 64-bit lane masks in LDS (OR own bit, read back, clear: 3 LDS instructions) and
//     only the bits above 7 use ballots.
//   * branch-free loads (clamped index + pad fix-up on the one ragged tile), no barrier at the top of the loop (each
//     wave clears its own count row after the last barrier of a tile), one barrier inside the block scan.
//
// Stability, column walk, XCD-aware block -> column map and the count / scan kernels are those of gsx_radix.hpp.
#pragma once
#include "gsx_radix.hpp"

enum { GSX_IN_KEYS_IOTA = 0, GSX_IN_KEYS_VALS = 1, GSX_IN_PAIRS = 2 };
enum { GSX_OUT_PAIRS = 0, GSX_OUT_VALS = 1, GSX_OUT_BOTH = 2 };

#define GSX_MATCH_BITS 7                         // digit bits resolved by the LDS mask table; the rest by ballots
#define GSX_MATCH_SLOTS (1 << GSX_MATCH_BITS)
#ifndef GSX_RANK_BATCH
#define GSX_RANK_BATCH 6                         // items whose match round trips are in flight together
#endif

// Lanes of the wave holding the same digit as this lane (all 64 lanes call).  The low GSX_MATCH_BITS bits of the digit
// pick a slot of the wave's own mask table: every lane ORs its lane bit into its slot, reads the slot back and clears
// it — three LDS instructions, executed in order for the whole wave, instead of 5 VALU + 2 s_nop per bit.  Bits above
// GSX_MATCH_BITS (two of a 9-bit digit) are resolved with ballots as in gsx_match_digit.  Split in two so that a
// caller can put the LDS round trips of several items in flight before it consumes the first mask.
__device__ __forceinline__ uint64_t gsx_match_issue(uint32_t d, unsigned long long* table, uint32_t lane) {
    unsigned long long* slot = table + (d & (GSX_MATCH_SLOTS - 1u));
    __hip_atomic_fetch_or(slot, 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    const uint64_t cand = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __hip_atomic_store(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    return cand;
}
__device__ __forceinline__ uint64_t gsx_match_finish(uint64_t cand, uint32_t d, uint32_t bits) {
    if (bits > GSX_MATCH_BITS) {                 // (wave-uniform)
        uint32_t mis_lo = 0u, mis_hi = 0u;
#pragma unroll
        for (int b = GSX_MATCH_BITS; b < GSX_RADIX_MAX_BITS; ++b) {
            if (static_cast<uint32_t>(b) < bits) {
                const uint32_t mine = static_cast<uint32_t>(static_cast<int32_t>(d << (31 - b)) >> 31);   // 0 or ~0
                const uint64_t bal = __builtin_amdgcn_ballot_w64(mine != 0u);
                mis_lo |= static_cast<uint32_t>(bal) ^ mine;
                mis_hi |= static_cast<uint32_t>(bal >> 32) ^ mine;
            }
        }
        cand &= ~((static_cast<uint64_t>(mis_hi) << 32) | mis_lo);
    }
    return cand;
}

#ifdef GSX_STAMP   // diagnostic build of tools/sortbench2 only: per-phase shader-clock totals of every wave
__device__ unsigned long long g_stamps[GSX_RADIX_MAX_BLOCKS * 8 * 8];
#define GSX_STAMP_AT(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[k] += now_ - st_last; st_last = now_; } while (0)
#else
#define GSX_STAMP_AT(k) do {} while (0)
#endif

template <int NT, int IPT, int MODE_IN, int MODE_OUT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_radix_scatter2(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, const uint2* __restrict__ pairs_in,
    uint2* __restrict__ pairs_out, uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ count_ptr, uint32_t shift,
    uint32_t bits, const uint32_t* __restrict__ cnt_scanned, const uint32_t* __restrict__ totals, uint32_t max_cols) {
    constexpr int NW = NT / 64;
    constexpr int TILE = NT * IPT;
    constexpr int ND = GSX_RADIX_MAX_DIGITS;
    static_assert(NT == ND, "thread t owns digit t");
    __shared__ __attribute__((aligned(16))) uint2 s_pairs[TILE];
    __shared__ __attribute__((aligned(16))) uint32_t s_whist[NW][ND];   // per-wave digit counts, then the wave's first in-tile position per digit
    __shared__ __attribute__((aligned(16))) unsigned long long s_match[NW][GSX_MATCH_SLOTS];
    __shared__ uint32_t s_gofs[ND];                                      // global index = s_gofs[d] + in-tile position
    __shared__ uint32_t s_tmp[2][NW];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(tid >> 6)));
    const uint32_t nd = 1u << bits, mask = nd - 1u;
    const uint32_t count = *count_ptr;
    const uint32_t n_cols = gsx_n_cols(count, TILE, max_cols);
    uint32_t col = gsx_logical_block();
    uint32_t rbegin, rend;
    if (col >= n_cols) return;
    gsx_block_range(count, n_cols, col, rbegin, rend);
    if (rbegin >= rend) return;
    uint32_t tile_base = rbegin;

    uint32_t key[IPT], val[IPT];
    // wave-striped: item i of lane l = wave_base + i*64 + l.  Branch-free: a lane past the range reads the range's last
    // element and turns its key into the pad value (all ones: highest digit, last in memory order, so pads take the tile's
    // last LDS slots and are never written out).
    auto request = [&](uint32_t tb, uint32_t re) {
        const uint32_t wave_base = tb + wave * (64 * IPT), last = re - 1u;
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const uint32_t g = wave_base + i * 64 + lane, gc = g < last ? g : last;
#ifdef GSX_ABL2_NOLOAD
            if (true) { key[i] = (g * 2654435761u) >> 5; val[i] = gc; } else
#endif
            if (MODE_IN == GSX_IN_PAIRS) {
                const uint2 p = pairs_in[gc];
                key[i] = p.x; val[i] = p.y;
            } else {
                key[i] = keys_in[gc];
                val[i] = MODE_IN == GSX_IN_KEYS_IOTA ? g : vals_in[gc];
            }
        }
    };
    auto pad = [&](uint32_t tb, uint32_t re) {
        if (re - tb >= static_cast<uint32_t>(TILE)) return;          // (uniform: full tiles need nothing)
        const uint32_t wave_base = tb + wave * (64 * IPT);
#pragma unroll
        for (int i = 0; i < IPT; ++i)
            if (wave_base + i * 64 + lane >= re) key[i] = 0xFFFFFFFFu;
    };
    // thread t: base (exclusive scan of the digit totals), the column's scanned count and the running output offset of digit t
    uint32_t col_cnt = 0, base = 0, run = 0;
    auto request_col = [&](uint32_t c) { col_cnt = (tid < nd) ? cnt_scanned[static_cast<uint64_t>(tid) * n_cols + c] : 0u; };
    request(tile_base, rend);
    request_col(col);
    {   // own count row and own mask table start at zero (each wave clears its own; nobody else touches them before barrier A)
        uint4* row = reinterpret_cast<uint4*>(&s_whist[wave][0]);
        row[lane] = make_uint4(0u, 0u, 0u, 0u);
        row[64 + lane] = make_uint4(0u, 0u, 0u, 0u);
        reinterpret_cast<uint4*>(&s_match[wave][0])[lane] = make_uint4(0u, 0u, 0u, 0u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // (compiler only)
    }
    {
        const uint32_t tot = (tid < nd) ? totals[tid] : 0u;
        base = gsx_block_excl_scan<NT>(tot, s_tmp[0], nullptr);
    }
    bool new_col = true;
    uint32_t flip = 0;
#ifdef GSX_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif

    for (;;) {
        pad(tile_base, rend);
        GSX_STAMP_AT(0);   // waiting for the tile's keys (pad touches them)
        // ---- rank inside the wave: LDS mask match + one returning LDS add per (item, digit) ----
        uint32_t* wh = s_whist[wave];
        // The LDS round trips are batched by hand (GSX_RANK_BATCH items in flight): written item by item, the
        // compiler emits or / read / clear / wait / add / wait per item and a wave spends the phase waiting for LDS.
        uint32_t lead_prev[IPT], pk[IPT];   // pk = lanes below with the same digit | leader lane << 16
        constexpr int RB = GSX_RANK_BATCH;
        static_assert(IPT % RB == 0, "whole batches");
#pragma unroll
        for (int i0 = 0; i0 < IPT; i0 += RB) {
            uint64_t cand[RB];
#pragma unroll
            for (int k = 0; k < RB; ++k)
#ifdef GSX_ABL2_NOMATCH
                cand[k] = 1ull << lane;
#else
                cand[k] = gsx_match_issue((key[i0 + k] >> shift) & mask, s_match[wave], lane);
#endif
            uint32_t cnt[RB];
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const uint32_t d = (key[i0 + k] >> shift) & mask;
                const uint64_t peers = gsx_match_finish(cand[k], d, bits);
                pk[i0 + k] = gsx_mbcnt(peers) | (static_cast<uint32_t>(__builtin_ctzll(peers)) << 16);
                cnt[k] = static_cast<uint32_t>(__popcll(peers));
            }
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const uint32_t d = (key[i0 + k] >> shift) & mask;
                lead_prev[i0 + k] = 0;
#ifdef GSX_ABL2_NOADD
                lead_prev[i0 + k] = d + cnt[k];
#else
                if ((pk[i0 + k] & 0xFFFFu) == 0u) lead_prev[i0 + k] = atomicAdd(&wh[d], cnt[k]);
#endif
            }
        }
        uint32_t rank[IPT];
#pragma unroll
        for (int i = 0; i < IPT; ++i) rank[i] = static_cast<uint32_t>(__shfl(lead_prev[i], pk[i] >> 16)) + (pk[i] & 0xFFFFu);
        GSX_STAMP_AT(1);   // ranking
        __syncthreads();   // (A) counts complete; every wave has finished writing out the previous tile
        GSX_STAMP_AT(2);   // barrier A

        // ---- per digit (thread d): prefix over waves, bin base, global offset ----
        {
            uint32_t c[NW], acc = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { c[w] = s_whist[w][tid]; acc += c[w]; }
            // exclusive scan of acc over the block; the wave totals alternate between two LDS rows, so one barrier suffices
            uint32_t incl = acc;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(incl, off);
                if (lane >= static_cast<uint32_t>(off)) incl += t;
            }
            if (lane == 63) s_tmp[flip][wave] = incl;
            __syncthreads();
            uint32_t wave_off = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const uint32_t t = s_tmp[flip][w];
                if (static_cast<uint32_t>(w) < wave) wave_off += t;
            }
            flip ^= 1u;
            uint32_t pre = wave_off + incl - acc;        // bin base of digit tid
            if (new_col) run = base + col_cnt;
            s_gofs[tid] = run - pre;
            run += acc;
#pragma unroll
            for (int w = 0; w < NW; ++w) { s_whist[w][tid] = pre; pre += c[w]; }
        }
        __syncthreads();   // (C)
        GSX_STAMP_AT(3);   // prefix step incl. its barrier and barrier C

        // ---- scatter into LDS in digit order (all first positions fetched, then all pairs stored) ----
        {
            uint32_t pos[IPT];
#pragma unroll
            for (int i = 0; i < IPT; ++i) pos[i] = wh[(key[i] >> shift) & mask];
#pragma unroll
            for (int i = 0; i < IPT; ++i)
#ifdef GSX_ABL2_LINSCATTER
                s_pairs[wave * (64 * IPT) + i * 64 + lane + (pos[i] + rank[i] == 0x7FFFFFFFu)] = make_uint2(key[i], val[i]);
#else
                s_pairs[pos[i] + rank[i]] = make_uint2(key[i], val[i]);
#endif
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
        GSX_STAMP_AT(4);   // scatter into LDS + next tile's requests
        __syncthreads();   // (B) the tile is in LDS in digit order, s_gofs is set; nobody reads the count rows any more
        GSX_STAMP_AT(5);   // barrier B

        {   // clear the own count row for the next tile (the mask table clears itself)
            uint4* row = reinterpret_cast<uint4*>(&s_whist[wave][0]);
            row[lane] = make_uint4(0u, 0u, 0u, 0u);
            row[64 + lane] = make_uint4(0u, 0u, 0u, 0u);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
        // ---- write-out: consecutive LDS slots of one digit are consecutive in HBM.  Three batched steps (pairs,
        //      their digits' global offsets, stores) instead of twelve dependent read -> read -> store chains. ----
        {
            uint2 pr[IPT];
#pragma unroll
            for (int i = 0; i < IPT; ++i) pr[i] = s_pairs[i * NT + tid];
            uint32_t dst[IPT];
#pragma unroll
            for (int i = 0; i < IPT; ++i) dst[i] = s_gofs[(pr[i].x >> shift) & mask];
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                const uint32_t p = i * NT + tid;
#if defined(GSX_ABL2_NOSTORE)          // (GSX_ABL2_*: ablation switches of tools/sortbench2, never defined in the library build)
                if (p < valid && pr[i].x == 0x12345677u && dst[i] == 0x7FFFFFFFu)
#elif defined(GSX_ABL2_LINSTORE)
                dst[i] = tile_base - TILE + (dst[i] == 0xFFFFFFFFu);
                if (p < valid)
#else
                if (p < valid)
#endif
                {
                    if (MODE_OUT != GSX_OUT_VALS) pairs_out[dst[i] + p] = pr[i];
                    if (MODE_OUT != GSX_OUT_PAIRS) vals_out[dst[i] + p] = pr[i].y;
                }
            }
        }
        GSX_STAMP_AT(6);   // write-out (issue only: the stores drain asynchronously)
        if (!more) break;
    }
#ifdef GSX_STAMP
    if (lane == 0)
        for (int k = 0; k < 8; ++k) g_stamps[(static_cast<uint64_t>(blockIdx.x) * 8 + wave) * 8 + k] = st_acc[k];
#endif
}

// ---- H for packed pairs: digit counts of each column range (keys are the .x words) ----
template <int NT>
__global__ __launch_bounds__(NT) void k_radix_hist_pairs(const uint2* __restrict__ pairs, const uint32_t* __restrict__ count_ptr,
                                                         uint32_t shift, uint32_t bits, uint32_t* __restrict__ cnt, uint32_t tile,
                                                         uint32_t max_cols) {
    __shared__ uint32_t s_hist[GSX_RADIX_MAX_DIGITS * GSX_HREP];
    const uint32_t nd = 1u << bits, mask = nd - 1u;
    const uint32_t count = *count_ptr;
    if (count == 0u) return;
    const uint32_t n_cols = gsx_n_cols(count, tile, max_cols);
    for (uint32_t col = gsx_logical_block(); col < n_cols; col += gridDim.x) {
        gsx_hist_zero(s_hist, NT);
        uint32_t begin, end;
        gsx_block_range(count, n_cols, col, begin, end);           // begin is a multiple of 256 pairs: 16-byte aligned
        const uint32_t n_vec = (end - begin) / 2u;
        const uint4* pv = reinterpret_cast<const uint4*>(pairs + begin);
        constexpr int UN = 6;                                      // a 6144-pair column is 6 loads per thread, all in flight
        for (uint32_t i0 = threadIdx.x; i0 < n_vec; i0 += NT * UN) {
            uint4 v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const uint32_t i = i0 + u * NT;
                v[u] = (i < n_vec) ? pv[i] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (i0 + u * NT < n_vec) {
                    gsx_hist_add(s_hist, (v[u].x >> shift) & mask);
                    gsx_hist_add(s_hist, (v[u].z >> shift) & mask);
                }
            }
        }
        for (uint32_t g = begin + n_vec * 2u + threadIdx.x; g < end; g += NT) gsx_hist_add(s_hist, (pairs[g].x >> shift) & mask);
        gsx_hist_store(s_hist, nd, NT, cnt, n_cols, col);
        __syncthreads();
    }
}
