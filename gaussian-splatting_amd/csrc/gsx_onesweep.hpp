// gsx_onesweep.hpp — one stable 8-bit radix pass over (key, u32 value) pairs
// with chained-scan ("onesweep") decoupled look-back, written for gfx950
// wave64: ranking uses v_cmp ballots + v_mbcnt, no LDS atomics.
//
// Replaces the reference's std::sort(indices, distances[a] > distances[b])
// (SplatMesh.h:97-108); the same pass also orders the binner's (tile, splat)
// pairs.  Stability is what turns "input in ascending index order" into the
// contract's tie rule (distance descending, then index ascending).
//
// Inter-workgroup protocol: tile ids are handed out by an atomic ticket, so a
// workgroup only ever waits on tiles whose owners are already running (forward
// progress without any dispatch-order assumption).  Each (tile, digit) status
// word is ONE 32-bit agent-scope atomic carrying flag+count together — the data
// is the flag, so no fence is needed around it.  Every spin is bounded; an
// expired spin raises GsxDevFlags::spin_expired instead of hanging the GPU.
#pragma once
#include "gsx_internal.h"

#define GSX_ST_AGG 0x40000000u
#define GSX_ST_PFX 0x80000000u
#define GSX_ST_VAL 0x3FFFFFFFu
#define GSX_SPIN_LIMIT (1u << 22)

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

// One radix pass over tile `tile` (already ticketed).  Must be called by all
// NT threads of the block.  KeyT is uint32_t or uint16_t; `pad_key` is the key
// used for slots past `count` (its digit must be the maximum so padding sorts
// behind real keys of the same digit).
template <typename KeyT, int NT, int IPT, bool IOTA_VALS>
__device__ __forceinline__ void gsx_onesweep_tile(
    const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
    uint32_t count, uint32_t shift, uint32_t tile,
    uint32_t* __restrict__ status /* n_tiles x 256 */, const uint32_t* __restrict__ gbase /* 256 */,
    uint32_t* __restrict__ spin_flag,
    KeyT* s_keys, uint32_t* s_vals, uint32_t (*s_whist)[GSX_RADIX], uint32_t* s_binbase,
    uint32_t* s_gofs, uint32_t* s_tmp) {
    constexpr int NW = NT / 64;
    constexpr int TILE = NT * IPT;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tile_base = tile * TILE;
    const uint32_t wave_base = tile_base + wave * (64 * IPT);

    // ---- load (wave-striped: item i of lane l is key wave_base + i*64 + l) ----
    KeyT key[IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const uint32_t g = wave_base + i * 64 + lane;
        key[i] = (g < count) ? keys_in[g] : static_cast<KeyT>(~static_cast<KeyT>(0));
    }

    // ---- rank inside the wave, items in order: ballot match + mbcnt ----
    volatile uint32_t* wh = s_whist[wave];
    uint16_t rank[IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const uint32_t d = (static_cast<uint32_t>(key[i]) >> shift) & 0xFFu;
        uint64_t peers = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __builtin_amdgcn_ballot_w64(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t lower = gsx_mbcnt(peers);
        const uint32_t cnt = static_cast<uint32_t>(__popcll(peers));
        const uint32_t prev = wh[d];
        if (lower == 0) wh[d] = prev + cnt;   // one writer per digit; LDS ops of a wave retire in order
        rank[i] = static_cast<uint16_t>(prev + lower);
    }
    __syncthreads();

    // ---- per digit: prefix over waves, tile count, publish, look back ----
    uint32_t my_count = 0, my_gbase = 0;
    if (tid < GSX_RADIX) {
        uint32_t acc = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const uint32_t t = s_whist[w][tid];
            s_whist[w][tid] = acc;
            acc += t;
        }
        my_count = acc;
        my_gbase = gbase[tid];
        const uint32_t word = (tile == 0 ? GSX_ST_PFX : GSX_ST_AGG) | my_count;
        __hip_atomic_store(&status[static_cast<uint64_t>(tile) * GSX_RADIX + tid], word, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint32_t bin_base = gsx_block_excl_scan<NT>(tid < GSX_RADIX ? my_count : 0u, s_tmp, nullptr);
    if (tid < GSX_RADIX) {
        uint32_t excl = 0;
        if (tile > 0) {
            int64_t t = static_cast<int64_t>(tile) - 1;
            uint32_t spins = 0;
            while (true) {
                const uint32_t s = __hip_atomic_load(&status[static_cast<uint64_t>(t) * GSX_RADIX + tid],
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (s & GSX_ST_PFX) { excl += s & GSX_ST_VAL; break; }
                if (s & GSX_ST_AGG) { excl += s & GSX_ST_VAL; --t; spins = 0; continue; }
                if (++spins > GSX_SPIN_LIMIT) { atomicOr(spin_flag, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            __hip_atomic_store(&status[static_cast<uint64_t>(tile) * GSX_RADIX + tid],
                               GSX_ST_PFX | ((excl + my_count) & GSX_ST_VAL), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        s_binbase[tid] = bin_base;
        s_gofs[tid] = my_gbase + excl - bin_base;  // global index = s_gofs[d] + local position
    }
    __syncthreads();

    // ---- scatter into LDS in digit order ----
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const uint32_t d = (static_cast<uint32_t>(key[i]) >> shift) & 0xFFu;
        const uint32_t pos = s_binbase[d] + s_whist[wave][d] + rank[i];
        const uint32_t g = wave_base + i * 64 + lane;
        s_keys[pos] = key[i];
        uint32_t v = 0;
        if (g < count) v = IOTA_VALS ? g : vals_in[g];
        s_vals[pos] = v;
    }
    __syncthreads();

    // ---- coalesced write-out: consecutive LDS slots of one digit are consecutive in HBM ----
    const uint32_t valid = (count - tile_base < static_cast<uint32_t>(TILE)) ? (count - tile_base) : TILE;
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const uint32_t p = i * NT + tid;
        if (p < valid) {
            const KeyT k = s_keys[p];
            const uint32_t d = (static_cast<uint32_t>(k) >> shift) & 0xFFu;
            const uint32_t dst = s_gofs[d] + p;
            keys_out[dst] = k;
            vals_out[dst] = s_vals[p];
        }
    }
}
