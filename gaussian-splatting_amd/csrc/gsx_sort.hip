// gsx_sort.hip — scene upload (AoS -> SoA), depth-key generation fused with the
// digit histograms, the sort plan, and the onesweep radix passes.
//
// Reference path replaced: SplatMesh::sortSplats (SplatMesh.h:84-114) and the
// per-frame index upload (SplatMesh.h:56-57).
#include "gsx_internal.h"
#include "gsx_onesweep.hpp"

// ---------------------------------------------------------------------------
// Upload: 80-B AoS `Splat` (Splat.h:19-29) -> SoA float4 arrays.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_upload_soa(const float4* __restrict__ aos, uint64_t n,
                                                    float4* __restrict__ pos4, float4* __restrict__ covA,
                                                    float4* __restrict__ covB, float4* __restrict__ color4,
                                                    GsxDevFlags* __restrict__ flags) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c0 = aos[i * 5 + 0], c1 = aos[i * 5 + 1], c2 = aos[i * 5 + 2];
    pos4[i] = aos[i * 5 + 3];
    color4[i] = aos[i * 5 + 4];
    // column-major: c0 = (s00,s10,s20,_), c1 = (s01,s11,s21,_), c2 = (s02,s12,s22,_)
    covA[i] = make_float4(c0.x, c1.x, c2.x, c1.y);
    covB[i] = make_float4(c2.y, c2.z, 0.0f, 0.0f);
    const bool sym = (__float_as_uint(c0.y) == __float_as_uint(c1.x)) && (__float_as_uint(c0.z) == __float_as_uint(c2.x)) &&
                     (__float_as_uint(c1.z) == __float_as_uint(c2.y));
    // +0/-0 and NaN payload differences do not matter to the projection; compare by value too
    const bool symv = (c0.y == c1.x || (c0.y != c0.y && c1.x != c1.x)) && (c0.z == c2.x || (c0.z != c0.z && c2.x != c2.x)) &&
                      (c1.z == c2.y || (c1.z != c1.z && c2.y != c2.y));
    if (!sym && !symv) atomicOr(&flags->asym_cov, 1u);
}

int gsx_launch_upload(gsx_ctx* c, const void* aos80_dev, uint64_t n) {
    if (n == 0) return GSX_OK;
    hipLaunchKernelGGL(k_upload_soa, dim3(gsx_div_up(n, 256)), dim3(256), 0, c->stream,
                       static_cast<const float4*>(aos80_dev), n, c->pos4, c->covA, c->covB, c->color4, c->flags);
    GSX_HIP(c, hipGetLastError());
    return GSX_OK;
}

// ---------------------------------------------------------------------------
// Depth key.  glm::distance(position, cameraPos) = sqrt((dx*dx + dy*dy) + dz*dz)
// with d = cameraPos - position (glm/detail/func_geometric.inl:8-23,47-55), each
// operation rounded once (gsx_mul/gsx_add/... in gsx_internal.h: contraction off).  Ascending
// u32 key = bit-complement of the (non-negative) float, top bit cleared, so a
// stable ascending radix sort yields distance descending, index ascending.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float gsx_distance(float4 p, float cx, float cy, float cz) {
    const float dx = gsx_sub(cx, p.x), dy = gsx_sub(cy, p.y), dz = gsx_sub(cz, p.z);
    const float d2 = gsx_add(gsx_add(gsx_mul(dx, dx), gsx_mul(dy, dy)), gsx_mul(dz, dz));
    return gsx_sqrt(d2);
}
__device__ __forceinline__ uint32_t gsx_key_from_distance(float d) { return (~__float_as_uint(d)) & 0x7FFFFFFFu; }

// LDS histogram add with a wave-uniform fast path (the high digits of distance
// keys are nearly constant, which would serialise 64-way on one LDS word).
__device__ __forceinline__ void gsx_hist_add(uint32_t* hist, uint32_t d, bool active) {
    const uint64_t act = __builtin_amdgcn_ballot_w64(active);
    if (act == 0) return;
    const uint32_t first = static_cast<uint32_t>(__builtin_ctzll(act));
    const uint32_t d0 = __shfl(d, first);
    const uint64_t same = __builtin_amdgcn_ballot_w64(active && d == d0);
    if (same == act) {
        if (gsx_lane() == first) atomicAdd(&hist[d0], static_cast<uint32_t>(__popcll(act)));
    } else if (active) {
        atomicAdd(&hist[d], 1u);
    }
}

template <bool IOTA>
__global__ __launch_bounds__(256) void k_keygen_hist(const float4* __restrict__ pos4, const uint32_t* __restrict__ sel,
                                                     const uint32_t* __restrict__ m_ptr, float cx, float cy, float cz,
                                                     uint32_t* __restrict__ keys, uint32_t* __restrict__ ghist) {
    __shared__ uint32_t sh[GSX_PASSES][GSX_RADIX];
    for (uint32_t t = threadIdx.x; t < GSX_PASSES * GSX_RADIX; t += blockDim.x) (&sh[0][0])[t] = 0;
    __syncthreads();
    const uint32_t m = *m_ptr;
    constexpr int UNROLL = 4;
    const uint32_t stride = gridDim.x * blockDim.x;
    // Lanes whose k0 runs past m leave the loop; ballots below only see live lanes.
    for (uint32_t k0 = blockIdx.x * blockDim.x + threadIdx.x; k0 < m; k0 += stride * UNROLL) {
        float4 p[UNROLL];
        bool ok[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const uint64_t k = static_cast<uint64_t>(k0) + static_cast<uint64_t>(j) * stride;
            ok[j] = k < m;
            if (ok[j]) p[j] = pos4[IOTA ? static_cast<uint32_t>(k) : sel[k]];
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const uint64_t k = static_cast<uint64_t>(k0) + static_cast<uint64_t>(j) * stride;
            uint32_t key = 0;
            if (ok[j]) {
                key = gsx_key_from_distance(gsx_distance(p[j], cx, cy, cz));
                keys[k] = key;
            }
#pragma unroll
            for (int q = 0; q < GSX_PASSES; ++q) gsx_hist_add(sh[q], (key >> (8 * q)) & 0xFFu, ok[j]);
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < GSX_PASSES * GSX_RADIX; t += blockDim.x) {
        const uint32_t v = (&sh[0][0])[t];
        if (v) atomicAdd(&ghist[t], v);
    }
}

// Distances only (parity aid for gsx_get_keys).
template <bool IOTA>
__global__ __launch_bounds__(256) void k_distances(const float4* __restrict__ pos4, const uint32_t* __restrict__ sel,
                                                   const uint32_t* __restrict__ m_ptr, float cx, float cy, float cz,
                                                   float* __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *m_ptr) return;
    out[k] = gsx_distance(pos4[IOTA ? k : sel[k]], cx, cy, cz);
}

// One block: exclusive scans of the four global histograms + which passes to run.
__global__ __launch_bounds__(256) void k_sort_plan(const uint32_t* __restrict__ ghist, const uint32_t* __restrict__ m_ptr,
                                                   GsxSortPlan* __restrict__ plan) {
    __shared__ uint32_t s_tmp[4];
    __shared__ uint32_t s_skip[GSX_PASSES];
    const uint32_t tid = threadIdx.x, m = *m_ptr;
    if (tid < GSX_PASSES) s_skip[tid] = 0;
    __syncthreads();
    for (int p = 0; p < GSX_PASSES; ++p) {
        const uint32_t v = ghist[p * GSX_RADIX + tid];
        if (p > 0 && m > 0 && v == m) s_skip[p] = 1;  // pass 0 always runs: it materialises the index payload
        plan->gbase[p][tid] = gsx_block_excl_scan<256>(v, s_tmp, nullptr);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t cur = 1, run = 1;  // pass 0: keys[0] -> keys[1]
        plan->skip[0] = 0;
        plan->src[0] = 0;
        for (int p = 1; p < GSX_PASSES; ++p) {
            plan->skip[p] = s_skip[p];
            plan->src[p] = cur;
            if (!s_skip[p]) { cur ^= 1u; ++run; }
        }
        plan->final_buf = cur;
        plan->m = m;
        plan->passes_run = run;
    }
}

template <bool IOTA>
__global__ __launch_bounds__(GSX_SORT_NT) void k_sort_pass(uint32_t* __restrict__ k0, uint32_t* __restrict__ k1,
                                                           uint32_t* __restrict__ v0, uint32_t* __restrict__ v1,
                                                           const uint32_t* __restrict__ sel, int pass,
                                                           const GsxSortPlan* __restrict__ plan,
                                                           GsxSortZero* __restrict__ zero, uint32_t* __restrict__ status,
                                                           uint32_t* __restrict__ spin_flag) {
    __shared__ uint32_t s_keys[GSX_SORT_TILE];
    __shared__ uint32_t s_vals[GSX_SORT_TILE];
    __shared__ uint32_t s_whist[GSX_SORT_NT / 64][GSX_RADIX];
    __shared__ uint32_t s_binbase[GSX_RADIX];
    __shared__ uint32_t s_gofs[GSX_RADIX];
    __shared__ uint32_t s_tmp[GSX_SORT_NT / 64];
    __shared__ uint32_t s_tile;
    if (plan->skip[pass]) return;
    const uint32_t m = plan->m;
    if (threadIdx.x == 0) s_tile = atomicAdd(&zero->tile_ctr[pass], 1u);
    for (uint32_t t = threadIdx.x; t < (GSX_SORT_NT / 64) * GSX_RADIX; t += GSX_SORT_NT) (&s_whist[0][0])[t] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    if (static_cast<uint64_t>(tile) * GSX_SORT_TILE >= m) return;
    const uint32_t src = plan->src[pass];
    const uint32_t* kin = src ? k1 : k0;
    uint32_t* kout = src ? k0 : k1;
    const uint32_t* vin = (pass == 0) ? sel : (src ? v1 : v0);
    uint32_t* vout = src ? v0 : v1;
    gsx_onesweep_tile<uint32_t, GSX_SORT_NT, GSX_SORT_IPT, IOTA>(kin, vin, kout, vout, m, 8u * pass, tile, status,
                                                                 plan->gbase[pass], spin_flag, s_keys, s_vals, s_whist,
                                                                 s_binbase, s_gofs, s_tmp);
}

static inline uint64_t sort_status_words(uint64_t cap) { return static_cast<uint64_t>(gsx_div_up(cap, GSX_SORT_TILE)) * GSX_RADIX; }

int gsx_launch_sort(gsx_ctx* c) {
    const uint32_t cap = c->m_cap;
    const uint32_t n_tiles = gsx_div_up(cap, GSX_SORT_TILE);
    const uint64_t zero_bytes = sizeof(GsxSortZero) + static_cast<uint64_t>(GSX_PASSES) * sort_status_words(cap) * 4;
    if (zero_bytes > c->sort_zero_bytes) { c->err = "sort workspace too small"; return GSX_E_INTERNAL; }
    if (c->profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_KEY0], c->stream));
    GSX_HIP(c, hipMemsetAsync(c->sort_zero, 0, zero_bytes, c->stream));
    if (cap > 0) {
        const uint32_t blocks = min(gsx_div_up(cap, 256 * 4), 2048u);
        if (c->sel_iota)
            hipLaunchKernelGGL(k_keygen_hist<true>, dim3(blocks), dim3(256), 0, c->stream, c->pos4, nullptr, c->m_dev,
                               c->cam[0], c->cam[1], c->cam[2], c->keys[0], &c->sort_zero->ghist[0][0]);
        else
            hipLaunchKernelGGL(k_keygen_hist<false>, dim3(blocks), dim3(256), 0, c->stream, c->pos4, c->sel_idx, c->m_dev,
                               c->cam[0], c->cam[1], c->cam[2], c->keys[0], &c->sort_zero->ghist[0][0]);
    }
    hipLaunchKernelGGL(k_sort_plan, dim3(1), dim3(256), 0, c->stream, &c->sort_zero->ghist[0][0], c->m_dev, c->plan);
    if (c->profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_KEY1], c->stream));
    uint32_t* status0 = reinterpret_cast<uint32_t*>(c->sort_zero + 1);
    for (int p = 0; p < GSX_PASSES && n_tiles > 0; ++p) {
        uint32_t* status = status0 + static_cast<uint64_t>(p) * sort_status_words(cap);
        if (p == 0 && c->sel_iota)
            hipLaunchKernelGGL(k_sort_pass<true>, dim3(n_tiles), dim3(GSX_SORT_NT), 0, c->stream, c->keys[0], c->keys[1],
                               c->vals[0], c->vals[1], nullptr, p, c->plan, c->sort_zero, status, &c->flags->spin_expired);
        else
            hipLaunchKernelGGL(k_sort_pass<false>, dim3(n_tiles), dim3(GSX_SORT_NT), 0, c->stream, c->keys[0], c->keys[1],
                               c->vals[0], c->vals[1], c->sel_idx, p, c->plan, c->sort_zero, status, &c->flags->spin_expired);
    }
    if (c->profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_SORT1], c->stream));
    GSX_HIP(c, hipGetLastError());
    c->sorted_valid = true;
    return GSX_OK;
}

int gsx_launch_keys_only(gsx_ctx* c, float* dist_dev) {
    if (c->m_cap == 0) return GSX_OK;
    if (c->sel_iota)
        hipLaunchKernelGGL(k_distances<true>, dim3(gsx_div_up(c->m_cap, 256)), dim3(256), 0, c->stream, c->pos4, nullptr,
                           c->m_dev, c->cam[0], c->cam[1], c->cam[2], dist_dev);
    else
        hipLaunchKernelGGL(k_distances<false>, dim3(gsx_div_up(c->m_cap, 256)), dim3(256), 0, c->stream, c->pos4,
                           c->sel_idx, c->m_dev, c->cam[0], c->cam[1], c->cam[2], dist_dev);
    GSX_HIP(c, hipGetLastError());
    return GSX_OK;
}

uint64_t gsx_sort_workspace_bytes(uint64_t cap) {
    return sizeof(GsxSortZero) + static_cast<uint64_t>(GSX_PASSES) * sort_status_words(cap) * 4;
}
