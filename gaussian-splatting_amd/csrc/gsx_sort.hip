// gsx_sort.hip — scene upload (AoS -> SoA), depth-key generation fused with the
// digit histograms, the sort plan, and the stable LSD radix passes (gsx_radix.hpp).
//
// Reference path replaced: SplatMesh::sortSplats (SplatMesh.h:84-114) and the
// per-frame index upload (SplatMesh.h:56-57).
#include "gsx_internal.h"
#include "gsx_radix.hpp"
#include "gsx_sort2.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

// ---------------------------------------------------------------------------
// Upload: 80-B AoS `Splat` (Splat.h:19-29) -> SoA positions (key pass) + 64-B AoS records (frame loop).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_upload_soa(const float4* __restrict__ aos, uint64_t n,
                                                    float4* __restrict__ pos4, float* __restrict__ pos3, GsxGeo* __restrict__ geo,
                                                    GsxDevFlags* __restrict__ flags) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c0 = aos[i * 5 + 0], c1 = aos[i * 5 + 1], c2 = aos[i * 5 + 2];
    const float4 pos = aos[i * 5 + 3];
    // SoA record of the key pass: position + a conservative radius for the cull-by-band selection: the square root of the
    // covariance's largest eigenvalue (closed form for a symmetric 3x3, a little margin; the trace where that breaks down).
    // The 64-B record below keeps transform[3] as it is.
    float rho;
    {
        const float s00 = c0.x, s11 = c1.y, s22 = c2.z, s01 = c1.x, s02 = c2.x, s12 = c2.y;
        const float tr = s00 + s11 + s22;
        float lmax = tr;
        const float p1 = s01 * s01 + s02 * s02 + s12 * s12, q = tr * (1.0f / 3.0f);
        const float d0 = s00 - q, d1 = s11 - q, d2 = s22 - q;
        const float p2 = d0 * d0 + d1 * d1 + d2 * d2 + 2.0f * p1;
        if (p2 > 0.0f) {
            const float pp = sqrtf(p2 * (1.0f / 6.0f)), ip = 1.0f / pp;
            const float b00 = d0 * ip, b11 = d1 * ip, b22 = d2 * ip, b01 = s01 * ip, b02 = s02 * ip, b12 = s12 * ip;
            float r = 0.5f * (b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02));
            r = fminf(fmaxf(r, -1.0f), 1.0f);
            const float lam = q + 2.0f * pp * cosf(acosf(r) * (1.0f / 3.0f));
            if (lam >= 0.0f && lam <= tr) lmax = fminf(tr, lam * 1.002f + 1.0e-12f);
        } else if (p2 == 0.0f) {
            lmax = fmaxf(q, 0.0f);
        }
        rho = (lmax >= 0.0f && lmax < 3.0e38f) ? sqrtf(lmax) * 1.0001f : __builtin_huge_valf();
    }
    pos4[i] = make_float4(pos.x, pos.y, pos.z, rho);
    pos3[i * 3 + 0] = pos.x; pos3[i * 3 + 1] = pos.y; pos3[i * 3 + 2] = pos.z;     // the two-pass sort's key kernel reads 12 B per splat
    // column-major: c0 = (s00,s10,s20,_), c1 = (s01,s11,s21,_), c2 = (s02,s12,s22,_)
    float4* g = reinterpret_cast<float4*>(geo + i);
    g[0] = pos;
    g[1] = make_float4(c0.x, c1.x, c2.x, c1.y);
    g[2] = make_float4(c2.y, c2.z, 0.0f, 0.0f);
    g[3] = aos[i * 5 + 4];
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
                       static_cast<const float4*>(aos80_dev), n, c->pos4, c->pos3, c->geo, c->flags);
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
// (gsx_distance / gsx_key_from_distance: gsx_internal.h, shared with the two-pass sort of gsx_sort2.hpp)

// ---------------------------------------------------------------------------
// K1: depth keys of the selection + the digit counts of radix pass 0 (fused H0).
// Block b owns the same contiguous tile range the radix passes give it.  Keys are
// stored relative to key_lo (the key of the host-computed upper distance bound),
// which makes the number of significant key bits — and so the number of radix
// passes — a host-side constant for the frame.
// ---------------------------------------------------------------------------
template <bool IOTA>
__global__ __launch_bounds__(GSX_SORT_NT) void k_keygen_hist0(const float4* __restrict__ pos4, const uint32_t* __restrict__ sel,
                                                              const uint32_t* __restrict__ m_ptr, float cx, float cy, float cz,
                                                              uint32_t key_lo, uint32_t key_span, uint32_t shift, uint32_t bits,
                                                              uint32_t wide_mask, uint32_t* __restrict__ keys, uint32_t* __restrict__ cnt, uint32_t max_cols,
                                                              GsxDevFlags* __restrict__ flags, uint32_t* __restrict__ m_copy) {
    __shared__ uint32_t s_hist[GSX_RADIX_MAX_DIGITS * GSX_HREP];
    const uint32_t nd = 1u << bits, mask = nd - 1u;
    const uint32_t m = *m_ptr;
    // the frame drawn from this permutation reads the selection's size from a word of its own (the next selection may
    // overwrite *m_ptr as soon as this sort is done)
    if (m_copy && blockIdx.x == 0 && threadIdx.x == 0) *m_copy = m;
    const uint32_t n_cols = gsx_n_cols(m, GSX_SORT_TILE, max_cols);
    uint32_t top = 0;                                  // OR of the key bits above the ungated passes
    for (uint32_t col = gsx_logical_block(); col < n_cols; col += gridDim.x) {
    gsx_hist_zero(s_hist, GSX_SORT_NT);
    uint32_t begin, end;
    gsx_block_range(m, n_cols, col, begin, end);
    constexpr int UNROLL = 4;
    for (uint32_t base = begin + threadIdx.x; base < end; base += GSX_SORT_NT * UNROLL) {
        float4 p[UNROLL];
        bool ok[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const uint32_t k = base + j * GSX_SORT_NT;
            ok[j] = k < end;
            if (ok[j]) p[j] = pos4[IOTA ? k : sel[k]];
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            if (!ok[j]) continue;
            const uint32_t k = base + j * GSX_SORT_NT;
            const uint32_t key = gsx_key_from_distance(gsx_distance(p[j], cx, cy, cz));
            uint32_t rel = key - key_lo;
            if (key < key_lo || rel > key_span) {   // outside the bound: NaN/inf position
                atomicOr(&flags->key_bound, 1u);
                rel = (key < key_lo) ? 0u : key_span;
            }
            keys[k] = rel;
            top |= rel & wide_mask;
            gsx_hist_add(s_hist, (rel >> shift) & mask);
        }
    }
    gsx_hist_store(s_hist, nd, GSX_SORT_NT, cnt, n_cols, col);
    __syncthreads();
    }
    // A key with bits above the ungated passes (a splat within ~2^-16 of the far bound's distance from the camera)
    // switches the gated top-digit pass on: its kernels take their key count from wide_m.
    if (top != 0u) atomicMax(&flags->wide_m, m);
}

// Gated copy of the permutation (the top-digit pass leaves it in a scratch array): count 0 = nothing to do.
__global__ __launch_bounds__(512) void k_copy_gated(const uint32_t* __restrict__ count_ptr, const uint32_t* __restrict__ src,
                                                    uint32_t* __restrict__ dst) {
    const uint32_t n = *count_ptr;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
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

// Axis-aligned bounds of the uploaded positions (ordered-int atomics), used by the
// host to bound the distance range of a frame.
__device__ __forceinline__ int gsx_f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__global__ __launch_bounds__(256) void k_aabb(const float4* __restrict__ pos4, uint64_t n, int* __restrict__ out6) {
    int mn[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const float4 p = pos4[i];
        const float v[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (v[k] == v[k]) { const int o = gsx_f2ord(v[k]); mn[k] = min(mn[k], o); mx[k] = max(mx[k], o); }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn[k] = min(mn[k], __shfl_xor(mn[k], off)); mx[k] = max(mx[k], __shfl_xor(mx[k], off)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&out6[k], mn[k]); atomicMax(&out6[3 + k], mx[k]); }
    }
}

int gsx_launch_aabb(gsx_ctx* c) {
    const int init[6] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, (int)0x80000000, (int)0x80000000, (int)0x80000000};
    GSX_HIP(c, hipMemcpyAsync(c->aabb_dev, init, sizeof init, hipMemcpyHostToDevice, c->stream));
    if (c->n) hipLaunchKernelGGL(k_aabb, dim3(min(gsx_div_up(c->n, 256 * 8), 2048u)), dim3(256), 0, c->stream, c->pos4, c->n,
                                 reinterpret_cast<int*>(c->aabb_dev));
    int out[6];
    GSX_HIP(c, hipMemcpyAsync(out, c->aabb_dev, sizeof out, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    c->aabb_valid = c->n > 0;
    for (int k = 0; k < 3; ++k) {
        if (out[k] > out[3 + k]) { c->aabb_valid = false; break; }
        const int a = out[k] >= 0 ? out[k] : out[k] ^ 0x7FFFFFFF, b = out[3 + k] >= 0 ? out[3 + k] : out[3 + k] ^ 0x7FFFFFFF;
        memcpy(&c->aabb_min[k], &a, 4);
        memcpy(&c->aabb_max[k], &b, 4);
        if (!std::isfinite(c->aabb_min[k]) || !std::isfinite(c->aabb_max[k])) c->aabb_valid = false;
    }
    return GSX_OK;
}

static uint32_t host_key(float d) {
    uint32_t b;
    memcpy(&b, &d, 4);
    return (~b) & 0x7FFFFFFFu;
}

// Conservative key range of the frame from the scene bounds: every distance lies in
// [dist(cam, box), dist(cam, farthest corner)], widened by 1e-5 relative.
static void key_bounds(const gsx_ctx* c, uint32_t* key_lo, uint32_t* key_span) {
    *key_lo = 0;
    *key_span = 0x7FFFFFFFu;
    if (!c->aabb_valid) return;
    double near2 = 0.0, far2 = 0.0;
    for (int k = 0; k < 3; ++k) {
        const double cam = c->cam[k], lo = c->aabb_min[k], hi = c->aabb_max[k];
        if (!std::isfinite(cam)) return;
        const double dn = cam < lo ? lo - cam : (cam > hi ? cam - hi : 0.0);
        const double df = std::fmax(std::fabs(cam - lo), std::fabs(cam - hi));
        near2 += dn * dn;
        far2 += df * df;
    }
    const double dmin = std::sqrt(near2) * (1.0 - 1e-5), dmax = std::sqrt(far2) * (1.0 + 1e-5) + 1e-30;
    if (!(dmax < 3.0e38)) return;
    const uint32_t lo = host_key(static_cast<float>(dmax)), hi = host_key(static_cast<float>(dmin));
    if (hi < lo) return;
    *key_lo = lo;
    *key_span = hi - lo;
}

template <typename KeyT, int NT, int IPT>
static void launch_pass(gsx_ctx* c, hipStream_t st, uint32_t g, const KeyT* kin, const uint32_t* vin, KeyT* kout, uint32_t* vout,
                        const uint32_t* count_ptr, uint32_t shift, uint32_t bits, uint32_t* ws, uint32_t cols, bool need_hist,
                        bool iota, hipEvent_t ev0, hipEvent_t ev1 /* null = no events */) {
    constexpr uint32_t TILE = NT * IPT;
    uint32_t* cnt = ws;
    uint32_t* totals = ws + static_cast<uint64_t>(GSX_RADIX_MAX_DIGITS) * cols;
    if (need_hist)
        hipLaunchKernelGGL((k_radix_hist<KeyT, NT>), dim3(g), dim3(NT), 0, st, kin, count_ptr, shift, bits, cnt, TILE, cols);
    hipLaunchKernelGGL((k_radix_scan<0>), dim3(1u << bits), dim3(256), 0, st, cnt, count_ptr, TILE, cols, totals);
    if (ev0) (void)hipEventRecord(ev0, st);
    if (iota)
        hipLaunchKernelGGL((k_radix_scatter<KeyT, NT, IPT, true, GSX_RADIX_MAX_DIGITS, GSX_SORT_WPE>), dim3(g), dim3(NT), 0, st, kin, vin, kout, vout, count_ptr, shift,
                           bits, cnt, totals, cols);
    else
        hipLaunchKernelGGL((k_radix_scatter<KeyT, NT, IPT, false, GSX_RADIX_MAX_DIGITS, GSX_SORT_WPE>), dim3(g), dim3(NT), 0, st, kin, vin, kout, vout, count_ptr, shift,
                           bits, cnt, totals, cols);
    if (ev1) (void)hipEventRecord(ev1, st);
}

// The buffers one depth sort works in: key / value ping-pong arrays, the radix count matrix, and where the
// permutation goes.  gsx_sort uses the context's own arrays; gsx_sort_noalloc carves them out of `scratch` below.
struct GsxSortBuffers {
    uint32_t* keys[2];
    uint32_t* vals[2];
    uint32_t* ws;
    uint64_t ws_words;
    uint32_t* result;
    bool with_events;      // profiling events / sampled scatter launches (the context's own sort only)
    bool two_pass_ok = false;   // the context's pair arrays may be used (the context's own sort only)
    uint32_t* m_copy = nullptr;   // device word that receives the selection's size (the context's own sort only)
};

// GSX_DEBUG_SORT2=1: the bucket statistics of every two-pass sort, the plan of every sampled sort and every sample, on stderr
static const bool g_dbg_sort2 = getenv("GSX_DEBUG_SORT2") != nullptr;

static int launch_sort_in(gsx_ctx* c, const GsxSortBuffers& sb, hipStream_t st) {
    const uint32_t cap = c->m_cap;
    const uint32_t g = gsx_radix_blocks(cap, GSX_SORT_TILE, c->sort_slots);
    const uint32_t cols = gsx_radix_columns(c->sort_slots);
    if (gsx_radix_workspace_words(cols) > sb.ws_words) { c->err = "sort workspace too small"; return GSX_E_INTERNAL; }
    const bool profiling = c->profiling && sb.with_events;
    uint32_t key_lo, key_span;
    key_bounds(c, &key_lo, &key_span);
    uint32_t total_bits = 0;
    while (total_bits < 32 && (static_cast<uint64_t>(key_span) >> total_bits) != 0) ++total_bits;
    // A host bound wider than three full digits (the camera stands inside the scene box, so the nearest distance the
    // host can promise is 0) does not buy a fourth full pass: the low 27 bits are sorted as usual and the top bits
    // get a GATED pass whose kernels take their key count from flags->wide_m — 0 unless the key kernel met a key
    // that uses those bits (a splat closer than ~2^-16 of the far bound).  Decided on the device, exact either way.
    constexpr uint32_t LOW_BITS = 3 * GSX_RADIX_MAX_BITS;
    const uint32_t top_bits = total_bits > LOW_BITS ? total_bits - LOW_BITS : 0u;
    const GsxRadixPlan plan = gsx_radix_plan(top_bits ? LOW_BITS : total_bits);
    const uint32_t wide_mask = top_bits ? ~((1u << LOW_BITS) - 1u) : 0u;
    // The two-pass form (gsx_sort2.hpp: one wide partition + one kernel that finishes every bucket in LDS) when its plan
    // allows; the LSD passes below otherwise (small sorts, narrow key ranges, gsx_sort_noalloc) and for a while after a
    // sort whose buckets were too dense for LDS (history, checked without waiting).
    Gsx2Plan p2;
    if (sb.two_pass_ok && c->s2_pairs && c->opt_sort_algo != 1) {
        if (c->s2_stats_rec && hipEventQuery(c->ev_s2_stats) == hipSuccess) {
            c->s2_stats_rec = false;
            if (g_dbg_sort2) fprintf(stderr, "[gsx] two-pass stats: big_buckets %u big_keys %u max_bucket %u keys %u (cap %u)\n", c->s2_stats_host->big_buckets, c->s2_stats_host->big_keys, c->s2_stats_host->max_bucket, c->s2_stats_host->keys, cap);
            // the size of that selection sizes the next sorts' buckets (a cut's size never reaches the host otherwise); buckets that
            // overflowed although the hint was below the capacity mean the selection has grown: forget the hint
            const uint32_t seen = c->s2_stats_host->keys;
            const bool hinted = c->s2_hint_cap == cap && c->s2_m_hint != 0 && c->s2_m_hint < cap;
            // buckets over GSX2_CAP: the following sorts launch the third k2_local instance (up to GSX2_CAP2 in LDS); it is
            // dropped again after ~1000 sorts that did not need it
            if (c->s2_stats_host->max_bucket > GSX2_CAP) c->s2_mid_left = 256u;
            else if (c->s2_mid_left) --c->s2_mid_left;
            uint32_t big_keys = c->s2_stats_host->big_keys;      // keys of that sort no instance took in LDS ...
            if (!c->s2_stats_mid && c->s2_stats_host->max_bucket <= GSX2_CAP2) big_keys = 0u;   // ... all of which the third instance takes from now on
            if (big_keys != 0 && hinted) {
                c->s2_m_hint = 0;
                if (c->s2_hint_scale < 64u) c->s2_hint_scale *= 2u;         // (or the scene is denser than its size says: size the buckets for more)
            } else {
                if (static_cast<uint64_t>(big_keys) * 64ull > cap) {
                    uint32_t& len = c->s2_backoff_len[c->s2_stats_r0];
                    len = len ? (len < 512u ? len * 2u : 1024u) : 16u;
                    c->s2_backoff[c->s2_stats_r0] = len;
                } else if (big_keys == 0) {
                    c->s2_backoff_len[c->s2_stats_r0] = 0;
                }
                const uint64_t want = (static_cast<uint64_t>(seen) + seen / 2u) * c->s2_hint_scale;   // (half as much again: cuts along an orbit vary by ~20 %)
                c->s2_m_hint = want < cap ? static_cast<uint32_t>(want) : cap;
                c->s2_hint_cap = cap;
            }
        }
        p2 = gsx2_plan(key_span, cap, c->n, 0, c->opt_sort_algo == 2, c->s2_hint_cap == cap ? c->s2_m_hint : 0u);
        const uint32_t bo = p2.map.rc == p2.map.r0 + 2u ? 23u : p2.map.r0;      // (the wide 13-bit plan keeps a memory of its own)
        if (p2.ok && c->opt_sort_algo == 0 && c->s2_backoff[bo]) { --c->s2_backoff[bo]; p2.ok = false; }
        if (g_dbg_sort2) fprintf(stderr, "[gsx] plan: two-pass %d r0 %u nd %u key_span %#x back-off left %u mid %u\n", (int)p2.ok, p2.map.r0, p2.nd, key_span, c->s2_backoff[bo], c->s2_mid_left);
    }
    const bool gated = top_bits != 0 && !p2.ok;      // (the two-pass form orders all 31 key bits itself: its buckets widen towards the camera)
    if (gated) GSX_HIP(c, hipMemsetAsync(&c->flags->wide_m, 0, sizeof(uint32_t), st));
    const int n_stage = p2.ok ? 2 : plan.n_passes;    // launches a level-2 sample can bracket: radix passes, or partition + bucket sort
    // Profiling level 2: every opt_sample_every-th sort is a SAMPLED one, in turn (option sample_whole_every, default alternately) (a) every scatter launch of the radix
    // passes -- or the partition and the bucket-sort launches of the two-pass form -- each bracketed by its own pair of events
    // from the ring, and (b) keys + sort as a whole (slot GSX_SAMPLE_WHOLE), from before the key kernel to behind the last
    // pass.  A sampled sort runs on the main stream (gsx_launch_sort), i.e. NOT beside the previous frame's rasterizer, so
    // that a bracket measures the kernel and not its share of a GPU it has to split.
    hipEvent_t samp_a[GSX_SAMPLE_WHOLE + 1] = {}, samp_b[GSX_SAMPLE_WHOLE + 1] = {};
    auto take_slot = [&](int what) {
        const uint32_t slot = c->samp_pos++ % GSX_SAMPLE_RING;
        gsx_sample_drain(c, slot);                  // (a slot comes round again 32 samples later: long finished)
        c->samp_pass[slot] = what;
        samp_a[what] = c->samp_ev[slot][0];
        samp_b[what] = c->samp_ev[slot][1];
    };
    int sampled_pass = -1;
    if (g_dbg_sort2 && c->prof_level == 2 && c->prof_frame % c->opt_sample_every == 0u)
        fprintf(stderr, "[gsx] %.1f sampled sort %llu: two-pass %d nd %u ncols %u key_lo %#x key_span %#x\n", gsx_dbg_now(), (unsigned long long)c->prof_frame, (int)p2.ok, p2.nd, p2.ncols, key_lo, key_span);
    if (sb.with_events) {
        if (c->prof_level == 2 && n_stage > 0 && c->prof_frame % c->opt_sample_every == 0u) {
            if ((c->prof_frame / c->opt_sample_every) % c->opt_sample_whole_every == c->opt_sample_whole_every - 1u) {
                sampled_pass = GSX_SAMPLE_WHOLE;
                take_slot(GSX_SAMPLE_WHOLE);
            } else {
                sampled_pass = 0;
                for (int p = 0; p < n_stage && p < GSX_SAMPLE_WHOLE; ++p) take_slot(p);
            }
        }
        c->sampled_pass = sampled_pass;
        ++c->prof_frame;
    }
    if (sampled_pass == GSX_SAMPLE_WHOLE) GSX_HIP(c, hipEventRecord(samp_a[GSX_SAMPLE_WHOLE], st));
    if (profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_KEY0], st));
    uint32_t cur = 0;
    uint32_t* result = sb.result;
    bool want_stats = false;
    const bool mid = p2.ok && c->s2_mid_left > 0u;
    const uint32_t cap_lds = mid ? GSX2_CAP2 : GSX2_CAP;
    if (p2.ok) {
        const Gsx2Workspace w = gsx2_ws_carve(c->s2_ws, c->sort_cap);
        if (c->sel_iota)
            hipLaunchKernelGGL(k2_keygen_hist<true>, dim3(p2.ncols), dim3(GSX2_NT), 0, st, reinterpret_cast<const Gsx2Pos*>(c->pos3), nullptr, c->m_dev, c->cam[0], c->cam[1],
                               c->cam[2], key_lo, key_span, p2.map, p2.nd, sb.keys[0], w.cnt, c->flags, sb.m_copy, w.stats);
        else
            hipLaunchKernelGGL(k2_keygen_hist<false>, dim3(p2.ncols), dim3(GSX2_NT), 0, st, reinterpret_cast<const Gsx2Pos*>(c->pos3), c->sel_idx, c->m_dev, c->cam[0], c->cam[1],
                               c->cam[2], key_lo, key_span, p2.map, p2.nd, sb.keys[0], w.cnt, c->flags, sb.m_copy, w.stats);
        if (profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_KEY1], st));
        hipLaunchKernelGGL(k2_scan, dim3(p2.nd / 16), dim3(512), 0, st, w.cnt, w.pre, w.totals, p2.nd, p2.ncols);
        hipEvent_t e0 = profiling ? c->ev[EV_X0] : samp_a[0];
        hipEvent_t e1 = profiling ? c->ev[EV_X0 + 1] : samp_b[0];
        if (e0) GSX_HIP(c, hipEventRecord(e0, st));
        if (c->sel_iota)
            hipLaunchKernelGGL(k2_partition<true>, dim3(p2.ncols), dim3(GSX2_PNT), 0, st, sb.keys[0], nullptr, c->m_dev, p2.map, p2.nd,
                               w.cnt, w.pre, w.totals, w.base, c->s2_pairs, w.stats, c->flags, cap_lds);
        else
            hipLaunchKernelGGL(k2_partition<false>, dim3(p2.ncols), dim3(GSX2_PNT), 0, st, sb.keys[0], c->sel_idx, c->m_dev, p2.map,
                               p2.nd, w.cnt, w.pre, w.totals, w.base, c->s2_pairs, w.stats, c->flags, cap_lds);
        if (e1) GSX_HIP(c, hipEventRecord(e1, st));
        e0 = profiling ? c->ev[EV_X0 + 2] : samp_a[1];
        e1 = profiling ? c->ev[EV_X0 + 3] : samp_b[1];
        if (e0) GSX_HIP(c, hipEventRecord(e0, st));
        // two instances by bucket size; the larger one also sorts what does not fit LDS (through the scratch twin of the pairs)
        hipLaunchKernelGGL((k2_local<GSX2_IPT>), dim3(p2.nd), dim3(GSX2_NT), 0, st, c->s2_pairs, c->s2_scratch, w.base, p2.nd, p2.map, p2.idx_bits, result, cap_lds);
        if (mid)   // (only while recent sorts had buckets of 9217 .. 16384 pairs: a launch that finds nothing to do costs 5 us)
            hipLaunchKernelGGL((k2_local<GSX2_IPT2>), dim3(p2.nd), dim3(GSX2_NT), 0, st, c->s2_pairs, c->s2_scratch, w.base, p2.nd, p2.map, p2.idx_bits, result, cap_lds);
        hipLaunchKernelGGL((k2_local<8>), dim3(p2.nd), dim3(GSX2_NT), 0, st, c->s2_pairs, c->s2_scratch, w.base, p2.nd, p2.map, p2.idx_bits, result, cap_lds);
        if (e1) GSX_HIP(c, hipEventRecord(e1, st));
        want_stats = true;
    } else {
    if (c->sel_iota)
        hipLaunchKernelGGL(k_keygen_hist0<true>, dim3(g), dim3(GSX_SORT_NT), 0, st, c->pos4, nullptr, c->m_dev, c->cam[0],
                           c->cam[1], c->cam[2], key_lo, key_span, plan.shift[0], plan.bits[0], wide_mask, sb.keys[0], sb.ws, cols,
                           c->flags, sb.m_copy);
    else
        hipLaunchKernelGGL(k_keygen_hist0<false>, dim3(g), dim3(GSX_SORT_NT), 0, st, c->pos4, c->sel_idx, c->m_dev, c->cam[0],
                           c->cam[1], c->cam[2], key_lo, key_span, plan.shift[0], plan.bits[0], wide_mask, sb.keys[0], sb.ws, cols,
                           c->flags, sb.m_copy);
    if (profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_KEY1], st));
    for (int p = 0; p < plan.n_passes; ++p) {
        const bool first = p == 0, last = p + 1 == plan.n_passes;
        hipEvent_t e0 = profiling ? c->ev[EV_X0 + 2 * p] : samp_a[p];
        hipEvent_t e1 = profiling ? c->ev[EV_X0 + 2 * p + 1] : samp_b[p];
        // (the last pass's keys are dead unless the gated pass sorts on their top bits: 4 B per key less to write)
        launch_pass<uint32_t, GSX_SORT_NT, GSX_SORT_IPT>(c, st, g, sb.keys[cur], first ? c->sel_idx : sb.vals[cur],
                                                          (last && !gated) ? nullptr : sb.keys[cur ^ 1],
                                                          last ? result : sb.vals[cur ^ 1], c->m_dev, plan.shift[p], plan.bits[p], sb.ws,
                                                          cols, /*need_hist=*/!first, /*iota=*/first && c->sel_iota, e0, e1);
        cur ^= 1;
    }
    }
    if (gated) {   // gated top digit: result -> scratch values, then back (all four kernels see count 0 in the common case)
        static_assert(EV_X0 + 2 * 3 + 1 < EV_X_END, "event slots of the gated pass");
        launch_pass<uint32_t, GSX_SORT_NT, GSX_SORT_IPT>(c, st, g, sb.keys[cur], result, sb.keys[cur ^ 1], sb.vals[cur ^ 1],
                                                          &c->flags->wide_m, LOW_BITS, top_bits, sb.ws, cols, /*need_hist=*/true,
                                                          /*iota=*/false, profiling ? c->ev[EV_X0 + 6] : nullptr,
                                                          profiling ? c->ev[EV_X0 + 7] : nullptr);
        hipLaunchKernelGGL(k_copy_gated, dim3(g), dim3(512), 0, st, &c->flags->wide_m, sb.vals[cur ^ 1], result);
    }
    if (sampled_pass == GSX_SAMPLE_WHOLE) GSX_HIP(c, hipEventRecord(samp_b[GSX_SAMPLE_WHOLE], st));
    // bucket statistics of a two-pass sort for the next sorts' choice of form: every 4th sort, read when the copy has landed,
    // never waited for (a 16-byte copy is a 4 us blit on the stream)
    const bool stats_always = g_dbg_sort2;     // (debug: the statistics of every sort)
    if (want_stats && !c->s2_stats_rec && ((c->s2_stats_tick++ & 3u) == 0u || stats_always)) {
        const Gsx2Workspace w = gsx2_ws_carve(c->s2_ws, c->sort_cap);
        GSX_HIP(c, hipMemcpyAsync(c->s2_stats_host, w.stats, sizeof(Gsx2Stats), hipMemcpyDeviceToHost, st));
        GSX_HIP(c, hipEventRecord(c->ev_s2_stats, st));
        c->s2_stats_rec = true;
        c->s2_stats_mid = mid;
        c->s2_stats_r0 = p2.map.rc == p2.map.r0 + 2u ? 23u : p2.map.r0;
    }
    if (profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_SORT1], st));
    GSX_HIP(c, hipGetLastError());
    if (sb.with_events) {   // (the context's own sort: what gsx_get_timings reports)
        c->sort_passes = static_cast<uint32_t>(n_stage) + (gated ? 1u : 0u);   // (the gated pass counts as launched)
        c->sort_gated = gated;
        c->sort_two_pass = p2.ok;
        c->sort_launches = p2.ok ? (mid ? 6u : 5u) : 3u * static_cast<uint32_t>(plan.n_passes) - 1u;   // (pass 0's count is fused into the key kernel)
    }
    return GSX_OK;
}

// Work about to be queued on `st` starts behind the latest sampled sort, if that one has not ended yet (the device runs far
// behind the host: a continuation the host launches two frames later can still meet it).
int gsx_behind_sample(gsx_ctx* c, hipStream_t st) {
    if (!c->sample_pending) return GSX_OK;
    if (hipEventQuery(c->ev_sample_done) == hipSuccess) { c->sample_pending = false; return GSX_OK; }
    if (st != c->stream) GSX_HIP(c, hipStreamWaitEvent(st, c->ev_sample_done, 0));
    return GSX_OK;
}

void gsx_join_sort(gsx_ctx* c) {
    if (!c->sort_join_pending) return;
    (void)hipStreamWaitEvent(c->stream, c->ev_sorted, 0);
    c->sort_join_pending = false;
}

int gsx_launch_sort(gsx_ctx* c) {
    GsxSortBuffers sb;
    sb.keys[0] = c->keys[0]; sb.keys[1] = c->keys[1];
    sb.vals[0] = c->vals[0]; sb.vals[1] = c->vals[1];
    sb.ws = c->sort_ws;
    sb.ws_words = c->sort_ws_words;
    const uint32_t target = (c->perm_cur + 1u) % GSX_PERM_RING;
    sb.result = c->perm[target];
    sb.with_events = true;
    sb.two_pass_ok = true;
    sb.m_copy = c->m_perm + target;
    // (with every stage bracketed by events - profiling level 1 - the stages run one after another on the main stream, and so
    // does a sort that carries a level-2 sample)
    const bool sampled = c->prof_level == 2 && c->prof_frame % c->opt_sample_every == 0u;
    const bool async = c->opt_async_sort && c->sort_stream && !c->profiling && !sampled;
    hipStream_t st = c->stream;
    if (async) {
        st = c->sort_stream;
        if (c->main_dirty) {             // a selection or an upload queued on the main stream since the last sort
            GSX_HIP(c, hipEventRecord(c->ev_pre, c->stream));
            c->pre_wait = true;
            c->main_dirty = false;
        }
        // ev_pre: the point of the main stream this sort must be behind (selection / upload, or a sort that ran on the main
        // stream itself - a sampled one - and used the same key / value / count arrays)
        if (c->pre_wait) { GSX_HIP(c, hipStreamWaitEvent(st, c->ev_pre, 0)); c->pre_wait = false; }
    } else {
        gsx_join_sort(c);                // behind the sort stream's latest sort: same arrays
        if (sampled) {
            // a sampled sort is timed as the kernels' own: it starts behind every frame that is queued (the other slot's frame
            // runs on a stream of its own), and whatever is queued on another stream before it has ended starts behind it
            // (gsx_behind_sample) - one bubble per sample
            for (GsxFrameSlot* sl : {static_cast<GsxFrameSlot*>(c), &c->parked})
                if (sl->frame_pending && sl->fstream != c->stream) GSX_HIP(c, hipStreamWaitEvent(c->stream, sl->frame_ev, 0));
        }
    }
    // the last frame that drew from the buffer this sort overwrites (four buffers: older than the two frames that may be in flight and the permutation that may wait to be drawn)
    if (c->frame_done_rec[target]) GSX_HIP(c, hipStreamWaitEvent(st, c->ev_frame_done[target], 0));
    const int rc = launch_sort_in(c, sb, st);
    if (rc != GSX_OK) return rc;
    GSX_HIP(c, hipEventRecord(c->ev_sorted, st));   // what a frame's stream waits for before it draws this permutation
    c->sorted_rec = true;
    if (sampled && c->ev_sample_done) {
        GSX_HIP(c, hipEventRecord(c->ev_sample_done, st));
        c->sample_pending = true;
    }
    c->perm_rec[target] = false;
    if (c->opt_sort_ahead && c->ev_perm[target]) {  // (a frame may draw this permutation after the NEXT sort has re-recorded ev_sorted)
        GSX_HIP(c, hipEventRecord(c->ev_perm[target], st));
        c->perm_rec[target] = true;
    }
    c->undrawn = c->undrawn + 1u > (c->opt_sort_ahead ? 2u : 1u) ? (c->opt_sort_ahead ? 2u : 1u) : c->undrawn + 1u;
    if (async) {
        c->sort_join_pending = true;
    } else if (c->sort_stream) {
        GSX_HIP(c, hipEventRecord(c->ev_pre, c->stream));   // the next sort on the sort stream starts behind this one
        c->pre_wait = true;
    }
    c->perm_cur = target;
    c->sorted_valid = true;
    return GSX_OK;
}

// ---------------------------------------------------------------------------
// gsx_sort_noalloc: the depth sort for callers that cannot allocate (HIP graph capture, serving loops).
// Its key / value ping-pong arrays and the radix count matrix live in `scratch`, a file-scope device array
// that every GPU the library runs on holds until the process ends.  In host code the array's NAME is only
// a stand-in: its device address is looked up per device with hipGetSymbolAddress at gsx_create time
// (gsx_scratch_init) and only that pointer is ever handed to kernels.
// ---------------------------------------------------------------------------
#define GSX_NOALLOC_WS_WORDS (static_cast<uint64_t>(GSX_RADIX_MAX_DIGITS) * GSX_RADIX_MAX_BLOCKS + GSX_RADIX_MAX_DIGITS)
#define GSX_NOALLOC_SCRATCH_BYTES (4ull * GSX_NOALLOC_MAX_KEYS * sizeof(uint32_t) + GSX_NOALLOC_WS_WORDS * sizeof(uint32_t))
static __device__ __align__(256) unsigned char scratch[GSX_NOALLOC_SCRATCH_BYTES];

#define GSX_MAX_DEVICES 64
static unsigned char* g_scratch_ptr[GSX_MAX_DEVICES] = {};   // per device: each GPU holds its own copy at its own address
static std::once_flag g_scratch_once[GSX_MAX_DEVICES];        // two host threads may create contexts on one device at the same time
static hipError_t g_scratch_err[GSX_MAX_DEVICES] = {};

// Called by gsx_create with `device` current.
int gsx_scratch_init(gsx_ctx* c, int device) {
    if (device < 0 || device >= GSX_MAX_DEVICES) return GSX_OK;   // (everything else works there; gsx_sort_noalloc says GSX_E_STATE with the reason)
    std::call_once(g_scratch_once[device], [device] {
        void* p = nullptr;
        g_scratch_err[device] = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (g_scratch_err[device] == hipSuccess) g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    });
    GSX_HIP(c, g_scratch_err[device]);
    return GSX_OK;
}

int gsx_launch_sort_noalloc(gsx_ctx* c, uint32_t* idx_out_dev) {
    if (c->device < 0 || c->device >= GSX_MAX_DEVICES || !g_scratch_ptr[c->device]) {
        c->err = "gsx_sort_noalloc: no scratch address for device " + std::to_string(c->device) + " (the table covers devices 0.." + std::to_string(GSX_MAX_DEVICES - 1) + ")";
        return GSX_E_STATE;
    }
    if (c->m_cap > GSX_NOALLOC_MAX_KEYS) {
        c->err = "gsx_sort_noalloc: selection of " + std::to_string(c->m_cap) + " splats exceeds the scratch array's " +
                 std::to_string(GSX_NOALLOC_MAX_KEYS) + " (use gsx_sort)";
        return GSX_E_UNSUPPORTED;
    }
    uint32_t* base = reinterpret_cast<uint32_t*>(g_scratch_ptr[c->device]);
    GsxSortBuffers sb;
    sb.keys[0] = base;
    sb.keys[1] = base + GSX_NOALLOC_MAX_KEYS;
    sb.vals[0] = base + 2ull * GSX_NOALLOC_MAX_KEYS;
    sb.vals[1] = base + 3ull * GSX_NOALLOC_MAX_KEYS;
    sb.ws = base + 4ull * GSX_NOALLOC_MAX_KEYS;
    sb.ws_words = GSX_NOALLOC_WS_WORDS;
    sb.result = idx_out_dev;
    sb.with_events = false;
    // Nothing in the scratch has to start at zero: every count-matrix column a pass reads was written by the same
    // pass's count kernel (their column count derives from the same device-side key count), and keys / values are
    // written before they are read; what an earlier, larger call left behind is never looked at.
    return launch_sort_in(c, sb, c->stream);
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

uint64_t gsx_sort_workspace_words(uint64_t) { return gsx_radix_workspace_words(GSX_RADIX_MAX_BLOCKS); }
uint64_t gsx_sort2_ws_bytes(uint64_t cap) { return gsx2_ws_bytes(cap); }

// Resident workgroups of the depth sort's scatter kernel on this device (CUs x blocks per CU).
uint32_t gsx_sort_slots(int cus) {
    int per = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_radix_scatter<uint32_t, GSX_SORT_NT, GSX_SORT_IPT, false, GSX_RADIX_MAX_DIGITS, GSX_SORT_WPE>, GSX_SORT_NT, 0) != hipSuccess || per < 1) per = 2;
    return static_cast<uint32_t>(cus) * static_cast<uint32_t>(per);
}

// Level-2 profiling: folds the sample in ring slot `slot` into the per-pass sums (waits for its stop event if needed).
void gsx_sample_drain(gsx_ctx* c, uint32_t slot) {
    const int p = c->samp_pass[slot];
    if (p < 0) return;
    c->samp_pass[slot] = -1;
    float ms = 0.0f;
    if (hipEventSynchronize(c->samp_ev[slot][1]) != hipSuccess) return;
    if (hipEventElapsedTime(&ms, c->samp_ev[slot][0], c->samp_ev[slot][1]) != hipSuccess) return;
    if (g_dbg_sort2) fprintf(stderr, "[gsx] sample: what %d  %.4f ms\n", p, ms);
    if (p <= GSX_SAMPLE_WHOLE) { c->samp_sum_ms[p] += ms; ++c->samp_n[p]; }
    if (p < GSX_SAMPLE_WHOLE) { c->samp_last_ms = ms; c->samp_last_pass = p; }
}
