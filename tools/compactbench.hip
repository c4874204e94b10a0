// tools/compactbench.hip — the one-launch flag compaction of gsx_lod.hip on its own: HIP-event time per launch beside the
// three-launch form, a check against the CPU, and (instrumented copy) where a workgroup's time goes, by the 100 MHz clock.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I gaussian-splatting_amd/csrc tools/compactbench.hip -o tools/compactbench
//   tools/compactbench [flags=8484260] [reps=30]
#include "../gaussian-splatting_amd/csrc/gsx_lod.hip"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// instrumented copy: NT threads, LAYOUT 0 = a thread owns upt consecutive units, 1 = unit (tile, thread): coalesced loads, a scan per tile
template <int NT, int LAYOUT, bool STAGE>
__global__ __launch_bounds__(NT) void k_cp_exp(const uint8_t* __restrict__ flags, uint32_t n, uint32_t match, uint32_t upt,
                                               uint64_t* __restrict__ agg, uint32_t seq, uint32_t* __restrict__ out,
                                               uint32_t* __restrict__ m_out, uint64_t* __restrict__ ts) {
    __shared__ uint32_t s_tmp[NT / 64];
    __shared__ uint32_t s_prefix;
    __shared__ uint32_t s_ids[8192];
    const uint32_t tid = threadIdx.x, units = (n + 15u) / 16u;
    const uint64_t t0 = wall_clock64();
    uint32_t t[8][4], ub[8];
    uint32_t cnt = 0, tile_cnt[8];
    const uint32_t match4 = match * 0x01010101u;
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) {
        ub[k] = LAYOUT ? (blockIdx.x * upt + k) * NT + tid : (blockIdx.x * NT + tid) * upt + k;
        uint4 w = make_uint4(match4 ^ 0x01010101u, match4 ^ 0x01010101u, match4 ^ 0x01010101u, match4 ^ 0x01010101u);
        if (k < upt && ub[k] < units) w = *reinterpret_cast<const uint4*>(flags + static_cast<uint64_t>(ub[k]) * 16u);
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t y = ww[q] ^ match4;
            t[k][q] = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
        }
        if (k < upt && ub[k] * 16u + 16u > n) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t first = ub[k] * 16u + 4u * q, valid = first >= n ? 0u : (n - first >= 4u ? 4u : n - first);
                t[k][q] &= valid >= 4u ? 0xFFFFFFFFu : ((1u << (8u * valid)) - 1u);
            }
        }
        tile_cnt[k] = __popc(t[k][0]) + __popc(t[k][1]) + __popc(t[k][2]) + __popc(t[k][3]);
        cnt += tile_cnt[k];
    }
    const uint64_t t1 = wall_clock64();
    uint32_t total, excl = 0, tile_excl[8], tile_tot[8];
    if (LAYOUT) {
        total = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k)
            if (k < upt) { tile_excl[k] = total + gsx_block_excl_scan<NT>(tile_cnt[k], s_tmp, &tile_tot[k]); total += tile_tot[k]; }
    } else {
        excl = gsx_block_excl_scan<NT>(cnt, s_tmp, &total);
    }
    if (tid == 0) __hip_atomic_store(&agg[blockIdx.x], (static_cast<uint64_t>(seq) << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t t2 = wall_clock64();
    if (tid < 64u) {
        uint32_t sum = 0;
        for (uint32_t p = tid; p < blockIdx.x; p += 64u) {
            uint64_t v = __hip_atomic_load(&agg[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (uint32_t spins = 0; static_cast<uint32_t>(v >> 32) != seq && spins < (1u << 20); ++spins) {
                __builtin_amdgcn_s_sleep(2);
                v = __hip_atomic_load(&agg[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (static_cast<uint32_t>(v >> 32) == seq) sum += static_cast<uint32_t>(v);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        if (tid == 0) s_prefix = sum;
    }
    __syncthreads();
    const uint64_t t3 = wall_clock64();
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *m_out = s_prefix + total;
    const uint32_t run = s_prefix;
    if (run + total <= n) {
        if (STAGE) {
            for (uint32_t win = 0; win < total; win += 8192u) {
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) {
                    if (k < upt) {
                        uint32_t pos = (LAYOUT ? tile_excl[k] : excl) - win;
                        if (!LAYOUT) { for (uint32_t kk = 0; kk < k; ++kk) pos += tile_cnt[kk]; }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            uint32_t bits = t[k][q];
                            while (bits) {
                                const uint32_t b = static_cast<uint32_t>(__ffs(static_cast<int>(bits))) - 1u;
                                if (pos < 8192u) s_ids[pos] = ub[k] * 16u + 4u * q + (b >> 3);
                                ++pos;
                                bits &= bits - 1u;
                            }
                        }
                    }
                }
                __syncthreads();
                const uint32_t m = total - win < 8192u ? total - win : 8192u;
                for (uint32_t k = tid; k < m; k += NT) out[run + win + k] = s_ids[k];
                __syncthreads();
            }
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
                if (k < upt) {
                    uint32_t pos = run + (LAYOUT ? tile_excl[k] : excl);
                    if (!LAYOUT) { for (uint32_t kk = 0; kk < k; ++kk) pos += tile_cnt[kk]; }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t bits = t[k][q];
                        while (bits) {
                            const uint32_t b = static_cast<uint32_t>(__ffs(static_cast<int>(bits))) - 1u;
                            out[pos++] = ub[k] * 16u + 4u * q + (b >> 3);
                            bits &= bits - 1u;
                        }
                    }
                }
            }
        }
    }
    const uint64_t t4 = wall_clock64();
    if (tid == 0 && ts) { uint64_t* o = ts + blockIdx.x * 5u; o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4; }
}

// the walk's stores: one byte per emitted node, ids in the order a walk visits them (shuffled in blocks of 8)
template <int MODE>
__global__ __launch_bounds__(256) void k_stamp(const uint32_t* __restrict__ ids, uint32_t m, uint8_t stamp, uint8_t* __restrict__ flags) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    if (MODE == 1) __builtin_nontemporal_store(stamp, flags + ids[i]);
    else flags[ids[i]] = stamp;
}

// experiment: one BIT per node (set with an atomic OR, cleared by the compaction that reads it) instead of a stamped byte
__global__ __launch_bounds__(256) void k_stamp_bits(const uint32_t* __restrict__ ids, uint32_t m, uint32_t* __restrict__ bits) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const uint32_t id = ids[i];
    atomicOr(&bits[id >> 5], 1u << (id & 31u));
}
template <int NT>
__global__ __launch_bounds__(NT) void k_compact_bits(uint32_t* __restrict__ bits, uint32_t n, uint64_t* __restrict__ agg, uint32_t seq,
                                                     uint32_t* __restrict__ out, uint32_t* __restrict__ m_out) {
    __shared__ uint32_t s_tmp[NT / 64];
    __shared__ uint32_t s_prefix;
    const uint32_t tid = threadIdx.x, units = (n + 127u) / 128u;           // 16 bytes = 128 nodes per thread
    const uint32_t u = blockIdx.x * NT + tid;
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    if (u < units) {
        w = reinterpret_cast<const uint4*>(bits)[u];
        if (w.x | w.y | w.z | w.w) reinterpret_cast<uint4*>(bits)[u] = make_uint4(0u, 0u, 0u, 0u);   // cleared for the next cut
    }
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
    const uint32_t cnt = __popc(w.x) + __popc(w.y) + __popc(w.z) + __popc(w.w);
    uint32_t total;
    const uint32_t excl = gsx_block_excl_scan<NT>(cnt, s_tmp, &total);
    if (tid == 0) __hip_atomic_store(&agg[blockIdx.x], (static_cast<uint64_t>(seq) << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64u) {
        uint32_t sum = 0;
        for (uint32_t p = tid; p < blockIdx.x; p += 64u) {
            uint64_t v = __hip_atomic_load(&agg[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (uint32_t spins = 0; static_cast<uint32_t>(v >> 32) != seq && spins < (1u << 20); ++spins) {
                __builtin_amdgcn_s_sleep(2);
                v = __hip_atomic_load(&agg[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (static_cast<uint32_t>(v >> 32) == seq) sum += static_cast<uint32_t>(v);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        if (tid == 0) s_prefix = sum;
    }
    __syncthreads();
    const uint32_t run = s_prefix;
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *m_out = run + total;
    if (run + total > n) return;
    uint32_t pos = run + excl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t b = ww[q];
        while (b) {
            const uint32_t k = static_cast<uint32_t>(__ffs(static_cast<int>(b))) - 1u;
            const uint32_t id = u * 128u + 32u * q + k;
            if (id < n) out[pos] = id;
            ++pos;
            b &= b - 1u;
        }
    }
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 8484260u;
    const int reps = argc > 2 ? atoi(argv[2]) : 30;
    std::mt19937 rng(3);
    std::vector<uint8_t> h((n + 15) / 16 * 16 + 16, 0);
    const float dens[4] = {0.0f, 0.05f, 0.2f, 0.45f};
    std::vector<uint32_t> ref;
    for (uint32_t b = 0; b < n; b += 4096) {
        const float d = dens[rng() & 3];
        for (uint32_t i = b; i < b + 4096 && i < n; ++i) {
            h[i] = (rng() & 0xFFFF) < d * 65536.f ? 7 : (rng() & 1 ? 6 : 0);
            if (h[i] == 7) ref.push_back(i);
        }
    }
    for (size_t i = n; i < h.size(); ++i) h[i] = 7;      // (bytes behind n must not count)
    printf("flags %u set %zu (%.1f %%)\n", n, ref.size(), 100.0 * ref.size() / n);
    uint8_t* d_f; uint32_t *d_out, *d_m, *d_bs; uint64_t *d_agg, *d_ts; GsxDevFlags* d_fl;
    CK(hipMalloc(&d_f, h.size())); CK(hipMalloc(&d_out, (size_t)n * 4)); CK(hipMalloc(&d_m, 4)); CK(hipMalloc(&d_bs, (n / 4096 + 2) * 4));
    CK(hipMalloc(&d_agg, 2048 * 8)); CK(hipMalloc(&d_ts, 2048 * 5 * 8)); CK(hipMalloc(&d_fl, sizeof(GsxDevFlags)));
    CK(hipMemcpy(d_f, h.data(), h.size(), hipMemcpyHostToDevice)); CK(hipMemset(d_agg, 0, 2048 * 8)); CK(hipMemset(d_fl, 0, sizeof(GsxDevFlags)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint32_t seq = 0;
    const uint32_t units = (n + 15) / 16;
    std::vector<uint32_t> got(n);
    auto check = [&](const char* what) -> bool {
        uint32_t m = 0;
        if (hipMemcpy(&m, d_m, 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
        if (m != ref.size()) { printf("  %s: count %u != %zu\n", what, m, ref.size()); return false; }
        if (hipMemcpy(got.data(), d_out, (size_t)m * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
        for (uint32_t i = 0; i < m; ++i) if (got[i] != ref[i]) { printf("  %s: id %u differs\n", what, i); return false; }
        return true;
    };
    auto time_it = [&](const char* what, auto launch) -> int {
        CK(hipMemset(d_out, 0xFF, (size_t)n * 4));
        launch(); CK(hipDeviceSynchronize());
        const bool ok = check(what);
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-58s %7.2f us per compaction  %s\n", what, ms * 1000.f / reps, ok ? "ok" : "WRONG");
        return 0;
    };
    const uint32_t nb = (n + GSX_CMP_TILE - 1) / GSX_CMP_TILE;
    time_it("three launches (count, scan, write)", [&] {
        hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(GSX_CMP_NT), 0, 0, d_f, n, 7u, d_bs);
        hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, 0, d_bs, nb, d_m, (uint32_t*)nullptr);
        hipLaunchKernelGGL(k_compact_write, dim3(nb), dim3(GSX_CMP_NT), 0, 0, d_f, n, 7u, d_bs, d_out);
    });
    {
        const uint32_t upt = (units + GSX_CP1_NT * GSX_CP1_MAXG - 1) / (GSX_CP1_NT * GSX_CP1_MAXG), g = (units + GSX_CP1_NT * upt - 1) / (GSX_CP1_NT * upt);
        char name[128]; snprintf(name, sizeof name, "one launch, shipped (upt %u, %u workgroups)", upt, g);
        time_it(name, [&] { hipLaunchKernelGGL(k_compact_onepass, dim3(g), dim3(GSX_CP1_NT), 0, 0, d_f, n, 7u, upt, d_agg, ++seq, d_out, d_m, (uint32_t*)nullptr, d_fl); });
    }
    auto exp = [&](auto kernel, int nt, uint32_t upt, const char* label) {
        const uint32_t g = (units + nt * upt - 1) / (nt * upt);
        if (g > 2048) return;
        char name[160]; snprintf(name, sizeof name, "%s NT %d upt %u: %u workgroups", label, nt, upt, g);
        time_it(name, [&] { hipLaunchKernelGGL(kernel, dim3(g), dim3(nt), 0, 0, d_f, n, 7u, upt, d_agg, ++seq, d_out, d_m, (uint64_t*)nullptr); });
        // one instrumented launch: phases of the first, a middle and the last workgroup, and the span of the whole grid
        (void)hipMemset(d_ts, 0, 2048 * 5 * 8);
        hipLaunchKernelGGL(kernel, dim3(g), dim3(nt), 0, 0, d_f, n, 7u, upt, d_agg, ++seq, d_out, d_m, d_ts);
        (void)hipDeviceSynchronize();
        std::vector<uint64_t> ts(g * 5);
        (void)hipMemcpy(ts.data(), d_ts, ts.size() * 8, hipMemcpyDeviceToHost);
        uint64_t lo = ~0ull, hi = 0;
        for (uint32_t b = 0; b < g; ++b) { lo = std::min(lo, ts[b * 5]); hi = std::max(hi, ts[b * 5 + 4]); }
        printf("    grid first start -> last end %.2f us;", (hi - lo) * 0.01);
        for (uint32_t b : {0u, g / 2, g - 1})
            printf("  wg %u: start +%.2f load %.2f scan+publish %.2f look-back %.2f write %.2f |", b, (ts[b * 5] - lo) * 0.01, (ts[b * 5 + 1] - ts[b * 5]) * 0.01,
                   (ts[b * 5 + 2] - ts[b * 5 + 1]) * 0.01, (ts[b * 5 + 3] - ts[b * 5 + 2]) * 0.01, (ts[b * 5 + 4] - ts[b * 5 + 3]) * 0.01);
        printf("\n");
    };
    {   // the shipped launch behind scattered stamp stores (what the frame loop does): events around the compaction only
        std::vector<uint32_t> order(ref);
        for (size_t b = 0; b + 8 <= order.size(); b += 8) { const size_t o = (rng() % (order.size() / 8)) * 8; for (int k = 0; k < 8; ++k) std::swap(order[b + k], order[o + k]); }
        uint32_t* d_ids; CK(hipMalloc(&d_ids, order.size() * 4)); CK(hipMemcpy(d_ids, order.data(), order.size() * 4, hipMemcpyHostToDevice));
        const uint32_t upt = (units + GSX_CP1_NT * GSX_CP1_MAXG - 1) / (GSX_CP1_NT * GSX_CP1_MAXG), g = (units + GSX_CP1_NT * upt - 1) / (GSX_CP1_NT * upt);
        const uint32_t m = (uint32_t)order.size();
        for (int mode = 0; mode < 2; ++mode) {
            float sum = 0, sum_st = 0;
            for (int i = 0; i < reps + 3; ++i) {
                hipEvent_t a, b, c0; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c0));
                CK(hipEventRecord(c0, 0));
                if (mode) hipLaunchKernelGGL(k_stamp<1>, dim3((m + 255) / 256), dim3(256), 0, 0, d_ids, m, (uint8_t)7, d_f);
                else hipLaunchKernelGGL(k_stamp<0>, dim3((m + 255) / 256), dim3(256), 0, 0, d_ids, m, (uint8_t)7, d_f);
                CK(hipEventRecord(a, 0));
                hipLaunchKernelGGL(k_compact_onepass, dim3(g), dim3(GSX_CP1_NT), 0, 0, d_f, n, 7u, upt, d_agg, ++seq, d_out, d_m, (uint32_t*)nullptr, d_fl);
                CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
                float ms, ms2; CK(hipEventElapsedTime(&ms, a, b)); CK(hipEventElapsedTime(&ms2, c0, a));
                if (i >= 3) { sum += ms; sum_st += ms2; }
                CK(hipEventDestroy(a)); CK(hipEventDestroy(b)); CK(hipEventDestroy(c0));
            }
            printf("behind %zu scattered byte stores (%s): stores %.2f us, compaction %.2f us  %s\n", order.size(), mode ? "nontemporal" : "plain", sum_st * 1000.f / reps,
                   sum * 1000.f / reps, check("behind stores") ? "ok" : "WRONG");
        }
    }
    {   // one bit per node: atomic OR stores, a compaction that reads 1 MB instead of 8.5 and clears what it read
        std::vector<uint32_t> order(ref);
        for (size_t b = 0; b + 8 <= order.size(); b += 8) { const size_t o = (rng() % (order.size() / 8)) * 8; for (int k = 0; k < 8; ++k) std::swap(order[b + k], order[o + k]); }
        uint32_t *d_ids, *d_bits; CK(hipMalloc(&d_ids, order.size() * 4)); CK(hipMemcpy(d_ids, order.data(), order.size() * 4, hipMemcpyHostToDevice));
        const uint32_t words = ((n + 127) / 128) * 4; CK(hipMalloc(&d_bits, words * 4)); CK(hipMemset(d_bits, 0, words * 4));
        const uint32_t m = (uint32_t)order.size(), units128 = (n + 127) / 128;
        for (int nt : {256, 512}) {
            const uint32_t g = (units128 + nt - 1) / nt;
            float sum = 0, sum_st = 0; bool ok = true;
            for (int i = 0; i < reps + 3; ++i) {
                hipEvent_t a, b, c0; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c0));
                if (i == 0) CK(hipMemset(d_out, 0xFF, (size_t)n * 4));
                CK(hipEventRecord(c0, 0));
                hipLaunchKernelGGL(k_stamp_bits, dim3((m + 255) / 256), dim3(256), 0, 0, d_ids, m, d_bits);
                CK(hipEventRecord(a, 0));
                if (nt == 256) hipLaunchKernelGGL(k_compact_bits<256>, dim3(g), dim3(256), 0, 0, d_bits, n, d_agg, ++seq, d_out, d_m);
                else hipLaunchKernelGGL(k_compact_bits<512>, dim3(g), dim3(512), 0, 0, d_bits, n, d_agg, ++seq, d_out, d_m);
                CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
                float ms, ms2; CK(hipEventElapsedTime(&ms, a, b)); CK(hipEventElapsedTime(&ms2, c0, a));
                if (i >= 3) { sum += ms; sum_st += ms2; }
                if (i == 0 || i == reps + 2) ok = ok && check("bits");
                CK(hipEventDestroy(a)); CK(hipEventDestroy(b)); CK(hipEventDestroy(c0));
            }
            printf("one BIT per node, NT %d (%u workgroups): %zu atomic ORs %.2f us, compaction %.2f us  %s\n", nt, g, order.size(), sum_st * 1000.f / reps, sum * 1000.f / reps, ok ? "ok" : "WRONG");
        }
    }
    for (uint32_t upt : {3u, 2u, 1u}) {
        exp(k_cp_exp<512, 0, true>, 512, upt, "consecutive units, staged,");
        exp(k_cp_exp<512, 1, true>, 512, upt, "tile-coalesced, staged,   ");
        exp(k_cp_exp<512, 1, false>, 512, upt, "tile-coalesced, direct,   ");
        exp(k_cp_exp<512, 0, false>, 512, upt, "consecutive units, direct,");
        exp(k_cp_exp<256, 0, false>, 256, upt * 2, "consecutive units, direct,");
        exp(k_cp_exp<256, 1, false>, 256, upt * 2, "tile-coalesced, direct,   ");
        exp(k_cp_exp<256, 1, true>, 256, upt * 2, "tile-coalesced, staged,   ");
        exp(k_cp_exp<1024, 1, true>, 1024, upt, "tile-coalesced, staged,   ");
    }
    return 0;
}
