// tools/sortbench2.hip — a full 3-pass depth sort of random keys with the packed-pair scatter (gsx_radix2.hpp) next to
// the round-1 kernels (gsx_radix.hpp), every kernel timed with HIP events, result checked against std::stable_sort.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I gaussian-splatting_amd/csrc -I tools tools/sortbench2.hip -o tools/sortbench2
//   tools/sortbench2 [keys=6000000] [key_bits=27] [reps=20]
#include "gsx_radix2_experiment.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

constexpr int NT = GSX_SORT_NT, IPT = GSX_SORT_IPT, TILE = NT * IPT;

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 6000000u;
    const uint32_t key_bits = argc > 2 ? (uint32_t)atoi(argv[2]) : 27u;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int per_old = 0, per_new = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_old, k_radix_scatter<uint32_t, NT, IPT, false>, NT, 0));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_new, k_radix_scatter2<NT, IPT, GSX_IN_PAIRS, GSX_OUT_PAIRS>, NT, 0));
    const GsxRadixPlan plan = gsx_radix_plan(key_bits);
    printf("n=%u key_bits=%u passes=%d occupancy old %d / new %d per CU, %d CUs, tile %d\n", n, key_bits, plan.n_passes, per_old, per_new,
           prop.multiProcessorCount, TILE);
    std::vector<uint32_t> hk(n);
    std::mt19937 rng(7);
    for (uint32_t i = 0; i < n; ++i) hk[i] = rng() & ((key_bits >= 32 ? 0u : (1u << key_bits)) - 1u);
    uint32_t *k0, *k1, *v0, *v1, *ws, *cnt_dev, *perm;
    uint2 *p0, *p1;
    CK(hipMalloc(&k0, (size_t)n * 4 + 64)); CK(hipMalloc(&k1, (size_t)n * 4 + 64)); CK(hipMalloc(&v0, (size_t)n * 4 + 64)); CK(hipMalloc(&v1, (size_t)n * 4 + 64));
    CK(hipMalloc(&p0, (size_t)n * 8 + 64)); CK(hipMalloc(&p1, (size_t)n * 8 + 64)); CK(hipMalloc(&perm, (size_t)n * 4 + 64));
    CK(hipMalloc(&ws, gsx_radix_workspace_words(GSX_RADIX_MAX_BLOCKS) * 4)); CK(hipMalloc(&cnt_dev, 4));
    CK(hipMemcpy(k0, hk.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(cnt_dev, &n, 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> ref(n);
    std::iota(ref.begin(), ref.end(), 0u);
    std::stable_sort(ref.begin(), ref.end(), [&](uint32_t a, uint32_t b) { return hk[a] < hk[b]; });
    hipEvent_t ev[16]; for (auto& x : ev) CK(hipEventCreate(&x));
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2] * 1e3f; };

    for (int variant = 0; variant < 2; ++variant) {
        const uint32_t slots = (uint32_t)prop.multiProcessorCount * (uint32_t)(variant ? per_new : per_old);
        const uint32_t g = gsx_radix_blocks(n, TILE, slots), cols = gsx_radix_columns(slots);
        uint32_t* cnt = ws; uint32_t* totals = ws + (size_t)GSX_RADIX_MAX_DIGITS * cols;
        std::vector<std::vector<float>> t(9);
        std::vector<float> whole;
        for (int r = 0; r < reps + 3; ++r) {
            int e = 0;
            CK(hipEventRecord(ev[e++], st));
            for (int p = 0; p < plan.n_passes; ++p) {
                const bool first = p == 0, last = p + 1 == plan.n_passes;
                const uint32_t sh = plan.shift[p], b = plan.bits[p];
                if (variant == 0) {
                    const uint32_t* kin = (p & 1) ? k1 : k0; uint32_t* kout = (p & 1) ? k0 : k1;
                    const uint32_t* vin = (p & 1) ? v1 : v0; uint32_t* vout = last ? perm : ((p & 1) ? v0 : v1);
                    hipLaunchKernelGGL((k_radix_hist<uint32_t, NT>), dim3(g), dim3(NT), 0, st, kin, cnt_dev, sh, b, cnt, (uint32_t)TILE, cols);
                    CK(hipEventRecord(ev[e++], st));
                    hipLaunchKernelGGL((k_radix_scan<0>), dim3(1u << b), dim3(256), 0, st, cnt, cnt_dev, (uint32_t)TILE, cols, totals);
                    CK(hipEventRecord(ev[e++], st));
                    if (first) hipLaunchKernelGGL((k_radix_scatter<uint32_t, NT, IPT, true>), dim3(g), dim3(NT), 0, st, kin, vin, kout, vout, cnt_dev, sh, b, cnt, totals, cols);
                    else hipLaunchKernelGGL((k_radix_scatter<uint32_t, NT, IPT, false>), dim3(g), dim3(NT), 0, st, kin, vin, kout, vout, cnt_dev, sh, b, cnt, totals, cols);
                    CK(hipEventRecord(ev[e++], st));
                } else {
                    const uint2* pin = (p & 1) ? p0 : p1; uint2* pout = (p & 1) ? p1 : p0;   // pass 0 writes p0, pass 1 reads p0 writes p1, ...
                    if (first) hipLaunchKernelGGL((k_radix_hist<uint32_t, NT>), dim3(g), dim3(NT), 0, st, k0, cnt_dev, sh, b, cnt, (uint32_t)TILE, cols);
                    else hipLaunchKernelGGL((k_radix_hist_pairs<NT>), dim3(g), dim3(NT), 0, st, pin, cnt_dev, sh, b, cnt, (uint32_t)TILE, cols);
                    CK(hipEventRecord(ev[e++], st));
                    hipLaunchKernelGGL((k_radix_scan<0>), dim3(1u << b), dim3(256), 0, st, cnt, cnt_dev, (uint32_t)TILE, cols, totals);
                    CK(hipEventRecord(ev[e++], st));
                    if (first && last) hipLaunchKernelGGL((k_radix_scatter2<NT, IPT, GSX_IN_KEYS_IOTA, GSX_OUT_VALS>), dim3(g), dim3(NT), 0, st, k0, nullptr, nullptr, nullptr, perm, cnt_dev, sh, b, cnt, totals, cols);
                    else if (first) hipLaunchKernelGGL((k_radix_scatter2<NT, IPT, GSX_IN_KEYS_IOTA, GSX_OUT_PAIRS>), dim3(g), dim3(NT), 0, st, k0, nullptr, nullptr, pout, nullptr, cnt_dev, sh, b, cnt, totals, cols);
                    else if (last) hipLaunchKernelGGL((k_radix_scatter2<NT, IPT, GSX_IN_PAIRS, GSX_OUT_VALS>), dim3(g), dim3(NT), 0, st, nullptr, nullptr, pin, nullptr, perm, cnt_dev, sh, b, cnt, totals, cols);
                    else hipLaunchKernelGGL((k_radix_scatter2<NT, IPT, GSX_IN_PAIRS, GSX_OUT_PAIRS>), dim3(g), dim3(NT), 0, st, nullptr, nullptr, pin, pout, nullptr, cnt_dev, sh, b, cnt, totals, cols);
                    CK(hipEventRecord(ev[e++], st));
                }
            }
            CK(hipStreamSynchronize(st));
            if (r >= 3) {
                for (int k = 0; k + 1 < e; ++k) { float ms; CK(hipEventElapsedTime(&ms, ev[k], ev[k + 1])); t[k].push_back(ms); }
                float ms; CK(hipEventElapsedTime(&ms, ev[0], ev[e - 1])); whole.push_back(ms);
            }
        }
        printf("%s: ", variant ? "pairs/2-barrier" : "round-1        ");
        for (int p = 0; p < plan.n_passes; ++p) printf("| p%d hist %.1f scan %.1f scatter %.1f ", p, med(t[3 * p]), med(t[3 * p + 1]), med(t[3 * p + 2]));
        printf("| total %.1f us\n", med(whole));
        std::vector<uint32_t> out(n);
        CK(hipMemcpy(out.data(), perm, (size_t)n * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (uint32_t i = 0; i < n; ++i) bad += out[i] != ref[i];
        printf("   check: %zu mismatches\n", bad);
        if (variant == 0) CK(hipMemcpy(k0, hk.data(), (size_t)n * 4, hipMemcpyHostToDevice));
#ifdef GSX_STAMP
        if (variant == 1) {   // the last launch's (final pass) per-phase clocks, mean over the waves that ran
            std::vector<unsigned long long> stamps((size_t)GSX_RADIX_MAX_BLOCKS * 64);
            CK(hipMemcpyFromSymbol(stamps.data(), HIP_SYMBOL(g_stamps), stamps.size() * 8));
            double sum[8] = {0}; size_t waves = 0;
            for (uint32_t b = 0; b < g; ++b) for (int w = 0; w < 8; ++w) { ++waves; for (int k = 0; k < 8; ++k) sum[k] += (double)stamps[((size_t)b * 8 + w) * 8 + k]; }
            const char* names[7] = {"wait keys", "rank", "barrier A", "prefix+C", "lds scatter+req", "barrier B", "write-out"};
            double tot = 0; for (int k = 0; k < 7; ++k) tot += sum[k] / waves;
            printf("   stamps (s_memtime ticks per wave, last pass, %u blocks): total %.0f\n", g, tot);
            for (int k = 0; k < 7; ++k) printf("      %-16s %9.0f  %5.1f %%\n", names[k], sum[k] / waves, 100.0 * sum[k] / waves / tot);
        }
#endif
    }
    return 0;
}
