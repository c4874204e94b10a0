// tools/sortbench.hip — one radix pass (hist, scan, scatter) over random keys, timed per kernel with HIP events and
// checked against a CPU stable counting sort.  Experiment harness for gsx_radix.hpp:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I gaussian-splatting_amd/csrc [-DGSX_...] tools/sortbench.hip -o tools/sortbench
//   tools/sortbench [keys=6000000] [bits=9] [reps=20] [blocks=0 (auto)]
#include "gsx_radix.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 6000000u;
    const uint32_t bits = argc > 2 ? (uint32_t)atoi(argv[2]) : 9u;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    uint32_t g = argc > 4 ? (uint32_t)atoi(argv[4]) : 0u;
    uint32_t ncols = argc > 5 ? (uint32_t)atoi(argv[5]) : 0u;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int per = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_radix_scatter<uint32_t, GSX_SORT_NT, GSX_SORT_IPT, false>, GSX_SORT_NT, 0));
    if (!g) g = gsx_radix_blocks(n, GSX_SORT_TILE, (uint32_t)prop.multiProcessorCount * (uint32_t)per);
    if (!ncols) ncols = gsx_radix_columns((uint32_t)prop.multiProcessorCount * (uint32_t)per);
    printf("n=%u bits=%u cols=%u blocks=%u (occupancy %d/CU, %d CUs) tile=%d\n", n, bits, ncols, g, per, prop.multiProcessorCount, GSX_SORT_TILE);
    std::vector<uint32_t> hk(n), hv(n);
    std::mt19937 rng(7);
    for (uint32_t i = 0; i < n; ++i) { hk[i] = rng() & 0x07FFFFFFu; hv[i] = i; }
    uint32_t *ki, *vi, *ko, *vo, *ws, *cnt_dev;
    CK(hipMalloc(&ki, (size_t)n * 4 + 64)); CK(hipMalloc(&vi, (size_t)n * 4 + 64)); CK(hipMalloc(&ko, (size_t)n * 4 + 64)); CK(hipMalloc(&vo, (size_t)n * 4 + 64));
    CK(hipMalloc(&ws, gsx_radix_workspace_words(GSX_RADIX_MAX_BLOCKS) * 4)); CK(hipMalloc(&cnt_dev, 4));
    CK(hipMemcpy(ki, hk.data(), (size_t)n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(vi, hv.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(cnt_dev, &n, 4, hipMemcpyHostToDevice));
    uint32_t* cnt = ws; uint32_t* totals = ws + (size_t)GSX_RADIX_MAX_DIGITS * ncols;
    hipEvent_t ev[4]; for (auto& x : ev) CK(hipEventCreate(&x));
    std::vector<float> th, ts, tx;
    for (int r = 0; r < reps + 3; ++r) {
        CK(hipEventRecord(ev[0], st));
        hipLaunchKernelGGL((k_radix_hist<uint32_t, GSX_SORT_NT>), dim3(g), dim3(GSX_SORT_NT), 0, st, ki, cnt_dev, 0u, bits, cnt, (uint32_t)GSX_SORT_TILE, ncols);
        CK(hipEventRecord(ev[1], st));
        hipLaunchKernelGGL((k_radix_scan<0>), dim3(1u << bits), dim3(256), 0, st, cnt, cnt_dev, (uint32_t)GSX_SORT_TILE, ncols, totals);
        CK(hipEventRecord(ev[2], st));
        hipLaunchKernelGGL((k_radix_scatter<uint32_t, GSX_SORT_NT, GSX_SORT_IPT, false>), dim3(g), dim3(GSX_SORT_NT), 0, st, ki, vi, ko, vo, cnt_dev, 0u, bits, cnt, totals, ncols);
        CK(hipEventRecord(ev[3], st));
        CK(hipStreamSynchronize(st));
        float a, b, c; CK(hipEventElapsedTime(&a, ev[0], ev[1])); CK(hipEventElapsedTime(&b, ev[1], ev[2])); CK(hipEventElapsedTime(&c, ev[2], ev[3]));
        if (r >= 3) { th.push_back(a); ts.push_back(b); tx.push_back(c); }
    }
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2] * 1e3f; };
    printf("hist %.1f us  scan %.1f us  scatter %.1f us (min %.1f)  -> %.2f TB/s algorithmic (16 B/key)\n", med(th), med(ts), med(tx),
           *std::min_element(tx.begin(), tx.end()) * 1e3f, 16.0 * n / (med(tx) * 1e-6) / 1e12);
    // check
    std::vector<uint32_t> ok(n), ov(n);
    CK(hipMemcpy(ok.data(), ko, (size_t)n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ov.data(), vo, (size_t)n * 4, hipMemcpyDeviceToHost));
    const uint32_t nd = 1u << bits;
    std::vector<uint32_t> off(nd + 1, 0);
    for (uint32_t i = 0; i < n; ++i) ++off[(hk[i] & (nd - 1)) + 1];
    for (uint32_t d = 0; d < nd; ++d) off[d + 1] += off[d];
    std::vector<uint32_t> rk(n), rv(n);
    for (uint32_t i = 0; i < n; ++i) { const uint32_t p = off[hk[i] & (nd - 1)]++; rk[p] = hk[i]; rv[p] = i; }
    size_t bad = 0;
    for (uint32_t i = 0; i < n; ++i) bad += (rk[i] != ok[i]) || (rv[i] != ov[i]);
    printf("check: %zu mismatches\n", bad);
    return 0;
}
