// tools/rocprim_calib.hip — calibration point for the depth sort: rocprim::radix_sort_pairs (the vendor's device radix
// sort, onesweep on gfx9) on the same box, same shapes: u32 key + u32 value, 25 / 27 / 32 significant key bits, 1 M / 6 M /
// 20 M pairs.  Tools only: nothing of rocprim is linked into libgsx.so.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rocprim_calib.hip -o tools/rocprim_calib
//   tools/rocprim_calib [reps=20]
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    hipStream_t st; CK(hipStreamCreate(&st));
    const uint32_t sizes[3] = {1000000u, 6000000u, 20000000u};
    const unsigned bit_sets[3] = {25u, 27u, 32u};
    const uint32_t nmax = sizes[2];
    std::vector<uint32_t> hk(nmax), hv(nmax);
    std::mt19937 rng(7);
    for (uint32_t i = 0; i < nmax; ++i) { hk[i] = rng(); hv[i] = i; }
    uint32_t *ki, *vi, *ko, *vo;
    CK(hipMalloc(&ki, (size_t)nmax * 4)); CK(hipMalloc(&vi, (size_t)nmax * 4)); CK(hipMalloc(&ko, (size_t)nmax * 4)); CK(hipMalloc(&vo, (size_t)nmax * 4));
    CK(hipMemcpy(vi, hv.data(), (size_t)nmax * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (unsigned bits : bit_sets) {
        std::vector<uint32_t> masked(nmax);
        const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
        for (uint32_t i = 0; i < nmax; ++i) masked[i] = hk[i] & mask;
        CK(hipMemcpy(ki, masked.data(), (size_t)nmax * 4, hipMemcpyHostToDevice));
        for (uint32_t n : sizes) {
            size_t tmp_bytes = 0;
            CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, ki, ko, vi, vo, n, 0u, bits, st));
            void* tmp = nullptr; CK(hipMalloc(&tmp, tmp_bytes));
            std::vector<float> t;
            for (int r = 0; r < reps + 3; ++r) {
                CK(hipEventRecord(e0, st));
                CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, ki, ko, vi, vo, n, 0u, bits, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 3) t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            const double med_us = t[t.size() / 2] * 1e3, min_us = t[0] * 1e3;
            // check order on a sample
            std::vector<uint32_t> ok(n);
            CK(hipMemcpy(ok.data(), ko, (size_t)n * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (uint32_t i = 1; i < n; ++i) bad += ok[i - 1] > ok[i];
            printf("{\"tool\": \"rocprim::radix_sort_pairs\", \"pairs\": %u, \"key_bits\": %u, \"median_us\": %.1f, \"min_us\": %.1f, \"pairs_per_s\": %.3e, \"temp_bytes\": %zu, \"unsorted\": %zu}\n",
                   n, bits, med_us, min_us, n / (med_us * 1e-6), tmp_bytes, bad);
            CK(hipFree(tmp));
        }
    }
    return 0;
}
