// tools/gatherbench.hip — what a conservative pre-cull on the 16-byte position record could save k_chunk_count: random gathers
// by sorted rank of (a) one 64-byte GsxGeo line per rank from a 6 M-record array, (b) one 16-byte float4 per rank from the SoA
// position array, (c) the 16-byte record for every rank and the 64-byte line for a fraction of them (the survivors).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gatherbench.hip -o tools/gatherbench
//   tools/gatherbench [records=6000000] [ranks=1300000] [survive_percent=25]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(512) void k_gather64(const uint32_t* __restrict__ idx, uint32_t m, const float4* __restrict__ geo, float* __restrict__ out) {
    const uint32_t k = blockIdx.x * 1024u + threadIdx.x * 2u;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (k + j < m) {
            const float4* g = geo + static_cast<uint64_t>(idx[k + j]) * 4;
            const float4 a = g[0], b = g[1], c = g[2], d = g[3];
            acc += a.x + b.y + c.z + d.w;
        }
    }
    if (acc == 12345.678f) out[k] = acc;
}
__global__ __launch_bounds__(512) void k_gather16(const uint32_t* __restrict__ idx, uint32_t m, const float4* __restrict__ pos, float* __restrict__ out) {
    const uint32_t k = blockIdx.x * 1024u + threadIdx.x * 2u;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (k + j < m) { const float4 a = pos[idx[k + j]]; acc += a.x + a.w; }
    if (acc == 12345.678f) out[k] = acc;
}
__global__ __launch_bounds__(512) void k_gather_two_step(const uint32_t* __restrict__ idx, uint32_t m, const float4* __restrict__ pos, const float4* __restrict__ geo,
                                                         float thr, float* __restrict__ out) {
    const uint32_t k = blockIdx.x * 1024u + threadIdx.x * 2u;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (k + j < m) {
            const uint32_t i = idx[k + j];
            const float4 a = pos[i];
            if (a.w < thr) {          // survivor: the full record
                const float4* g = geo + static_cast<uint64_t>(i) * 4;
                const float4 b = g[1], c = g[2], d = g[3];
                acc += b.y + c.z + d.w;
            }
            acc += a.x;
        }
    }
    if (acc == 12345.678f) out[k] = acc;
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 6000000u, m = argc > 2 ? (uint32_t)atoll(argv[2]) : 1300000u;
    const float survive = (argc > 3 ? atof(argv[3]) : 25.0) / 100.0;
    std::vector<uint32_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(3);
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<float4> pos(n);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    for (auto& p : pos) p = make_float4(1.f, 2.f, 3.f, U(rng));
    uint32_t* d_idx; float4 *d_pos, *d_geo; float* d_out;
    CK(hipMalloc(&d_idx, (size_t)m * 4)); CK(hipMalloc(&d_pos, (size_t)n * 16)); CK(hipMalloc(&d_geo, (size_t)n * 64)); CK(hipMalloc(&d_out, (size_t)m * 4 + 4096));
    CK(hipMemcpy(d_idx, perm.data(), (size_t)m * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_pos, pos.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    CK(hipMemset(d_geo, 0, (size_t)n * 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 g((m + 1023) / 1024), b(512);
    auto time = [&](int which) -> float {
        std::vector<float> t;
        for (int r = 0; r < 13; ++r) {
            (void)hipEventRecord(e0, 0);
            if (which == 0) hipLaunchKernelGGL(k_gather64, g, b, 0, 0, d_idx, m, d_geo, d_out);
            else if (which == 1) hipLaunchKernelGGL(k_gather16, g, b, 0, 0, d_idx, m, d_pos, d_out);
            else hipLaunchKernelGGL(k_gather_two_step, g, b, 0, 0, d_idx, m, d_pos, d_geo, survive, d_out);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (r >= 3) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        return t[t.size() / 2] * 1e3f;
    };
    const float a = time(0), c = time(1), d = time(2);
    printf("{\"records\": %u, \"ranks\": %u, \"gather_64B_us\": %.1f, \"gather_16B_us\": %.1f, \"two_step_us\": %.1f, \"survivors\": %.2f, "
           "\"GBps_64B\": %.0f, \"GBps_16B\": %.0f}\n", n, m, a, c, d, survive, 68.0 * m / a * 1e-3, 20.0 * m / c * 1e-3);
    return 0;
}
