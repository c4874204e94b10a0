// tools/membw.hip — calibration: achieved streaming bandwidth of trivial kernels at the
// buffer sizes the sort works on, and the cost of back-to-back dependent launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_copy16(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_copy4(const float* __restrict__ a, float* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read16(const float4* __restrict__ a, float* __restrict__ out, size_t n) {
    float s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) out[0] = s;
}
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

int main(int argc, char** argv) {
    hipStream_t st; CK(hipStreamCreate(&st));
    if (argc == 3) {
        // calibration mode for PMC passes: `membw <MiB> <copy16|copy4>` = 5 launches of one kernel over <MiB> MiB
        const size_t sz = (size_t)atoi(argv[1]) << 20;
        float *a, *b; CK(hipMalloc(&a, sz)); CK(hipMalloc(&b, sz));
        CK(hipMemset(a, 1, sz)); CK(hipMemset(b, 0, sz));
        for (int it = 0; it < 5; ++it) {
            if (argv[2][4] == '1') hipLaunchKernelGGL(k_copy16, dim3((int)((sz / 16 + 255) / 256)), dim3(256), 0, st, (const float4*)a, (float4*)b, sz / 16);
            else hipLaunchKernelGGL(k_copy4, dim3((int)((sz / 4 + 255) / 256)), dim3(256), 0, st, a, b, sz / 4);
        }
        CK(hipStreamSynchronize(st));
        printf("calibration: %zu bytes read + %zu bytes written per launch\n", sz, sz);
        return 0;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t maxb = 1ull << 30;
    float *a, *b; CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb));
    CK(hipMemset(a, 1, maxb)); CK(hipMemset(b, 0, maxb));
    const size_t sizes[] = {24u << 20, 48u << 20, 96u << 20, 192u << 20, 480u << 20, 1000u << 20};
    for (size_t sz : sizes) {
        for (int grid : {1024, 2048, 8192, 0}) {
            for (int kind = 0; kind < 3; ++kind) {
                float best = 1e9f;
                for (int it = 0; it < 12; ++it) {
                    CK(hipEventRecord(e0, st));
                    size_t n16 = sz / 16, n4 = sz / 4;
                    int g = grid ? grid : (int)((n16 + 255) / 256);
                    if (kind == 0) hipLaunchKernelGGL(k_copy16, dim3(g), dim3(256), 0, st, (const float4*)a, (float4*)b, n16);
                    else if (kind == 1) hipLaunchKernelGGL(k_copy4, dim3(grid ? grid : (int)((n4 + 255) / 256)), dim3(256), 0, st, a, b, n4);
                    else hipLaunchKernelGGL(k_read16, dim3(g), dim3(256), 0, st, (const float4*)a, b, n16);
                    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it >= 2 && ms < best) best = ms;
                }
                const double bytes = (kind == 2 ? 1.0 : 2.0) * sz;
                printf("size %4zu MiB grid %5d %-7s best %8.1f us  %7.2f TB/s\n", sz >> 20, grid, kind == 0 ? "copy16" : kind == 1 ? "copy4" : "read16", best * 1e3, bytes / (best * 1e-3) / 1e12);
            }
        }
    }
    for (int nk : {1, 10, 100}) {
        float best = 1e9f;
        for (int it = 0; it < 10; ++it) {
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < nk; ++k) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, (int*)nullptr);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%3d empty launches: %8.1f us (%.2f us each)\n", nk, best * 1e3, best * 1e3 / nk);
    }
    return 0;
}
