// tools/sort2bench.hip — the two-pass depth sort of gsx_sort2.hpp on its own: per-kernel HIP-event times and a check
// against std::sort of (key, index) on the CPU.  Scenes: "clustered" (cfg 2 shape: 70 % in 64 Gaussian clusters of
// sigma 0.25 + uniform background), "tight" (cfg 5 shape: 90 % in 4 clusters of sigma 0.08: oversized buckets, k2_big),
// "uniform", "equal" (every splat at the same place: one bucket).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I gaussian-splatting_amd/csrc tools/sort2bench.hip -o tools/sort2bench
//   tools/sort2bench [splats=6000000] [scene=clustered] [reps=20]
#include "gsx_sort2.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static uint32_t host_key(float d) { uint32_t b; memcpy(&b, &d, 4); return (~b) & 0x7FFFFFFFu; }

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 6000000u;
    const std::string scene = argc > 2 ? argv[2] : "clustered";
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    std::mt19937 rng(6);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::normal_distribution<float> N(0.f, 1.f);
    std::vector<float4> pos(n);
    const int ncl = scene == "tight" ? 4 : 64;
    const float sigma = scene == "tight" ? 0.08f : 0.25f, frac = scene == "tight" ? 0.9f : 0.7f;
    std::vector<float> cc(ncl * 3);
    for (auto& v : cc) v = 2.5f * U(rng);
    for (uint32_t i = 0; i < n; ++i) {
        float x, y, z;
        if (scene == "equal") { x = 0.5f; y = -0.25f; z = 1.0f; }
        else if (scene != "uniform" && (rng() & 0xFFFF) < frac * 65536.f) {
            const int c = rng() % ncl;
            x = cc[c * 3] + sigma * N(rng); y = cc[c * 3 + 1] + sigma * N(rng); z = cc[c * 3 + 2] + sigma * N(rng);
        } else { x = 3.f * U(rng); y = 3.f * U(rng); z = 3.f * U(rng); }
        pos[i] = make_float4(x, y, z, 1.f);
    }
    if (scene == "equal") { pos[0] = make_float4(3, 3, -3, 1); pos[1] = make_float4(0, 0, 3, 1); }   // (a key range wide enough for the plan)
    const bool inside = argc > 5 && atoi(argv[5]);
    const bool mid = argc > 6 && atoi(argv[6]);          // launch k2_local's third instance (buckets of 9217 .. 16384 pairs in LDS)
    const uint32_t cap_lds = mid ? GSX2_CAP2 : GSX2_CAP;
    const float cx = inside ? 0.4f : 0.f, cy = inside ? -0.3f : 0.f, cz = inside ? 0.9f : 5.f;
    std::vector<uint32_t> hkey(n);
    float dmin = 1e30f, dmax = 0.f;
    for (uint32_t i = 0; i < n; ++i) {
        const float dx = cx - pos[i].x, dy = cy - pos[i].y, dz = cz - pos[i].z;
        const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
        hkey[i] = host_key(d);
        dmin = std::min(dmin, d); dmax = std::max(dmax, d);
    }
    const uint32_t key_lo = host_key(dmax * 1.00001f), key_span = host_key(inside ? 0.f : dmin * 0.99999f) - key_lo;   // (inside the box the host can only promise distance 0)
    const Gsx2Plan plan = gsx2_plan(key_span, n, n, argc > 4 ? (uint32_t)atoi(argv[4]) : 0u);
    printf("n=%u scene=%s d in [%.3f, %.3f] key_span=0x%x plan ok=%d r0=%u nd=%u idx_bits=%u ncols=%u\n", n, scene.c_str(), dmin, dmax, key_span, plan.ok, plan.map.r0,
           plan.nd, plan.idx_bits, plan.ncols);
    if (!plan.ok) return 2;
    hipStream_t st; CK(hipStreamCreate(&st));
    float4* d_pos; Gsx2Pos* d_pos3; uint32_t *d_keys, *d_m, *d_pre, *d_tot, *d_base, *d_res, *d_keys_out; uint16_t* d_cnt; uint2 *d_pairs, *d_scratch;
    GsxDevFlags* d_flags; Gsx2Stats* d_stats;
    CK(hipMalloc(&d_pos, (size_t)n * 16)); CK(hipMalloc(&d_keys, (size_t)n * 4)); CK(hipMalloc(&d_m, 4));
    CK(hipMalloc(&d_cnt, (size_t)GSX2_MAX_COLS * GSX2_MAX_D * 2)); CK(hipMalloc(&d_pre, (size_t)GSX2_MAX_COLS * GSX2_MAX_D * 4));
    CK(hipMalloc(&d_tot, GSX2_MAX_D * 4)); CK(hipMalloc(&d_base, (GSX2_MAX_D + 64) * 4)); CK(hipMalloc(&d_res, (size_t)n * 4)); CK(hipMalloc(&d_keys_out, (size_t)n * 4));
    CK(hipMalloc(&d_pairs, (size_t)n * 8)); CK(hipMalloc(&d_scratch, (size_t)n * 8));
    CK(hipMalloc(&d_flags, sizeof(GsxDevFlags))); CK(hipMalloc(&d_stats, sizeof(Gsx2Stats)));
    CK(hipMemset(d_flags, 0, sizeof(GsxDevFlags))); CK(hipMemset(d_stats, 0, sizeof(Gsx2Stats)));
    CK(hipMemcpy(d_pos, pos.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    {   // the library's key kernel streams packed 12-byte positions
        std::vector<float> p3((size_t)n * 3);
        for (size_t i = 0; i < (size_t)n; ++i) { p3[3 * i] = pos[i].x; p3[3 * i + 1] = pos[i].y; p3[3 * i + 2] = pos[i].z; }
        CK(hipMalloc(&d_pos3, (size_t)n * 12)); CK(hipMemcpy(d_pos3, p3.data(), (size_t)n * 12, hipMemcpyHostToDevice));
    } CK(hipMemcpy(d_m, &n, 4, hipMemcpyHostToDevice));
    hipEvent_t ev[7]; for (int k = 0; k < 7; ++k) CK(hipEventCreate(&ev[k]));
    std::vector<float> t[7];
    for (int rep = 0; rep < reps + 3; ++rep) {
        CK(hipEventRecord(ev[0], st));
        hipLaunchKernelGGL(k2_keygen_hist<true>, dim3(plan.ncols), dim3(GSX2_NT), 0, st, d_pos3, nullptr, d_m, cx, cy, cz, key_lo, key_span, plan.map, plan.nd,
                           d_keys, d_cnt, d_flags, nullptr, d_stats);
        CK(hipEventRecord(ev[1], st));
        hipLaunchKernelGGL(k2_scan, dim3(plan.nd / 16), dim3(512), 0, st, d_cnt, d_pre, d_tot, plan.nd, plan.ncols);
        CK(hipEventRecord(ev[2], st));
        hipLaunchKernelGGL(k2_partition<true>, dim3(plan.ncols), dim3(GSX2_PNT), 0, st, d_keys, nullptr, d_m, plan.map, plan.nd, d_cnt, d_pre, d_tot, d_base, d_pairs, d_stats, d_flags, cap_lds);
        CK(hipEventRecord(ev[3], st));
        hipLaunchKernelGGL((k2_local<GSX2_IPT>), dim3(plan.nd), dim3(GSX2_NT), 0, st, d_pairs, d_scratch, d_base, plan.nd, plan.map, plan.idx_bits, d_res, cap_lds);
        if (mid) hipLaunchKernelGGL((k2_local<GSX2_IPT2>), dim3(plan.nd), dim3(GSX2_NT), 0, st, d_pairs, d_scratch, d_base, plan.nd, plan.map, plan.idx_bits, d_res, cap_lds);
        CK(hipEventRecord(ev[6], st));
        CK(hipEventRecord(ev[4], st));
        hipLaunchKernelGGL((k2_local<8>), dim3(plan.nd), dim3(GSX2_NT), 0, st, d_pairs, d_scratch, d_base, plan.nd, plan.map, plan.idx_bits, d_res, cap_lds);
        CK(hipEventRecord(ev[5], st));
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        if (rep >= 3) {
            for (int k = 0; k < 5; ++k) { float ms; CK(hipEventElapsedTime(&ms, ev[k], ev[k + 1])); t[k].push_back(ms); }
            float ms; CK(hipEventElapsedTime(&ms, ev[0], ev[5])); t[5].push_back(ms); CK(hipEventElapsedTime(&ms, ev[3], ev[6])); t[6].push_back(ms);
        }
    }
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2] * 1e3f; };
    Gsx2Stats hs; CK(hipMemcpy(&hs, d_stats, sizeof hs, hipMemcpyDeviceToHost));
    GsxDevFlags hf; CK(hipMemcpy(&hf, d_flags, sizeof hf, hipMemcpyDeviceToHost));
    printf("keygen+hist %.1f us  scan %.1f  partition %.1f  local: large %.1f (+%.1f) + small %.1f  | whole %.1f us = %.2f G splats/s (92 B/splat: %.2f of 8 TB/s)\n", med(t[0]), med(t[1]),
           med(t[2]), med(t[6]), med(t[3]) - med(t[6]), med(t[4]), med(t[5]), n / med(t[5]) * 1e-3, 92.0 * n / (med(t[5]) * 1e-6) / 8e12);
    printf("stats: big_buckets=%u big_keys=%u max_bucket=%u key_bound_flag=%u\n", hs.big_buckets, hs.big_keys, hs.max_bucket, hf.key_bound);
    std::vector<uint32_t> res(n);
    CK(hipMemcpy(res.data(), d_res, (size_t)n * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> ref(n);
    for (uint32_t i = 0; i < n; ++i) ref[i] = i;
    std::sort(ref.begin(), ref.end(), [&](uint32_t a, uint32_t b) { return hkey[a] != hkey[b] ? hkey[a] < hkey[b] : a < b; });
    size_t bad = 0, first = n;
    for (uint32_t i = 0; i < n; ++i) if (res[i] != ref[i]) { if (!bad) first = i; ++bad; }
    printf("check: %zu mismatches (first at %zu)\n", bad, first);
    return bad ? 1 : 0;
}
