// timing breakdown of the host GridHC build: total, and the pair-error function on its own
#include <chrono>
#include <cstdio>
#include <random>
#include "../../gaussian-splatting_amd/host/gsx_lod_host.hpp"
using namespace gsx;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t n = argc > 1 ? std::atol(argv[1]) : 1000000;
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f), S(0.005f, 0.03f), C(0.0f, 1.0f);
    SplatVector sp(n);
    for (auto& s : sp) {
        SplatSplit ss;
        ss.position = vec3(U(rng) * 3, U(rng) * 3, U(rng) * 3);
        ss.scale = vec3(S(rng), S(rng), S(rng));
        ss.color = vec4(C(rng), C(rng), C(rng), 0.2f + 0.8f * C(rng));
        ss.rotation = normalize(vec4(U(rng), U(rng), U(rng), U(rng)));
        s = split_to_splat(ss);
    }
    double t = now();
    GridHC g;
    g.build(sp);
    std::printf("GridHC build of %zu splats: %.3f s, %zu nodes, %zu cells\n", n, now() - t, g.splats.size(), g.cell_count());
    // pair error alone: HC of 2000 splats, single thread
    SplatVector one(sp.begin(), sp.begin() + 1500);
    t = now();
    HC h; h.build(one, false, 1);
    std::printf("HC build of 1500 (1 thread): %.3f s  (%.1f ns per initial pair)\n", now() - t, (now() - t) / (1500.0 * 1499 / 2) * 1e9);
    t = now();
    float acc = 0;
    const size_t reps = 2000000;
    for (size_t i = 0; i < reps; ++i) {
        const Splat& a = sp[i % 1000], &b = sp[(i * 7 + 1) % 1000 + 1000];
        float wa = a.weight(), wb = b.weight(); const float tot = wa + wb; wa /= tot; wb /= tot;
        const Splat c = merge_splats(a, b, wa, wb);
        acc += wa * splat_divergence(a, c) + wb * splat_divergence(b, c);
    }
    std::printf("pair error: %.1f ns each (%g)\n", (now() - t) / reps * 1e9, acc);
    return 0;
}
