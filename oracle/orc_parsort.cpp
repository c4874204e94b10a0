// orc_parsort.cpp — TEST/BENCH INFRASTRUCTURE ONLY (CPU oracle), never linked into the product.
//
// Multi-threaded stand-in for the reference's PARALLEL build of SplatMesh::sortSplats
// (SplatMesh.h:97-103: tbb::task_arena arena(8); arena.execute([&]{ std::sort(std::execution::par, ...) })).
// This image has libtbb.so but no TBB headers, so std::execution::par cannot be compiled; the closest
// thing the toolchain offers is libstdc++'s parallel mode (OpenMP): __gnu_parallel::sort with the
// reference's comparator on the reference's thread count.  Same ordering contract (distance descending,
// ties unspecified); a different parallel algorithm (multiway merge sort instead of TBB's parallel
// quicksort), so bench.py labels it "libstdc++ parallel mode, not TBB".
#include <parallel/algorithm>

#include <chrono>
#include <cstdint>
#include <vector>

#include <omp.h>

extern "C" double orc_sort_reference_port_parallel(const float* splat20, uint64_t n, uint32_t* idx, uint64_t m,
                                                   const float cam[3], int threads) {
    if (threads < 1) threads = 8;                                      // SplatMesh.h:98
    auto start = std::chrono::high_resolution_clock::now();          // SplatMesh.h:88
    std::vector<float> distances(n);                                  // SplatMesh.h:90
    for (uint64_t k = 0; k < m; ++k) {                                // SplatMesh.h:91-94 (serial in the reference too)
        const uint32_t i = idx[k];
        const float* p = splat20 + static_cast<uint64_t>(i) * 20 + 12;   // transform[3].xyz (Splat.h:19-29)
        const float dx = cam[0] - p[0], dy = cam[1] - p[1], dz = cam[2] - p[2];
        distances[i] = __builtin_sqrtf((dx * dx + dy * dy) + dz * dz);   // glm/detail/func_geometric.inl:8-23,47-55
    }
    omp_set_num_threads(threads);
    __gnu_parallel::sort(idx, idx + m, [&](uint32_t a, uint32_t b) {   // SplatMesh.h:100-102
        return distances[a] > distances[b];
    }, __gnu_parallel::default_parallel_tag(static_cast<unsigned>(threads)));
    auto end = std::chrono::high_resolution_clock::now();             // SplatMesh.h:111
    return std::chrono::duration<double>(end - start).count();
}
