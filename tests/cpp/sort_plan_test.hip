// tests/cpp/sort_plan_test.hip - host-only check of the two-pass depth sort's plans and bucket maps (gsx_sort2.hpp): for every
// plan shape (narrow / wide key range x selection size) the bucket of a key is monotone in the key and below the table size,
// and every bucket is exactly the aligned block of keys its low bits span (what k2_local relies on).  Launches nothing: runs
// without a GPU (tests/test_host.py).
#include "gsx_sort2.hpp"
#include <cstdio>
#include <random>
int main() {
    std::mt19937 rng(1);
    int bad = 0;
    const uint32_t spans[] = {0x13c3271u, 0x1ffffffu, 0x2000000u, 0x3b67902u, 0x41340ef6u, 0x7FFFFFFFu, 0x1283967u};
    for (uint32_t span : spans) for (uint32_t m : {6000000u, 13000000u, 20000000u, 24000000u}) {
        const Gsx2Plan p = gsx2_plan(span, m, m);
        printf("span %#x m %u: ok %d r0 %u rc %u nf %u base %u flen %#x nd %u\n", span, m, (int)p.ok, p.map.r0, p.map.rc, p.map.nf, p.map.base, p.map.flen, p.nd);
        if (!p.ok) continue;
        // monotone, below nd, and every key of a bucket shares the bits above bucket_bits
        uint32_t prev = 0; 
        for (uint64_t rel = 0; rel <= span; rel += 1 + (rng() % 4099)) {
            const uint32_t d = gsx2_bucket((uint32_t)rel, p.map);
            if (d < prev || d >= p.nd) { if (bad++ < 5) printf("  not monotone / out of range at %#llx: %u after %u (nd %u)\n", (unsigned long long)rel, d, prev, p.nd); }
            prev = d;
            const uint32_t r = gsx2_bucket_bits(d, p.map);
            const uint32_t lo = (uint32_t)rel & ~((r >= 32 ? 0xFFFFFFFFu : (1u << r) - 1u));
            const uint64_t hi = (uint64_t)lo + (1ull << r) - 1;
            if (gsx2_bucket(lo, p.map) != d || (hi <= span && gsx2_bucket((uint32_t)hi, p.map) != d)) { if (bad++ < 5) printf("  bucket %u (bits %u) does not hold its whole aligned block at %#llx\n", d, r, (unsigned long long)rel); }
            if (lo > 0 && gsx2_bucket(lo - 1, p.map) == d) { if (bad++ < 5) printf("  bucket %u wider than its bits %u at %#llx\n", d, r, (unsigned long long)rel); }
            if (hi < span && gsx2_bucket((uint32_t)hi + 1, p.map) == d) { if (bad++ < 5) printf("  bucket %u wider (above) than its bits %u at %#llx\n", d, r, (unsigned long long)rel); }
        }
    }
    printf("bad %d\n", bad);
    return bad != 0;
}
