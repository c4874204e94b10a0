// gsx_lod_host.hpp — host-side LOD structures behind the reference's class names
// (Octree, HC, GridHC, BB), built on FLAT arrays:
//
//   Octree  one id array per breadth-first level in which every node owns a contiguous, ascending range (stable
//           8-way counting partitions from level to level, single-child chains folded as they appear),
//           nodes as a structure of arrays in the order Octree::generate emits splats; partitioning
//           and the per-node moment-matched merge run on all host cores.
//   HC      node arrays (children, error, depth) indexed by splat id, the active set as an
//           index-linked list, the pair queue as a heap of 12-byte (error, a, b) candidates whose
//           merged splat is recomputed when a candidate wins instead of being stored per pair.
//   GridHC  cells as one counting sort of splat ids; non-empty cells are clustered in parallel.
//
// What is kept from the reference is the OUTPUT, bit for bit (tests/test_host.py compares every
// array with the compiled reference): node order, node boxes, merged splats, errors, cut lists.
// That pins the floating-point expressions (cited per function: Splat.h:98-195, BB.hpp:15-106,
// Octree.cpp:9-31, HC.hpp:64-94, GridHC.cpp:14-18, glm's operation order) and the order in which
// candidates enter and leave the pair heap (HC.cpp:25-35,41-90), not the data structures.
// Host-only; compile with -ffp-contract=off.  Nothing is printed.
#pragma once
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <thread>
#include <vector>

#include "gsx_scene.hpp"
#include "gsx_splat.hpp"

namespace gsx {

// ---- a minimal fork-join helper: fn(i) for i in [0, n), dynamic chunks, deterministic results as long as
// ---- fn(i) only writes slots that belong to i ----------------------------------------------------------
namespace lod_detail {
inline unsigned host_threads() {
    unsigned t = std::thread::hardware_concurrency();
    if (t == 0) t = 4;
    return t > 16 ? 16 : t;
}
// fn(i, t): t < threads identifies the executing thread (for per-thread scratch)
template <typename Fn>
inline void parallel_for_t(size_t n, size_t grain, unsigned threads, Fn fn) {
    if (threads <= 1 || n <= grain) {
        for (size_t i = 0; i < n; ++i) fn(i, 0u);
        return;
    }
    std::atomic<size_t> next{0};
    auto work = [&](unsigned t) {
        for (;;) {
            const size_t b = next.fetch_add(grain);
            if (b >= n) return;
            const size_t e = b + grain < n ? b + grain : n;
            for (size_t i = b; i < e; ++i) fn(i, t);
        }
    };
    std::vector<std::thread> pool;
    const unsigned extra = static_cast<unsigned>(std::min<size_t>(threads - 1, (n + grain - 1) / grain));
    for (unsigned t = 0; t < extra; ++t) pool.emplace_back(work, t + 1);
    work(0u);
    for (auto& th : pool) th.join();
}
template <typename Fn>
inline void parallel_for(size_t n, size_t grain, unsigned threads, Fn fn) {
    parallel_for_t(n, grain, threads, [&](size_t i, unsigned) { fn(i); });
}
// std::vector::resize constructs new elements on one thread, and on fresh memory most of that time is page faults.
// Growing a vector inside its reserved capacity, the new part is touched by all threads first.
template <typename T>
inline void grow_prefaulted(std::vector<T>& v, size_t total, unsigned threads) {
    if (total <= v.size()) { v.resize(total); return; }
    if (total <= v.capacity()) {
        unsigned char* raw = reinterpret_cast<unsigned char*>(v.data() + v.size());
        const size_t bytes = (total - v.size()) * sizeof(T), piece = size_t(1) << 20;
        parallel_for((bytes + piece - 1) / piece, 1, threads, [&](size_t k) {
            std::memset(raw + k * piece, 0, std::min(piece, bytes - k * piece));
        });
    }
    v.resize(total);
}
// uninitialised storage for plain records that every thread fills for itself
template <typename T>
struct RawArray {
    T* p = nullptr;
    explicit RawArray(size_t n) : p(static_cast<T*>(std::malloc((n ? n : 1) * sizeof(T)))) { if (!p) throw std::bad_alloc(); }
    ~RawArray() { std::free(p); }
    RawArray(const RawArray&) = delete;
    RawArray& operator=(const RawArray&) = delete;
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    T* data() { return p; }
    void swap(RawArray& o) { std::swap(p, o.p); }
};
}  // namespace lod_detail

// ---- moments of a splat record: cov = transform's 3x3 block (column-major), pos = column 3 --------------
// The merge / divergence expressions below are scalar restatements of Splat.h:98-195 in glm's operation order
// (type_mat3x3.inl operator+ / operator*, func_matrix.inl outerProduct / determinant / inverse): the results
// must equal the reference's to the last bit, so the order of every +, *, / is the reference's.
inline float splat_weight(const SplatSplit& s) { return s.color.w * (s.scale.x * s.scale.y * s.scale.z); }   // Splat.h:98-102

namespace lod_detail {
struct Cov3 { float m[3][3]; };   // m[column][row]
inline Cov3 cov_of(const Splat& s) {
    Cov3 c;
    for (int j = 0; j < 3; ++j)
        for (int r = 0; r < 3; ++r) c.m[j][r] = s.transform[j][r];
    return c;
}
inline void store(Splat& out, const Cov3& c, vec3 center, vec4 color) {
    out.transform = mat4(0.0f);
    for (int j = 0; j < 3; ++j)
        for (int r = 0; r < 3; ++r) out.transform[j][r] = c.m[j][r];
    out.transform[3] = vec4(center, 1.0f);
    out.color = color;
}
// func_matrix.inl:211-219
inline float det3(const Cov3& a) {
    return +a.m[0][0] * (a.m[1][1] * a.m[2][2] - a.m[2][1] * a.m[1][2]) - a.m[1][0] * (a.m[0][1] * a.m[2][2] - a.m[2][1] * a.m[0][2]) +
           a.m[2][0] * (a.m[0][1] * a.m[1][2] - a.m[1][1] * a.m[0][2]);
}
// func_matrix.inl:269-291
inline Cov3 inv3_det(const Cov3& a, float det);
inline Cov3 inv3(const Cov3& a) { return inv3_det(a, det3(a)); }
// the same with the determinant already known (HC keeps one per node)
inline Cov3 inv3_det(const Cov3& a, float det) {
    const float ood = 1.0f / det;
    Cov3 r;
    r.m[0][0] = +(a.m[1][1] * a.m[2][2] - a.m[2][1] * a.m[1][2]) * ood;
    r.m[1][0] = -(a.m[1][0] * a.m[2][2] - a.m[2][0] * a.m[1][2]) * ood;
    r.m[2][0] = +(a.m[1][0] * a.m[2][1] - a.m[2][0] * a.m[1][1]) * ood;
    r.m[0][1] = -(a.m[0][1] * a.m[2][2] - a.m[2][1] * a.m[0][2]) * ood;
    r.m[1][1] = +(a.m[0][0] * a.m[2][2] - a.m[2][0] * a.m[0][2]) * ood;
    r.m[2][1] = -(a.m[0][0] * a.m[2][1] - a.m[2][0] * a.m[0][1]) * ood;
    r.m[0][2] = +(a.m[0][1] * a.m[1][2] - a.m[1][1] * a.m[0][2]) * ood;
    r.m[1][2] = -(a.m[0][0] * a.m[1][2] - a.m[1][0] * a.m[0][2]) * ood;
    r.m[2][2] = +(a.m[0][0] * a.m[1][1] - a.m[1][0] * a.m[0][1]) * ood;
    return r;
}
// one-directional term of Splat.h:169-183: 0.5 (tr(B^-1 A) + d^T B^-1 d - 3 + ln(det B / det A)), d = a - b
inline float kl_term_pre(const Cov3& A, vec3 pa, float detA, const Cov3& Bi, float detB, vec3 pb);
inline float kl_term(const Cov3& A, vec3 pa, const Cov3& B, vec3 pb) { return kl_term_pre(A, pa, det3(A), inv3(B), det3(B), pb); }
// the same with B's inverse and both determinants already known: pure functions of A and B, so caching them changes no bit
inline float kl_term_pre(const Cov3& A, vec3 pa, float detA, const Cov3& Bi, float detB, vec3 pb) {
    const vec3 d = pa - pb;
    float diag[3];                                       // diagonal of Bi * A (type_mat3x3.inl:486-519)
    for (int j = 0; j < 3; ++j) diag[j] = Bi.m[0][j] * A.m[j][0] + Bi.m[1][j] * A.m[j][1] + Bi.m[2][j] * A.m[j][2];
    const float trace_term = diag[0] + diag[1] + diag[2];
    const vec3 Bd(Bi.m[0][0] * d.x + Bi.m[1][0] * d.y + Bi.m[2][0] * d.z, Bi.m[0][1] * d.x + Bi.m[1][1] * d.y + Bi.m[2][1] * d.z,
                  Bi.m[0][2] * d.x + Bi.m[1][2] * d.y + Bi.m[2][2] * d.z);
    const float det_term = std::log(detB / detA);
    return 0.5f * (trace_term + dot(d, Bd) - 3.0f + det_term);
}
}  // namespace lod_detail

// Splat.h:185-195: symmetric KL + colour distance + alpha distance
inline float splat_divergence(const Splat& a, const Splat& b) {
    using namespace lod_detail;
    const Cov3 A = cov_of(a), B = cov_of(b);
    const vec3 pa = xyz(a.transform[3]), pb = xyz(b.transform[3]);
    const float kl = 0.5f * (kl_term(A, pa, B, pb) + kl_term(B, pb, A, pa));
    const vec4 cd = a.color - b.color;
    return kl + length(xyz(cd)) + std::fabs(a.color.w - b.color.w);
}

// Splat.h:146-163: moment-matched merge of two splats with normalized weights
inline Splat merge_splats(const Splat& a, const Splat& b, float w_a, float w_b) {
    using namespace lod_detail;
    const vec3 pa = xyz(a.transform[3]), pb = xyz(b.transform[3]);
    const vec3 center = pa * w_a + pb * w_b;
    const vec3 da = pa - center, db = pb - center;
    Cov3 c;
    for (int j = 0; j < 3; ++j)
        for (int r = 0; r < 3; ++r)
            c.m[j][r] = w_a * (a.transform[j][r] + da[r] * da[j]) + w_b * (b.transform[j][r] + db[r] * db[j]);
    Splat out;
    store(out, c, center, a.color * w_a + b.color * w_b);
    return out;
}

// Splat.h:104-144 over PRE-CONVERTED records: sp[id] = split_to_splat(raw[id]) and wt[id] = splat_weight(raw[id]) are pure
// functions of one input record, so computing them once per splat instead of once per (node, splat) changes no bit.
// ids[i * stride] is the i-th id (the octree keeps its ids inside 16-byte records); `w` is scratch for n normalized weights.
inline Splat merge_range(const Splat* sp, const float* wt, const uint32_t* ids, size_t n, size_t stride, float* w) {
    using namespace lod_detail;
    float total = 0.0f;
    for (size_t i = 0; i < n; ++i) total += wt[ids[i * stride]];
    for (size_t i = 0; i < n; ++i) w[i] = wt[ids[i * stride]] / total;
    vec4 color(0.0f);
    for (size_t i = 0; i < n; ++i) color = color + sp[ids[i * stride]].color * w[i];
    vec3 center(0.0f);
    for (size_t i = 0; i < n; ++i) center = center + xyz(sp[ids[i * stride]].transform[3]) * w[i];
    Cov3 c;
    for (int j = 0; j < 3; ++j)
        for (int r = 0; r < 3; ++r) c.m[j][r] = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        const Splat& s = sp[ids[i * stride]];
        const vec3 d = xyz(s.transform[3]) - center;
        for (int j = 0; j < 3; ++j)
            for (int r = 0; r < 3; ++r) c.m[j][r] = c.m[j][r] + (s.transform[j][r] + d[r] * d[j]) * w[i];
    }
    Splat out;
    store(out, c, center, color);
    return out;
}
// The reference's signature (used by hosts that merge a handful of raw splats).
inline Splat merge(const SplatSplitVector& raw) {
    SplatVector sp(raw.size());
    std::vector<float> wt(raw.size());
    std::vector<uint32_t> ids(raw.size());
    for (size_t i = 0; i < raw.size(); ++i) { sp[i] = split_to_splat(raw[i]); wt[i] = splat_weight(raw[i]); ids[i] = static_cast<uint32_t>(i); }
    std::vector<float> w(raw.size());
    return merge_range(sp.data(), wt.data(), ids.data(), raw.size(), 1, w.data());
}

// ---- BB.hpp: an axis-aligned box kept as (centre, edge lengths) — what the reference's mat4 holds on its diagonal
// ---- and in column 3.  size() re-derives the edge from the stored value through glm::length, as BB.hpp:28-33 does. ----
class BB {
public:
    vec3 mid{0.0f}, edge{1.0f};
    static BB from_aabb(vec3 mn, vec3 mx) {               // BB.hpp:15-23
        BB b;
        b.mid = (mn + mx) / 2.0f;
        b.edge = vec3(mx.x - mn.x, mx.y - mn.y, mx.z - mn.z);
        return b;
    }
    vec3 center() const { return mid; }
    vec3 size() const { return {length(vec3(edge.x, 0.0f, 0.0f)), length(vec3(0.0f, edge.y, 0.0f)), length(vec3(0.0f, 0.0f, edge.z))}; }
    vec3 min() const { return center() - size() / 2.0f; }
    vec3 max() const { return center() + size() / 2.0f; }
    // Child octant `oct` (bit 0: +x, bit 1: +y, bit 2: +z) of this box (Octree.cpp:83-101).
    BB octant(int oct) const {
        vec3 mn = min(), mx = max();
        const vec3 c = center();
        for (int k = 0; k < 3; ++k) {
            if (oct & (1 << k)) mn[k] = c[k];
            else mx[k] = c[k];
        }
        return from_aabb(mn, mx);
    }
    // NDC bounding-rectangle area of the 8 projected corners (BB.hpp:41-77); viewproj = proj * view.
    float screen_area(const mat4& viewproj) const {
        const vec3 h = size() / 2.0f, c = center();
        static const int sx[8] = {-1, 1, 1, -1, -1, 1, 1, -1}, sy[8] = {-1, -1, 1, 1, -1, -1, 1, 1}, sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
        float lo_x = 10000000.0f, lo_y = 10000000.0f, hi_x = -10000000.0f, hi_y = -10000000.0f;
        for (int i = 0; i < 8; ++i) {
            const vec3 corner = c + vec3(sx[i] < 0 ? -h.x : h.x, sy[i] < 0 ? -h.y : h.y, sz[i] < 0 ? -h.z : h.z);
            const vec4 clip = viewproj * vec4(corner, 1.0f);
            if (clip.w == 0.0f) continue;
            const float nx = clip.x / clip.w, ny = clip.y / clip.w;
            lo_x = (nx < lo_x) ? nx : lo_x;               // glm::min / glm::max as ternaries
            lo_y = (ny < lo_y) ? ny : lo_y;
            hi_x = (hi_x < nx) ? nx : hi_x;
            hi_y = (hi_y < ny) ? ny : hi_y;
        }
        return (hi_x - lo_x) * (hi_y - lo_y);
    }
    float screen_area(Camera::Ptr camera) const { return screen_area(camera->getProjectionMatrix() * camera->getViewMatrix()); }
    // Bounds of a point set; both corners start at the origin (Octree.cpp:9-31, BB.hpp:79-106); `cube` widens the box
    // to its longest edge around the centre.
    template <typename PosAt>
    static BB around(size_t n, PosAt pos_at, bool cube) {
        vec3 mn(0.0f), mx(0.0f);
        for (size_t i = 0; i < n; ++i) {
            const vec3 p = pos_at(i);
            for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], p[k]); mx[k] = std::max(mx[k], p[k]); }
        }
        if (!cube) return from_aabb(mn, mx);
        const vec3 c = (mx + mn) / 2.0f;
        const float longest = std::max({mx.x - mn.x, mx.y - mn.y, mx.z - mn.z});
        return from_aabb(c - vec3(longest / 2.0f), c + vec3(longest / 2.0f));
    }
    static BB from_splats(const SplatVector& splats, bool expand = true) {
        return around(splats.size(), [&](size_t i) { return xyz(splats[i].transform[3]); }, expand);
    }
};

// Flattened trees in the layout include/gsx.h takes.
struct OctreeFlat {
    std::vector<float> center3, half3;
    std::vector<uint32_t> parent, first_child, n_children;
    std::vector<uint8_t> is_leaf;
};
struct HcFlat {
    std::vector<uint32_t> parent, child2;   // child2: UINT32_MAX pair = leaf
    std::vector<uint8_t> is_leaf;
    std::vector<float> error, weight;
    std::vector<uint32_t> roots, cell_root_start;  // roots in list order; GridHC: per non-empty cell
};

// ---- Octree (Octree.hpp:57-173, Octree.cpp:33-216) ------------------------------------------------------
class Octree {
public:
    SplatVector splats;                 // one merged splat per node, node i <-> splats[i]
    uint32_t max_depth{10};
    uint32_t max_splats_per_node{1};

    size_t node_count() const { return box.size(); }

    // Octree.cpp:33-147.  The reference splits breadth first into pointer nodes and then folds every node that has a
    // single child into that child; here a node whose splats all fall into one octant simply shrinks to that octant and
    // is examined again, so only the folded tree is ever materialised.  Nodes come out in breadth-first order of the
    // folded tree with children in octant order — the order generate() numbers the splats in.
    void build(SplatSplitVector raw_in) {
        raw = std::move(raw_in);
        splats.clear();
        box.clear(); depth.clear(); first.clear(); last.clear(); child0.clear(); n_child.clear(); up.clear();
        const size_t n = raw.size();
        const unsigned threads = lod_detail::host_threads();
        // two id arrays alternate between levels (the next level is a stable partition of the current one)
        lod_detail::RawArray<Item> ping(n), pong(n);
        lod_detail::RawArray<Splat> sp(n);                  // split_to_splat / splat_weight of every input record, once
        lod_detail::RawArray<float> wt(n);
        lod_detail::parallel_for(n, 1 << 14, threads, [&](size_t i) {
            ping[i] = Item{static_cast<uint32_t>(i), raw[i].position};
            sp[i] = split_to_splat(raw[i]);
            wt[i] = splat_weight(raw[i]);
        });
        std::vector<std::vector<float>> scratch(threads);   // per thread: normalized weights of the node being merged
        reserve_nodes(2 * n + 16);                           // (a folded tree has at most 2n - 1 nodes; reserving touches no memory)
        merged.clear();
        merged.reserve(2 * n + 16);
        grow_nodes(1, 1);
        set_node(0, BB::around(n, [&](size_t i) { return raw[i].position; }, true), 0u, 0u, static_cast<uint32_t>(n), 0xFFFFFFFFu);
        // The root's merge is one sequential sum over all n splats (a fifth of the whole build on 8 cores) and nothing else
        // depends on it: it runs beside the level loop on a thread of its own (its ids are simply 0 .. n-1).
        Splat root_merged;
        std::thread root_job([&]() {
            std::vector<uint32_t> all(n);
            std::iota(all.begin(), all.end(), 0u);
            std::vector<float> w(n);
            root_merged = merge_range(sp.data(), wt.data(), all.data(), n, 1, w.data());
        });
        struct JoinGuard { std::thread& t; ~JoinGuard() { if (t.joinable()) t.join(); } } root_guard{root_job};
        struct Split { BB box; uint32_t depth; uint32_t count[8]; uint32_t out, kids; bool leaf; };
        for (size_t level_begin = 0; level_begin < box.size();) {
            const size_t level_end = box.size();
            const Item* cur = ping.data();
            std::vector<Split> split(level_end - level_begin);
            // (1) per node, concurrently: fold chain links, decide leaf / split, count the octants
            lod_detail::parallel_for(level_end - level_begin, 16, threads, [&](size_t k) {
                const size_t node = level_begin + k;
                Split& s = split[k];
                s.box = box[node];
                s.depth = depth[node];
                const uint32_t b = first[node], e = last[node];
                for (;;) {
                    s.leaf = s.depth >= max_depth || (e - b) <= max_splats_per_node;
                    if (s.leaf) return;
                    const vec3 c = s.box.center();
                    for (int o = 0; o < 8; ++o) s.count[o] = 0;
                    for (uint32_t i = b; i < e; ++i) ++s.count[octant_of(cur[i].pos, c)];
                    int used = 0, only = 0;
                    for (int o = 0; o < 8; ++o)
                        if (s.count[o]) { ++used; only = o; }
                    if (used != 1) return;
                    s.box = s.box.octant(only);           // a chain link: the node becomes its only child (Octree.cpp:124-135)
                    s.depth += 1;
                }
            });
            // (2) in node order: where every split node's children start in the node arrays and in the next level's id
            //     array; then the children themselves are filled in concurrently
            // (chunked two-pass prefix sum: per-chunk totals concurrently, chunk offsets serially, per-node offsets concurrently)
            constexpr size_t CHUNK = 4096;
            const size_t n_chunks = (split.size() + CHUNK - 1) / CHUNK;
            std::vector<uint32_t> chunk_ids(n_chunks + 1, 0u), chunk_kids(n_chunks + 1, 0u);
            lod_detail::parallel_for(n_chunks, 1, threads, [&](size_t ch) {
                uint32_t sum_ids = 0, sum_kids = 0;
                for (size_t k = ch * CHUNK; k < std::min(split.size(), (ch + 1) * CHUNK); ++k) {
                    const Split& s = split[k];
                    if (s.leaf) continue;
                    for (int o = 0; o < 8; ++o)
                        if (s.count[o]) { sum_ids += s.count[o]; ++sum_kids; }
                }
                chunk_ids[ch + 1] = sum_ids;
                chunk_kids[ch + 1] = sum_kids;
            });
            chunk_kids[0] = static_cast<uint32_t>(box.size());
            for (size_t ch = 0; ch < n_chunks; ++ch) { chunk_ids[ch + 1] += chunk_ids[ch]; chunk_kids[ch + 1] += chunk_kids[ch]; }
            lod_detail::parallel_for(n_chunks, 1, threads, [&](size_t ch) {
                uint32_t at_ids = chunk_ids[ch], at_kids = chunk_kids[ch];
                for (size_t k = ch * CHUNK; k < std::min(split.size(), (ch + 1) * CHUNK); ++k) {
                    Split& s = split[k];
                    s.out = at_ids;
                    s.kids = at_kids;
                    if (s.leaf) continue;
                    for (int o = 0; o < 8; ++o)
                        if (s.count[o]) { at_ids += s.count[o]; ++at_kids; }
                }
            });
            const uint32_t next_nodes = chunk_kids[n_chunks];
            grow_nodes(next_nodes, threads);
            lod_detail::parallel_for(split.size(), 64, threads, [&](size_t k) {
                const size_t node = level_begin + k;
                const Split& s = split[k];
                box[node] = s.box;
                depth[node] = s.depth;
                if (s.leaf) return;
                child0[node] = s.kids;
                uint32_t kid = s.kids, at = s.out;
                for (int o = 0; o < 8; ++o) {
                    if (!s.count[o]) continue;
                    set_node(kid++, s.box.octant(o), s.depth + 1, at, at + s.count[o], static_cast<uint32_t>(node));
                    at += s.count[o];
                }
                n_child[node] = kid - s.kids;
            });
            // (3) concurrently: every node's merged splat (Splat.h:104-144 over its ids in ascending order, which is the order
            //     they have in the CURRENT level's array: a node's own slice is never reordered), and the stable 8-way
            //     partition of every split node's slice into its children's slices of the other array
            lod_detail::grow_prefaulted(merged, level_end, threads);
            Item* nxt = pong.data();
            const size_t merge_grain = std::max<size_t>(1, std::min<size_t>(64, split.size() / (static_cast<size_t>(threads) * 8)));
            lod_detail::parallel_for_t(split.size(), merge_grain, threads, [&](size_t k, unsigned thread) {
                const size_t node = level_begin + k;
                const uint32_t b = first[node], e = last[node];
                std::vector<float>& w = scratch[thread];
                if (w.size() < static_cast<size_t>(e - b)) w.resize(e - b);
                if (node != 0) merged[node] = merge_range(sp.data(), wt.data(), &cur[b].id, e - b, sizeof(Item) / sizeof(uint32_t), w.data());
                const Split& s = split[k];
                if (s.leaf) return;
                const vec3 c = s.box.center();
                uint32_t cursor[8], at = s.out;
                for (int o = 0; o < 8; ++o) { cursor[o] = at; at += s.count[o]; }
                for (uint32_t i = b; i < e; ++i) {
                    const Item it = cur[i];
                    nxt[cursor[octant_of(it.pos, c)]++] = it;
                }
            });
            ping.swap(pong);
            level_begin = level_end;
        }
        root_job.join();
        merged[0] = root_merged;
    }

    // Octree.cpp:149-187: the merged splat of every node, in node order.  build() already computed them (while each node's
    // ids were at hand in ascending order); this publishes them.
    void generate() {
        if (!merged.empty()) { splats = std::move(merged); merged.clear(); }
    }

    // Octree.cpp:189-216: breadth-first cut; a node is emitted (its splat id = its number) when it is a leaf or its
    // box projects to less than min_screen_area, else its children are visited.
    Indices get_indices(Camera::Ptr camera, float min_screen_area) const {
        Indices out;
        if (box.empty()) return out;
        const mat4 viewproj = camera->getProjectionMatrix() * camera->getViewMatrix();   // BB.hpp:60-61, the same for every node
        std::vector<uint32_t> visit{0u};
        for (size_t at = 0; at < visit.size(); ++at) {
            const uint32_t node = visit[at];
            const float area = box[node].screen_area(viewproj);
            if (n_child[node] == 0 || area < min_screen_area) { out.push_back(node); continue; }
            for (uint32_t c = 0; c < n_child[node]; ++c) visit.push_back(child0[node] + c);
        }
        return out;
    }

    // Octree.hpp:131-162: depth-first from the back of a stack; nodes deeper than `depth_limit` or leaves are emitted.
    Indices get_indices_depth(uint32_t depth_limit) const {
        Indices out;
        if (box.empty()) return out;
        std::vector<uint32_t> stack{0u};
        while (!stack.empty()) {
            const uint32_t node = stack.back();
            stack.pop_back();
            if (depth[node] > depth_limit || n_child[node] == 0) { out.push_back(node); continue; }
            for (uint32_t c = 0; c < n_child[node]; ++c) stack.push_back(child0[node] + c);
        }
        return out;
    }

    SplatVector* data() { return &splats; }

    OctreeFlat flatten() const {
        OctreeFlat f;
        const size_t nodes = box.size();
        f.center3.resize(nodes * 3); f.half3.resize(nodes * 3);
        f.parent = up; f.n_children.assign(n_child.begin(), n_child.end());
        f.first_child.resize(nodes); f.is_leaf.resize(nodes);
        uint32_t numbered = 1;                             // nodes numbered so far in breadth-first order: where node i's children start
        for (size_t i = 0; i < nodes; ++i) {               // (for a leaf: where they would start)
            const vec3 c = box[i].center(), h = box[i].size() / 2.0f;
            for (int k = 0; k < 3; ++k) { f.center3[3 * i + k] = c[k]; f.half3[3 * i + k] = h[k]; }
            f.is_leaf[i] = n_child[i] == 0 ? 1 : 0;
            f.first_child[i] = numbered;
            numbered += n_child[i];
        }
        return f;
    }

private:
    static int octant_of(vec3 p, vec3 c) { return (p.x > c.x ? 1 : 0) | (p.y > c.y ? 2 : 0) | (p.z > c.z ? 4 : 0); }   // Octree.cpp:70-73
    void reserve_nodes(size_t total) {
        box.reserve(total); depth.reserve(total); first.reserve(total); last.reserve(total);
        child0.reserve(total); n_child.reserve(total); up.reserve(total);
    }
    void grow_nodes(size_t total, unsigned threads) {
        using lod_detail::grow_prefaulted;
        grow_prefaulted(box, total, threads); grow_prefaulted(depth, total, threads); grow_prefaulted(first, total, threads);
        grow_prefaulted(last, total, threads); grow_prefaulted(child0, total, threads); grow_prefaulted(n_child, total, threads);
        grow_prefaulted(up, total, threads);
    }
    void set_node(size_t i, const BB& b, uint32_t d, uint32_t lo, uint32_t hi, uint32_t parent) {
        box[i] = b; depth[i] = d; first[i] = lo; last[i] = hi; child0[i] = 0u; n_child[i] = 0u; up[i] = parent;
    }
    SplatSplitVector raw;
    struct Item { uint32_t id; vec3 pos; };           // id + position, so that counting and partitioning stream
    SplatVector merged;                                // per node, computed by build()
    std::vector<BB> box;
    std::vector<uint32_t> depth, first, last, child0, n_child, up;
};

// ---- HC (HC.hpp:16-95, HC.cpp:4-168): greedy agglomerative clustering --------------------------------------
class HC {
public:
    struct Params { float max_error{std::numeric_limits<float>::infinity()}; };
    struct MetricWeights { float e{0.0f}, w{0.0f}, d{0.0f}; };
    static constexpr uint32_t NONE = 0xFFFFFFFFu;

    Params params;
    SplatVector splats;                                // leaves first (input order), then merged nodes in creation order

    HC() = default;
    explicit HC(const Params& p) : params(p) {}

    const std::vector<uint32_t>& roots() const { return root_ids; }
    size_t node_count() const { return splats.size(); }

    // HC.cpp:4-98.  `threads` > 1 evaluates pair costs concurrently; candidates still enter the heap in the reference's
    // order (list order of the active set), so the heap — and with it every tie between equal errors — evolves identically.
    void build(SplatVector splats_init, bool /*verbose*/ = true, unsigned threads = 0) {
        if (threads == 0) threads = lod_detail::host_threads();
        const uint32_t n = static_cast<uint32_t>(splats_init.size());
        splats = std::move(splats_init);
        splats.reserve(2 * static_cast<size_t>(n));
        kid[0].assign(n, NONE); kid[1].assign(n, NONE);
        err.assign(n, 0.0f); level.assign(n, 0u); wt.resize(n);
        for (uint32_t i = 0; i < n; ++i) wt[i] = splats[i].weight();
        cov_det.clear(); cov_inv.clear();
        cov_det.reserve(2 * static_cast<size_t>(n)); cov_inv.reserve(2 * static_cast<size_t>(n));
        for (uint32_t i = 0; i < n; ++i) remember_cov(splats[i]);
        // active set: index-linked list in insertion order; slot 2n-1 nodes at most, sentinel = `cap`
        const uint32_t cap = n ? 2 * n - 1 : 0;
        std::vector<uint32_t> next(cap + 1), prev(cap + 1);
        std::vector<uint8_t> merged_away(cap, 0);
        for (uint32_t i = 0; i <= cap; ++i) { next[i] = cap; prev[i] = cap; }
        uint32_t tail = cap;                                            // sentinel: next[cap] = head, prev[cap] = tail
        auto append = [&](uint32_t id) { prev[id] = tail; next[id] = cap; next[tail] = id; prev[cap] = id; tail = id; };
        auto unlink = [&](uint32_t id) {
            next[prev[id]] = next[id]; prev[next[id]] = prev[id];
            if (tail == id) tail = prev[id];
        };
        for (uint32_t i = 0; i < n; ++i) append(i);

        std::vector<Candidate> heap;
        auto push = [&](const Candidate& c) { heap.push_back(c); std::push_heap(heap.begin(), heap.end(), worse); };
        // all pairs (i < j in list order), HC.cpp:25-35: costs row by row in parallel, pushes in order
        if (n >= 2) {
            heap.reserve(static_cast<size_t>(n) * (n - 1) / 2);
            std::vector<float> row(n);
            for (uint32_t i = 0; i + 1 < n; ++i) {
                lod_detail::parallel_for(n - i - 1, 64, threads, [&](size_t k) { row[k] = pair_error(i, i + 1 + static_cast<uint32_t>(k)); });
                for (uint32_t j = i + 1; j < n; ++j) push(Candidate{row[j - i - 1], i, j});
            }
        }
        std::vector<uint32_t> others;
        std::vector<float> cost;
        while (!heap.empty()) {                                          // HC.cpp:41-90
            std::pop_heap(heap.begin(), heap.end(), worse);
            const Candidate top = heap.back();
            heap.pop_back();
            if (merged_away[top.a] || merged_away[top.b]) continue;      // stale: one side already belongs to a cluster
            if (top.a == top.b) throw std::runtime_error("HC: a cluster was paired with itself");
            const uint32_t id = static_cast<uint32_t>(splats.size());
            float wa = wt[top.a], wb = wt[top.b];
            const float total = wa + wb;
            wa /= total; wb /= total;
            splats.push_back(merge_splats(splats[top.a], splats[top.b], wa, wb));
            kid[0].push_back(top.a); kid[1].push_back(top.b);
            err.push_back(top.error);
            level.push_back(std::max(level[top.a], level[top.b]) + 1u);
            wt.push_back(splats[id].weight());
            remember_cov(splats[id]);
            merged_away[top.a] = merged_away[top.b] = 1;
            unlink(top.b); unlink(top.a);
            others.clear();
            for (uint32_t x = next[cap]; x != cap; x = next[x]) others.push_back(x);
            cost.resize(others.size());
            lod_detail::parallel_for(others.size(), 64, threads, [&](size_t k) { cost[k] = pair_error(id, others[k]); });
            for (size_t k = 0; k < others.size(); ++k) {
                if (cost[k] > params.max_error) continue;
                push(Candidate{cost[k], id, others[k]});
            }
            append(id);
        }
        root_ids.clear();
        for (uint32_t x = next[cap]; x != cap; x = next[x]) root_ids.push_back(x);
        std::vector<float>().swap(cov_det);             // (only the build needs them)
        std::vector<lod_detail::Cov3>().swap(cov_inv);
    }

    // HC.cpp:100-132.  The reference erases the examined entry and appends its two children, which is a first-in
    // first-out walk: entries in front of the cursor are final, so the output is the order in which nodes are accepted.
    Indices get_indices(Camera::Ptr camera, float threshold, MetricWeights) const {
        const vec3 cam = xyz(camera->worldMatrix[3]);
        Indices out, fifo(root_ids);
        for (size_t head = 0; head < fifo.size(); ++head) {
            const uint32_t node = fifo[head];
            const float d = length(xyz(splats[node].transform[3]) - cam);
            const float inv_d2 = 1 / (d * d);
            const float metric = err[node] * wt[node] * inv_d2;
            if (metric < threshold || kid[0][node] == NONE) { out.push_back(node); continue; }
            fifo.push_back(kid[0][node]);
            fifo.push_back(kid[1][node]);
        }
        return out;
    }

    // HC.cpp:134-168: the same walk limited to `steps` examined entries (accepted leaves count as steps); whatever is
    // still waiting when the steps run out is emitted behind the accepted ones.
    Indices get_indices_depth(uint32_t steps) const {
        Indices out, fifo(root_ids);
        size_t head = 0;
        for (uint32_t step = 0; head < fifo.size() && step < steps; ++step, ++head) {
            const uint32_t node = fifo[head];
            if (kid[0][node] == NONE) { out.push_back(node); continue; }
            fifo.push_back(kid[0][node]);
            fifo.push_back(kid[1][node]);
        }
        out.insert(out.end(), fifo.begin() + static_cast<std::ptrdiff_t>(head), fifo.end());
        return out;
    }

    // Appends this tree to a flat forest; ids are offset by `offset` (GridHC.cpp:79-88).
    void flatten_into(HcFlat& f, uint32_t offset) const {
        const size_t base = f.parent.size(), n = splats.size();
        f.parent.resize(base + n, NONE);
        f.child2.resize(2 * (base + n), NONE);
        f.is_leaf.resize(base + n, 1);
        f.error.insert(f.error.end(), err.begin(), err.end());
        f.weight.insert(f.weight.end(), wt.begin(), wt.end());
        for (uint32_t r : root_ids) f.roots.push_back(r + offset);
        for (size_t i = 0; i < n; ++i) {
            if (kid[0][i] == NONE) continue;
            f.is_leaf[base + i] = 0;
            for (int k = 0; k < 2; ++k) {
                f.child2[2 * (base + i) + k] = kid[k][i] + offset;
                f.parent[base + kid[k][i]] = static_cast<uint32_t>(i) + offset;
            }
        }
    }
    HcFlat flatten() const { HcFlat f; flatten_into(f, 0); return f; }

private:
    struct Candidate { float error; uint32_t a, b; };
    // std::priority_queue's comparator of HC.hpp (smallest error on top); the heap calls are the ones priority_queue makes
    static bool worse(const Candidate& x, const Candidate& y) { return x.error > y.error; }
    // HC.hpp:64-94: the error a merge of a and b would carry = weight-averaged divergence of both from the merged splat
    float pair_error(uint32_t a, uint32_t b) const {
        float wa = wt[a], wb = wt[b];
        const float total = wa + wb;
        wa /= total; wb /= total;
        const Splat c = merge_splats(splats[a], splats[b], wa, wb);
        // = wa * splat_divergence(splats[a], c) + wb * splat_divergence(splats[b], c) (Splat.h:169-195) with what depends on one
        // node only (its covariance's determinant and inverse) taken from the per-node cache and the merged splat's computed once
        using namespace lod_detail;
        const Cov3 C = cov_of(c);
        const float detC = det3(C);
        const Cov3 Ci = inv3_det(C, detC);
        const vec3 pc = xyz(c.transform[3]);
        auto divergence = [&](uint32_t x) {
            const Splat& sx = splats[x];
            const Cov3 X = cov_of(sx);
            const vec3 px = xyz(sx.transform[3]);
            const float kl = 0.5f * (kl_term_pre(X, px, cov_det[x], Ci, detC, pc) + kl_term_pre(C, pc, detC, cov_inv[x], cov_det[x], px));
            const vec4 cd = sx.color - c.color;
            return kl + length(xyz(cd)) + std::fabs(sx.color.w - c.color.w);
        };
        return wa * divergence(a) + wb * divergence(b);
    }
    void remember_cov(const Splat& s) {
        const lod_detail::Cov3 X = lod_detail::cov_of(s);
        cov_det.push_back(lod_detail::det3(X));
        cov_inv.push_back(lod_detail::inv3_det(X, cov_det.back()));
    }
    std::vector<float> cov_det;                        // per node: det and inverse of its covariance (pair_error)
    std::vector<lod_detail::Cov3> cov_inv;
    std::vector<uint32_t> kid[2], level, root_ids;
    std::vector<float> err, wt;
};

// ---- GridHC (GridHC.hpp:9-39, GridHC.cpp:10-111): one HC per non-empty cell of a uniform grid ---------------
class GridHC {
public:
    uint32_t subdivisions[3] = {20, 20, 20};
    std::vector<Splat> splats;                         // the cells' HC::splats, concatenated in cell order

    size_t cell_count() const { return trees.size(); } // non-empty cells

    void build(SplatVector splats_init) {              // GridHC.cpp:10-74
        const BB bb = BB::from_splats(splats_init, false);
        const vec3 size = bb.size();
        const vec3 scaled = size * 1.1f;               // 1.1f * size / vec3(subdivisions), GridHC.cpp:17
        const vec3 cell(scaled.x / static_cast<float>(subdivisions[0]), scaled.y / static_cast<float>(subdivisions[1]),
                        scaled.z / static_cast<float>(subdivisions[2]));
        const vec3 origin = bb.min() - cell * (0.1f / 1.1f);
        const size_t n_cells = static_cast<size_t>(subdivisions[0]) * subdivisions[1] * subdivisions[2];
        // counting sort of the splat ids by cell (ids stay in input order inside a cell); out-of-grid splats are dropped
        const uint32_t n = static_cast<uint32_t>(splats_init.size());
        std::vector<uint32_t> cell_of(n, HC::NONE), start(n_cells + 1, 0u);
        for (uint32_t s = 0; s < n; ++s) {
            const vec3 p = xyz(splats_init[s].transform[3]);
            const float f[3] = {(p.x - origin.x) / cell.x, (p.y - origin.y) / cell.y, (p.z - origin.z) / cell.z};
            uint32_t ijk[3];
            bool inside = true;
            for (int k = 0; k < 3; ++k) {
                // the reference converts float -> uint32 unchecked; negative / non-finite / huge values end up out of range there
                if (!(f[k] >= 0.0f) || f[k] >= 4294967296.0f) { inside = false; break; }
                ijk[k] = static_cast<uint32_t>(f[k]);
                if (ijk[k] >= subdivisions[k]) { inside = false; break; }
            }
            if (!inside) continue;
            cell_of[s] = get_index(ijk[0], ijk[1], ijk[2]);
            ++start[cell_of[s] + 1];
        }
        for (size_t c = 0; c < n_cells; ++c) start[c + 1] += start[c];
        std::vector<uint32_t> by_cell(start[n_cells]), cursor(start.begin(), start.end() - 1);
        for (uint32_t s = 0; s < n; ++s)
            if (cell_of[s] != HC::NONE) by_cell[cursor[cell_of[s]]++] = s;
        std::vector<uint32_t> occupied;
        for (size_t c = 0; c < n_cells; ++c)
            if (start[c + 1] > start[c]) occupied.push_back(static_cast<uint32_t>(c));
        trees.assign(occupied.size(), HC());
        // cells are independent: cluster them concurrently (each tree single-threaded), biggest work first by dynamic chunks
        lod_detail::parallel_for(occupied.size(), 1, lod_detail::host_threads(), [&](size_t t) {
            const uint32_t c = occupied[t];
            SplatVector members(start[c + 1] - start[c]);
            for (uint32_t k = start[c]; k < start[c + 1]; ++k) members[k - start[c]] = splats_init[by_cell[k]];
            trees[t].build(std::move(members), false, 1);
        });
        splats.clear();
        size_t total = 0;
        for (const HC& t : trees) total += t.splats.size();
        splats.reserve(total);
        for (const HC& t : trees) splats.insert(splats.end(), t.splats.begin(), t.splats.end());
    }

    Indices get_indices_error(uint32_t depth, float /*error*/) const {   // GridHC.cpp:76-92
        Indices out;
        out.reserve(splats.size());
        uint32_t offset = 0;
        for (const HC& t : trees) {
            for (uint32_t id : t.get_indices_depth(depth)) out.push_back(id + offset);
            offset += static_cast<uint32_t>(t.splats.size());
        }
        return out;
    }

    Indices get_indices(Camera::Ptr camera, float threshold, HC::MetricWeights w) const {   // GridHC.cpp:94-111
        Indices out;
        out.reserve(splats.size());
        uint32_t offset = 0;
        for (const HC& t : trees) {
            for (uint32_t id : t.get_indices(camera, threshold, w)) out.push_back(id + offset);
            offset += static_cast<uint32_t>(t.splats.size());
        }
        return out;
    }

    HcFlat flatten() const {
        HcFlat f;
        uint32_t offset = 0;
        for (const HC& t : trees) {
            f.cell_root_start.push_back(static_cast<uint32_t>(f.roots.size()));
            t.flatten_into(f, offset);
            offset += static_cast<uint32_t>(t.splats.size());
        }
        f.cell_root_start.push_back(static_cast<uint32_t>(f.roots.size()));
        return f;
    }

private:
    uint32_t get_index(uint32_t i, uint32_t j, uint32_t k) const {  // GridHC.hpp:32-38
        const uint64_t index = static_cast<uint64_t>(i) * subdivisions[1] * subdivisions[2] + static_cast<uint64_t>(j) * subdivisions[2] + k;
        if (index >= static_cast<uint64_t>(subdivisions[0]) * subdivisions[1] * subdivisions[2]) throw std::out_of_range("Cell index out of range");
        return static_cast<uint32_t>(index);
    }
    std::vector<HC> trees;                             // non-empty cells in cell-index order
};

}  // namespace gsx
