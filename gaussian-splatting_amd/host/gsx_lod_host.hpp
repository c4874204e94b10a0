// gsx_lod_host.hpp — host-side LOD structures with the reference's public API:
// BB (BB.hpp), Octree (Octree.hpp/.cpp), HC (HC.hpp/.cpp), GridHC (GridHC.hpp/.cpp)
// and the merge / divergence math they use (Splat.h:98-195), on gsx_math (glm
// operation order) so the built trees and their `splats` are bit-identical to the
// reference's.  Construction stays on the CPU (it is load-time work);
// what the GPU path consumes are the flatten() outputs (include/gsx.h
// gsx_octree_flat / gsx_hc_flat) and the view-independent depth cuts.
// Host-only; compile with -ffp-contract=off.  Unlike the reference nothing is
// printed to stdout.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <deque>
#include <limits>
#include <list>
#include <memory>
#include <numeric>
#include <queue>
#include <stdexcept>
#include <vector>

#include "gsx_scene.hpp"
#include "gsx_splat.hpp"

namespace gsx {

// ---- mat3 helpers in glm order --------------------------------------------------
inline mat3 operator+(const mat3& a, const mat3& b) { mat3 r(0.0f); for (int j = 0; j < 3; ++j) r[j] = a[j] + b[j]; return r; }
inline mat3 operator*(const mat3& a, float s) { mat3 r(0.0f); for (int j = 0; j < 3; ++j) r[j] = a[j] * s; return r; }
inline mat3 operator*(float s, const mat3& a) { return a * s; }
inline vec3 operator*(float s, vec3 v) { return v * s; }
inline vec4 operator*(float s, vec4 v) { return v * s; }
// glm/detail/func_matrix.inl:363-371
inline mat3 outerProduct(vec3 c, vec3 r) { mat3 m(0.0f); for (int i = 0; i < 3; ++i) m[i] = c * r[i]; return m; }
// glm/detail/func_matrix.inl:211-219
inline float determinant(const mat3& m) {
    return +m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2]) +
           m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]);
}
// glm/detail/func_matrix.inl:269-291
inline mat3 inverse(const mat3& m) {
    const float ood = 1.0f / (+m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2]) +
                              m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]));
    mat3 r(0.0f);
    r[0][0] = +(m[1][1] * m[2][2] - m[2][1] * m[1][2]) * ood;
    r[1][0] = -(m[1][0] * m[2][2] - m[2][0] * m[1][2]) * ood;
    r[2][0] = +(m[1][0] * m[2][1] - m[2][0] * m[1][1]) * ood;
    r[0][1] = -(m[0][1] * m[2][2] - m[2][1] * m[0][2]) * ood;
    r[1][1] = +(m[0][0] * m[2][2] - m[2][0] * m[0][2]) * ood;
    r[2][1] = -(m[0][0] * m[2][1] - m[2][0] * m[0][1]) * ood;
    r[0][2] = +(m[0][1] * m[1][2] - m[1][1] * m[0][2]) * ood;
    r[1][2] = -(m[0][0] * m[1][2] - m[1][0] * m[0][2]) * ood;
    r[2][2] = +(m[0][0] * m[1][1] - m[1][0] * m[0][1]) * ood;
    return r;
}

// ---- Splat.h:98-195 ---------------------------------------------------------------
inline float splat_weight(const SplatSplit& s) { return s.color.w * (s.scale.x * s.scale.y * s.scale.z); }

inline Splat merge(const SplatSplitVector& raw) {  // Splat.h:104-144
    std::vector<float> w(raw.size());
    float total = 0.0f;
    for (size_t i = 0; i < raw.size(); ++i) { w[i] = splat_weight(raw[i]); total += w[i]; }
    for (size_t i = 0; i < raw.size(); ++i) w[i] /= total;
    SplatVector splats(raw.size());
    for (size_t i = 0; i < raw.size(); ++i) splats[i] = split_to_splat(raw[i]);
    vec4 color(0.0f);
    for (size_t i = 0; i < splats.size(); ++i) color = color + splats[i].color * w[i];
    vec3 center(0.0f);
    for (size_t i = 0; i < splats.size(); ++i) center = center + xyz(splats[i].transform[3]) * w[i];
    mat3 cov(0.0f);
    for (size_t i = 0; i < splats.size(); ++i) {
        const vec3 diff = xyz(splats[i].transform[3]) - center;
        cov = cov + (mat3_of(splats[i].transform) + outerProduct(diff, diff)) * w[i];
    }
    Splat s;
    s.transform = mat4(cov);
    s.transform[3] = vec4(center, 1.0f);
    s.color = color;
    return s;
}

inline Splat merge_splats(const Splat& a, const Splat& b, float w_a, float w_b) {  // Splat.h:146-163
    const vec3 ac = xyz(a.transform[3]), bc = xyz(b.transform[3]);
    const vec3 center = w_a * ac + w_b * bc;
    const vec3 ad = ac - center, bd = bc - center;
    const mat3 cov = w_a * (mat3_of(a.transform) + outerProduct(ad, ad)) + w_b * (mat3_of(b.transform) + outerProduct(bd, bd));
    Splat s;
    s.transform = mat4(cov);
    s.transform[3] = vec4(center, 1.0f);
    s.color = w_a * a.color + w_b * b.color;
    return s;
}

inline float trace(const mat3& a) { return a[0][0] + a[1][1] + a[2][2]; }
inline float kl_divergence(const Splat& a, const Splat& b) {  // Splat.h:169-183
    const mat3 a_cov = mat3_of(a.transform), b_cov = mat3_of(b.transform);
    const vec3 diff = xyz(a.transform[3]) - xyz(b.transform[3]);
    const mat3 cov_inv = inverse(b_cov);
    const float trace_term = trace(cov_inv * a_cov);
    const float det_term = std::log(determinant(b_cov) / determinant(a_cov));
    return 0.5f * (trace_term + dot(diff, cov_inv * diff) - 3.0f + det_term);
}
inline float sym_kl_divergence(const Splat& a, const Splat& b) { return 0.5f * (kl_divergence(a, b) + kl_divergence(b, a)); }
inline float splat_divergence(const Splat& a, const Splat& b) {  // Splat.h:189-195
    const float kl = sym_kl_divergence(a, b);
    const vec4 cd = a.color - b.color;
    return kl + length(xyz(cd)) + std::fabs(a.color.w - b.color.w);
}

// ---- BB.hpp -------------------------------------------------------------------------
class BB {
public:
    mat4 transform{1.0f};
    static BB from_aabb(vec3 mn, vec3 mx) {  // BB.hpp:15-23
        BB bb;
        bb.transform = mat4(1.0f);
        bb.transform[3] = vec4((mn + mx) / 2.0f, 1.0f);
        bb.transform[0][0] = mx.x - mn.x;
        bb.transform[1][1] = mx.y - mn.y;
        bb.transform[2][2] = mx.z - mn.z;
        return bb;
    }
    vec3 center() const { return xyz(transform[3]); }
    vec3 size() const { return {length(xyz(transform[0])), length(xyz(transform[1])), length(xyz(transform[2]))}; }
    vec3 min() const { return center() - size() / 2.0f; }
    vec3 max() const { return center() + size() / 2.0f; }
    std::array<vec3, 8> corners() const {  // BB.hpp:41-57
        const vec3 h = size() / 2.0f, c = center();
        const float x = h.x, y = h.y, z = h.z;
        return {c + vec3(-x, -y, -z), c + vec3(x, -y, -z), c + vec3(x, y, -z), c + vec3(-x, y, -z),
                c + vec3(-x, -y, z),  c + vec3(x, -y, z),  c + vec3(x, y, z),  c + vec3(-x, y, z)};
    }
    float screen_area(Camera::Ptr camera) const {  // BB.hpp:59-77
        const mat4 vp = camera->getProjectionMatrix() * camera->getViewMatrix();
        float mnx = 10000000.0f, mny = 10000000.0f, mxx = -10000000.0f, mxy = -10000000.0f;
        for (const vec3& p : corners()) {
            const vec4 clip = vp * vec4(p, 1.0f);
            if (clip.w == 0.0f) continue;
            const float nx = clip.x / clip.w, ny = clip.y / clip.w;
            mnx = (nx < mnx) ? nx : mnx; mny = (ny < mny) ? ny : mny;
            mxx = (mxx < nx) ? nx : mxx; mxy = (mxy < ny) ? ny : mxy;
        }
        return (mxx - mnx) * (mxy - mny);
    }
    template <typename PosOf, typename Vec>
    static BB from_positions(const Vec& items, PosOf pos, bool expand) {  // BB.hpp:79-106 / Octree.cpp:9-31 (min/max start at 0)
        vec3 mn(0.0f), mx(0.0f);
        for (const auto& it : items) {
            const vec3 p = pos(it);
            mn.x = std::min(mn.x, p.x); mn.y = std::min(mn.y, p.y); mn.z = std::min(mn.z, p.z);
            mx.x = std::max(mx.x, p.x); mx.y = std::max(mx.y, p.y); mx.z = std::max(mx.z, p.z);
        }
        const vec3 center = (mx + mn) / 2.0f;
        if (!expand) return from_aabb(mn, mx);
        const float maxSize = std::max({mx.x - mn.x, mx.y - mn.y, mx.z - mn.z});
        return from_aabb(center - vec3(maxSize / 2.0f), center + vec3(maxSize / 2.0f));
    }
    static BB from_splats(const SplatVector& splats, bool expand = true) {
        return from_positions(splats, [](const Splat& s) { return xyz(s.transform[3]); }, expand);
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

// ---- Octree.hpp / Octree.cpp ------------------------------------------------------
class Octree {
public:
    class Node {
    public:
        using Ptr = std::shared_ptr<Node>;
        uint32_t depth = 0;
        BB bb;
        Indices indices, indices_raw;
        std::vector<Ptr> children;
        bool is_leaf() const { return children.empty(); }
    };
    using NodeVector = std::vector<Node::Ptr>;
    Node::Ptr root;
    SplatVector splats;
    uint32_t max_depth{10};
    uint32_t max_splats_per_node{1};

    void build(SplatSplitVector raw) {  // Octree.cpp:33-147
        splats.clear();
        queue.clear();
        splats_raw = std::move(raw);
        root = std::make_shared<Node>();
        root->depth = 0;
        root->bb = BB::from_positions(splats_raw, [](const SplatSplit& s) { return s.position; }, true);
        root->indices_raw.resize(splats_raw.size());
        std::iota(root->indices_raw.begin(), root->indices_raw.end(), 0);
        uint32_t counter = 0;
        queue.push_back(root);
        while (counter < queue.size()) {
            Node::Ptr node = queue[counter++];
            if (node->depth >= max_depth || node->indices_raw.size() <= max_splats_per_node) continue;
            const vec3 center = node->bb.center();
            std::vector<Node::Ptr> children(8);
            for (int i = 0; i < 8; ++i) { children[i] = std::make_shared<Node>(); children[i]->depth = node->depth + 1; }
            for (uint32_t idx : node->indices_raw) {
                const SplatSplit& s = splats_raw[idx];
                int index = 0;
                if (s.position.x > center.x) index |= 1;
                if (s.position.y > center.y) index |= 2;
                if (s.position.z > center.z) index |= 4;
                children[index]->indices_raw.push_back(idx);
            }
            for (int i = 0; i < 8; ++i) {
                Node::Ptr child = children[i];
                if (child->indices_raw.empty()) continue;
                vec3 mn = node->bb.min(), mx = node->bb.max();
                if (i & 1) mn.x = center.x; else mx.x = center.x;
                if (i & 2) mn.y = center.y; else mx.y = center.y;
                if (i & 4) mn.z = center.z; else mx.z = center.z;
                child->bb = BB::from_aabb(mn, mx);
                node->children.push_back(child);
                queue.push_back(child);
            }
        }
        // collapse single-child chains (Octree.cpp:110-142)
        counter = 0;
        uint32_t size = 1;
        queue[0] = root;
        while (counter < size) {
            Node::Ptr node = queue[counter++];
            if (node->is_leaf()) continue;
            if (node->children.size() == 1) {
                Node::Ptr child = node->children[0];
                node->depth = child->depth;
                node->bb = child->bb;
                node->indices = child->indices;
                node->indices_raw = child->indices_raw;
                node->children = child->children;
                counter--;
                continue;
            }
            for (auto& child : node->children) queue[size++] = child;
        }
        queue.resize(counter);
    }

    void generate() {  // Octree.cpp:149-187: one merged splat per node in BFS order
        if (queue.empty()) return;
        queue[0] = root;
        uint32_t counter = 0, size = 1;
        splats.clear();
        while (counter < size) {
            Node::Ptr node = queue[counter++];
            SplatSplitVector sub(node->indices_raw.size());
            for (size_t i = 0; i < sub.size(); ++i) sub[i] = splats_raw[node->indices_raw[i]];
            splats.push_back(merge(sub));
            node->indices.push_back(static_cast<uint32_t>(splats.size() - 1));
            if (node->is_leaf()) continue;
            for (auto& child : node->children) queue[size++] = child;
        }
    }

    Indices get_indices(Camera::Ptr camera, float min_screen_area) {  // Octree.cpp:189-216
        Indices out;
        if (queue.empty()) return out;
        queue[0] = root;
        uint32_t counter = 0, size = 1;
        while (counter < size) {
            Node::Ptr node = queue[counter++];
            const float area = node->bb.screen_area(camera);
            if (node->is_leaf() || (area < min_screen_area)) {
                out.insert(out.end(), node->indices.begin(), node->indices.end());
                continue;
            }
            for (auto& child : node->children) queue[size++] = child;
        }
        return out;
    }

    Indices get_indices_depth(uint32_t depth) {  // Octree.hpp:131-162 (DFS from the back of the queue)
        Indices out;
        std::vector<Node::Ptr> stack{root}, picked;
        while (!stack.empty()) {
            Node::Ptr node = stack.back();
            stack.pop_back();
            if (node->depth > depth || node->is_leaf()) { picked.push_back(node); continue; }
            for (auto& child : node->children) stack.push_back(child);
        }
        for (auto& n : picked) out.insert(out.end(), n->indices.begin(), n->indices.end());
        return out;
    }

    SplatVector* data() { return &splats; }

    // BFS flattening in generate()'s order: node i owns splat i.
    OctreeFlat flatten() const {
        OctreeFlat f;
        if (!root) return f;
        std::vector<const Node*> nodes{root.get()};
        f.parent.push_back(0xFFFFFFFFu);
        for (size_t i = 0; i < nodes.size(); ++i) {
            const Node* n = nodes[i];
            f.first_child.push_back(static_cast<uint32_t>(nodes.size()));
            f.n_children.push_back(static_cast<uint32_t>(n->children.size()));
            f.is_leaf.push_back(n->is_leaf() ? 1 : 0);
            const vec3 c = n->bb.center(), h = n->bb.size() / 2.0f;
            for (int k = 0; k < 3; ++k) { f.center3.push_back(c[k]); f.half3.push_back(h[k]); }
            for (auto& ch : n->children) { nodes.push_back(ch.get()); f.parent.push_back(static_cast<uint32_t>(i)); }
        }
        return f;
    }

private:
    SplatSplitVector splats_raw;
    NodeVector queue;
};

// ---- HC.hpp / HC.cpp ----------------------------------------------------------------
class HC {
public:
    struct Params { float max_error{std::numeric_limits<float>::infinity()}; };
    struct MetricWeights { float e{0.0f}, w{0.0f}, d{0.0f}; };
    class Node {
    public:
        using Ptr = std::shared_ptr<Node>;
        uint32_t depth = 0;
        Splat splat;
        uint32_t index = 0;
        std::array<Ptr, 2> children;
        float error{0.0f};
        bool is_leaf() const { return children[0] == nullptr && children[1] == nullptr; }
        std::list<Ptr>::iterator nodes_it;
        bool processed{false};
    };
    Params params;
    std::list<Node::Ptr> nodes;
    SplatVector splats;

    HC() = default;
    explicit HC(const Params& p) : params(p) {}

    void build(SplatVector splats_init, bool verbose = true) {  // HC.cpp:4-98
        (void)verbose;
        std::priority_queue<Node::Ptr, std::vector<Node::Ptr>, Cmp> queue;
        splats.reserve(2 * splats_init.size());
        for (const auto& s : splats_init) {
            auto node = std::make_shared<Node>();
            node->depth = 0;
            node->index = static_cast<uint32_t>(splats.size());
            node->splat = s;
            node->error = 0.0f;
            node->nodes_it = nodes.insert(nodes.end(), node);
            splats.push_back(s);
        }
        for (auto it = nodes.begin(); it != nodes.end(); ++it)
            for (auto jt = std::next(it); jt != nodes.end(); ++jt) queue.push(merge_nodes(*it, *jt));
        while (!queue.empty()) {
            auto merged = queue.top();
            queue.pop();
            if (merged->children[0]->processed || merged->children[1]->processed) continue;
            merged->index = static_cast<uint32_t>(splats.size());
            splats.push_back(merged->splat);
            auto a = merged->children[0], b = merged->children[1];
            a->processed = true;
            b->processed = true;
            if (a->nodes_it == b->nodes_it) throw std::runtime_error("Same node, this should not happen in HC.");
            nodes.erase(b->nodes_it);
            nodes.erase(a->nodes_it);
            for (auto it = nodes.begin(); it != nodes.end(); ++it) {
                auto node = merge_nodes(merged, *it);
                if (node->error > params.max_error) continue;
                queue.push(node);
            }
            merged->nodes_it = nodes.insert(nodes.end(), merged);
        }
    }

    Indices get_indices(Camera::Ptr camera, float threshold, MetricWeights) {  // HC.cpp:100-132
        std::vector<Node::Ptr> queue(nodes.begin(), nodes.end());
        size_t pos = 0;
        const vec3 cam = xyz(camera->worldMatrix[3]);
        while (!queue.empty() && pos < queue.size()) {
            auto node = queue[pos];
            const float d = length(xyz(node->splat.transform[3]) - cam);
            const float dist = 1 / (d * d);
            const float metric = node->error * node->splat.weight() * dist;
            if (metric < threshold || node->is_leaf()) { pos++; continue; }
            queue.erase(queue.begin() + static_cast<long>(pos));
            queue.push_back(node->children[0]);
            queue.push_back(node->children[1]);
        }
        Indices out;
        for (auto& n : queue) out.push_back(n->index);
        return out;
    }

    // HC.cpp:134-168.  erase(begin+pos) + two push_backs is a FIFO: entries before
    // `pos` are settled leaves, the entry at `pos` is the FIFO head.
    Indices get_indices_depth(uint32_t depth) {
        std::vector<Node::Ptr> settled;
        std::deque<Node::Ptr> fifo(nodes.begin(), nodes.end());
        uint32_t counter = 0;
        while (!fifo.empty()) {
            if (counter >= depth) break;
            counter++;
            auto node = fifo.front();
            fifo.pop_front();
            if (node->is_leaf()) { settled.push_back(node); continue; }
            fifo.push_back(node->children[0]);
            fifo.push_back(node->children[1]);
        }
        Indices out;
        for (auto& n : settled) out.push_back(n->index);
        for (auto& n : fifo) out.push_back(n->index);
        return out;
    }

    // Appends this tree to a flat forest; ids are offset by `offset` (GridHC.cpp:79-88).
    void flatten_into(HcFlat& f, uint32_t offset) const {
        const size_t base = f.parent.size(), n = splats.size();
        f.parent.resize(base + n, 0xFFFFFFFFu);
        f.child2.resize(2 * (base + n), 0xFFFFFFFFu);
        f.is_leaf.resize(base + n, 1);
        f.error.resize(base + n, 0.0f);
        f.weight.resize(base + n, 0.0f);
        for (size_t i = 0; i < n; ++i) f.weight[base + i] = splats[i].weight();
        std::vector<const Node*> stack;
        for (auto& r : nodes) { f.roots.push_back(r->index + offset); stack.push_back(r.get()); }
        while (!stack.empty()) {
            const Node* nd = stack.back();
            stack.pop_back();
            f.error[base + nd->index] = nd->error;
            if (!nd->is_leaf()) {
                f.is_leaf[base + nd->index] = 0;
                for (int k = 0; k < 2; ++k) {
                    f.child2[2 * (base + nd->index) + k] = nd->children[k]->index + offset;
                    f.parent[base + nd->children[k]->index] = nd->index + offset;
                    stack.push_back(nd->children[k].get());
                }
            }
        }
    }
    HcFlat flatten() const { HcFlat f; flatten_into(f, 0); return f; }

private:
    struct Cmp { bool operator()(const Node::Ptr& a, const Node::Ptr& b) const { return a->error > b->error; } };
    Node::Ptr merge_nodes(const Node::Ptr& a, const Node::Ptr& b) {  // HC.hpp:64-94
        const Splat sa = splats[a->index], sb = splats[b->index];
        float w_a = sa.weight(), w_b = sb.weight();
        const float total = w_a + w_b;
        w_a /= total;
        w_b /= total;
        const Splat sc = merge_splats(sa, sb, w_a, w_b);
        const float div_a = splat_divergence(sa, sc), div_b = splat_divergence(sb, sc);
        auto node = std::make_shared<Node>();
        node->depth = std::max(a->depth, b->depth) + 1;
        node->splat = sc;
        node->error = (w_a * div_a + w_b * div_b);
        node->children[0] = a;
        node->children[1] = b;
        return node;
    }
};

// ---- GridHC.hpp / GridHC.cpp --------------------------------------------------------
class GridHC {
public:
    struct Cell {
        using Ptr = std::shared_ptr<Cell>;
        std::vector<Splat> splats;
        HC hc;
        bool empty() const { return splats.empty(); }
    };
    uint32_t subdivisions[3] = {20, 20, 20};
    std::vector<Cell::Ptr> cells;
    std::vector<Splat> splats;

    void build(SplatVector splats_init) {  // GridHC.cpp:10-74
        const BB bb = BB::from_splats(splats_init, false);
        const vec3 size = bb.size();
        const vec3 sub(static_cast<float>(subdivisions[0]), static_cast<float>(subdivisions[1]), static_cast<float>(subdivisions[2]));
        const vec3 s11 = 1.1f * size;
        const vec3 cell_size(s11.x / sub.x, s11.y / sub.y, s11.z / sub.z);
        const vec3 mn = bb.min() - cell_size * (0.1f / 1.1f);
        cells.clear();
        cells.resize(static_cast<size_t>(subdivisions[0]) * subdivisions[1] * subdivisions[2]);
        splats.clear();
        splats.reserve(2 * splats_init.size());
        for (auto& c : cells) c = std::make_shared<Cell>();
        for (const auto& s : splats_init) {
            const vec3 p = xyz(s.transform[3]);
            const float fi = (p.x - mn.x) / cell_size.x, fj = (p.y - mn.y) / cell_size.y, fk = (p.z - mn.z) / cell_size.z;
            // negative / non-finite -> the reference's float->uint32 conversion lands out of range: splat ignored
            if (!(fi >= 0.0f) || !(fj >= 0.0f) || !(fk >= 0.0f) || fi >= 4294967296.0f || fj >= 4294967296.0f || fk >= 4294967296.0f) continue;
            const uint32_t i = static_cast<uint32_t>(fi), j = static_cast<uint32_t>(fj), k = static_cast<uint32_t>(fk);
            if (i < subdivisions[0] && j < subdivisions[1] && k < subdivisions[2]) cells[get_index(i, j, k)]->splats.push_back(s);
        }
        for (auto& cell : cells) {
            if (cell->empty()) continue;
            cell->hc.build(cell->splats, false);
            splats.insert(splats.end(), cell->hc.splats.begin(), cell->hc.splats.end());
        }
    }

    Indices get_indices_error(uint32_t depth, float /*error*/) {  // GridHC.cpp:76-92
        Indices out;
        out.reserve(splats.size());
        uint32_t offset = 0;
        for (const auto& cell : cells) {
            if (cell->empty()) continue;
            for (uint32_t id : cell->hc.get_indices_depth(depth)) out.push_back(id + offset);
            offset += static_cast<uint32_t>(cell->hc.splats.size());
        }
        return out;
    }

    Indices get_indices(Camera::Ptr camera, float threshold, HC::MetricWeights w) {  // GridHC.cpp:94-111
        Indices out;
        out.reserve(splats.size());
        uint32_t offset = 0;
        for (const auto& cell : cells) {
            if (cell->empty()) continue;
            for (uint32_t id : cell->hc.get_indices(camera, threshold, w)) out.push_back(id + offset);
            offset += static_cast<uint32_t>(cell->hc.splats.size());
        }
        return out;
    }

    HcFlat flatten() const {
        HcFlat f;
        uint32_t offset = 0;
        for (const auto& cell : cells) {
            if (cell->empty()) continue;
            f.cell_root_start.push_back(static_cast<uint32_t>(f.roots.size()));
            cell->hc.flatten_into(f, offset);
            offset += static_cast<uint32_t>(cell->hc.splats.size());
        }
        f.cell_root_start.push_back(static_cast<uint32_t>(f.roots.size()));
        return f;
    }

private:
    uint32_t get_index(uint32_t i, uint32_t j, uint32_t k) const {  // GridHC.hpp:32-38
        const uint32_t index = i * subdivisions[1] * subdivisions[2] + j * subdivisions[2] + k;
        if (index >= cells.size()) throw std::out_of_range("Cell index out of range");
        return index;
    }
};

}  // namespace gsx
