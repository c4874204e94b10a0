// orc_lod.cpp — CPU ORACLE (test infrastructure): per-frame LOD cuts on
// flattened trees. Build with -ffp-contract=off.
#include "orc.h"

#include <cmath>
#include <cstdint>
#include <deque>
#include <vector>

extern "C" {

// BB.hpp:41-57 corners(), BB.hpp:59-77 screen_area(); mat4*vec4 per
// glm/detail/type_mat4x4.inl:561-572: (m0*x + m1*y) + (m2*z + m3*w)
float orc_bb_screen_area(const float c[3], const float h[3], const float vp[16]) {
    const float x = h[0], y = h[1], z = h[2];
    const float off[8][3] = {{-x, -y, -z}, {x, -y, -z}, {x, y, -z}, {-x, y, -z},
                             {-x, -y, z},  {x, -y, z},  {x, y, z},  {-x, y, z}};
    float mn[2] = {10000000.0f, 10000000.0f};
    float mx[2] = {-10000000.0f, -10000000.0f};
    for (int i = 0; i < 8; ++i) {
        const float px = c[0] + off[i][0], py = c[1] + off[i][1], pz = c[2] + off[i][2];
        float clip[4];
        for (int r = 0; r < 4; ++r) {
            const float a0 = vp[0 * 4 + r] * px + vp[1 * 4 + r] * py;
            const float a1 = vp[2 * 4 + r] * pz + vp[3 * 4 + r] * 1.0f;
            clip[r] = a0 + a1;
        }
        if (clip[3] == 0.0f) continue;
        const float nx = clip[0] / clip[3], ny = clip[1] / clip[3];
        // glm::min(a,b) = (b < a) ? b : a ; glm::max(a,b) = (a < b) ? b : a
        mn[0] = (nx < mn[0]) ? nx : mn[0];
        mn[1] = (ny < mn[1]) ? ny : mn[1];
        mx[0] = (mx[0] < nx) ? nx : mx[0];
        mx[1] = (mx[1] < ny) ? ny : mx[1];
    }
    const float sx = mx[0] - mn[0], sy = mx[1] - mn[1];
    return sx * sy;
}

// Octree.cpp:189-216
uint64_t orc_octree_cut(uint32_t n_nodes, const float* center3, const float* half3,
                        const uint32_t* first_child, const uint32_t* n_children,
                        const float vp[16], float min_screen_area, uint32_t* out) {
    if (n_nodes == 0) return 0;
    std::vector<uint32_t> queue(n_nodes);
    queue[0] = 0;
    uint32_t counter = 0, size = 1;
    uint64_t m = 0;
    while (counter < size) {
        const uint32_t node = queue[counter++];
        const float area = orc_bb_screen_area(center3 + 3 * node, half3 + 3 * node, vp);
        if (n_children[node] == 0 || (area < min_screen_area)) {
            out[m++] = node;  // node->indices == {BFS position} (Octree.cpp:174-176)
            continue;
        }
        for (uint32_t k = 0; k < n_children[node]; ++k) queue[size++] = first_child[node] + k;
    }
    return m;
}

// HC.cpp:134-168. The reference's vector::erase(begin+pos) + two push_backs is a
// FIFO: entries before pos are settled leaves, the entry at pos is the FIFO head.
uint64_t orc_hc_cut_depth(uint32_t n_roots, const uint32_t* roots, const uint32_t* child2,
                          uint32_t depth, uint32_t* out) {
    std::vector<uint32_t> queue(roots, roots + n_roots);
    uint32_t counter = 0;
    size_t pos = 0;
    while (!queue.empty() && pos < queue.size()) {
        if (counter >= depth) break;
        counter++;
        const uint32_t node = queue[pos];
        const bool leaf = child2[2 * node] == UINT32_MAX && child2[2 * node + 1] == UINT32_MAX;
        if (leaf) { pos++; continue; }
        queue.erase(queue.begin() + static_cast<long>(pos));
        queue.push_back(child2[2 * node]);
        queue.push_back(child2[2 * node + 1]);
    }
    for (size_t i = 0; i < queue.size(); ++i) out[i] = queue[i];
    return queue.size();
}

// HC.cpp:100-132
uint64_t orc_hc_cut_metric(uint32_t n_roots, const uint32_t* roots, const uint32_t* child2,
                           const float* error, const float* weight, const float* pos3,
                           const float cam[3], float threshold, uint32_t* out) {
    std::vector<uint32_t> queue(roots, roots + n_roots);
    size_t pos = 0;
    while (!queue.empty() && pos < queue.size()) {
        const uint32_t node = queue[pos];
        const float dx = pos3[3 * node] - cam[0], dy = pos3[3 * node + 1] - cam[1],
                    dz = pos3[3 * node + 2] - cam[2];
        const float splat_dist = std::sqrt(dx * dx + dy * dy + dz * dz);
        const float dist = 1 / (splat_dist * splat_dist);
        const float metric = error[node] * weight[node] * dist;
        const bool leaf = child2[2 * node] == UINT32_MAX && child2[2 * node + 1] == UINT32_MAX;
        if (metric < threshold || leaf) { pos++; continue; }
        queue.erase(queue.begin() + static_cast<long>(pos));
        queue.push_back(child2[2 * node]);
        queue.push_back(child2[2 * node + 1]);
    }
    for (size_t i = 0; i < queue.size(); ++i) out[i] = queue[i];
    return queue.size();
}

// Splat.h:24-28; glm/detail/func_matrix.inl:223-244 determinant(mat4)
float orc_splat_weight(const float* s) {
    auto m = [&](int c, int r) { return s[c * 4 + r]; };
    const float s00 = m(2,2) * m(3,3) - m(3,2) * m(2,3);
    const float s01 = m(2,1) * m(3,3) - m(3,1) * m(2,3);
    const float s02 = m(2,1) * m(3,2) - m(3,1) * m(2,2);
    const float s03 = m(2,0) * m(3,3) - m(3,0) * m(2,3);
    const float s04 = m(2,0) * m(3,2) - m(3,0) * m(2,2);
    const float s05 = m(2,0) * m(3,1) - m(3,0) * m(2,1);
    const float d0 = +(m(1,1) * s00 - m(1,2) * s01 + m(1,3) * s02);
    const float d1 = -(m(1,0) * s00 - m(1,2) * s03 + m(1,3) * s04);
    const float d2 = +(m(1,0) * s01 - m(1,1) * s03 + m(1,3) * s05);
    const float d3 = -(m(1,0) * s02 - m(1,1) * s04 + m(1,2) * s05);
    const float det = m(0,0) * d0 + m(0,1) * d1 + m(0,2) * d2 + m(0,3) * d3;
    const float alpha = s[19];
    const float volume = static_cast<float>(1.333 * static_cast<double>(std::sqrt(det)));
    return alpha * volume;
}

}  // extern "C"
