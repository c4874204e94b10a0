// ref_harness.cpp — builds oracle/_ref/libgsxref.so from the REAL reference
// sources where they lie under /root/reference (never copied into this repo):
//   Splat.h, BB.hpp, Camera.h, Node.{h,cpp}, Octree.{hpp,cpp}, HC.{hpp,cpp},
//   GridHC.{hpp,cpp} and the vendored header-only glm 0.9.9.8.
// TEST INFRASTRUCTURE ONLY: used to pin oracle/orc_*.cpp and to generate the
// fixtures under tests/golden (tests/golden/make_golden.py).  It exists only in
// the build container; the GPU box receives the prebuilt .so.
//
// NOT buildable here (needs <webgpu/webgpu.hpp> + <tbb/task_arena.h>, both
// absent, and no stand-ins are written): SplatMesh.h, SplatMesh{Octree,HC,GridHC}
// and ResourceManager.cpp.  For SplatMesh::sortSplats (SplatMesh.h:84-114) this
// harness therefore evaluates the reference's key expression with the reference's
// own glm (glm::distance on splatData[i].transform[3]) and applies std::sort with
// the reference's comparator; the ordering is implementation-independent except
// inside equal-key runs.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <limits>
#include <list>
#include <map>
#include <memory>
#include <stdexcept>
#include <vector>

#include <glm/glm.hpp>
#include <glm/ext.hpp>

#include "Splat.h"
#include "BB.hpp"
#include "Camera.h"
#include "Octree.hpp"
#include "HC.hpp"
#include "GridHC.hpp"
#include "OrbitCamera.h"

static_assert(sizeof(Splat) == 80, "Splat is the 80-B GPU record");
static_assert(sizeof(SplatRaw) == 32, "SplatRaw is the 32-B file record");

namespace {

Camera::Ptr make_camera(const float world16[16], float fov, float aspect, float znear, float zfar) {
    auto cam = std::make_shared<Camera>();
    std::memcpy(&cam->worldMatrix[0][0], world16, 64);
    cam->fov = fov; cam->aspect = aspect; cam->near = znear; cam->far = zfar;
    return cam;
}

struct OctreeFlat {
    std::vector<Octree::Node::Ptr> nodes;  // BFS order as Octree::generate walks it
    std::vector<uint32_t> first_child, n_children;
};

OctreeFlat flatten(Octree& oc) {
    OctreeFlat f;
    f.nodes.push_back(oc.root);
    for (size_t i = 0; i < f.nodes.size(); ++i) {
        auto node = f.nodes[i];
        f.first_child.push_back(static_cast<uint32_t>(f.nodes.size()));
        f.n_children.push_back(static_cast<uint32_t>(node->children.size()));
        for (auto& c : node->children) f.nodes.push_back(c);
    }
    return f;
}

// One HC tree flattened by node->index (== position in hc.splats).
void flatten_hc(HC& hc, uint32_t offset, std::vector<uint32_t>& child2, std::vector<float>& error,
                std::vector<uint32_t>& roots) {
    const size_t n = hc.splats.size();
    const size_t base = child2.size() / 2;
    child2.resize((base + n) * 2, UINT32_MAX);
    error.resize(base + n, 0.0f);
    std::vector<HC::Node::Ptr> stack;
    for (auto& r : hc.nodes) { roots.push_back(r->index + offset); stack.push_back(r); }
    while (!stack.empty()) {
        auto nd = stack.back(); stack.pop_back();
        error[base + nd->index] = nd->error;
        if (!nd->is_leaf()) {
            child2[2 * (base + nd->index)] = nd->children[0]->index + offset;
            child2[2 * (base + nd->index) + 1] = nd->children[1]->index + offset;
            stack.push_back(nd->children[0]);
            stack.push_back(nd->children[1]);
        }
    }
}

std::map<int, std::unique_ptr<Octree>> g_octrees;
std::map<int, std::unique_ptr<HC>> g_hcs;
std::map<int, std::unique_ptr<GridHC>> g_grids;
int g_next = 1;

}  // namespace

extern "C" {

// ---- Splat.h ----
void ref_raw_to_split(const void* raw32, uint64_t n, float* split14) {
    const SplatRaw* r = static_cast<const SplatRaw*>(raw32);
    for (uint64_t i = 0; i < n; ++i) {
        SplatSplit s = raw_to_split(r[i]);
        std::memcpy(split14 + i * 14, &s, 56);
    }
}
void ref_split_to_splat(const float* split14, uint64_t n, float* splat20) {
    for (uint64_t i = 0; i < n; ++i) {
        SplatSplit s;
        std::memcpy(&s, split14 + i * 14, 56);
        Splat sp = split_to_splat(s);
        std::memcpy(splat20 + i * 20, &sp, 80);
    }
}
float ref_splat_weight(const float* splat20) {
    Splat sp;
    std::memcpy(&sp, splat20, 80);
    return sp.weight();
}

// ---- SplatMesh.h:90-94 key expression with the reference's glm ----
void ref_distance_keys(const float* splat20, uint64_t n, const uint32_t* idx, uint64_t m,
                       const float cam[3], float* dist_out) {
    const Splat* splatData = reinterpret_cast<const Splat*>(splat20);
    (void)n;
    glm::vec3 cameraPos(cam[0], cam[1], cam[2]);
    for (uint64_t k = 0; k < m; ++k) {
        uint32_t i = idx ? idx[k] : static_cast<uint32_t>(k);
        glm::vec3 position = glm::vec3(splatData[i].transform[3]);
        dist_out[i] = glm::distance(position, cameraPos);
    }
}
// SplatMesh.h:90-107 (sequential branch) on a caller-owned, persistent index vector
void ref_sort(const float* splat20, uint64_t n, uint32_t* idx_inout, uint64_t m, const float cam[3]) {
    std::vector<float> distances(n);
    ref_distance_keys(splat20, n, idx_inout, m, cam, distances.data());
    std::sort(idx_inout, idx_inout + m, [&](uint32_t a, uint32_t b) { return distances[a] > distances[b]; });
}

// ---- Camera.h / Node.cpp ----
void ref_camera_matrices(const float world16[16], float fov, float aspect, float znear, float zfar,
                         float view16[16], float proj16[16]) {
    auto cam = make_camera(world16, fov, aspect, znear, zfar);
    glm::mat4 v = cam->getViewMatrix(), p = cam->getProjectionMatrix();
    std::memcpy(view16, &v[0][0], 64);
    std::memcpy(proj16, &p[0][0], 64);
}
// The reference's OrbitCamera (OrbitCamera.h:8-75, header only) driven by window events the way Application forwards
// them (main.cpp:139-155,235-236): rows of 4 doubles {0,x,y,dt} cursor, {1,button,action,-}, {2,yoffset,-,dt} scroll.
// Every sequence starts with a cursor event while no button is down (the reference leaves p_xpos / p_ypos
// uninitialised until the first one).
void ref_orbit_events(float distance, float orbit_rate, float scroll_rate, const double* ev, uint32_t n, float world16[16]) {
    auto scene = std::make_shared<Node>();
    auto orbit = std::make_shared<OrbitCamera>(distance);
    scene->addChild(orbit);
    orbit->orbitRate = orbit_rate;
    orbit->scrollRate = scroll_rate;
    for (uint32_t i = 0; i < n; ++i) {
        const double* e = ev + 4 * static_cast<size_t>(i);
        if (e[0] == 0.0) orbit->onMouseMove(e[1], e[2], e[3]);
        else if (e[0] == 1.0) orbit->onMouseButton(static_cast<int>(e[1]), static_cast<int>(e[2]));
        else orbit->onScroll(e[1], e[3]);
    }
    scene->updateWorldMatrix(glm::mat4(1.0f));
    std::memcpy(world16, &orbit->camera->worldMatrix[0][0], 64);
}

// Orbit pose exactly as main.cpp:90-93,469-486 builds it: scene -> OrbitCamera(distance)
// -> camera; `steps` successive Node::rotate(axis, angle) calls on the orbit node.
void ref_orbit_world(float distance, const float axis[3], float angle, uint32_t steps, float world16[16],
                     float rot3_out[9]) {
    auto scene = std::make_shared<Node>();
    auto orbit = std::make_shared<Node>();
    auto cam = std::make_shared<Camera>();
    orbit->children.push_back(cam);
    cam->position = glm::vec3(0.0f, 0.0f, distance);
    scene->addChild(orbit);
    for (uint32_t s = 0; s < steps; ++s) orbit->rotate(glm::vec3(axis[0], axis[1], axis[2]), angle);
    scene->updateWorldMatrix(glm::mat4(1.0f));
    std::memcpy(world16, &cam->worldMatrix[0][0], 64);
    if (rot3_out) std::memcpy(rot3_out, &orbit->rotation[0][0], 36);
}
void ref_viewproj(const float world16[16], float fov, float aspect, float znear, float zfar, float vp16[16]) {
    auto cam = make_camera(world16, fov, aspect, znear, zfar);
    glm::mat4 vp = cam->getProjectionMatrix() * cam->getViewMatrix();  // BB.hpp:60-61
    std::memcpy(vp16, &vp[0][0], 64);
}

// ---- BB.hpp ----
float ref_bb_screen_area(const float bb16[16], const float world16[16], float fov, float aspect,
                         float znear, float zfar) {
    BB bb;
    std::memcpy(&bb.transform[0][0], bb16, 64);
    return bb.screen_area(make_camera(world16, fov, aspect, znear, zfar));
}

// ---- Octree ----
int ref_octree_build(const float* split14, uint64_t n) {
    SplatSplitVector v(n);
    std::memcpy(v.data(), split14, n * 56);
    auto oc = std::make_unique<Octree>();
    oc->build(v);
    oc->generate();
    int h = g_next++;
    g_octrees[h] = std::move(oc);
    return h;
}
uint64_t ref_octree_size(int h) { return g_octrees.at(h)->splats.size(); }
// bb16: n x 16 (node->bb.transform), center3/half3: n x 3 (bb.center(), bb.size()/2)
void ref_octree_export(int h, float* splat20, float* bb16, float* center3, float* half3,
                       uint32_t* first_child, uint32_t* n_children, uint32_t* node_splat) {
    Octree& oc = *g_octrees.at(h);
    OctreeFlat f = flatten(oc);
    std::memcpy(splat20, oc.splats.data(), oc.splats.size() * 80);
    for (size_t i = 0; i < f.nodes.size(); ++i) {
        std::memcpy(bb16 + 16 * i, &f.nodes[i]->bb.transform[0][0], 64);
        glm::vec3 c = f.nodes[i]->bb.center(), hs = f.nodes[i]->bb.size() / 2.0f;
        for (int k = 0; k < 3; ++k) { center3[3 * i + k] = c[k]; half3[3 * i + k] = hs[k]; }
        first_child[i] = f.first_child[i];
        n_children[i] = f.n_children[i];
        node_splat[i] = f.nodes[i]->indices.size() == 1 ? f.nodes[i]->indices[0] : UINT32_MAX;
    }
}
uint64_t ref_octree_get_indices(int h, const float world16[16], float fov, float aspect, float znear,
                                float zfar, float min_screen_area, uint32_t* out) {
    Indices ind = g_octrees.at(h)->get_indices(make_camera(world16, fov, aspect, znear, zfar), min_screen_area);
    std::memcpy(out, ind.data(), ind.size() * 4);
    return ind.size();
}

// ---- HC ----
int ref_hc_build(const float* splat20, uint64_t n) {
    SplatVector v(n);
    std::memcpy(v.data(), splat20, n * 80);
    auto hc = std::make_unique<HC>();
    hc->build(v, false);
    int h = g_next++;
    g_hcs[h] = std::move(hc);
    return h;
}
uint64_t ref_hc_size(int h) { return g_hcs.at(h)->splats.size(); }
uint64_t ref_hc_roots(int h) { return g_hcs.at(h)->nodes.size(); }
void ref_hc_export(int h, float* splat20, uint32_t* child2, float* error, uint32_t* roots) {
    HC& hc = *g_hcs.at(h);
    std::memcpy(splat20, hc.splats.data(), hc.splats.size() * 80);
    std::vector<uint32_t> c2, r;
    std::vector<float> e;
    flatten_hc(hc, 0, c2, e, r);
    std::memcpy(child2, c2.data(), c2.size() * 4);
    std::memcpy(error, e.data(), e.size() * 4);
    std::memcpy(roots, r.data(), r.size() * 4);
}
uint64_t ref_hc_get_indices_depth(int h, uint32_t depth, uint32_t* out) {
    Indices ind = g_hcs.at(h)->get_indices_depth(depth);
    std::memcpy(out, ind.data(), ind.size() * 4);
    return ind.size();
}
uint64_t ref_hc_get_indices(int h, const float world16[16], float threshold, uint32_t* out) {
    Indices ind = g_hcs.at(h)->get_indices(make_camera(world16, 45.0f, 1.0f, 0.1f, 100.0f), threshold,
                                           HC::MetricWeights{});
    std::memcpy(out, ind.data(), ind.size() * 4);
    return ind.size();
}

// ---- GridHC ----
int ref_gridhc_build(const float* splat20, uint64_t n, uint32_t sx, uint32_t sy, uint32_t sz) {
    SplatVector v(n);
    std::memcpy(v.data(), splat20, n * 80);
    auto g = std::make_unique<GridHC>();
    g->subdivisions = glm::uvec3(sx, sy, sz);
    g->build(v);
    int h = g_next++;
    g_grids[h] = std::move(g);
    return h;
}
uint64_t ref_gridhc_size(int h) { return g_grids.at(h)->splats.size(); }
uint64_t ref_gridhc_roots(int h) {
    uint64_t r = 0;
    for (auto& c : g_grids.at(h)->cells) if (!c->empty()) r += c->hc.nodes.size();
    return r;
}
uint64_t ref_gridhc_cells(int h) {
    uint64_t r = 0;
    for (auto& c : g_grids.at(h)->cells) if (!c->empty()) r++;
    return r;
}
// cell_root_start has (non-empty cells + 1) entries into roots; ids are global
// (cell offset applied, GridHC.cpp:79-88).
void ref_gridhc_export(int h, float* splat20, uint32_t* child2, float* error, uint32_t* roots,
                       uint32_t* cell_root_start) {
    GridHC& g = *g_grids.at(h);
    std::memcpy(splat20, g.splats.data(), g.splats.size() * 80);
    std::vector<uint32_t> c2, r;
    std::vector<float> e;
    uint32_t offset = 0, ci = 0;
    for (auto& cell : g.cells) {
        if (cell->empty()) continue;
        cell_root_start[ci++] = static_cast<uint32_t>(r.size());
        flatten_hc(cell->hc, offset, c2, e, r);
        offset += cell->hc.splats.size();
    }
    cell_root_start[ci] = static_cast<uint32_t>(r.size());
    std::memcpy(child2, c2.data(), c2.size() * 4);
    std::memcpy(error, e.data(), e.size() * 4);
    std::memcpy(roots, r.data(), r.size() * 4);
}
uint64_t ref_gridhc_get_indices_error(int h, uint32_t depth, float error, uint32_t* out) {
    Indices ind = g_grids.at(h)->get_indices_error(depth, error);
    std::memcpy(out, ind.data(), ind.size() * 4);
    return ind.size();
}
uint64_t ref_gridhc_get_indices(int h, const float world16[16], float threshold, uint32_t* out) {
    Indices ind = g_grids.at(h)->get_indices(make_camera(world16, 45.0f, 1.0f, 0.1f, 100.0f), threshold,
                                             HC::MetricWeights{});
    std::memcpy(out, ind.data(), ind.size() * 4);
    return ind.size();
}

void ref_release(int h) { g_octrees.erase(h); g_hcs.erase(h); g_grids.erase(h); }

}  // extern "C"
