// gsx_host.cpp — host-only entry points of libgsx (no GPU needed): the load-time
// half of ResourceManager and the camera/scene math that produces the path's
// inputs.  Plain C++ on gaussian-splatting_amd/host/*.hpp; built with
// -ffp-contract=off so every f32 operation is rounded once, like the reference's
// g++ -O3 build.
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/gsx.h"
#include "../host/gsx_lod_host.hpp"
#include "../host/gsx_scene.hpp"
#include "../host/gsx_splat.hpp"
#include "../host/gsx_ply.hpp"

using namespace gsx;

extern "C" {

int gsx_host_convert_raw(const void* raw32, uint64_t n, int center, void* aos80_out) {
    if ((!raw32 || !aos80_out) && n) return GSX_E_INVALID;
    const SplatRaw* raw = static_cast<const SplatRaw*>(raw32);
    SplatSplitVector split;
    try {
        split.resize(n);
    } catch (const std::bad_alloc&) {
        return GSX_E_NOMEM;
    } catch (const std::exception&) {      // e.g. std::length_error from an absurd element count: nothing crosses the C ABI
        return GSX_E_INVALID;
    }
    // per-record conversions are independent (and dominated by three powf per record): all host cores; the mean for
    // centring is a sequential sum in record order, as in ResourceManager.cpp:52-62
    const unsigned threads = lod_detail::host_threads();
    lod_detail::parallel_for(n, 1 << 14, threads, [&](size_t i) { split[i] = raw_to_split(raw[i]); });
    if (center && n) center_splats(split);
    Splat* out = static_cast<Splat*>(aos80_out);
    lod_detail::parallel_for(n, 1 << 14, threads, [&](size_t i) { out[i] = split_to_splat(split[i]); });
    return GSX_OK;
}

int gsx_host_load_splat_file(const char* path, int center, void** aos80_out, uint64_t* n_out) {
    if (!path || !aos80_out || !n_out) return GSX_E_INVALID;
    *aos80_out = nullptr;
    *n_out = 0;
    SplatSplitVector split;
    try {
        split = loadSplatsRaw(path, center != 0);
    } catch (const std::bad_alloc&) {
        return GSX_E_NOMEM;
    } catch (const std::exception&) {      // e.g. std::length_error from an absurd element count: nothing crosses the C ABI
        return GSX_E_INVALID;
    }
    if (split.empty()) return GSX_OK;  // missing file == empty scene, like the reference
    Splat* out = static_cast<Splat*>(std::malloc(split.size() * sizeof(Splat)));
    if (!out) return GSX_E_NOMEM;
    for (size_t i = 0; i < split.size(); ++i) out[i] = split_to_splat(split[i]);
    *aos80_out = out;
    *n_out = split.size();
    return GSX_OK;
}

void gsx_host_free(void* p) { std::free(p); }

int gsx_host_load_ply_raw(const char* path, void** raw32_out, uint64_t* n_out) {
    if (!path || !raw32_out || !n_out) return GSX_E_INVALID;
    *raw32_out = nullptr;
    *n_out = 0;
    std::vector<SplatRaw> raw;
    PlyStatus st;
    try {
        st = load_ply_raw(path, raw);
    } catch (const std::bad_alloc&) {
        return GSX_E_NOMEM;
    } catch (const std::exception&) {      // e.g. std::length_error from an absurd element count: nothing crosses the C ABI
        return GSX_E_INVALID;
    }
    if (st == PlyStatus::CannotOpen) return GSX_OK;              // missing file == empty scene, like loadSplatsRaw
    if (st == PlyStatus::Unsupported) return GSX_E_UNSUPPORTED;
    if (st != PlyStatus::Ok) return GSX_E_INVALID;
    if (raw.empty()) return GSX_OK;
    void* out = std::malloc(raw.size() * sizeof(SplatRaw));
    if (!out) return GSX_E_NOMEM;
    std::memcpy(out, raw.data(), raw.size() * sizeof(SplatRaw));
    *raw32_out = out;
    *n_out = raw.size();
    return GSX_OK;
}

int gsx_host_load_ply_file(const char* path, int center, void** aos80_out, uint64_t* n_out) {
    if (!aos80_out || !n_out) return GSX_E_INVALID;
    *aos80_out = nullptr;
    void* raw = nullptr;
    uint64_t n = 0;
    int rc = gsx_host_load_ply_raw(path, &raw, &n);
    *n_out = n;
    if (rc != GSX_OK || n == 0) return rc;
    void* out = std::malloc(n * sizeof(Splat));
    if (!out) { std::free(raw); *n_out = 0; return GSX_E_NOMEM; }
    rc = gsx_host_convert_raw(raw, n, center, out);
    std::free(raw);
    if (rc != GSX_OK) { std::free(out); *n_out = 0; return rc; }
    *aos80_out = out;
    return GSX_OK;
}

int gsx_host_camera(const float world16[16], float fov_deg, float aspect, float znear, float zfar, float view16[16],
                    float proj16[16], float viewproj16[16]) {
    if (!world16) return GSX_E_INVALID;
    Camera cam;
    std::memcpy(cam.worldMatrix.data(), world16, 64);
    cam.fov = fov_deg; cam.aspect = aspect; cam.near = znear; cam.far = zfar;
    const mat4 v = cam.getViewMatrix(), p = cam.getProjectionMatrix();
    if (view16) std::memcpy(view16, v.data(), 64);
    if (proj16) std::memcpy(proj16, p.data(), 64);
    if (viewproj16) {
        const mat4 vp = p * v;  // BB.hpp:60-61
        std::memcpy(viewproj16, vp.data(), 64);
    }
    return GSX_OK;
}

int gsx_host_orbit_pose(float distance, const float* axes3, const float* angles, uint32_t n_rot, float world16[16]) {
    if (!world16 || (n_rot && (!axes3 || !angles))) return GSX_E_INVALID;
    auto scene = std::make_shared<Node>();
    auto orbit = std::make_shared<OrbitCamera>(distance);
    scene->addChild(orbit);
    for (uint32_t i = 0; i < n_rot; ++i) orbit->rotate(vec3(axes3[3 * i], axes3[3 * i + 1], axes3[3 * i + 2]), angles[i]);
    scene->updateWorldMatrix(mat4(1.0f));   // main.cpp:483-486
    std::memcpy(world16, orbit->camera->worldMatrix.data(), 64);
    return GSX_OK;
}

int gsx_host_orbit_events(float distance, float orbit_rate, float scroll_rate, const double* events4, uint32_t n_events, float world16[16]) {
    if (!world16 || (n_events && !events4)) return GSX_E_INVALID;
    auto scene = std::make_shared<Node>();
    auto orbit = std::make_shared<OrbitCamera>(distance);
    scene->addChild(orbit);
    orbit->orbitRate = orbit_rate;       // main.cpp:235-236
    orbit->scrollRate = scroll_rate;
    for (uint32_t i = 0; i < n_events; ++i) {
        const double* e = events4 + 4 * static_cast<size_t>(i);
        if (e[0] == 0.0) orbit->onMouseMove(e[1], e[2], e[3]);                                              // main.cpp:139-141
        else if (e[0] == 1.0) orbit->onMouseButton(static_cast<int>(e[1]), static_cast<int>(e[2]));        // main.cpp:143-151
        else if (e[0] == 2.0) orbit->onScroll(e[1], e[3]);                                                  // main.cpp:153-155
        else return GSX_E_INVALID;
    }
    scene->updateWorldMatrix(mat4(1.0f));   // main.cpp:483-486
    std::memcpy(world16, orbit->camera->worldMatrix.data(), 64);
    return GSX_OK;
}

// ---- host-side LOD structures (construction is load-time CPU work) -----------------
}  // extern "C"

struct gsx_lod {
    int kind = 0;  // 1 octree, 2 hc, 3 gridhc
    Octree octree;
    HC hc;
    GridHC grid;
    OctreeFlat oflat;
    HcFlat hflat;
    const SplatVector& splats() const { return kind == 1 ? octree.splats : (kind == 2 ? hc.splats : grid.splats); }
};

extern "C" {

int gsx_host_octree_build(const void* raw32, uint64_t n, int center, gsx_lod** out) {
    if (!out || (!raw32 && n)) return GSX_E_INVALID;
    *out = nullptr;
    try {
        const SplatRaw* raw = static_cast<const SplatRaw*>(raw32);
        SplatSplitVector split(n);
        lod_detail::parallel_for(n, 1 << 14, lod_detail::host_threads(), [&](size_t i) { split[i] = raw_to_split(raw[i]); });
        if (center && n) center_splats(split);
        auto* l = new gsx_lod();
        l->kind = 1;
        l->octree.build(std::move(split));      // SplatMeshOctree.hpp:72-76
        l->octree.generate();
        l->oflat = l->octree.flatten();
        *out = l;
    } catch (const std::bad_alloc&) { return GSX_E_NOMEM; } catch (const std::exception&) { return GSX_E_INTERNAL; }
    return GSX_OK;
}

int gsx_host_hc_build(const void* aos80, uint64_t n, gsx_lod** out) {
    if (!out || (!aos80 && n)) return GSX_E_INVALID;
    *out = nullptr;
    try {
        const Splat* sp = static_cast<const Splat*>(aos80);
        auto* l = new gsx_lod();
        l->kind = 2;
        l->hc.build(SplatVector(sp, sp + n), false);   // SplatMeshHC.hpp:85
        l->hflat = l->hc.flatten();
        *out = l;
    } catch (const std::bad_alloc&) { return GSX_E_NOMEM; } catch (const std::exception&) { return GSX_E_INTERNAL; }
    return GSX_OK;
}

int gsx_host_gridhc_build(const void* aos80, uint64_t n, uint32_t sx, uint32_t sy, uint32_t sz, gsx_lod** out) {
    if (!out || (!aos80 && n) || !sx || !sy || !sz) return GSX_E_INVALID;
    *out = nullptr;
    try {
        const Splat* sp = static_cast<const Splat*>(aos80);
        auto* l = new gsx_lod();
        l->kind = 3;
        l->grid.subdivisions[0] = sx; l->grid.subdivisions[1] = sy; l->grid.subdivisions[2] = sz;
        l->grid.build(SplatVector(sp, sp + n));        // SplatMeshGridHC.cpp:46
        l->hflat = l->grid.flatten();
        *out = l;
    } catch (const std::bad_alloc&) { return GSX_E_NOMEM; } catch (const std::exception&) { return GSX_E_INTERNAL; }
    return GSX_OK;
}

void gsx_host_lod_free(gsx_lod* l) { delete l; }

uint64_t gsx_host_lod_size(const gsx_lod* l) { return l ? l->splats().size() : 0; }
uint64_t gsx_host_lod_roots(const gsx_lod* l) { return l ? l->hflat.roots.size() : 0; }
uint64_t gsx_host_lod_cells(const gsx_lod* l) { return (l && !l->hflat.cell_root_start.empty()) ? l->hflat.cell_root_start.size() - 1 : 0; }

int gsx_host_lod_splats(const gsx_lod* l, void* aos80_out) {
    if (!l || !aos80_out) return GSX_E_INVALID;
    std::memcpy(aos80_out, l->splats().data(), l->splats().size() * sizeof(Splat));
    return GSX_OK;
}

int gsx_host_lod_flat_octree(const gsx_lod* l, float* center3, float* half3, uint32_t* parent, uint8_t* is_leaf,
                             uint32_t* first_child, uint32_t* n_children) {
    if (!l || l->kind != 1) return GSX_E_INVALID;
    const OctreeFlat& f = l->oflat;
    if (center3) std::memcpy(center3, f.center3.data(), f.center3.size() * 4);
    if (half3) std::memcpy(half3, f.half3.data(), f.half3.size() * 4);
    if (parent) std::memcpy(parent, f.parent.data(), f.parent.size() * 4);
    if (is_leaf) std::memcpy(is_leaf, f.is_leaf.data(), f.is_leaf.size());
    if (first_child) std::memcpy(first_child, f.first_child.data(), f.first_child.size() * 4);
    if (n_children) std::memcpy(n_children, f.n_children.data(), f.n_children.size() * 4);
    return GSX_OK;
}

int gsx_host_lod_flat_hc(const gsx_lod* l, uint32_t* parent, uint8_t* is_leaf, float* error, float* weight, uint32_t* child2,
                         uint32_t* roots, uint32_t* cell_root_start) {
    if (!l || (l->kind != 2 && l->kind != 3)) return GSX_E_INVALID;
    const HcFlat& f = l->hflat;
    if (parent) std::memcpy(parent, f.parent.data(), f.parent.size() * 4);
    if (is_leaf) std::memcpy(is_leaf, f.is_leaf.data(), f.is_leaf.size());
    if (error) std::memcpy(error, f.error.data(), f.error.size() * 4);
    if (weight) std::memcpy(weight, f.weight.data(), f.weight.size() * 4);
    if (child2) std::memcpy(child2, f.child2.data(), f.child2.size() * 4);
    if (roots) std::memcpy(roots, f.roots.data(), f.roots.size() * 4);
    if (cell_root_start && !f.cell_root_start.empty()) std::memcpy(cell_root_start, f.cell_root_start.data(), f.cell_root_start.size() * 4);
    return GSX_OK;
}

int gsx_host_lod_cut_depth(gsx_lod* l, uint32_t depth, uint32_t* idx_out, uint64_t capacity, uint64_t* m_out) {
    if (!l || !m_out) return GSX_E_INVALID;
    Indices ind;
    try {
        if (l->kind == 1) ind = l->octree.get_indices_depth(depth);            // Octree.hpp:131-162
        else if (l->kind == 2) ind = l->hc.get_indices_depth(depth);            // HC.cpp:134-168
        else ind = l->grid.get_indices_error(depth, 0.0f);                      // GridHC.cpp:76-92
    } catch (const std::bad_alloc&) { return GSX_E_NOMEM; } catch (const std::exception&) { return GSX_E_INTERNAL; }
    *m_out = ind.size();
    if (idx_out) {
        if (capacity < ind.size()) return GSX_E_INVALID;
        std::memcpy(idx_out, ind.data(), ind.size() * 4);
    }
    return GSX_OK;
}

// CPU reference cuts of the host structures (what the reference's render() computes per frame);
// the device path is gsx_select_octree / gsx_select_hc.
int gsx_host_lod_cut_camera(gsx_lod* l, const float world16[16], float fov_deg, float aspect, float znear, float zfar,
                            float threshold, uint32_t* idx_out, uint64_t capacity, uint64_t* m_out) {
    if (!l || !m_out || !world16) return GSX_E_INVALID;
    auto cam = std::make_shared<Camera>();
    std::memcpy(cam->worldMatrix.data(), world16, 64);
    cam->fov = fov_deg; cam->aspect = aspect; cam->near = znear; cam->far = zfar;
    Indices ind;
    try {
        if (l->kind == 1) ind = l->octree.get_indices(cam, threshold);                         // Octree.cpp:189-216
        else if (l->kind == 2) ind = l->hc.get_indices(cam, threshold, HC::MetricWeights{});   // HC.cpp:100-132
        else ind = l->grid.get_indices(cam, threshold, HC::MetricWeights{});                   // GridHC.cpp:94-111
    } catch (const std::bad_alloc&) { return GSX_E_NOMEM; } catch (const std::exception&) { return GSX_E_INTERNAL; }
    *m_out = ind.size();
    if (idx_out) {
        if (capacity < ind.size()) return GSX_E_INVALID;
        std::memcpy(idx_out, ind.data(), ind.size() * 4);
    }
    return GSX_OK;
}

}  // extern "C"
