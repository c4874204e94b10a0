// gsx_host.cpp — host-only entry points of libgsx (no GPU needed): the load-time
// half of ResourceManager and the camera/scene math that produces the path's
// inputs.  Plain C++ on gaussian-splatting_amd/host/*.hpp; built with
// -ffp-contract=off so every f32 operation is rounded once, like the reference's
// g++ -O3 build.
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/gsx.h"
#include "../host/gsx_scene.hpp"
#include "../host/gsx_splat.hpp"

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
    }
    for (uint64_t i = 0; i < n; ++i) split[i] = raw_to_split(raw[i]);
    if (center && n) center_splats(split);
    Splat* out = static_cast<Splat*>(aos80_out);
    for (uint64_t i = 0; i < n; ++i) out[i] = split_to_splat(split[i]);
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

}  // extern "C"
