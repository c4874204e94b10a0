// gsx_splat.hpp — the reference's splat data model (Splat.h:5-29) and the load-time
// conversions that feed the device path (Splat.h:38-91, ResourceManager.cpp:36-65),
// on gsx_math (glm operation order).  Host-only; compile with -ffp-contract=off.
#pragma once
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "gsx_math.hpp"

namespace gsx {

#pragma pack(push, 1)
struct SplatRaw {          // 32-byte .splat file record (Splat.h:5-10)
    float position[3];
    float scale[3];
    uint8_t color[4];
    uint8_t rotation[4];   // (w, x, y, z) as (b - 128) / 128
};
#pragma pack(pop)
static_assert(sizeof(SplatRaw) == 32, "SplatRaw is the on-disk record");

struct SplatSplit {        // Splat.h:12-17
    vec3 position, scale;
    vec4 color, rotation;
};
static_assert(sizeof(SplatSplit) == 56, "SplatSplit layout");

struct Splat {             // Splat.h:19-29 — the 80-byte record gsx_upload_splats takes
    mat4 transform;
    vec4 color;
    float weight() const {  // Splat.h:24-28 (1.333 is a double literal there)
        const float alpha = color.w;
        const float volume = static_cast<float>(1.333 * static_cast<double>(std::sqrt(determinant(transform))));
        return alpha * volume;
    }
};
static_assert(sizeof(Splat) == 80, "Splat is the 80-byte GPU record");

using SplatSplitVector = std::vector<SplatSplit>;
using SplatVector = std::vector<Splat>;
using Indices = std::vector<uint32_t>;

// Splat.h:38-58
inline mat4 quat_to_rot(vec4 q) {
    mat4 rot(1.0f);
    const float qr = q[0], qi = q[1], qj = q[2], qk = q[3];
    rot[0][0] = 1 - 2 * (qj * qj + qk * qk);
    rot[0][1] = 2 * (qi * qj + qk * qr);
    rot[0][2] = 2 * (qi * qk - qj * qr);
    rot[1][0] = 2 * (qi * qj - qk * qr);
    rot[1][1] = 1 - 2 * (qi * qi + qk * qk);
    rot[1][2] = 2 * (qj * qk + qi * qr);
    rot[2][0] = 2 * (qi * qk + qj * qr);
    rot[2][1] = 2 * (qj * qk - qi * qr);
    rot[2][2] = 1 - 2 * (qi * qi + qj * qj);
    return rot;
}

// Splat.h:60-75
inline SplatSplit raw_to_split(const SplatRaw& r) {
    SplatSplit s;
    s.position = {r.position[0], r.position[1], r.position[2]};
    s.scale = {r.scale[0], r.scale[1], r.scale[2]};
    vec4 rot(static_cast<float>(r.rotation[0]), static_cast<float>(r.rotation[1]), static_cast<float>(r.rotation[2]),
             static_cast<float>(r.rotation[3]));
    rot = (rot - vec4(128.0f)) / 128.0f;
    s.rotation = normalize(rot);
    const float expo[4] = {2.2f, 2.2f, 2.2f, 1.0f};
    for (int k = 0; k < 4; ++k) s.color[k] = std::pow(static_cast<float>(r.color[k]) / 255.0f, expo[k]);
    return s;
}

// Splat.h:77-91
inline Splat split_to_splat(const SplatSplit& s) {
    mat4 xyz = quat_to_rot(s.rotation);
    xyz[0] = xyz[0] * s.scale.x;
    xyz[1] = xyz[1] * s.scale.y;
    xyz[2] = xyz[2] * s.scale.z;
    Splat out;
    out.transform = xyz * transpose(xyz);
    out.transform[3] = vec4(s.position, 1.0f);
    out.color = s.color;
    return out;
}

// ResourceManager.cpp:52-62
inline void center_splats(SplatSplitVector& splats) {
    vec3 c(0.0f);
    for (const auto& s : splats) c = c + s.position;
    c = c / static_cast<float>(splats.size());
    for (auto& s : splats) s.position = s.position - c;
}

// ResourceManager::loadSplatsRaw (ResourceManager.cpp:36-65): empty vector when the
// file cannot be opened; a trailing partial record is ignored, as `file.read` does.
inline SplatSplitVector loadSplatsRaw(const std::string& path, bool center) {
    std::ifstream file(path, std::ios::binary);
    if (!file.is_open()) return {};
    SplatSplitVector splats;
    SplatRaw raw;
    while (file.read(reinterpret_cast<char*>(&raw), sizeof(SplatRaw))) splats.push_back(raw_to_split(raw));
    if (center && !splats.empty()) center_splats(splats);
    return splats;
}

}  // namespace gsx
