// orc_core.cpp — CPU ORACLE (test infrastructure, never shipped in the product
// path): data model, depth keys, sort order, camera matrices.
// Build: g++ -std=c++17 -O2 -ffp-contract=off (one rounding per f32 operation).
// See orc.h for the pinning status of each function.
#include "orc.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

// glm/detail/type_mat4x4.inl:630-648 — column-major, Result[j] =
// ((A0*B[j][0] + A1*B[j][1]) + A2*B[j][2]) + A3*B[j][3]
void mul44(const float* a, const float* b, float* out) {
    float r[16];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float t = a[0 * 4 + i] * b[j * 4 + 0];
            t = t + a[1 * 4 + i] * b[j * 4 + 1];
            t = t + a[2 * 4 + i] * b[j * 4 + 2];
            t = t + a[3 * 4 + i] * b[j * 4 + 3];
            r[j * 4 + i] = t;
        }
    std::memcpy(out, r, sizeof r);
}

}  // namespace

extern "C" {

// Splat.h:60-75 raw_to_split
void orc_raw_to_split(const void* raw32, uint64_t n, float* split14) {
    const uint8_t* p = static_cast<const uint8_t*>(raw32);
    for (uint64_t i = 0; i < n; ++i, p += 32) {
        float f[6];
        std::memcpy(f, p, 24);
        const uint8_t* col = p + 24;
        const uint8_t* rot = p + 28;
        float* o = split14 + i * 14;
        for (int k = 0; k < 6; ++k) o[k] = f[k];  // position, scale
        // color = pow(color / 255, (2.2, 2.2, 2.2, 1.0))   (Splat.h:71-72)
        const float expo[4] = {2.2f, 2.2f, 2.2f, 1.0f};
        for (int k = 0; k < 4; ++k) o[6 + k] = std::pow(static_cast<float>(col[k]) / 255.0f, expo[k]);
        // rotation = normalize((rotation - 128) / 128)     (Splat.h:67-68)
        float r[4];
        for (int k = 0; k < 4; ++k) r[k] = (static_cast<float>(rot[k]) - 128.0f) / 128.0f;
        // glm normalize: v * (1 / sqrt(dot)), vec4 dot = (x+y)+(z+w)
        float d = (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
        float inv = 1.0f / std::sqrt(d);
        for (int k = 0; k < 4; ++k) o[10 + k] = r[k] * inv;
    }
}

// ResourceManager.cpp:52-62 mean-centering
void orc_center_split(float* split14, uint64_t n) {
    float c[3] = {0.0f, 0.0f, 0.0f};
    for (uint64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) c[k] += split14[i * 14 + k];
    const float fn = static_cast<float>(n);
    for (int k = 0; k < 3; ++k) c[k] /= fn;
    for (uint64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) split14[i * 14 + k] -= c[k];
}

// Splat.h:38-58 quat_to_rot, Splat.h:77-91 split_to_splat
void orc_split_to_splat(const float* split14, uint64_t n, float* splat20) {
    for (uint64_t i = 0; i < n; ++i) {
        const float* s = split14 + i * 14;
        const float qr = s[10], qi = s[11], qj = s[12], qk = s[13];
        float m[16] = {0};  // mat4(mat3(1)) then overwritten
        m[15] = 1.0f;
        m[0] = 1 - 2 * (qj * qj + qk * qk);
        m[1] = 2 * (qi * qj + qk * qr);
        m[2] = 2 * (qi * qk - qj * qr);
        m[4] = 2 * (qi * qj - qk * qr);
        m[5] = 1 - 2 * (qi * qi + qk * qk);
        m[6] = 2 * (qj * qk + qi * qr);
        m[8] = 2 * (qi * qk + qj * qr);
        m[9] = 2 * (qj * qk - qi * qr);
        m[10] = 1 - 2 * (qi * qi + qj * qj);
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 4; ++r) m[c * 4 + r] *= s[3 + c];  // xyz[c] *= scale[c]
        float t[16];  // transpose(xyz)
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) t[c * 4 + r] = m[r * 4 + c];
        float cov[16];
        mul44(m, t, cov);  // mat4 * mat4
        float* o = splat20 + i * 20;
        std::memcpy(o, cov, sizeof cov);
        o[12] = s[0]; o[13] = s[1]; o[14] = s[2]; o[15] = 1.0f;
        for (int k = 0; k < 4; ++k) o[16 + k] = s[6 + k];
    }
}

// SplatMesh.h:90-94 + glm distance = length(p1 - p0) = sqrt((dx*dx + dy*dy) + dz*dz)
static inline float key_distance(const float* splat20, uint32_t i, const float cam[3]) {
    const float* p = splat20 + static_cast<uint64_t>(i) * 20 + 12;
    const float dx = cam[0] - p[0], dy = cam[1] - p[1], dz = cam[2] - p[2];
    return std::sqrt(dx * dx + dy * dy + dz * dz);
}

void orc_distance_keys(const float* splat20, uint64_t n, const uint32_t* idx, uint64_t m,
                       const float cam[3], float* dist_out) {
    (void)n;
    for (uint64_t k = 0; k < m; ++k) {
        const uint32_t i = idx ? idx[k] : static_cast<uint32_t>(k);
        dist_out[i] = key_distance(splat20, i, cam);
    }
}

void orc_sort_contract(const float* splat20, uint64_t n, const uint32_t* idx, uint64_t m,
                       const float cam[3], uint32_t* out) {
    std::vector<float> dist(n);
    orc_distance_keys(splat20, n, idx, m, cam, dist.data());
    if (idx) std::memcpy(out, idx, m * sizeof(uint32_t));
    else std::iota(out, out + m, 0u);
    // total order: distance descending (SplatMesh.h:105-107), ties by ascending
    // index (north_star contract; the reference's std::sort leaves ties unspecified)
    std::sort(out, out + m, [&](uint32_t a, uint32_t b) {
        if (dist[a] != dist[b]) return dist[a] > dist[b];
        return a < b;
    });
}

double orc_sort_reference_port(const float* splat20, uint64_t n, uint32_t* idx, uint64_t m,
                               const float cam[3]) {
    auto start = std::chrono::high_resolution_clock::now();          // SplatMesh.h:88
    std::vector<float> distances(n);                                  // SplatMesh.h:90
    for (uint64_t k = 0; k < m; ++k) {                                // SplatMesh.h:91-94
        const uint32_t i = idx[k];
        distances[i] = key_distance(splat20, i, cam);
    }
    std::sort(idx, idx + m, [&](uint32_t a, uint32_t b) {             // SplatMesh.h:105-107
        return distances[a] > distances[b];
    });
    auto end = std::chrono::high_resolution_clock::now();             // SplatMesh.h:111
    return std::chrono::duration<double>(end - start).count();
}

// glm/ext/matrix_clip_space.inl:249-262 perspectiveRH_NO; Camera.h:32-34
void orc_perspective(float fovy_deg, float aspect, float zn, float zf, float out[16]) {
    // glm::radians(deg) = deg * 0.01745329251994329576923690768489 (as float)
    const float fovy = fovy_deg * static_cast<float>(0.01745329251994329576923690768489);
    const float t = std::tan(fovy / 2.0f);
    std::memset(out, 0, 16 * sizeof(float));
    out[0] = 1.0f / (aspect * t);
    out[5] = 1.0f / t;
    out[10] = -(zf + zn) / (zf - zn);
    out[11] = -1.0f;
    out[14] = -(2.0f * zf * zn) / (zf - zn);
}

// glm/detail/func_matrix.inl:294-351
void orc_mat4_inverse(const float m_[16], float out[16]) {
    auto m = [&](int c, int r) { return m_[c * 4 + r]; };
    float c00 = m(2,2) * m(3,3) - m(3,2) * m(2,3);
    float c02 = m(1,2) * m(3,3) - m(3,2) * m(1,3);
    float c03 = m(1,2) * m(2,3) - m(2,2) * m(1,3);
    float c04 = m(2,1) * m(3,3) - m(3,1) * m(2,3);
    float c06 = m(1,1) * m(3,3) - m(3,1) * m(1,3);
    float c07 = m(1,1) * m(2,3) - m(2,1) * m(1,3);
    float c08 = m(2,1) * m(3,2) - m(3,1) * m(2,2);
    float c10 = m(1,1) * m(3,2) - m(3,1) * m(1,2);
    float c11 = m(1,1) * m(2,2) - m(2,1) * m(1,2);
    float c12 = m(2,0) * m(3,3) - m(3,0) * m(2,3);
    float c14 = m(1,0) * m(3,3) - m(3,0) * m(1,3);
    float c15 = m(1,0) * m(2,3) - m(2,0) * m(1,3);
    float c16 = m(2,0) * m(3,2) - m(3,0) * m(2,2);
    float c18 = m(1,0) * m(3,2) - m(3,0) * m(1,2);
    float c19 = m(1,0) * m(2,2) - m(2,0) * m(1,2);
    float c20 = m(2,0) * m(3,1) - m(3,0) * m(2,1);
    float c22 = m(1,0) * m(3,1) - m(3,0) * m(1,1);
    float c23 = m(1,0) * m(2,1) - m(2,0) * m(1,1);
    const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07};
    const float f2[4] = {c08, c08, c10, c11}, f3[4] = {c12, c12, c14, c15};
    const float f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    const float v0[4] = {m(1,0), m(0,0), m(0,0), m(0,0)};
    const float v1[4] = {m(1,1), m(0,1), m(0,1), m(0,1)};
    const float v2[4] = {m(1,2), m(0,2), m(0,2), m(0,2)};
    const float v3[4] = {m(1,3), m(0,3), m(0,3), m(0,3)};
    const float sa[4] = {+1, -1, +1, -1}, sb[4] = {-1, +1, -1, +1};
    float inv[16];
    for (int k = 0; k < 4; ++k) {
        inv[0 * 4 + k] = (v1[k] * f0[k] - v2[k] * f1[k] + v3[k] * f2[k]) * sa[k];
        inv[1 * 4 + k] = (v0[k] * f0[k] - v2[k] * f3[k] + v3[k] * f4[k]) * sb[k];
        inv[2 * 4 + k] = (v0[k] * f1[k] - v1[k] * f3[k] + v3[k] * f5[k]) * sa[k];
        inv[3 * 4 + k] = (v0[k] * f2[k] - v1[k] * f4[k] + v2[k] * f5[k]) * sb[k];
    }
    const float d0 = m(0,0) * inv[0], d1 = m(0,1) * inv[4], d2 = m(0,2) * inv[8], d3 = m(0,3) * inv[12];
    const float dot1 = (d0 + d1) + (d2 + d3);
    const float ood = 1.0f / dot1;
    for (int k = 0; k < 16; ++k) out[k] = inv[k] * ood;
}

void orc_mat4_mul(const float a[16], const float b[16], float out[16]) { mul44(a, b, out); }

// Node.cpp:25-27 rotate; glm/ext/quaternion_trigonometric.inl:27-33 angleAxis;
// glm/gtc/quaternion.inl:41-66 mat3_cast; glm/detail/type_mat3x3.inl:486-519
void orc_node_rotate(float rot3[9], const float axis[3], float angle) {
    const float s = std::sin(angle * 0.5f);
    const float qw = std::cos(angle * 0.5f);
    const float qx = axis[0] * s, qy = axis[1] * s, qz = axis[2] * s;
    const float qxx = qx * qx, qyy = qy * qy, qzz = qz * qz, qxz = qx * qz, qxy = qx * qy,
                qyz = qy * qz, qwx = qw * qx, qwy = qw * qy, qwz = qw * qz;
    float a[9];
    a[0] = 1.0f - 2.0f * (qyy + qzz); a[1] = 2.0f * (qxy + qwz); a[2] = 2.0f * (qxz - qwy);
    a[3] = 2.0f * (qxy - qwz); a[4] = 1.0f - 2.0f * (qxx + qzz); a[5] = 2.0f * (qyz + qwx);
    a[6] = 2.0f * (qxz + qwy); a[7] = 2.0f * (qyz - qwx); a[8] = 1.0f - 2.0f * (qxx + qyy);
    float r[9];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i)
            r[j * 3 + i] = a[0 * 3 + i] * rot3[j * 3 + 0] + a[1 * 3 + i] * rot3[j * 3 + 1] +
                           a[2 * 3 + i] * rot3[j * 3 + 2];
    std::memcpy(rot3, r, sizeof r);
}

// Node.cpp:10-23 for scene(I) -> orbit node(rot3) -> camera(position (0,0,d));
// OrbitCamera.h:24-30
void orc_orbit_world(const float rot3[9], float distance, float world[16]) {
    auto local = [](const float* rot /*3x3 or null*/, const float pos[3], float* out) {
        float id[16] = {1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1};  // scale(I, (1,1,1))
        float r4[16] = {1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1};
        if (rot)
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 3; ++r) r4[c * 4 + r] = rot[c * 3 + r];
        mul44(r4, id, out);
        out[12] = pos[0]; out[13] = pos[1]; out[14] = pos[2]; out[15] = 1.0f;
    };
    const float I[16] = {1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1};
    const float zero[3] = {0, 0, 0};
    float scene_l[16], scene_w[16], orbit_l[16], orbit_w[16], cam_l[16];
    local(nullptr, zero, scene_l);
    mul44(I, scene_l, scene_w);
    local(rot3, zero, orbit_l);
    mul44(scene_w, orbit_l, orbit_w);
    const float cp[3] = {0.0f, 0.0f, distance};
    local(nullptr, cp, cam_l);
    mul44(orbit_w, cam_l, world);
}

}  // extern "C"
