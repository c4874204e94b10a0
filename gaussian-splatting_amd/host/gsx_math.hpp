// gsx_math.hpp — the small slice of vector/matrix math the host side of the path
// needs, with the operation order of the reference's vendored glm 0.9.9.8 so that
// host-computed inputs (camera matrices, Splat records, LOD predicates' constants)
// are bit-identical to the reference's.  Column-major, m[c][r] like glm.
// Compile host code using this header with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>

// GSX_FACADE_WITH_GLM: the caller's code keeps its glm types (the reference passes glm::vec3 to SplatMesh::sortSplats,
// SplatMesh.h:84-85, and fills a glm::mat4x4 Uniforms block from the camera, main.cpp:48-58,222-224): vec3 / vec4 / mat4
// then convert to and from their glm counterparts implicitly.  Needs <glm/glm.hpp> on the include path (the reference's own).
#if defined(GSX_FACADE_WITH_GLM)
#include <glm/glm.hpp>
#endif

namespace gsx {

struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
#if defined(GSX_FACADE_WITH_GLM)
    vec3(const glm::vec3& v) : x(v.x), y(v.y), z(v.z) {}
    operator glm::vec3() const { return glm::vec3(x, y, z); }
#endif
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
// glm/detail/func_geometric.inl:47-55
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
// glm/detail/func_geometric.inl:86-98
inline vec3 cross(vec3 x, vec3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
// glm/detail/func_geometric.inl:16-23: distance(p0, p1) = length(p1 - p0)
inline float distance(vec3 p0, vec3 p1) { return length(p1 - p0); }

struct vec4 {
    float x = 0, y = 0, z = 0, w = 0;
    vec4() = default;
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    vec4(vec3 v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    explicit vec4(float s) : x(s), y(s), z(s), w(s) {}
#if defined(GSX_FACADE_WITH_GLM)
    vec4(const glm::vec4& v) : x(v.x), y(v.y), z(v.z), w(v.w) {}
    operator glm::vec4() const { return glm::vec4(x, y, z, w); }
#endif
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator-(vec4 a, vec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline vec4 operator*(vec4 a, vec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline vec4 operator/(vec4 a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }
inline vec3 xyz(vec4 v) { return {v.x, v.y, v.z}; }
// glm/detail/func_geometric.inl:57-65
inline float dot(vec4 a, vec4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
// glm/detail/func_geometric.inl:82-90 + func_exponential.inl:134-139
inline vec4 normalize(vec4 v) { return v * (1.0f / std::sqrt(dot(v, v))); }

struct mat3 {
    vec3 c[3];
    mat3() : mat3(1.0f) {}
    explicit mat3(float d) { c[0] = {d, 0, 0}; c[1] = {0, d, 0}; c[2] = {0, 0, d}; }
    vec3& operator[](int i) { return c[i]; }
    const vec3& operator[](int i) const { return c[i]; }
};
// glm/detail/type_mat3x3.inl:486-519
inline mat3 operator*(const mat3& a, const mat3& b) {
    mat3 r(0.0f);
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) r[j][i] = a[0][i] * b[j][0] + a[1][i] * b[j][1] + a[2][i] * b[j][2];
    return r;
}
// glm/detail/type_mat3x3.inl:468-474
inline vec3 operator*(const mat3& m, vec3 v) {
    return {m[0][0] * v.x + m[1][0] * v.y + m[2][0] * v.z, m[0][1] * v.x + m[1][1] * v.y + m[2][1] * v.z,
            m[0][2] * v.x + m[1][2] * v.y + m[2][2] * v.z};
}

struct mat4 {
    vec4 c[4];
    mat4() : mat4(1.0f) {}
    explicit mat4(float d) { c[0] = {d, 0, 0, 0}; c[1] = {0, d, 0, 0}; c[2] = {0, 0, d, 0}; c[3] = {0, 0, 0, d}; }
    explicit mat4(const mat3& m) {
        c[0] = vec4(m[0], 0); c[1] = vec4(m[1], 0); c[2] = vec4(m[2], 0); c[3] = {0, 0, 0, 1};
    }
    vec4& operator[](int i) { return c[i]; }
    const vec4& operator[](int i) const { return c[i]; }
#if defined(GSX_FACADE_WITH_GLM)
    mat4(const glm::mat4& m) { for (int j = 0; j < 4; ++j) c[j] = vec4(m[j]); }
    operator glm::mat4() const { return glm::mat4(glm::vec4(c[0]), glm::vec4(c[1]), glm::vec4(c[2]), glm::vec4(c[3])); }
#endif
    const float* data() const { return &c[0].x; }
    float* data() { return &c[0].x; }
};
static_assert(sizeof(mat4) == 64, "mat4 is 16 packed floats");
inline mat3 mat3_of(const mat4& m) {
    mat3 r(0.0f);
    for (int j = 0; j < 3; ++j) r[j] = xyz(m[j]);
    return r;
}
// glm/detail/type_mat4x4.inl:630-648: ((A0*b0 + A1*b1) + A2*b2) + A3*b3
inline mat4 operator*(const mat4& a, const mat4& b) {
    mat4 r(0.0f);
    for (int j = 0; j < 4; ++j) r[j] = a[0] * b[j][0] + a[1] * b[j][1] + a[2] * b[j][2] + a[3] * b[j][3];
    return r;
}
// glm/detail/type_mat4x4.inl:561-572: (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
inline vec4 operator*(const mat4& m, vec4 v) { return (m[0] * v.x + m[1] * v.y) + (m[2] * v.z + m[3] * v.w); }
inline mat4 operator*(const mat4& m, float s) {
    mat4 r(0.0f);
    for (int j = 0; j < 4; ++j) r[j] = m[j] * s;
    return r;
}
inline mat4 transpose(const mat4& m) {
    mat4 r(0.0f);
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) r[j][i] = m[i][j];
    return r;
}

// glm/detail/func_matrix.inl:294-351
inline mat4 inverse(const mat4& m) {
    const float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3], c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3],
                c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3], c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3],
                c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3], c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3],
                c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2], c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2],
                c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2], c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3],
                c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3], c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3],
                c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2], c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2],
                c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2], c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1],
                c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1], c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    const vec4 f0(c00, c00, c02, c03), f1(c04, c04, c06, c07), f2(c08, c08, c10, c11), f3(c12, c12, c14, c15),
        f4(c16, c16, c18, c19), f5(c20, c20, c22, c23);
    const vec4 v0(m[1][0], m[0][0], m[0][0], m[0][0]), v1(m[1][1], m[0][1], m[0][1], m[0][1]),
        v2(m[1][2], m[0][2], m[0][2], m[0][2]), v3(m[1][3], m[0][3], m[0][3], m[0][3]);
    const vec4 i0 = v1 * f0 - v2 * f1 + v3 * f2, i1 = v0 * f0 - v2 * f3 + v3 * f4, i2 = v0 * f1 - v1 * f3 + v3 * f5,
               i3 = v0 * f2 - v1 * f4 + v2 * f5;
    const vec4 sa(+1, -1, +1, -1), sb(-1, +1, -1, +1);
    mat4 inv(0.0f);
    inv[0] = i0 * sa; inv[1] = i1 * sb; inv[2] = i2 * sa; inv[3] = i3 * sb;
    const vec4 row0(inv[0][0], inv[1][0], inv[2][0], inv[3][0]);
    const vec4 d0 = m[0] * row0;
    const float d1 = (d0.x + d0.y) + (d0.z + d0.w);
    return inv * (1.0f / d1);
}

// glm/detail/func_matrix.inl:223-244
inline float determinant(const mat4& m) {
    const float s00 = m[2][2] * m[3][3] - m[3][2] * m[2][3], s01 = m[2][1] * m[3][3] - m[3][1] * m[2][3],
                s02 = m[2][1] * m[3][2] - m[3][1] * m[2][2], s03 = m[2][0] * m[3][3] - m[3][0] * m[2][3],
                s04 = m[2][0] * m[3][2] - m[3][0] * m[2][2], s05 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    const float d0 = +(m[1][1] * s00 - m[1][2] * s01 + m[1][3] * s02), d1 = -(m[1][0] * s00 - m[1][2] * s03 + m[1][3] * s04),
                d2 = +(m[1][0] * s01 - m[1][1] * s03 + m[1][3] * s05), d3 = -(m[1][0] * s02 - m[1][1] * s04 + m[1][2] * s05);
    return m[0][0] * d0 + m[0][1] * d1 + m[0][2] * d2 + m[0][3] * d3;
}

// glm/trigonometric radians(); glm/ext/matrix_clip_space.inl:249-262 perspectiveRH_NO
inline float radians(float deg) { return deg * static_cast<float>(0.01745329251994329576923690768489); }
inline mat4 perspective(float fovy, float aspect, float zn, float zf) {
    const float t = std::tan(fovy / 2.0f);
    mat4 r(0.0f);
    r[0][0] = 1.0f / (aspect * t);
    r[1][1] = 1.0f / t;
    r[2][2] = -(zf + zn) / (zf - zn);
    r[2][3] = -1.0f;
    r[3][2] = -(2.0f * zf * zn) / (zf - zn);
    return r;
}

// glm/ext/quaternion_trigonometric.inl:27-33 + glm/gtc/quaternion.inl:41-66
inline mat3 rotation_about(vec3 axis, float angle) {
    const float s = std::sin(angle * 0.5f), qw = std::cos(angle * 0.5f);
    const float qx = axis.x * s, qy = axis.y * s, qz = axis.z * s;
    const float qxx = qx * qx, qyy = qy * qy, qzz = qz * qz, qxz = qx * qz, qxy = qx * qy, qyz = qy * qz, qwx = qw * qx,
                qwy = qw * qy, qwz = qw * qz;
    mat3 r(1.0f);
    r[0][0] = 1.0f - 2.0f * (qyy + qzz); r[0][1] = 2.0f * (qxy + qwz); r[0][2] = 2.0f * (qxz - qwy);
    r[1][0] = 2.0f * (qxy - qwz); r[1][1] = 1.0f - 2.0f * (qxx + qzz); r[1][2] = 2.0f * (qyz + qwx);
    r[2][0] = 2.0f * (qxz + qwy); r[2][1] = 2.0f * (qyz - qwx); r[2][2] = 1.0f - 2.0f * (qxx + qyy);
    return r;
}

// glm/ext/matrix_transform.inl scale(): columns scaled, translation kept
inline mat4 scale(const mat4& m, vec3 v) {
    mat4 r(0.0f);
    r[0] = m[0] * v.x; r[1] = m[1] * v.y; r[2] = m[2] * v.z; r[3] = m[3];
    return r;
}

}  // namespace gsx
