// hostmath.h -- minimal glm-style vec/mat types for the host side of the renderer (column-major like glm, so the
// matrices can be copied into the uniform block as they are).  Arithmetic is fp32, one rounding per operator
// (the library is compiled with -ffp-contract=off) so that the uniforms are reproducible.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>

#include "vr_math.h"

namespace vr {

struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
struct vec4 { float x = 0, y = 0, z = 0, w = 0; vec4() = default; vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {} };
struct ivec2 { int x = 0, y = 0; };
struct uvec3 { uint32_t x = 0, y = 0, z = 0; };

inline vec3 operator+(vec3 a, vec3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline vec3 operator-(vec3 a, vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline vec3 operator*(vec3 a, vec3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
inline vec3 operator*(vec3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline vec3 operator*(float s, vec3 a) { return { a.x * s, a.y * s, a.z * s }; }
inline vec3 operator-(vec3 a) { return { -a.x, -a.y, -a.z }; }
inline v3 to_v3(vec3 a) { return v3{ a.x, a.y, a.z }; }
inline vec3 from_v3(v3 a) { return { a.x, a.y, a.z }; }
inline vec3 normalize(vec3 a) { return from_v3(normalize(to_v3(a))); }
inline vec3 cross(vec3 a, vec3 b) { return from_v3(cross(to_v3(a), to_v3(b))); }
inline vec3 vmin(vec3 a, vec3 b) { return { min_(a.x, b.x), min_(a.y, b.y), min_(a.z, b.z) }; }
inline vec3 vmax(vec3 a, vec3 b) { return { max_(a.x, b.x), max_(a.y, b.y), max_(a.z, b.z) }; }

struct mat3 {
    float m[9];                     // m[3*c + r]
    mat3() : mat3(1.0f) {}
    explicit mat3(float d) { for (int i = 0; i < 9; ++i) m[i] = (i % 4 == 0) ? d : 0.0f; }
};
struct mat4 {
    float m[16];                    // m[4*c + r]
    mat4() : mat4(1.0f) {}
    explicit mat4(float d) { for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? d : 0.0f; }
};

inline mat4 operator*(const mat4& a, const mat4& b) {
    mat4 r(0.0f);
    for (int c = 0; c < 4; ++c)
        for (int rr = 0; rr < 4; ++rr)
            r.m[4 * c + rr] = a.m[rr] * b.m[4 * c] + a.m[4 + rr] * b.m[4 * c + 1] + a.m[8 + rr] * b.m[4 * c + 2] + a.m[12 + rr] * b.m[4 * c + 3];
    return r;
}
inline vec3 transform_point(const mat4& m, vec3 p) { return from_v3(mat4_point(m.m, to_v3(p))); }
inline mat3 upper3(const mat4& a) {
    mat3 r(0.0f);
    for (int c = 0; c < 3; ++c) for (int rr = 0; rr < 3; ++rr) r.m[3 * c + rr] = a.m[4 * c + rr];
    return r;
}
inline mat4 from3(const mat3& a) {
    mat4 r(1.0f);
    for (int c = 0; c < 3; ++c) for (int rr = 0; rr < 3; ++rr) r.m[4 * c + rr] = a.m[3 * c + rr];
    return r;
}

// glm::scale(mat4(1), s) followed by glm::translate(.., v): columns 0..2 scaled, column 3 = m0*v.x + m1*v.y + m2*v.z + m3
inline mat4 scale_then_translate(float s, vec3 v) {
    mat4 r(1.0f);
    r.m[0] = s; r.m[5] = s; r.m[10] = s;
    r.m[12] = s * v.x; r.m[13] = s * v.y; r.m[14] = s * v.z;
    return r;
}

// glm::rotate(mat4(1), angle, axis) for the three principal axes, as mat3
inline mat3 rotation_axis(float deg, int axis) {
    const float a = deg * 0.01745329251994329576923690768489f;   // glm::radians
    const float c = cos_(a), s = sin_(a);
    mat3 r(1.0f);
    if (axis == 0) { r.m[4] = c; r.m[5] = s; r.m[7] = -s; r.m[8] = c; }
    else if (axis == 1) { r.m[0] = c; r.m[2] = -s; r.m[6] = s; r.m[8] = c; }
    else { r.m[0] = c; r.m[1] = s; r.m[3] = -s; r.m[4] = c; }
    return r;
}

inline mat3 inverse(const mat3& M) {
    const float* m = M.m;
    const float a = m[0], b = m[3], c = m[6], d = m[1], e = m[4], f = m[7], g = m[2], h = m[5], i = m[8];
    const float det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const float id = 1.0f / det;
    mat3 R(0.0f);
    float* r = R.m;
    r[0] = (e * i - f * h) * id; r[3] = -(b * i - c * h) * id; r[6] = (b * f - c * e) * id;
    r[1] = -(d * i - f * g) * id; r[4] = (a * i - c * g) * id; r[7] = -(a * f - c * d) * id;
    r[2] = (d * h - e * g) * id; r[5] = -(a * h - b * g) * id; r[8] = (a * e - b * d) * id;
    return R;
}

inline mat4 inverse(const mat4& M) {
    const float* m = M.m;
    float inv[16];
    inv[0]  =  m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4]  = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8]  =  m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1]  = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5]  =  m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9]  = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] =  m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2]  =  m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6]  = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] =  m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3]  = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7]  =  m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] =  m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    const float id = 1.0f / det;
    mat4 R(0.0f);
    for (int i = 0; i < 16; ++i) R.m[i] = inv[i] * id;
    return R;
}

}  // namespace vr
