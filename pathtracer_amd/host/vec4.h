// vec4.h — 4-lane fp32 vector used by the host side (scene loading, BVH build).
// Counterpart of the reference's f4 (src/vec.h:8-209): 3-arg construction sets
// w = 0, scalar construction broadcasts into w, dot/length2 sum all four lanes.
#pragma once
#include <cmath>

namespace pt {

struct V4 {
    float x = 0, y = 0, z = 0, w = 0;
    V4() = default;
    explicit V4(float s) : x(s), y(s), z(s), w(s) {}
    V4(float x_, float y_, float z_) : x(x_), y(y_), z(z_), w(0) {}
    V4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    float operator[](int i) const { return i == 0 ? x : i == 1 ? y : i == 2 ? z : w; }
    float &operator[](int i) { return i == 0 ? x : i == 1 ? y : i == 2 ? z : w; }
};

inline V4 operator-(const V4 &a) { return {-a.x, -a.y, -a.z, -a.w}; }
inline V4 operator+(const V4 &a, const V4 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 operator-(const V4 &a, const V4 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline V4 operator*(const V4 &a, float f) { return {a.x * f, a.y * f, a.z * f, a.w * f}; }
inline V4 operator*(float f, const V4 &a) { return {f * a.x, f * a.y, f * a.z, f * a.w}; }

inline float dot(const V4 &a, const V4 &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline float length2(const V4 &a) { return dot(a, a); }
// vec.h:200-202 normalises with rsqrtf; evaluated here as 1/sqrt with two correctly
// rounded operations ("Numerics contract").
inline V4 normalize(const V4 &a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline V4 cross(const V4 &a, const V4 &b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline V4 vmin(const V4 &a, const V4 &b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z), std::fmin(a.w, b.w)}; }
inline V4 vmax(const V4 &a, const V4 &b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z), std::fmax(a.w, b.w)}; }

// Column-major 4x4 with the arithmetic association of glm 0.9.9.x (the reference's
// matrix library, pinned by cmake/glm.cmake:11,15 and not vendored), so transformed
// vertices round the same way: mat*vec = (c0*x + c1*y) + (c2*z + c3*w).
struct M4 {
    V4 c[4];
    static M4 identity() {
        M4 m;
        m.c[0] = {1, 0, 0, 0}; m.c[1] = {0, 1, 0, 0}; m.c[2] = {0, 0, 1, 0}; m.c[3] = {0, 0, 0, 1};
        return m;
    }
};
inline V4 operator*(const M4 &m, const V4 &v) {
    V4 add0 = m.c[0] * v.x + m.c[1] * v.y;
    V4 add1 = m.c[2] * v.z + m.c[3] * v.w;
    return add0 + add1;
}
inline M4 operator*(const M4 &a, const M4 &b) {
    M4 r;
    for (int j = 0; j < 4; j++)
        r.c[j] = a.c[0] * b.c[j].x + a.c[1] * b.c[j].y + a.c[2] * b.c[j].z + a.c[3] * b.c[j].w;
    return r;
}
inline M4 transpose(const M4 &m) {
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.c[i][j] = m.c[j][i];
    return r;
}
inline M4 translate(float x, float y, float z) {
    M4 m = M4::identity();
    m.c[3] = m.c[0] * x + m.c[1] * y + m.c[2] * z + m.c[3];
    return m;
}
inline M4 scale(float x, float y, float z) {
    M4 m = M4::identity();
    m.c[0] = m.c[0] * x; m.c[1] = m.c[1] * y; m.c[2] = m.c[2] * z;
    return m;
}
// glm::rotate(angle, axis) applied to the identity (gtx/transform.hpp ->
// gtc/matrix_transform rotate): Rodrigues with a normalised axis.
inline M4 rotate(float angle, float ax, float ay, float az) {
    const float c = std::cos(angle), s = std::sin(angle);
    const float il = 1.0f / std::sqrt(ax * ax + ay * ay + az * az);
    const float a[3] = {ax * il, ay * il, az * il};
    const float t[3] = {(1.0f - c) * a[0], (1.0f - c) * a[1], (1.0f - c) * a[2]};
    float R[3][3];
    R[0][0] = c + t[0] * a[0];        R[0][1] = t[0] * a[1] + s * a[2]; R[0][2] = t[0] * a[2] - s * a[1];
    R[1][0] = t[1] * a[0] - s * a[2]; R[1][1] = c + t[1] * a[1];        R[1][2] = t[1] * a[2] + s * a[0];
    R[2][0] = t[2] * a[0] + s * a[1]; R[2][1] = t[2] * a[1] - s * a[0]; R[2][2] = c + t[2] * a[2];
    const M4 m = M4::identity();
    M4 r;
    for (int j = 0; j < 3; j++) r.c[j] = m.c[0] * R[j][0] + m.c[1] * R[j][1] + m.c[2] * R[j][2];
    r.c[3] = m.c[3];
    return r;
}

} // namespace pt
