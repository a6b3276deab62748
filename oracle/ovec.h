// ORACLE — test infrastructure only.  Nothing in the product path may include, link or call this.
//
// ovec.h: 3-vector and the GLSL built-ins used by the reference shaders, restated on top of the
// numeric contract in include/vxrt_detmath.h.  Follows src/linear.rs:77-236 (Vec3: dot is a
// left-to-right sum, norm() is a true division by length()) and GLSL 4.50 §8.
#pragma once
#include "../include/vxrt_detmath.h"
#include <cstdint>

namespace orc {

struct V3 {
    float x, y, z;
};

static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 v3s(float s) { return V3{s, s, s}; }
static inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
static inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
static inline V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
static inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
static inline V3 operator/(float s, V3 a) { return V3{s / a.x, s / a.y, s / a.z}; }

// dot: ((x*x') + (y*y')) + (z*z')   (src/linear.rs:99-101 starts from 0.0, which adds nothing)
static inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float length(V3 a) { return vx_sqrt(dot(a, a)); }
// normalize = v / length(v)   (src/linear.rs:117-119; GLSL leaves the method to the driver)
static inline V3 normalize(V3 a) { return a / length(a); }
// cross (src/linear.rs:208-216)
static inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// reflect(I,N) = I - 2*dot(N,I)*N   (GLSL 4.50 §8.5)
static inline V3 reflect(V3 i, V3 n) { return i - (2.0f * dot(n, i)) * n; }
static inline V3 vabs(V3 a) { return V3{vx_abs(a.x), vx_abs(a.y), vx_abs(a.z)}; }
static inline V3 vsign(V3 a) { return V3{vx_sign(a.x), vx_sign(a.y), vx_sign(a.z)}; }
static inline V3 vmix(V3 a, V3 b, float t) {
    return V3{vx_mix(a.x, b.x, t), vx_mix(a.y, b.y, t), vx_mix(a.z, b.z, t)};
}

}  // namespace orc
