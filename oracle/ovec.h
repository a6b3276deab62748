// ORACLE — test infrastructure only.  Nothing in the product path may include, link or call this.
//
// ovec.h: 3-vector and the GLSL built-ins used by the reference shaders, restated on top of the
// numeric contract in include/vxrt_detmath.h.  Follows src/linear.rs:77-236 (Vec3: dot is a
// left-to-right sum, norm() is a true division by length()) and GLSL 4.50 §8.
#pragma once
#include "../include/vxrt_detmath.h"
#include <cstdint>

// -DORC_ALT_BUILTINS=1 (make alt -> _build/liboracle_alt.so; tests/test_oracle_builtin_sensitivity.py only): the choices this
// restatement makes where GLSL / Vulkan leave the value to the driver, made THE OTHER WAY, switched on at run time by
// orc_set_alt(mask) — to measure how far an image can move between two conforming implementations:
//   bit 0  U6  sin cos tan exp log pow  -> binary64 libm, rounded once to binary32 (instead of the vxrt_detmath.h polynomials)
//   bit 1  U6  normalize(v)             -> v * inversesqrt(dot(v, v)), the usual driver lowering, inversesqrt rounded once from
//                                          binary64 (instead of the true division v / length(v) of src/linear.rs:117-119)
//   bit 2  U4  texture()                -> bilinear weights at full binary32 precision (instead of 8 fractional bits)
//   bit 3  U5  inverse(mat4)            -> adjugate / determinant evaluated in binary32 (instead of binary64 rounded once)
// With mask 0 the library equals liboracle.so bit for bit (the test checks that first).
#ifndef ORC_ALT_BUILTINS
#define ORC_ALT_BUILTINS 0
#endif
#if ORC_ALT_BUILTINS
#include <cmath>
namespace orc { extern int g_alt_mask; }
static inline float orc_alt_sin(float x) { return (orc::g_alt_mask & 1) ? (float)std::sin((double)x) : vx_sin(x); }
static inline float orc_alt_cos(float x) { return (orc::g_alt_mask & 1) ? (float)std::cos((double)x) : vx_cos(x); }
static inline float orc_alt_tan(float x) { return (orc::g_alt_mask & 1) ? (float)std::tan((double)x) : vx_tan(x); }
static inline float orc_alt_exp(float x) { return (orc::g_alt_mask & 1) ? (float)std::exp((double)x) : vx_exp(x); }
static inline float orc_alt_log(float x) { return (orc::g_alt_mask & 1) ? (float)std::log((double)x) : vx_log(x); }
static inline float orc_alt_pow(float x, float y) { return (orc::g_alt_mask & 1) ? (float)std::pow((double)x, (double)y) : vx_pow(x, y); }
#define vx_sin orc_alt_sin
#define vx_cos orc_alt_cos
#define vx_tan orc_alt_tan
#define vx_exp orc_alt_exp
#define vx_log orc_alt_log
#define vx_pow orc_alt_pow
#endif

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
#if ORC_ALT_BUILTINS
static inline V3 normalize(V3 a) {
    if (g_alt_mask & 2) return a * (float)(1.0 / std::sqrt((double)dot(a, a)));
    return a / length(a);
}
#else
static inline V3 normalize(V3 a) { return a / length(a); }
#endif
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
