// vx_vec.h — 3-vector helpers of libvxrt, host and device.  Every operation is a fixed sequence of
// IEEE binary32 operations (see include/vxrt_detmath.h): sums are evaluated left to right, normalize
// is a true division by sqrt(dot), nothing may be contracted into an FMA (-ffp-contract=off).
#pragma once
#include "../../include/vxrt_detmath.h"

#if defined(__HIPCC__)
#define VXV __host__ __device__ inline __attribute__((always_inline))
#else
#define VXV static inline
#endif

struct f3 {
    float x, y, z;
};

VXV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
VXV f3 splat3(float s) { return mk3(s, s, s); }
VXV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
VXV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
VXV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
VXV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
VXV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
VXV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
VXV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
VXV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
VXV float len3(f3 a) { return vx_sqrt(dot3(a, a)); }
VXV f3 norm3(f3 a) { return a / len3(a); }
VXV f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
VXV f3 reflect3(f3 i, f3 n) { return i - (2.0f * dot3(n, i)) * n; }
VXV f3 mix3(f3 a, f3 b, float t) { return mk3(vx_mix(a.x, b.x, t), vx_mix(a.y, b.y, t), vx_mix(a.z, b.z, t)); }
