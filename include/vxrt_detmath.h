/*
 * vxrt_detmath.h — the numeric contract of the vxrt hot path.
 *
 * The reference's kernels (shaders/voxels.comp, temporal.comp, denoise.comp) lean on GLSL
 * built-ins whose precision the Vulkan/Metal driver decides (normalize, sin, cos, pow, exp,
 * log, min/max on NaN ...; call sites e.g. voxels.comp:282-283,296,299,378-381,
 * denoise.comp:66,80).  No reference test pins them.  A path tracer that offsets bounce
 * origins by 1e-5 on coordinates up to 64 turns a single differently-rounded operation into a
 * flipped hit/miss, so this build pins every such built-in to ONE definition made only of IEEE-754
 * binary32 +,-,*,/ ,sqrt, floor and integer bit operations.  Compiled with -ffp-contract=off and
 * no fast-math by both g++ (CPU oracle) and hipcc (gfx950 kernels, where HIP's default
 * correctly-rounded divide/sqrt is kept) the functions below return bit-identical results
 * on the host and on the device.
 *
 * Everything here is `static inline`, C99/C++/HIP compatible and has no state.
 */
#ifndef VXRT_DETMATH_H
#define VXRT_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define VX_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define VX_HD static inline
#endif

/* ---- bit casts ---------------------------------------------------------------------------- */
VX_HD uint32_t vx_f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
VX_HD float vx_u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

/* ---- GLSL min/max/sign/abs/clamp/mix, literally as the GLSL 4.50 spec words them ------------- */
/* min(x,y): "Returns y if y < x, otherwise it returns x" — NaN behaviour follows from that. */
VX_HD float vx_min(float x, float y) { return (y < x) ? y : x; }
/* min(0.0, v) — the form voxels.comp:285 uses — on integer bits: v if v < 0 (sign set, not -0, not NaN), else +0.0.
 * Written this way because the gfx950 backend turns `(v < 0.0f) ? v : 0.0f` into v_min_f32 0, v, which returns -0.0
 * for v = -0.0 where the GLSL wording (and g++) return +0.0; a bounce direction with a different sign of zero then
 * takes a different octant at the next cast.  Use this wherever the first argument of min() is the constant zero. */
VX_HD float vx_min0(float v) {
    uint32_t b = vx_f2u(v);
    return (b > 0x80000000u && b <= 0xff800000u) ? v : 0.0f;
}
/* max(x,y): "Returns y if x < y, otherwise it returns x". */
VX_HD float vx_max(float x, float y) { return (x < y) ? y : x; }
/* sign(x): 1.0 if x > 0, 0.0 if x = 0, -1.0 if x < 0 (NaN -> 0.0). */
VX_HD float vx_sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
VX_HD float vx_abs(float x) { return vx_u2f(vx_f2u(x) & 0x7fffffffu); }
VX_HD float vx_clamp(float x, float lo, float hi) { return vx_min(vx_max(x, lo), hi); }
/* mix(x,y,a) = x*(1-a) + y*a */
VX_HD float vx_mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }

/* float -> int with every input defined (C leaves NaN / out-of-range undefined; x86 and gfx950 differ). */
VX_HD int32_t vx_f2i(float x) {
    if (!(x == x)) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (-2147483647 - 1);
    return (int32_t)x;
}

VX_HD float vx_sqrt(float x) { return __builtin_sqrtf(x); }
VX_HD float vx_floor(float x) { return __builtin_floorf(x); }

/* ---- sin / cos ----------------------------------------------------------------------------
 * Cody–Waite reduction by pi/2 in three parts, then the classic degree-13/14 minimax
 * polynomials on [-pi/4, pi/4].  Domain |x| <= 32768 (NaN outside); meant for |x| < ~1e3 (the path feeds it sun yaw/pitch and
 * phi = 2*pi*u, u in [0,1)).  Max error ~1 ulp on that range. */
VX_HD float vx_ksin(float r) {
    float z = r * r;
    float p = -2.5050759689e-08f + z * 1.5896910177e-10f;
    p = 2.7557314297e-06f + z * p;
    p = -1.9841270114e-04f + z * p;
    p = 8.3333337680e-03f + z * p;
    p = -1.6666667163e-01f + z * p;
    return r + (r * z) * p;
}
VX_HD float vx_kcos(float r) {
    float z = r * r;
    float p = 2.0875723372e-09f + z * -1.1359647598e-11f;
    p = -2.7557314297e-07f + z * p;
    p = 2.4801587642e-05f + z * p;
    p = -1.3888889225e-03f + z * p;
    p = 4.1666667908e-02f + z * p;
    return (1.0f - 0.5f * z) + (z * z) * p;
}
VX_HD float vx_reduce_pio2(float x, int *quadrant) {
    if (!(vx_abs(x) <= 32768.0f)) {  /* outside the supported domain (also inf, NaN): NaN, deterministically */
        *quadrant = 0;
        return vx_u2f(0x7fc00000u);
    }
    float kf = vx_floor(x * 0.63661977236758134f + 0.5f);
    /* pi/2 = P1 + P2 + P3, P1 and P2 with trailing zero bits so k*P1, k*P2 are exact for |k| < 2^7 */
    float r = x - kf * 1.5707855225e+00f;     /* 0x3fc90f80 */
    r = r - kf * 1.0804273188e-05f;           /* 0x37354400 */
    r = r - kf * 6.0770999344e-11f;           /* 0x2e85a308 */
    *quadrant = ((int)kf) & 3;
    return r;
}
VX_HD float vx_sin(float x) {
    int q;
    float r = vx_reduce_pio2(x, &q);
    float s = vx_ksin(r), c = vx_kcos(r);
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
VX_HD float vx_cos(float x) {
    int q;
    float r = vx_reduce_pio2(x, &q);
    float s = vx_ksin(r), c = vx_kcos(r);
    float v = (q & 1) ? s : c;
    return ((q + 1) & 2) ? -v : v;
}
VX_HD float vx_tan(float x) { return vx_sin(x) / vx_cos(x); }

#ifndef VXRT_EXP_FMA
#define VXRT_EXP_FMA 1   /* round 4: exp's kernel is evaluated with fused multiply-adds (0: round 1-3's form, every product and sum rounded) */
#endif
/* ---- exp ------------------------------------------------------------------------------------
 * exp(x) = 2^k * e^r, k = round(x/ln2), |r| <= ln2/2, degree-7 Taylor–Horner.  Results below
 * the normal range are flushed to +0 (x < -87.3), above to +inf (x > 88.7); NaN -> NaN. */
VX_HD float vx_exp_in_range(float x);
VX_HD float vx_exp(float x) {
    if (!(x == x)) return x;
    if (x > 88.72f) return vx_u2f(0x7f800000u);
    if (x < -87.3f) return 0.0f;
    return vx_exp_in_range(x);
}
/* vx_exp for an argument the caller knows to lie in [-87.3, 88.72] (not NaN): the same operations without the three range tests */
/* round 1-3's form of the kernel: every product and sum rounded.  Kept because include/vxrt_bluenoise.h's filter table is specified
 * with it (a published table must not move when the renderer's exp does). */
VX_HD float vx_exp_in_range_unfused(float x) {
    float kf = vx_floor(x * 1.44269504088896341f + 0.5f);
    float r = x - kf * 6.93145752e-01f;   /* ln2 hi, 0x3f317200 */
    r = r - kf * 1.42860677e-06f;         /* ln2 lo */
    float p = 1.0f / 720.0f + r * (1.0f / 5040.0f);
    p = 1.0f / 120.0f + r * p;
    p = 1.0f / 24.0f + r * p;
    p = 1.0f / 6.0f + r * p;
    p = 0.5f + r * p;
    p = 1.0f + r * p;
    p = 1.0f + r * p;
    int k = (int)kf;
    return p * vx_u2f((uint32_t)(k + 127) << 23);
}
VX_HD float vx_exp_unfused(float x) {
    if (!(x == x)) return x;
    if (x > 88.72f) return vx_u2f(0x7f800000u);
    if (x < -87.3f) return 0.0f;
    return vx_exp_in_range_unfused(x);
}
VX_HD float vx_exp_in_range(float x) {
#if VXRT_EXP_FMA
    /* Every multiply-add fused: fmaf is correctly rounded on both sides (v_fma_f32; vfmadd / libm's fmaf on the host), so host and
     * device still agree bit for bit (tests/test_gpu_detmath.py), the result is within 1 ulp of round 3's form and nearer to the true
     * value (7 roundings fewer), and the denoiser's exact tap — one exp per tap, 289 taps per pixel at radius 8 — loses 10 of its 60
     * instructions: 2.27 -> 1.90 ms at 4K.  The shaders' own arithmetic is never contracted; this is
     * the inside of a built-in whose precision GLSL leaves to the driver (U6). */
    float kf = vx_floor(__builtin_fmaf(x, 1.44269504088896341f, 0.5f));
    float r = __builtin_fmaf(-kf, 6.93145752e-01f, x);
    r = __builtin_fmaf(-kf, 1.42860677e-06f, r);
    float p = __builtin_fmaf(r, 1.0f / 5040.0f, 1.0f / 720.0f);
    p = __builtin_fmaf(r, p, 1.0f / 120.0f);
    p = __builtin_fmaf(r, p, 1.0f / 24.0f);
    p = __builtin_fmaf(r, p, 1.0f / 6.0f);
    p = __builtin_fmaf(r, p, 0.5f);
    p = __builtin_fmaf(r, p, 1.0f);
    p = __builtin_fmaf(r, p, 1.0f);
    int k = (int)kf;
    return p * vx_u2f((uint32_t)(k + 127) << 23);
#else
    return vx_exp_in_range_unfused(x);
#endif
}

/* ---- log (natural) ----------------------------------------------------------------------------
 * x = 2^e * m, m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(s), s = (m-1)/(m+1), odd series to s^9.
 * log(0) = -inf, log(x<0) = NaN, log(inf) = inf, subnormals handled by pre-scaling. */
VX_HD float vx_log(float x) {
    uint32_t u = vx_f2u(x);
    if (!(x == x)) return x;
    if (x == 0.0f) return vx_u2f(0xff800000u);
    if (u & 0x80000000u) return vx_u2f(0x7fc00000u);
    if (u == 0x7f800000u) return x;
    int e = 0;
    if (u < 0x00800000u) { x = x * 8388608.0f; u = vx_f2u(x); e = -23; }
    e += (int)(u >> 23) - 127;
    float m = vx_u2f((u & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float p = 1.0f / 7.0f + z * (1.0f / 9.0f);
    p = 1.0f / 5.0f + z * p;
    p = 1.0f / 3.0f + z * p;
    float R = 2.0f * s + (2.0f * s) * (z * p);
    float ef = (float)e;
    return (ef * 1.42860677e-06f + R) + ef * 6.93145752e-01f;
}

/* ---- pow ----------------------------------------------------------------------------------------
 * pow(x,y) = exp(y*log(x)) for x > 0; pow(0,y>0) = 0; x < 0 is undefined in GLSL -> NaN here.
 * NOTE: the shaders' pow(v, 2) call sites (voxels.comp:380, denoise.comp:39-40,75) are restated
 * as v*v, not through this function: every LLVM/NIR based driver folds pow(x,2.0) to x*x, and the
 * reference's denoiser only works if that happens (its argument at denoise.comp:75 is negative
 * for half of all taps, for which GLSL pow() is undefined). */
VX_HD float vx_pow(float x, float y) {
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : vx_u2f(0x7f800000u));
    return vx_exp(y * vx_log(x));
}

#endif /* VXRT_DETMATH_H */
