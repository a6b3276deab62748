// post.hip — temporal accumulation and spatial denoise kernels of libvxrt (gfx950), plus two small
// utility kernels (noise-table fill, detmath probe).
//
//  temporal_kernel : shaders/temporal.comp:48-125 — reproject into the previous frame, validate by
//                    world-space distance, blend with the per-pixel decaying factor kept in .a.
//                    80 algorithmic bytes per pixel (64 read + 16 written); the per-pixel
//                    inverse(mat4) of the shader is hoisted to the host (one matrix per frame).
//  denoise_kernel  : shaders/denoise.comp:24-93 — (2r+1)^2 cross-bilateral window.  A 16x16 block
//                    stages its (16+2r)^2 apron once in LDS as 8 floats per pixel (rgb, log|depth| ;
//                    normal, material id), so a tap is two ds_read_b128 + ~45 flops + one exp instead
//                    of three 16-byte global loads and two logs; the per-offset distance term comes
//                    from a small LDS table; taps whose weight is exactly zero are skipped.
//                    radius 0 (the default) is a separate streaming kernel.
#include "halo_view.h"
#include "kernels.h"
#include "vx_vec.h"

namespace vxrt {
namespace {

__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ f3 xyz(float4 v) { return mk3(v.x, v.y, v.z); }

__device__ __forceinline__ f3 pixel_dir(const Cam& c, int x, int y) {
    return norm3((float(x) * ld3(c.r) - float(y) * ld3(c.u)) + ld3(c.f));
}

// One image of the temporal history as this rank can see it: its own rows, plus — when the halo unpacked after the previous
// frame's temporal stage is still there — halo.rows rows beyond each of its band edges (kernels.h: HaloView).
//   kColour: the accumulated colour (rgb + blending factor)      kDepth: normal/depth, of which temporal.comp:94-104 reads .w only
enum HistoryKind { kColour = 0, kDepth = 1 };
struct HistoryRows {
    const float4* img;
    HaloView halo;
    HistoryKind kind;
};
struct HistoryRow {
    const float4* own;   // a row of this rank's image, or null: the halo row
    HaloRow far;
};
// frame row y of the history; false when this rank cannot see it
__device__ __forceinline__ bool history_row(const BandMap& b, const HistoryRows& hs, int y, HistoryRow& out) {
    const int l = local_row(b, y);
    out.own = nullptr;
    if (l >= 0) { out.own = hs.img + size_t(l) * b.width; return true; }
    return halo_find(b, hs.halo, y, out.far);
}
__device__ __forceinline__ float4 history_texel(const HistoryRows& hs, const HistoryRow& r, int x) {
    if (r.own != nullptr) return r.own[x];
    const float4 a = r.far.a[x];
    if (hs.kind == kColour) return make_float4(a.x, a.y, a.z, r.far.c[x]);
    return make_float4(0.0f, 0.0f, 0.0f, a.w);   // the normal of an old texel is never read (temporal.comp:94-104)
}

// texture() through the reference's Linear / ClampToEdge sampler (src/context.rs:980-989): bilinear,
// weights quantised to 8 fractional bits (oracle U4).  A row this rank cannot see makes the lookup fail.
__device__ __forceinline__ bool sample_bilinear(const HistoryRows& hs, const BandMap& b, float u, float v, float4& out) {
    float fx = u * float(b.width) - 0.5f, fy = v * float(b.height) - 0.5f;
    float x0f = vx_floor(fx), y0f = vx_floor(fy);
    float ax = vx_floor((fx - x0f) * 256.0f + 0.5f) / 256.0f, ay = vx_floor((fy - y0f) * 256.0f + 0.5f) / 256.0f;
    int x0 = min(max(vx_f2i(x0f), -2), b.width), y0 = min(max(vx_f2i(y0f), -2), b.height);
    int xa = min(max(x0, 0), b.width - 1), xb = min(max(x0 + 1, 0), b.width - 1);
    int ya = min(max(y0, 0), b.height - 1), yb = min(max(y0 + 1, 0), b.height - 1);
    // a texel whose quantised weight is 0 is not read: only rows with a non-zero weight must be visible
    HistoryRow ra, rb;
    ra.own = rb.own = hs.img;
    if (ay != 1.0f && !history_row(b, hs, ya, ra)) return false;
    if (ay != 0.0f && !history_row(b, hs, yb, rb)) return false;
    float4 t00 = history_texel(hs, ra, xa), t10 = history_texel(hs, ra, xb);
    float4 t01 = history_texel(hs, rb, xa), t11 = history_texel(hs, rb, xb);
    const float* p00 = &t00.x; const float* p10 = &t10.x; const float* p01 = &t01.x; const float* p11 = &t11.x;
    float* o = &out.x;
    for (int k = 0; k < 4; k++) {
        float top = ax == 0.0f ? p00[k] : (ax == 1.0f ? p10[k] : (p00[k] * (1.0f - ax) + p10[k] * ax));
        float bot = ax == 0.0f ? p01[k] : (ax == 1.0f ? p11[k] : (p01[k] * (1.0f - ax) + p11[k] * ax));
        o[k] = ay == 0.0f ? top : (ay == 1.0f ? bot : (top * (1.0f - ay) + bot * ay));
    }
    return true;
}

__global__ __launch_bounds__(256) void temporal_kernel(const TemporalArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int lrow = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.band.width || lrow >= a.band.local_rows) return;
    const int y = frame_row(a.band, lrow);
    const size_t pix = size_t(lrow) * a.band.width + x;

    const f3 color = xyz(a.sampled_color[pix]);
    const float4 nd = a.new_nd[pix];
    const f3 normal = xyz(nd);
    const float depth = nd.w;
    const f3 cam_o = ld3(a.cam.o);
    const f3 world_pos = cam_o + depth * pixel_dir(a.cam, x, y);

    float4 old_color = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float blending = 1.0f;
    if (depth >= 0.0f && a.has_history) {
        const float* m = a.inv;  // old_world_to_screen * vec4(world_pos, 1)   temporal.comp:82-85
        float sx = ((m[0] * world_pos.x + m[1] * world_pos.y) + m[2] * world_pos.z) + m[3];
        float sy = ((m[4] * world_pos.x + m[5] * world_pos.y) + m[6] * world_pos.z) + m[7];
        float sz = ((m[8] * world_pos.x + m[9] * world_pos.y) + m[10] * world_pos.z) + m[11];
        sx = sx / sz;
        sy = sy / sz;
        float tu = (sx + 0.5f) * (1.0f / float(a.band.width));     // temporal.comp:89
        float tv = (sy + -0.5f) * (-1.0f / float(a.band.height));
        if (0.0f <= tu && tu <= 1.0f && 0.0f <= tv && tv <= 1.0f) {
            float4 old_nd;
            const HistoryRows nd_rows{a.old_nd, a.halo, kDepth}, color_rows{a.old_color, a.halo, kColour};
            if (sample_bilinear(nd_rows, a.band, tu, tv, old_nd)) {
                f3 old_dir = norm3((float(vx_f2i(sx + 0.5f)) * ld3(a.old_cam.r) + float(vx_f2i(sy - 0.5f)) * ld3(a.old_cam.u)) + ld3(a.old_cam.f));
                f3 old_position = ld3(a.old_cam.o) + old_nd.w * old_dir;
                f3 camera_dir = norm3(cam_o - world_pos);
                float bias = vx_max(0.0f, dot3(camera_dir, normal));
                float dist = len3(old_position - world_pos);
                if (dist < (bias * a.blending_distance_cutoff) * depth) {
                    sample_bilinear(color_rows, a.band, tu, tv, old_color);
                    blending = old_color.w;
                }
            }
        }
    }
    f3 blended = depth >= 0.0f ? mix3(xyz(old_color), color, blending) : color;
    float next_blending = vx_clamp((1.0f - a.sample_blending) * blending, 1.0f - a.maximum_blending, 1.0f);
    a.new_color[pix] = make_float4(blended.x, blended.y, blended.z, next_blending);
    if (a.albedo != nullptr) {  // denoise.comp:88-92 with radius 0, on the value just written
        const f3 alb = xyz(a.albedo[pix]);
        const f3 out = mix3(blended, alb * blended, a.albedo_factor);
        a.denoised[pix] = make_float4(out.x, out.y, out.z, 1.0f);
    }
}

// A staged pixel of the denoise apron is two float4 in two LDS arrays (each read is a conflict-free
// ds_read_b128: consecutive lanes, consecutive 16-byte slots):
//   A[i] = (r, g, b, log|depth|)      B[i] = (nx, ny, nz, bits(material id | flags))
constexpr int32_t kTapOutside = 0x7fffffff;   // not a pixel of the frame: denoise.comp:57 skips the tap
constexpr int32_t kTapNonFinite = 0x40000000; // colour holds an inf/NaN: its zero-weight taps may not be skipped

__device__ __forceinline__ bool finite3(float4 c) {
    return ((__float_as_uint(c.x) & 0x7f800000u) != 0x7f800000u) && ((__float_as_uint(c.y) & 0x7f800000u) != 0x7f800000u) &&
           ((__float_as_uint(c.z) & 0x7f800000u) != 0x7f800000u);
}

// radius 0 (the reference's default): out = mix(c, albedo * c, albedo_factor), a pure stream.
__global__ __launch_bounds__(256) void denoise_passthrough_kernel(const DenoiseArgs a) {
    const size_t n = size_t(a.band.local_rows) * a.band.width;
    const size_t pix = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (pix >= n) return;
    const f3 cc = xyz(a.colors[pix]);
    const f3 alb = xyz(a.albedo[pix]);
    const f3 out = mix3(cc, alb * cc, a.albedo_factor);
    a.output[pix] = make_float4(out.x, out.y, out.z, 1.0f);
}

// The GENERIC form of the windowed denoiser — one 16x16 output tile per block, one output per lane, the full formula for every tap —
// used when the fast form's premise does not hold (launch_denoise: a sigma_range beyond ~7, outside the reference's GUI range of
// 0.1..5, src/context.rs:1798) and as its cross-check in the tests (VXRT_OPT_DENOISE_MODE | 2).
// kTolerant = false: denoise.comp:64-80 operation for operation (IEEE division by sigma_range_2, the polynomial vx_exp of
// include/vxrt_detmath.h): bit-identical to the oracle; ~85 instructions per tap, of which the division and the exponential are 40.
// kTolerant = true (VXRT_OPT_DENOISE_MODE 1): the same weight as 2^(-(range terms) * log2(e) / sigma_range_2 - distance term * log2(e))
// with the reciprocal folded into one multiplier, fused multiply-adds and the hardware's v_exp_f32 — ~30 instructions per tap.
// The weight's relative error is ~|arg| * 2^-22 <= 2e-5; the filtered colour is a normalised average of such weights
// (tests/test_gpu_pipeline.py: RMSE and maximum error against the oracle at 3840x2160, radius 8).
template <bool kTolerant>
__global__ __launch_bounds__(256) void denoise_generic_kernel(const DenoiseArgs a) {
    extern __shared__ float4 lds_raw[];
    const int r = int(a.radius);
    const int tw = 16 + 2 * r, taps = 2 * r + 1;
    float4* tileA = lds_raw;
    float4* tileB = lds_raw + tw * tw;
    float* wdist = reinterpret_cast<float*>(lds_raw + 2 * tw * tw);  // (dx*dx + dy*dy) / sigma_distance_2 per window offset
    const int x0 = blockIdx.x * 16 - r;
    const int tile_row = a.tile_rows != nullptr ? int(a.tile_rows[blockIdx.y]) : int(blockIdx.y);
    const int lrow0 = tile_row * 16;  // band_rows is a multiple of 16: a tile never straddles two bands
    const int y0 = frame_row(a.band, lrow0) - r;
    const int lband = local_band_of(a.band, lrow0);
    const int band_y0 = band_first_row(a.band, lband * a.band.nranks + a.band.rank);   // first frame row of the band
    const int band_rows_here = band_nominal_rows(a.band, lband * a.band.nranks + a.band.rank);

    for (int i = threadIdx.x; i < taps * taps; i += 256) wdist[i] = a.wdist[i];   // denoise.comp:79, made by launch_denoise
    for (int i = threadIdx.x; i < tw * tw; i += 256) {
        int tx = i % tw, ty = i / tw;
        int gx = x0 + tx, gy = y0 + ty;
        float4 ta = make_float4(0.0f, 0.0f, 0.0f, 0.0f), tb = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(kTapOutside));
        if (gx >= 0 && gx < a.band.width && gy >= 0 && gy < a.band.height) {
            int l = local_row(a.band, gy);
            float4 c, nd;
            int32_t mat = 0;
            bool have = true;
            if (l >= 0) {
                size_t p = size_t(l) * a.band.width + gx;
                c = a.colors[p]; nd = a.nd[p];
                mat = (__float_as_int(a.albedo[p].w) >> 24) & 0xff;                             // denoise.comp:67
            } else if (a.halo.base != nullptr) {
                // a row of a neighbouring rank: side 0 = the rows above the band, side 1 = the rows below it
                const int side = gy < band_y0 ? 0 : 1;
                const int k = side == 0 ? gy - (band_y0 - a.halo.rows) : gy - (band_y0 + band_rows_here);
                const HaloRow row = halo_row(a.halo, a.band.width, side, lband, k);
                const float4 ha = row.a[gx], hb = row.b[gx];
                c = make_float4(ha.x, ha.y, ha.z, 0.0f);
                nd = make_float4(hb.x, hb.y, hb.z, ha.w);
                mat = __float_as_int(hb.w);
            } else {
                have = false;
            }
            if (have) {
                ta = make_float4(c.x, c.y, c.z, vx_log(vx_abs(nd.w)));                         // denoise.comp:66
                if (!finite3(c)) mat |= kTapNonFinite;
                tb = make_float4(nd.x, nd.y, nd.z, __int_as_float(mat));
            }
        }
        tileA[i] = ta;
        tileB[i] = tb;
    }
    __syncthreads();

    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int x = blockIdx.x * 16 + lx, lrow = lrow0 + ly;
    if (x >= a.band.width || lrow >= a.band.local_rows) return;
    const int y = frame_row(a.band, lrow);
    if (y >= a.band.height) return;
    const size_t pix = size_t(lrow) * a.band.width + x;

    const float4 ca = tileA[(ly + r) * tw + (lx + r)], cb = tileB[(ly + r) * tw + (lx + r)];
    const f3 cc = xyz(ca), cn = xyz(cb);
    const float clogd = ca.w;
    const int32_t cmat = __float_as_int(cb.w) & 0xff;
    const f3 ray_dir = pixel_dir(a.cam, x, y);
    const float depth_bias = vx_max(0.0f, dot3(cn, -ray_dir));

    float normalization = 0.0f;
    f3 sum = splat3(0.0f);
    const float range_scale = -1.44269504088896341f / a.sigma_range_2;   // tolerant mode: -log2(e) / sigma_range_2
    for (int dy = -r; dy <= r; dy++) {
        const float4* rowA = tileA + (ly + r + dy) * tw + lx;
        const float4* rowB = tileB + (ly + r + dy) * tw + lx;
        const float* rowW = wdist + (dy + r) * taps;
        for (int dx = 0; dx < taps; dx++) {
            const float4 wa = rowA[dx], wb = rowB[dx];
            const int32_t wflags = __float_as_int(wb.w);
            if (wflags == kTapOutside) continue;
            f3 wc = xyz(wa);
            f3 color_delta = cc - wc;
            f3 normal_delta = cn - xyz(wb);
            float depth_delta = clogd - wa.w;
            float material_delta = cmat != (wflags & 0xff) ? 1.0f : 0.0f;
            float bd = depth_bias * depth_delta;
            if (kTolerant) {
                float q = __builtin_fmaf(color_delta.z, color_delta.z, __builtin_fmaf(color_delta.y, color_delta.y, color_delta.x * color_delta.x));
                float n2 = __builtin_fmaf(normal_delta.z, normal_delta.z, __builtin_fmaf(normal_delta.y, normal_delta.y, normal_delta.x * normal_delta.x));
                n2 = __builtin_fmaf(bd, bd, n2) + material_delta;
                q = __builtin_fmaf(1e4f, n2, q);
                const float factor = __builtin_amdgcn_exp2f(__builtin_fmaf(q, range_scale, -rowW[dx]));
                normalization += factor;
                sum = mk3(__builtin_fmaf(wc.x, factor, sum.x), __builtin_fmaf(wc.y, factor, sum.y), __builtin_fmaf(wc.z, factor, sum.z));
                continue;
            }
            float factor_range = (((dot3(color_delta, color_delta) + 1e4f * dot3(normal_delta, normal_delta)) + 1e4f * (bd * bd)) +
                                  1e4f * material_delta) / a.sigma_range_2;
            float arg = -factor_range - rowW[dx];
            // exp(arg) is exactly +0 below -87.3 (vx_exp): such a tap adds +0 to the weight sum and colour * 0 to the
            // colour sum — nothing, unless the colour is inf/NaN (then 0 * colour = NaN must still poison the sum)
            if (arg < -87.3f && !(wflags & kTapNonFinite)) continue;
            float factor = vx_exp(arg);
            normalization += factor;
            sum = sum + wc * factor;
        }
    }
    f3 out = sum / normalization;
    f3 alb = xyz(a.albedo[pix]);
    out = mix3(out, alb * out, a.albedo_factor);
    a.output[pix] = make_float4(out.x, out.y, out.z, 1.0f);
}

// ---- the windowed denoiser, fast form: TWO outputs per lane ----------------------------------------------------------------------
// A block of 256 threads makes a 32x16 output tile; thread (tx, ty) of 32x8 makes the two vertically adjacent outputs (tx, 2 ty)
// and (tx, 2 ty + 1) and walks the UNION of their windows once: a staged tap is read from LDS once and weighed against both centres
// (LDS reads and address arithmetic per output halve).  Each centre still accumulates its own taps in the shader's order (dy outer,
// dx inner), so the exact mode stays bit-identical to the oracle.  Lanes of a wave are 32 consecutive pixels of a row: the 16-byte
// taps of a lane group are consecutive LDS slots (conflict-free ds_read_b128).
//
// What a staged pixel is: A = (r, g, b, log|depth|) and ONE code word = material id | the normal's components, 2 bits each
// (+-0, +1, -1, 2^30: every normal trace_kernel writes).  With 1e4 / sigma_range_2 > 100 (launch_denoise checks it) a tap whose
// code differs from the centre's has factor_range >= 1e4 / sigma_range_2, its weight exp(-factor_range - ..) is EXACTLY +0
// (vx_exp returns +0 below -87.3), and it adds nothing to either sum: one integer compare replaces the normal and material terms
// of denoise.comp:69-78, and the frame's outside (denoise.comp:57).  A tap whose code equals the centre's has normal_delta = 0 and
// material_delta = 0 exactly, so its factor_range is (dot(color_delta, color_delta) + 1e4 (bias depth_delta)^2) / sigma_range_2, bit
// for bit.  Pixels for which none of this can be said — a colour or log|depth| that is not finite (0 * inf and NaN must still
// poison the sums as they do in the shader), a normal outside that set — carry the sign bit ("exotic") and take the literal
// formula with their operands fetched from global memory again; they are a handful per frame (rays that graze a 0 * inf).
constexpr int32_t kCodeOutside = 0x7fffffff;            // not a pixel of the frame / a row this rank cannot see
constexpr int32_t kCodeIdle = 0x7ffffffe;               // a centre this lane does not make
constexpr int32_t kCodeExotic = int32_t(0x80000000u);

__device__ __forceinline__ int axis_code(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x7fffffffu) == 0u ? 0 : (u == 0x3f800000u ? 1 : (u == 0xbf800000u ? 2 : (u == 0x4e800000u ? 3 : -1)));
}
__device__ __forceinline__ bool finite1(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }

// colour, normal/depth and material id of frame pixel (gx, gy) as this rank sees it (its own rows, or the halo); false: not visible
__device__ __forceinline__ bool fetch_pixel(const DenoiseArgs& a, int gx, int gy, float4& c, float4& nd, int32_t& mat) {
    if (gx < 0 || gx >= a.band.width || gy < 0 || gy >= a.band.height) return false;
    const int l = local_row(a.band, gy);
    if (l >= 0) {
        const size_t p = size_t(l) * a.band.width + gx;
        c = a.colors[p]; nd = a.nd[p];
        mat = (__float_as_int(a.albedo[p].w) >> 24) & 0xff;                                 // denoise.comp:67
        return true;
    }
    HaloRow row;
    if (!halo_find(a.band, a.halo, gy, row)) return false;
    const float4 ha = row.a[gx], hb = row.b[gx];
    c = make_float4(ha.x, ha.y, ha.z, 0.0f);
    nd = make_float4(hb.x, hb.y, hb.z, ha.w);
    mat = __float_as_int(hb.w);
    return true;
}

// denoise.comp:64-80 for one tap, literally (the slow path of the fast form): returns false when the tap is not a pixel
__device__ __forceinline__ bool literal_tap(const DenoiseArgs& a, int gx, int gy, int dx, int dy, f3 cc, f3 cn, float clogd, int32_t cmat,
                                            float depth_bias, f3& wc, float& factor) {
    float4 c, nd;
    int32_t mat;
    if (!fetch_pixel(a, gx, gy, c, nd, mat)) return false;
    wc = xyz(c);
    const f3 color_delta = cc - wc, normal_delta = cn - xyz(nd);
    const float depth_delta = clogd - vx_log(vx_abs(nd.w));
    const float material_delta = cmat != mat ? 1.0f : 0.0f;
    const float bd = depth_bias * depth_delta;
    const float factor_range = (((dot3(color_delta, color_delta) + 1e4f * dot3(normal_delta, normal_delta)) + 1e4f * (bd * bd)) +
                                1e4f * material_delta) / a.sigma_range_2;
    const float factor_distance = float(dx * dx + dy * dy) / a.sigma_distance_2;
    factor = vx_exp(-factor_range - factor_distance);
    return true;
}

struct PairCentre {      // one of a lane's two outputs
    f3 cc;               // colour
    float clogd, bias;   // log|depth|, depth_bias (denoise.comp:49)
    int32_t code;        // its pixel code; kCodeIdle: not made by this lane, or exotic (made by literal_window)
    float norm;
    f3 sum;
};

// One tap against one centre.  kLean: no pixel of the block's apron is exotic, so a code that differs means "weighs nothing" and
// nothing else; in tolerant mode the weight is then computed by every lane and selected (no divergent region at all).
template <bool kTolerant, bool kLean>
__device__ __forceinline__ void pair_tap(const DenoiseArgs& a, PairCentre& ct, float4 wa, int32_t wcode, float wd, float range_scale, int gx, int gy,
                                         int dx, int dy, int cx, int cy) {
    if (kTolerant && kLean) {
        const f3 wc = xyz(wa);
        const f3 cd = ct.cc - wc;
        const float bd = ct.bias * (ct.clogd - wa.w);
        float q = __builtin_fmaf(cd.z, cd.z, __builtin_fmaf(cd.y, cd.y, cd.x * cd.x));
        q = __builtin_fmaf(1e4f * bd, bd, q);
        float f = __builtin_amdgcn_exp2f(__builtin_fmaf(q, range_scale, -wd));
        f = wcode == ct.code ? f : 0.0f;      // plain pixels only: every operand is finite, f is a number, and 0 adds nothing
        ct.norm += f;
        ct.sum = mk3(__builtin_fmaf(wc.x, f, ct.sum.x), __builtin_fmaf(wc.y, f, ct.sum.y), __builtin_fmaf(wc.z, f, ct.sum.z));
        return;
    }
    if (wcode == ct.code) {   // same material, same normal, both plain: the two remaining range terms
        const f3 wc = xyz(wa);
        const f3 cd = ct.cc - wc;
        const float bd = ct.bias * (ct.clogd - wa.w);
        if (kTolerant) {
            float q = __builtin_fmaf(cd.z, cd.z, __builtin_fmaf(cd.y, cd.y, cd.x * cd.x));
            q = __builtin_fmaf(1e4f * bd, bd, q);
            const float f = __builtin_amdgcn_exp2f(__builtin_fmaf(q, range_scale, -wd));
            ct.norm += f;
            ct.sum = mk3(__builtin_fmaf(wc.x, f, ct.sum.x), __builtin_fmaf(wc.y, f, ct.sum.y), __builtin_fmaf(wc.z, f, ct.sum.z));
        } else {
            const float factor_range = (dot3(cd, cd) + 1e4f * (bd * bd)) / a.sigma_range_2;
            const float arg = -factor_range - wd;
            if (!(arg < -87.3f)) {      // below: exp is exactly +0 and the (finite) colour times it adds nothing
                // plain pixels: factor_range and the distance term are finite and >= 0, so arg is a number in [-87.3, 0]
                const float f = vx_exp_in_range(arg);
                ct.norm += f;
                ct.sum = ct.sum + wc * f;
            }
        }
    } else if (!kLean && wcode < 0 && ct.code != kCodeIdle) {   // an exotic tap: the literal formula, operands from global memory
        const size_t pc = size_t(local_row(a.band, cy)) * a.band.width + cx;
        const float4 nd = a.nd[pc];
        const int32_t cmat = (__float_as_int(a.albedo[pc].w) >> 24) & 0xff;
        f3 wc;
        float f;
        if (literal_tap(a, gx, gy, dx, dy, ct.cc, xyz(nd), ct.clogd, cmat, ct.bias, wc, f)) {
            ct.norm += f;
            ct.sum = ct.sum + wc * f;
        }
    }
}

// an exotic centre: its whole window by the literal formula, in the shader's order
__device__ __forceinline__ void literal_window(const DenoiseArgs& a, int cx, int cy, float& norm, f3& sum) {
    const int r = int(a.radius);
    const size_t pc = size_t(local_row(a.band, cy)) * a.band.width + cx;
    const float4 c = a.colors[pc], nd = a.nd[pc];
    const int32_t cmat = (__float_as_int(a.albedo[pc].w) >> 24) & 0xff;
    const f3 cc = xyz(c), cn = xyz(nd);
    const float clogd = vx_log(vx_abs(nd.w));
    const float bias = vx_max(0.0f, dot3(cn, -pixel_dir(a.cam, cx, cy)));
    norm = 0.0f;
    sum = splat3(0.0f);
    for (int dy = -r; dy <= r; dy++)
        for (int dx = -r; dx <= r; dx++) {
            f3 wc;
            float f;
            if (literal_tap(a, cx + dx, cy + dy, dx, dy, cc, cn, clogd, cmat, bias, wc, f)) {
                norm += f;
                sum = sum + wc * f;
            }
        }
}

// The union window of a lane's two centres.  Rows R = -r .. r + 1 relative to the upper centre: row R is tap row dy = R of centre 0
// and dy = R - 1 of centre 1.  kR > 0: the radius at compile time — the taps of a row unroll, their LDS offsets become immediates
// and the row's distance terms arrive in a few wide scalar loads.
template <bool kTolerant, bool kLean, int kR>
__device__ __forceinline__ void pair_window(const DenoiseArgs& a, PairCentre& c0, PairCentre& c1, const float4* tileA, const int32_t* codes, int tw,
                                            int tx, int ty, int x, int ya, float range_scale) {
    const int r = kR > 0 ? kR : int(a.radius);
    const int taps = 2 * r + 1;
    constexpr int kUnroll = kR > 0 ? 2 * kR + 1 : 1;
    for (int R = -r; R <= r + 1; R++) {
        const float4* rowA = tileA + (2 * ty + r + R) * tw + tx;
        const int32_t* rowC = codes + (2 * ty + r + R) * tw + tx;
        const bool use0 = R <= r, use1 = R > -r;
        const float* w0 = a.wdist + (use0 ? R + r : 0) * taps;
        const float* w1 = a.wdist + (use1 ? R - 1 + r : 0) * taps;
#pragma unroll kUnroll
        for (int j = 0; j < (kR > 0 ? 2 * kR + 1 : taps); j++) {
            const float4 wa = rowA[j];
            const int32_t wcode = rowC[j];
            if (use0) pair_tap<kTolerant, kLean>(a, c0, wa, wcode, w0[j], range_scale, x + j - r, ya + R, j - r, R, x, ya);
            if (use1) pair_tap<kTolerant, kLean>(a, c1, wa, wcode, w1[j], range_scale, x + j - r, ya + R, j - r, R - 1, x, ya + 1);
        }
    }
}

template <bool kTolerant, int kR>
// Waves per SIMD the registers must leave room for: what the block's LDS allows (30 KB at radius 8: 5 blocks per CU, one wave of each
// per SIMD) or 6.  Without the bound the multi-rank row mapping inlined into the staging loop took the tolerant radius-8 kernel from 77
// to 106 VGPRs — 4 waves per SIMD, 13 % slower at the same instruction count (round 4; found in profiles/r04/post_stages_summary.json).
__global__ __launch_bounds__(256, (kR >= 6 ? 5 : 6)) void denoise_pair_kernel(const DenoiseArgs a) {
    extern __shared__ float4 lds_raw[];
    __shared__ int block_exotic;
    const int r = kR;
    const int tw = 32 + 2 * r, th = 16 + 2 * r;
    float4* tileA = lds_raw;
    int32_t* codes = reinterpret_cast<int32_t*>(lds_raw + tw * th);
    const int x0 = blockIdx.x * 32 - r;
    const int tile_row = a.tile_rows != nullptr ? int(a.tile_rows[blockIdx.y]) : int(blockIdx.y);
    const int lrow0 = tile_row * 16;  // band_rows is a multiple of 16: a tile never straddles two bands
    const int y0 = frame_row(a.band, lrow0) - r;

    if (threadIdx.x == 0) block_exotic = 0;
    __syncthreads();
    bool any_exotic = false;
    for (int i = threadIdx.x; i < tw * th; i += 256) {
        const int tx = i % tw, ty = i / tw;
        float4 c, nd;
        int32_t mat;
        float4 ta = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        int32_t code = kCodeOutside;
        if (fetch_pixel(a, x0 + tx, y0 + ty, c, nd, mat)) {
            const float logd = vx_log(vx_abs(nd.w));                                            // denoise.comp:66
            const int ax = axis_code(nd.x), ay = axis_code(nd.y), az = axis_code(nd.z);
            const bool plain = (ax | ay | az) >= 0 && finite3(c) && finite1(logd);
            ta = make_float4(c.x, c.y, c.z, logd);
            code = plain ? (mat | ax << 8 | ay << 10 | az << 12) : (kCodeExotic | mat);
            any_exotic |= !plain;
        }
        tileA[i] = ta;
        codes[i] = code;
    }
    if (any_exotic) block_exotic = 1;
    __syncthreads();

    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x = blockIdx.x * 32 + tx;
    const int lrow_a = lrow0 + 2 * ty;
    const int ya = frame_row(a.band, lrow_a);     // the pair's rows are neighbours in the frame too (one band)
    const bool in_x = x < a.band.width;
    const bool made0 = in_x && lrow_a < a.band.local_rows && ya < a.band.height;
    const bool made1 = in_x && lrow_a + 1 < a.band.local_rows && ya + 1 < a.band.height;
    if (!made0 && !made1) return;

    auto centre = [&](int k, bool made, bool& exotic) {
        PairCentre ct;
        const int idx = (2 * ty + k + r) * tw + (tx + r);
        const float4 ca = tileA[idx];
        const int32_t code = codes[idx];
        ct.cc = xyz(ca);
        ct.clogd = ca.w;
        ct.norm = 0.0f;
        ct.sum = splat3(0.0f);
        exotic = made && code < 0;
        ct.code = (made && code >= 0) ? code : kCodeIdle;
        ct.bias = 0.0f;
        if (made) {
            const float4 nd = a.nd[size_t(lrow_a + k) * a.band.width + x];
            ct.bias = vx_max(0.0f, dot3(xyz(nd), -pixel_dir(a.cam, x, ya + k)));                  // denoise.comp:49
        }
        return ct;
    };
    bool exotic0, exotic1;
    PairCentre c0 = centre(0, made0, exotic0), c1 = centre(1, made1, exotic1);

    const float range_scale = -1.44269504088896341f / a.sigma_range_2;   // tolerant mode: -log2(e) / sigma_range_2
    if (block_exotic == 0) {
        pair_window<kTolerant, true, kR>(a, c0, c1, tileA, codes, tw, tx, ty, x, ya, range_scale);
    } else {   // a non-finite colour or depth, or an odd normal, somewhere in this block's apron: the careful loop (radius at run time: small code)
        pair_window<kTolerant, false, 0>(a, c0, c1, tileA, codes, tw, tx, ty, x, ya, range_scale);
        if (exotic0) literal_window(a, x, ya, c0.norm, c0.sum);
        if (exotic1) literal_window(a, x, ya + 1, c1.norm, c1.sum);
    }
    auto finish = [&](const PairCentre& ct, int k) {
        const size_t pix = size_t(lrow_a + k) * a.band.width + x;
        f3 out = ct.sum / ct.norm;
        const f3 alb = xyz(a.albedo[pix]);
        out = mix3(out, alb * out, a.albedo_factor);
        a.output[pix] = make_float4(out.x, out.y, out.z, 1.0f);
    };
    if (made0) finish(c0, 0);
    if (made1) finish(c1, 1);
}

// N samples per pixel = the mean of N consecutive trace frames, summed left to right in binary32 and divided once.
__global__ __launch_bounds__(256) void spp_accumulate_kernel(const SppArgs a) {
    const size_t p = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (p >= a.pixels) return;
    float4 s = a.first ? a.frames[0][p] : a.sum[p];
    for (int k = a.first ? 1 : 0; k < a.count; k++) {
        const float4 c = a.frames[k][p];
        s = make_float4(s.x + c.x, s.y + c.y, s.z + c.z, s.w + c.w);
    }
    if (a.last) {
        const float n = float(a.total);
        a.out[p] = make_float4(s.x / n, s.y / n, s.z / n, s.w / n);
    } else {
        a.sum[p] = s;
    }
}

__global__ void detmath_probe_kernel(int fn, const float* x, const float* y, float* out, size_t n) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = 0.0f;
    switch (fn) {
        case 0: v = vx_sin(x[i]); break;
        case 1: v = vx_cos(x[i]); break;
        case 2: v = vx_exp(x[i]); break;
        case 3: v = vx_log(x[i]); break;
        case 4: v = vx_pow(x[i], y[i]); break;
        case 5: v = vx_sqrt(x[i]); break;
        case 6: v = x[i] / y[i]; break;
        case 7: v = vx_tan(x[i]); break;
        case 8: {  // normalize + dot + cross chain, the shape of most shading arithmetic
            f3 a = norm3(mk3(x[i], y[i], x[i] * y[i] + 0.25f));
            f3 b = cross3(a, mk3(y[i], x[i], 1.0f));
            v = dot3(a, b) + len3(b);
            break;
        }
        case 9: case 10: {  // the y / z of random_hemisphere (voxels.comp:277-287) before the flip: signs of zero matter there
            float phi = (2.0f * 3.14159265358979f) * x[i];
            float rx = 2.0f * y[i] - 1.0f;
            float plane_radius = vx_sqrt(1.0f - rx * rx);
            v = fn == 9 ? plane_radius * vx_cos(phi) : plane_radius * vx_sin(phi);
            break;
        }
        case 11: v = x[i] * y[i]; break;
        case 12: v = x[i] - y[i]; break;
        case 13: v = x[i] - y[i] * vx_min0(2.0f * x[i]); break;
        case 14: v = vx_min(x[i], y[i]); break;
        case 15: v = vx_max(x[i], y[i]); break;
        case 16: v = vx_max(0.0f, x[i]) * y[i]; break;
        case 17: v = vx_sign(x[i]) * y[i]; break;
        case 18: v = vx_clamp(x[i], y[i], 1.0f); break;
        case 19: v = vx_min0(x[i]) * y[i]; break;
    }
    out[i] = v;
}

}  // namespace

hipError_t launch_temporal(const TemporalArgs& a, hipStream_t s) {
    dim3 grid((a.band.width + 63) / 64, (a.band.local_rows + 3) / 4);
    hipLaunchKernelGGL(temporal_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_denoise(const DenoiseArgs& args, hipStream_t s) {
    if (args.radius == 0u) {
        const size_t n = size_t(args.band.local_rows) * args.band.width;
        hipLaunchKernelGGL(denoise_passthrough_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, args);
        return hipGetLastError();
    }
    DenoiseArgs a = args;
    const int r = int(a.radius), taps = 2 * r + 1;
    const bool tolerant = (a.mode & 1) != 0;
    // factor_distance of denoise.comp:79 per window offset (an IEEE division, here on the host: the same binary32 quotient); in
    // tolerant mode in units of log 2 — the weight is an exp2 there
    for (int dy = -r; dy <= r; dy++)
        for (int dx = -r; dx <= r; dx++) {
            float w = float(dx * dx + dy * dy) / a.sigma_distance_2;
            if (tolerant) w *= 1.44269504088896341f;
            a.wdist[(dy + r) * taps + (dx + r)] = w;
        }
    const unsigned tile_rows = a.tile_rows != nullptr ? a.tile_row_count : unsigned(a.band.local_rows + 15) / 16u;
    if (tile_rows == 0u) return hipSuccess;
    // The fast form's premise: a tap of another material or normal weighs exactly nothing — factor_range >= 1e4 / sigma_range_2 > 100
    // puts the exponent far below vx_exp's -87.3 (and v_exp_f32's -126 in units of log 2).  Holds for every sigma_range < 7.07; the
    // reference's GUI offers 0.1 .. 5 (src/context.rs:1798).  NaN / non-positive sigma_range_2: the generic form, which says what the shader says.
    const bool fast = (a.mode & 2) == 0 && 1e4f / a.sigma_range_2 > 100.0f && a.sigma_range_2 > 0.0f;
    if (fast) {
        dim3 grid((a.band.width + 31) / 32, tile_rows);
        const size_t lds = size_t(32 + 2 * r) * size_t(16 + 2 * r) * 20;
        switch (r * 2 + (tolerant ? 1 : 0)) {
#define VXRT_PAIR(R)                                                                                            \
            case R * 2: hipLaunchKernelGGL((denoise_pair_kernel<false, R>), grid, dim3(256), lds, s, a); break; \
            case R * 2 + 1: hipLaunchKernelGGL((denoise_pair_kernel<true, R>), grid, dim3(256), lds, s, a); break;
            VXRT_PAIR(1) VXRT_PAIR(2) VXRT_PAIR(3) VXRT_PAIR(4) VXRT_PAIR(5) VXRT_PAIR(6) VXRT_PAIR(7) VXRT_PAIR(8)
#undef VXRT_PAIR
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    dim3 grid((a.band.width + 15) / 16, tile_rows);
    int tw = 16 + 2 * r;
    size_t lds = size_t(tw) * tw * 32 + size_t(taps * taps + 3) / 4 * 16;
    if (tolerant)
        hipLaunchKernelGGL(denoise_generic_kernel<true>, grid, dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL(denoise_generic_kernel<false>, grid, dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_spp_accumulate(const SppArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(spp_accumulate_kernel, dim3(unsigned((a.pixels + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_detmath_probe(int fn, const float* x, const float* y, float* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(detmath_probe_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, fn, x, y, out, n);
    return hipGetLastError();
}

}  // namespace vxrt
