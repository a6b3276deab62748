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

// One 16x16 output tile per block, radius 1..8.
// kTolerant = false: denoise.comp:64-80 operation for operation (IEEE division by sigma_range_2, the polynomial vx_exp of
// include/vxrt_detmath.h): bit-identical to the oracle; ~85 instructions per tap, of which the division and the exponential are 40.
// kTolerant = true (VXRT_OPT_DENOISE_MODE 1): the same weight as 2^(-(range terms) * log2(e) / sigma_range_2 - distance term * log2(e))
// with the reciprocal folded into one multiplier, fused multiply-adds and the hardware's v_exp_f32 — ~30 instructions per tap.
// The weight's relative error is ~|arg| * 2^-22 <= 2e-5; the filtered colour is a normalised average of such weights
// (tests/test_gpu_pipeline.py: RMSE and maximum error against the oracle at 3840x2160, radius 8).
template <bool kTolerant>
__global__ __launch_bounds__(256) void denoise_kernel(const DenoiseArgs a) {
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
    const int lband = lrow0 / a.band.band_rows;
    const int band_y0 = (lband * a.band.nranks + a.band.rank) * a.band.band_rows;  // first frame row of the band

    for (int i = threadIdx.x; i < taps * taps; i += 256) {
        const int dx = i % taps - r, dy = i / taps - r;
        wdist[i] = float(dx * dx + dy * dy) / a.sigma_distance_2;   // denoise.comp:79
        if (kTolerant) wdist[i] *= 1.44269504088896341f;            // in units of log 2: the weight is an exp2 there
    }
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
                const int k = side == 0 ? gy - (band_y0 - a.halo.rows) : gy - (band_y0 + a.band.band_rows);
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

hipError_t launch_denoise(const DenoiseArgs& a, hipStream_t s) {
    if (a.radius == 0u) {
        const size_t n = size_t(a.band.local_rows) * a.band.width;
        hipLaunchKernelGGL(denoise_passthrough_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    dim3 grid((a.band.width + 15) / 16, a.tile_rows != nullptr ? a.tile_row_count : unsigned(a.band.local_rows + 15) / 16u);
    if (grid.y == 0u) return hipSuccess;
    int tw = 16 + 2 * int(a.radius), taps = 2 * int(a.radius) + 1;
    size_t lds = size_t(tw) * tw * 32 + size_t(taps * taps + 3) / 4 * 16;
    if (a.mode == 1)
        hipLaunchKernelGGL(denoise_kernel<true>, grid, dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL(denoise_kernel<false>, grid, dim3(256), lds, s, a);
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
