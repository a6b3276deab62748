// trace.hip — the path-trace kernel of libvxrt for gfx950 (CDNA4, wave64).
//
// Computes what shaders/voxels.comp computes (one thread per pixel: primary ray, up to max_bounces
// diffuse/specular bounces with one sun shadow ray each, blue-noise RNG, three rgba32f outputs), with
// the result contract of its octree walk cast_bounded_ray (voxels.comp:134-247): same visited
// (node, octant, time) sequence, same iteration cap, same tie-breaking — but re-shaped for the machine:
//
//  * scene = 8-byte SVO records (kernels.h) instead of 32-byte nodes: a sibling step touches no memory,
//    a descend is one 8-byte load, the leaf word is fetched once per hit;
//  * the per-ray stack (voxels.comp:127-130: 16 x {node, octant}) is split: the octant / next-octant /
//    has-next fields live in three registers as bit fields, the node records in LDS laid out
//    [level][thread] (conflict-free ds_write_b64 / ds_read_b64); a multi-level pop
//    (voxels.comp:227-234) is one find-first-set on the has-next mask instead of a loop;
//  * node centres are rebuilt from integer path coordinates — all cube geometry is dyadic, so this is
//    exact and equal to the shader's incremental float updates;
//  * primary, bounce and sun rays of a pixel run through ONE traversal loop (a small state machine),
//    so lanes that are in different shading phases still execute the traversal together;
//  * a wave covers an 8x8 pixel tile (coherent primary rays, 128-byte output segments).
//
// Arithmetic follows include/vxrt_detmath.h: every float operation that can change a result is the
// shader's operation, in the shader's order, never contracted.
#include "kernels.h"
#include "vx_vec.h"

namespace vxrt {
namespace {

constexpr float kAlmostInfinity = 1073741824.0f;  // float(1 << 30)  voxels.comp:8
constexpr int32_t kLeafBit = int32_t(0x80000000u);
constexpr int32_t kEmitBit = 1 << 30;
constexpr int kBlock = 256;
constexpr uint32_t kNoiseLayer = 128u * 128u;
constexpr uint32_t kNoiseTotal = kNoiseLayer * 512u;

struct RayHit {
    float time;
    int32_t node;
    f3 normal;
};

// ray_cube_intersection, voxels.comp:73-90
__device__ __forceinline__ bool slab(f3 o, f3 inv, f3 sg, f3 c, float half, float& entry, float& exit) {
    f3 hs = half * sg;
    f3 en = ((c - hs) - o) * inv;
    f3 ex = ((c + hs) - o) * inv;
    entry = vx_max(vx_max(en.x, en.y), en.z);
    exit = vx_min(vx_min(ex.x, ex.y), ex.z);
    return exit >= 0.0f && entry < exit;
}

// current_octant, voxels.comp:119-125 (strict >: ties go to the low side)
__device__ __forceinline__ uint32_t octant_of(f3 p, f3 c) {
    return ((p.x - c.x) > 0.0f ? 4u : 0u) + ((p.y - c.y) > 0.0f ? 2u : 0u) + ((p.z - c.z) > 0.0f ? 1u : 0u);
}

struct SceneView {
    const SvoRecord* svo;
    const int32_t* leaves;
    f3 root_center;
    f3 root_min;
    float root_size;
};

// cast_bounded_ray, voxels.comp:134-247.  `stack` points at this thread's column of the LDS stack
// (entry l at stack[l * kBlock]).  On the iteration cap the shader returns true without writing the
// normal; it is defined as 0 here (oracle U1).
__device__ __forceinline__ bool cast_ray(const SceneView& sc, f3 o, f3 d, float max_distance, uint2* stack, RayHit& hit) {
    const uint32_t dir_mask = (d.x < 0.0f ? 4u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 1u : 0u);
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const f3 sg = mk3(vx_sign(inv.x), vx_sign(inv.y), vx_sign(inv.z));

    float entry, exit;
    if (!slab(o, inv, sg, sc.root_center, 0.5f * sc.root_size, entry, exit)) return false;

    float time = vx_max(0.0f, entry);
    float size = sc.root_size;
    f3 center = sc.root_center;
    uint32_t ix = 0, iy = 0, iz = 0;  // integer path coordinates of the current node, `lvl` bits each
    uint32_t lvl = 0;
    uint32_t has_next_mask = 0;       // bit l: level l can still advance to a sibling (frame.node != -1)
    uint64_t next_octants = 0;        // 3 bits per level: the sibling to resume with
    SvoRecord rec = sc.svo[0];
    uint32_t octant = octant_of(o + d * time, center);

    hit.normal = splat3(0.0f);
    for (int iterations = 1;; iterations++) {
        if (iterations >= 2048) {  // voxels.comp:166-169
            hit.node = kLeafBit;
            hit.time = time;
            return true;
        }
        if (time > max_distance) return false;

        const uint32_t bit = 1u << octant;
        if (rec.masks & (bit << 8)) {  // value < 0: leaf                voxels.comp:177-189
            hit.node = sc.leaves[rec.base + __popc((rec.masks >> 8) & (bit - 1u))];
            hit.time = time;
            f3 p = o + time * d;
            f3 delta = mk3(float((octant >> 2) & 1u), float((octant >> 1) & 1u), float(octant & 1u));
            f3 oc = center + (0.5f * size) * (delta - splat3(0.5f));
            f3 dist = mk3(vx_abs(p.x - oc.x), vx_abs(p.y - oc.y), vx_abs(p.z - oc.z));
            float m = vx_max(vx_max(dist.x, dist.y), dist.z);
            f3 mask = mk3(dist.x == m ? 1.0f : 0.0f, dist.y == m ? 1.0f : 0.0f, dist.z == m ? 1.0f : 0.0f);
            hit.normal = mask * mk3(-vx_sign(d.x), -vx_sign(d.y), -vx_sign(d.z));
            return true;
        }

        // next sibling through the node's mid planes                     voxels.comp:191-203
        const f3 t_mid = (center - o) * inv;
        const uint32_t directional = octant ^ dir_mask;
        const float mx = (directional & 4u) ? kAlmostInfinity : t_mid.x;
        const float my = (directional & 2u) ? kAlmostInfinity : t_mid.y;
        const float mz = (directional & 1u) ? kAlmostInfinity : t_mid.z;
        const float next_time = vx_min(vx_min(mx, my), mz);
        const uint32_t transition = (mx == next_time) ? 4u : ((my == next_time) ? 2u : ((mz == next_time) ? 1u : 0u));
        const uint32_t next_octant = octant ^ transition;
        const bool has_next = next_time <= exit && transition != 0u && (directional & transition) == 0u;

        if (rec.masks & bit) {  // value > 0: descend                       voxels.comp:205-221
            if (has_next) {
                stack[lvl * kBlock] = make_uint2(rec.masks, rec.base);
                has_next_mask |= 1u << lvl;
                next_octants = (next_octants & ~(uint64_t(7) << (3u * lvl))) | (uint64_t(next_octant) << (3u * lvl));
            }
            const uint32_t child = rec.base + __popc(rec.masks & (bit - 1u));
            const uint2 raw = *reinterpret_cast<const uint2*>(sc.svo + child);
            rec.masks = raw.x;
            rec.base = raw.y;
            ix = (ix << 1) | ((octant >> 2) & 1u);
            iy = (iy << 1) | ((octant >> 1) & 1u);
            iz = (iz << 1) | (octant & 1u);
            lvl++;
            size *= 0.5f;
            center = sc.root_min + mk3(float(ix) + 0.5f, float(iy) + 0.5f, float(iz) + 0.5f) * size;
            octant = octant_of(o + d * time, center);
            float child_entry;
            slab(o, inv, sg, center, 0.5f * size, child_entry, exit);
            time = vx_max(time, child_entry);
        } else if (has_next) {  // empty slot, step to the sibling           voxels.comp:222-224
            octant = next_octant;
            time = next_time;
        } else {                // pop to the nearest level that can still advance   voxels.comp:225-243
            if (has_next_mask == 0u) return false;
            const uint32_t l = 31u - uint32_t(__clz(int(has_next_mask)));
            has_next_mask &= ~(1u << l);
            const uint32_t up = lvl - l;
            ix >>= up; iy >>= up; iz >>= up;
            lvl = l;
            size = __builtin_ldexpf(sc.root_size, -int(l));
            center = sc.root_min + mk3(float(ix) + 0.5f, float(iy) + 0.5f, float(iz) + 0.5f) * size;
            const uint2 raw = stack[l * kBlock];
            rec.masks = raw.x;
            rec.base = raw.y;
            time = exit;
            float unused_entry;
            slab(o, inv, sg, center, 0.5f * size, unused_entry, exit);
            octant = uint32_t(next_octants >> (3u * l)) & 7u;
        }
    }
}

__device__ __forceinline__ f3 node_rgb(int32_t node) {
    return mk3(float((node >> 16) & 0xff), float((node >> 8) & 0xff), float(node & 0xff));
}
// node_color, voxels.comp:253-258
__device__ __forceinline__ f3 node_color(int32_t node) { return node_rgb(node) / 255.0f; }
// node_emmitance, voxels.comp:260-266
__device__ __forceinline__ f3 node_emittance(int32_t node, float emit_strength) {
    float e = (node & kEmitBit) != 0 ? 1.0f : 0.0f;
    return ((e * emit_strength) * node_rgb(node)) / 255.0f;
}

struct Rng {  // rand(), voxels.comp:268-275
    uint32_t index;
    const float* noise;
    __device__ __forceinline__ float next() {
        index = (index + kNoiseLayer) % kNoiseTotal;
        return noise[index];
    }
};

// random_hemisphere, voxels.comp:277-287
__device__ __forceinline__ f3 random_hemisphere(f3 n, Rng& rng) {
    float phi = (2.0f * 3.14159265358979f) * rng.next();
    f3 r;
    r.x = 2.0f * rng.next() - 1.0f;
    float plane_radius = vx_sqrt(1.0f - r.x * r.x);
    r.y = plane_radius * vx_cos(phi);
    r.z = plane_radius * vx_sin(phi);
    return r - n * vx_min(0.0f, 2.0f * dot3(n, r));
}

__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

__global__ __launch_bounds__(kBlock) void trace_kernel(const TraceArgs a) {
    extern __shared__ uint2 lds_stack[];  // [stack_levels][kBlock]
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    // 16x16 pixel tile per block, one 8x8 sub-tile per wave
    const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int lrow = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    const int lband = lrow / a.band.band_rows;
    const int y = (lband * a.band.nranks + a.band.rank) * a.band.band_rows + (lrow - lband * a.band.band_rows);
    const bool active = x < a.band.width && lrow < a.band.local_rows && y < a.band.height;

    uint32_t rays = 0;
    if (active) {
        SceneView sc;
        sc.svo = a.svo;
        sc.leaves = a.leaves;
        sc.root_center = ld3(a.root_center);
        sc.root_size = a.root_size;
        sc.root_min = sc.root_center - splat3(0.5f * a.root_size);
        uint2* stack = lds_stack + tid;
        const size_t pix = size_t(lrow) * a.band.width + x;

        Rng rng;
        rng.noise = a.noise;
        rng.index = uint32_t(x) % 128u + (uint32_t(y) % 128u) * 128u + (a.frame_number % 512u) * kNoiseLayer;

        const f3 sun_dir = ld3(a.sun_dir), sun_color = ld3(a.sun_color), sky = ld3(a.sky_color);

        f3 o = ld3(a.cam.o);
        f3 d = norm3((float(x) * ld3(a.cam.r) - float(y) * ld3(a.cam.u)) + ld3(a.cam.f));  // voxels.comp:299-303

        f3 sample = splat3(0.0f), blend = splat3(1.0f);
        uint32_t ambient_rays = 1;
        int bounce = 0;
        bool sun_phase = false;
        f3 pend_sun = splat3(0.0f), pend_emit = splat3(0.0f), next_dir = splat3(0.0f);

        for (;;) {
            RayHit hit;
            rays++;
            const bool is_hit = cast_ray(sc, o, d, kAlmostInfinity, stack, hit);

            if (sun_phase) {  // back from the sun shadow ray                 voxels.comp:357-371
                if (!is_hit) sample = sample + pend_sun;
                sample = sample + pend_emit;
                d = next_dir;
                sun_phase = false;
                if (++bounce >= a.max_bounces) break;
                continue;
            }

            if (!is_hit) {  // sky                                              voxels.comp:373-388
                if (bounce == 0) {
                    blend = splat3(1.0f);
                    float sun_power = vx_pow(vx_max(0.0f, dot3(d, ld3(a.neg_sun_dir_n))), a.sun_exponent);
                    sample = sample + (sky + sun_color * sun_power) * blend;
                    a.out_nd[pix] = make_float4(kAlmostInfinity, kAlmostInfinity, kAlmostInfinity, -1.0f);
                    a.out_albedo[pix] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(0xffffff));
                } else {
                    sample = sample + sky * blend;
                }
                break;
            }

            const f3 n = hit.normal;
            const f3 hit_pos = o + d * hit.time;
            const f3 color = bounce == 0 ? splat3(1.0f) : node_color(hit.node);
            const f3 emit = node_emittance(hit.node, a.emit_strength);
            if (bounce == 0) {  // first-hit G-buffer                           voxels.comp:320-324,392-396
                a.out_nd[pix] = make_float4(n.x, n.y, n.z, hit.time);
                f3 alb = (hit.node & kEmitBit) == 0 ? node_color(hit.node) : splat3(1.0f);
                a.out_albedo[pix] = make_float4(alb.x, alb.y, alb.z, __int_as_float(hit.node));
            }

            if (rng.next() < a.specularity) {  // specular                     voxels.comp:326-334
                f3 refl = norm3(reflect3(d, n));
                sample = sample + emit * blend;
                blend = blend * ((2.0f * color) * dot3(refl, n));
                o = hit_pos + 1e-5f * n;
                d = refl;
            } else if (a.sun_strength > 0.0f) {  // diffuse + sun sample         voxels.comp:339-371
                float r0 = rng.next(), r1 = rng.next(), r2 = rng.next();
                f3 up_dir = norm3(cross3(mk3(r0, r1, r2), sun_dir));
                f3 right_dir = norm3(cross3(sun_dir, up_dir));
                float dx = 2.0f * rng.next() - 1.0f;
                float dy = 2.0f * rng.next() - 1.0f;
                f3 light_dir = ld3(a.sun_dir_n) + (dx * right_dir + dy * up_dir) * a.sun_size;
                f3 to_light = norm3(-light_dir);
                ambient_rays++;
                pend_sun = ((sun_color * color) * blend) * vx_max(0.0f, dot3(n, to_light));
                f3 refl = random_hemisphere(n, rng);
                pend_emit = emit * blend;
                blend = blend * (color * dot3(n, refl));
                o = hit_pos + 1e-5f * n;
                next_dir = refl;
                d = to_light;
                sun_phase = true;
                continue;
            } else {  // diffuse, sun switched off
                f3 refl = random_hemisphere(n, rng);
                sample = sample + emit * blend;
                blend = blend * (color * dot3(n, refl));
                o = hit_pos + 1e-5f * n;
                d = refl;
            }
            if (++bounce >= a.max_bounces) break;
        }

        f3 out = sample / float(ambient_rays);  // voxels.comp:391
        a.out_color[pix] = make_float4(out.x, out.y, out.z, 1.0f);
    }

    // rays cast by this wave -> one atomic
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
    if (lane == 0 && rays != 0) atomicAdd(a.ray_counter, (unsigned long long)rays);
}

}  // namespace

hipError_t launch_trace(const TraceArgs& a, hipStream_t s) {
    dim3 grid((a.band.width + 15) / 16, (a.band.local_rows + 15) / 16);
    size_t lds = size_t(a.stack_levels) * kBlock * sizeof(uint2);
    hipLaunchKernelGGL(trace_kernel, grid, dim3(kBlock), lds, s, a);
    return hipGetLastError();
}

}  // namespace vxrt
