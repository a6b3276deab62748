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
//
// Shape of the loop (what differs from the shader's text, none of it changes a result):
//  * descend (voxels.comp:205-221) and pop (:225-243) share one code path for everything they have in
//    common — new integer path coordinates, node size and centre, the slab test — so a wave whose lanes
//    are split between the two executes that code once, not twice;
//  * a saved frame is {masks | next_octant << 16, base}: the sibling to resume with travels with the
//    node record in LDS, and only frames that can still advance are stored (the shader's node == -1
//    "complete" frames are never read back: a pop goes straight to the highest level whose bit is set
//    in has_next_mask);
//  * leaving the loop (leaf, miss, iteration cap) only sets a status; the leaf word load and the
//    normal computation happen once after the loop for all lanes of the wave together, instead of
//    inside the loop each time a single lane hits.
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
    SvoRecord rec = sc.svo[0];
    uint32_t octant = octant_of(o + d * time, center);

    enum { kLeaf = 1, kMiss = 2, kCap = 3 };
    int status;
    for (int iterations = 1;; iterations++) {
        if (iterations >= 2048) { status = kCap; break; }        // voxels.comp:166-169
        if (time > max_distance) { status = kMiss; break; }      // voxels.comp:171-173
        const uint32_t bit = 1u << octant;
        if (rec.masks & (bit << 8)) { status = kLeaf; break; }   // value < 0

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
        const bool is_child = (rec.masks & bit) != 0u;           // value > 0

        if (is_child || !has_next) {
            uint2 raw;
            if (is_child) {  // descend: remember where to resume, fetch the child record   voxels.comp:205-214
                if (has_next) {
                    stack[lvl * kBlock] = make_uint2(rec.masks | next_octant << 16, rec.base);
                    has_next_mask |= 1u << lvl;
                }
                raw = *reinterpret_cast<const uint2*>(sc.svo + (rec.base + __popc(rec.masks & (bit - 1u))));
                ix = (ix << 1) | ((octant >> 2) & 1u);
                iy = (iy << 1) | ((octant >> 1) & 1u);
                iz = (iz << 1) | (octant & 1u);
                lvl++;
            } else {  // pop to the nearest level that can still advance                     voxels.comp:225-234
                if (has_next_mask == 0u) { status = kMiss; break; }
                const uint32_t l = 31u - uint32_t(__clz(int(has_next_mask)));
                has_next_mask &= ~(1u << l);
                const uint32_t up = lvl - l;
                ix >>= up; iy >>= up; iz >>= up;
                lvl = l;
                raw = stack[l * kBlock];
                // consume the LDS read here: left alone, the compiler merges it with the descend branch's global
                // load into one flat_load (either address space), which is slower and waits on both counters
                asm volatile("" : "+v"(raw.x), "+v"(raw.y));
            }
            size = __builtin_ldexpf(sc.root_size, -int(lvl));
            center = sc.root_min + mk3(float(ix) + 0.5f, float(iy) + 0.5f, float(iz) + 0.5f) * size;
            float node_entry, node_exit;
            slab(o, inv, sg, center, 0.5f * size, node_entry, node_exit);
            if (is_child) {  // voxels.comp:216-221
                octant = octant_of(o + d * time, center);
                time = vx_max(time, node_entry);
            } else {         // voxels.comp:236-242
                time = exit;
                octant = (raw.x >> 16) & 7u;
            }
            exit = node_exit;
            rec.masks = raw.x & 0xffffu;
            rec.base = raw.y;
        } else {  // empty slot, step to the sibling                                         voxels.comp:222-224
            octant = next_octant;
            time = next_time;
        }
    }

    hit.time = time;
    hit.normal = splat3(0.0f);
    if (status == kMiss) return false;
    if (status == kCap) {
        hit.node = kLeafBit;
        return true;
    }
    // leaf                                                                                   voxels.comp:177-189
    const uint32_t bit = 1u << octant;
    hit.node = sc.leaves[rec.base + __popc((rec.masks >> 8) & (bit - 1u))];
    f3 p = o + time * d;
    f3 delta = mk3(float((octant >> 2) & 1u), float((octant >> 1) & 1u), float(octant & 1u));
    f3 oc = center + (0.5f * size) * (delta - splat3(0.5f));
    f3 dist = mk3(vx_abs(p.x - oc.x), vx_abs(p.y - oc.y), vx_abs(p.z - oc.z));
    float m = vx_max(vx_max(dist.x, dist.y), dist.z);
    f3 mask = mk3(dist.x == m ? 1.0f : 0.0f, dist.y == m ? 1.0f : 0.0f, dist.z == m ? 1.0f : 0.0f);
    hit.normal = mask * mk3(-vx_sign(d.x), -vx_sign(d.y), -vx_sign(d.z));
    return true;
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

// Rays cast by this wave -> one atomic on one of kRaySlots counters, each on a 64-byte line of its own.
// (A single counter word saturates at ~88 atomics/us chip-wide: with one atomic per wave that alone
// put a 0.4 ms floor under a 1080p frame.)
__device__ __forceinline__ void count_rays(unsigned long long* slots, uint32_t rays, int lane) {
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
    if (lane == 0 && rays != 0) {
        const unsigned slot = (blockIdx.x + blockIdx.y * gridDim.x) * 4u + (threadIdx.x >> 6);
        atomicAdd(slots + size_t(slot % kRaySlots) * 8u, (unsigned long long)rays);
    }
}

__global__ __launch_bounds__(kBlock) void trace_kernel(const TraceArgs a) {
    extern __shared__ uint2 lds_stack[];  // [stack_levels][kBlock]
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    // 16x16 pixel tile per block, one 8x8 sub-tile per wave.  Blocks take tiles in the order of tile_order
    // (longest tile of the previous frame first): a frame's cost is concentrated in the tiles that see
    // geometry, and started last they would leave the chip idling behind a few long waves.
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    const unsigned tiles_x = unsigned(a.band.width + 15) / 16u;
    const unsigned tile = a.tile_order ? a.tile_order[blockIdx.x] : blockIdx.x;
    const int x = int(tile % tiles_x) * 16 + (wave & 1) * 8 + (lane & 7);
    const int lrow = int(tile / tiles_x) * 16 + (wave >> 1) * 8 + (lane >> 3);
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

    count_rays(a.ray_counter, rays, lane);
    if (a.tile_cost && lane == 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_start;
        atomicMax(a.tile_cost + tile, dt > 0xffffffffull ? 0xffffffffu : uint32_t(dt));
    }
}

// Counting sort of the tiles by descending cost (key = log2 of the cost with two mantissa bits), one block.
__global__ __launch_bounds__(1024) void tile_order_kernel(uint32_t* cost, uint32_t* order, uint32_t* last_cost, unsigned tiles) {
    __shared__ unsigned hist[128], offs[128];
    const unsigned tid = threadIdx.x;
    if (tid < 128) hist[tid] = 0;
    __syncthreads();
    auto key_of = [](uint32_t c) -> unsigned {
        if (c < 4u) return c;
        const unsigned e = 31u - unsigned(__clz(int(c)));
        return (e << 2 | ((c >> (e - 2u)) & 3u)) - 4u;   // 4..127, monotone in c
    };
    for (unsigned t = tid; t < tiles; t += blockDim.x) atomicAdd(&hist[127u - key_of(cost[t])], 1u);
    __syncthreads();
    if (tid == 0) {
        unsigned run = 0;
        for (int k = 0; k < 128; k++) { offs[k] = run; run += hist[k]; }
    }
    __syncthreads();
    for (unsigned t = tid; t < tiles; t += blockDim.x) {
        const unsigned pos = atomicAdd(&offs[127u - key_of(cost[t])], 1u);
        order[pos] = t;
    }
    __syncthreads();
    for (unsigned t = tid; t < tiles; t += blockDim.x) {
        last_cost[t] = cost[t];
        cost[t] = 0u;
    }
}



// ------------------------------------------------------------------------------------------------------
// Wavefront variant (the default): one launch per path segment, live paths compacted in between.
//
// Measured on the monolithic kernel above (menger 1080p): the first segment (primary ray, first shading,
// first sun ray) costs 0.14 ms, but bounces 2..4 add 0.4 ms for 0.4 rays/px, because only 25 % / 6 % / 2 %
// of a tile's lanes are still alive while every wave on the object keeps running.  So:
//
//   primary_kernel   one thread per pixel (8x8 tile per wave): primary ray, G-buffer; a miss is
//                    finished on the spot (sky), a hit is appended to a path queue;
//   bounce_kernel<k> one thread per queued path: shade hit k (RNG, sun sample), cast the sun ray, cast
//                    bounce ray k+1; finished paths write their radiance, hits go to the next queue.
//
// Queues hold 64-byte PathRec records in kShards (= wave width) shards; a wave appends with ONE wave64
// ballot + popcount + atomicAdd on its shard's counter (64 counters on 64 cache lines: a single counter
// word would cap the chip at ~88 appends/us) and an mbcnt prefix for the lane slots.  A consumer wave
// reads the 64 shard counts with its 64 lanes, prefix-sums them with shuffles and maps chunk index ->
// (shard, offset) without any further atomics.  The per-pixel operation order — hence every bit of the
// result — is that of trace_kernel; only WHICH lane executes a path changes.
// ------------------------------------------------------------------------------------------------------
constexpr unsigned kShards = 64;
constexpr unsigned kCountStride = 16;  // uints: one 64-byte line per shard counter

struct PathRec {  // 4 x float4
    f3 hit_pos; int32_t node;
    f3 dir; uint32_t normal_ambient;   // normal: 2 bits per axis (0:+0, 1:+1, 2:-1, 3:-0); ambient_rays << 8
    f3 sample; uint32_t rng_index;
    f3 blend; uint32_t pix;
};

__device__ __forceinline__ uint32_t pack_axis(float v) { return v == 0.0f ? ((vx_f2u(v) >> 31) ? 3u : 0u) : (v > 0.0f ? 1u : 2u); }
__device__ __forceinline__ float unpack_axis(uint32_t c) { return c == 0u ? 0.0f : (c == 1u ? 1.0f : (c == 2u ? -1.0f : -0.0f)); }

__device__ __forceinline__ void store_rec(float4* q, const PathRec& r) {
    q[0] = make_float4(r.hit_pos.x, r.hit_pos.y, r.hit_pos.z, __int_as_float(r.node));
    q[1] = make_float4(r.dir.x, r.dir.y, r.dir.z, __uint_as_float(r.normal_ambient));
    q[2] = make_float4(r.sample.x, r.sample.y, r.sample.z, __uint_as_float(r.rng_index));
    q[3] = make_float4(r.blend.x, r.blend.y, r.blend.z, __uint_as_float(r.pix));
}
__device__ __forceinline__ PathRec load_rec(const float4* q) {
    const float4 a = q[0], b = q[1], c = q[2], d = q[3];
    PathRec r;
    r.hit_pos = mk3(a.x, a.y, a.z); r.node = __float_as_int(a.w);
    r.dir = mk3(b.x, b.y, b.z); r.normal_ambient = __float_as_uint(b.w);
    r.sample = mk3(c.x, c.y, c.z); r.rng_index = __float_as_uint(c.w);
    r.blend = mk3(d.x, d.y, d.z); r.pix = __float_as_uint(d.w);
    return r;
}

// Append the records of the lanes with `keep` to shard `shard` of the queue (called by all 64 lanes).
__device__ __forceinline__ void queue_append(const PathQueue& q, unsigned shard, bool keep, const PathRec& rec, int lane) {
    const unsigned long long m = __ballot(keep);
    if (m == 0ull) return;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(q.counts + shard * kCountStride, unsigned(__popcll(m)));
    base = __builtin_amdgcn_readfirstlane(base);
    if (keep) {
        const unsigned rank = __builtin_amdgcn_mbcnt_hi(unsigned(m >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(m), 0u));
        store_rec(q.recs + (size_t(shard) * q.shard_capacity + base + rank) * 4u, rec);
    }
}

__device__ __forceinline__ SceneView make_scene(const TraceArgs& a) {
    SceneView sc;
    sc.svo = a.svo;
    sc.leaves = a.leaves;
    sc.root_center = ld3(a.root_center);
    sc.root_size = a.root_size;
    sc.root_min = sc.root_center - splat3(0.5f * a.root_size);
    return sc;
}

__device__ __forceinline__ void zero_counts(unsigned* counts, int tid) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid < int(kShards)) counts[tid * kCountStride] = 0u;
}

__global__ __launch_bounds__(kBlock) void primary_kernel(const TraceArgs a, const PathQueue out, unsigned* zero) {
    extern __shared__ uint2 lds_stack[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    zero_counts(zero, tid);
    const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int lrow = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    const int lband = lrow / a.band.band_rows;
    const int y = (lband * a.band.nranks + a.band.rank) * a.band.band_rows + (lrow - lband * a.band.band_rows);
    const bool active = x < a.band.width && lrow < a.band.local_rows && y < a.band.height;

    bool keep = false;
    PathRec rec;
    rec.node = 0; rec.normal_ambient = 0; rec.rng_index = 0; rec.pix = 0;
    rec.hit_pos = rec.dir = rec.sample = rec.blend = splat3(0.0f);
    if (active) {
        const SceneView sc = make_scene(a);
        const uint32_t pix = uint32_t(lrow) * uint32_t(a.band.width) + uint32_t(x);
        const f3 o = ld3(a.cam.o);
        const f3 d = norm3((float(x) * ld3(a.cam.r) - float(y) * ld3(a.cam.u)) + ld3(a.cam.f));  // voxels.comp:299-303
        RayHit hit;
        if (cast_ray(sc, o, d, kAlmostInfinity, lds_stack + tid, hit)) {
            const f3 n = hit.normal;
            a.out_nd[pix] = make_float4(n.x, n.y, n.z, hit.time);                                 // voxels.comp:320-324,395
            const f3 alb = (hit.node & kEmitBit) == 0 ? node_color(hit.node) : splat3(1.0f);
            a.out_albedo[pix] = make_float4(alb.x, alb.y, alb.z, __int_as_float(hit.node));       // voxels.comp:392,396
            rec.hit_pos = o + d * hit.time;
            rec.node = hit.node;
            rec.dir = d;
            rec.normal_ambient = pack_axis(n.x) | pack_axis(n.y) << 2 | pack_axis(n.z) << 4 | 1u << 8;
            rec.sample = splat3(0.0f);
            rec.blend = splat3(1.0f);
            rec.rng_index = uint32_t(x) % 128u + (uint32_t(y) % 128u) * 128u + (a.frame_number % 512u) * kNoiseLayer;
            rec.pix = pix;
            keep = true;
        } else {  // sky on the primary ray                                               voxels.comp:373-382,391
            float sun_power = vx_pow(vx_max(0.0f, dot3(d, ld3(a.neg_sun_dir_n))), a.sun_exponent);
            f3 out = (splat3(0.0f) + (ld3(a.sky_color) + ld3(a.sun_color) * sun_power) * splat3(1.0f)) / 1.0f;
            a.out_color[pix] = make_float4(out.x, out.y, out.z, 1.0f);
            a.out_nd[pix] = make_float4(kAlmostInfinity, kAlmostInfinity, kAlmostInfinity, -1.0f);
            a.out_albedo[pix] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(0xffffff));
        }
    }
    const unsigned wg = (blockIdx.y * gridDim.x + blockIdx.x) * 4u + unsigned(wave);
    queue_append(out, wg % kShards, keep, rec, lane);
    count_rays(a.ray_counter, active ? 1u : 0u, lane);
}

#ifndef VXRT_BOUNCE_WAVES
#define VXRT_BOUNCE_WAVES 4
#endif
__global__ __launch_bounds__(kBlock, VXRT_BOUNCE_WAVES) void bounce_kernel(const TraceArgs a, const PathQueue in, const PathQueue out, unsigned* zero,
                                                        int first_bounce, int last_bounce) {
    extern __shared__ uint2 lds_stack[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    zero_counts(zero, tid);
    const SceneView sc = make_scene(a);
    const f3 sun_dir = ld3(a.sun_dir), sun_color = ld3(a.sun_color), sky = ld3(a.sky_color);
    uint2* stack = lds_stack + tid;

    // chunk table: lane q owns shard q
    const unsigned my_count = in.counts[lane * kCountStride];
    const unsigned my_chunks = (my_count + 63u) / 64u;
    unsigned incl = my_chunks;
    for (int off = 1; off < 64; off <<= 1) {
        unsigned v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    const unsigned total_chunks = __shfl(incl, 63, 64);
    const unsigned total_waves = gridDim.x * 4u;
    uint32_t rays = 0;

    for (unsigned c = blockIdx.x * 4u + unsigned(wave); c < total_chunks; c += total_waves) {
        const unsigned long long above = __ballot(incl > c);
        const int q = __ffsll((long long)above) - 1;                       // shard that holds chunk c
        const unsigned first = __shfl(incl - my_chunks, q, 64);           // chunks before shard q
        const unsigned count_q = __shfl(my_count, q, 64);
        const unsigned entry = (c - first) * 64u + unsigned(lane);
        const bool valid = entry < count_q;

        bool keep = false;
        PathRec rec;
        rec.node = 0; rec.normal_ambient = 0; rec.rng_index = 0; rec.pix = 0;
        rec.hit_pos = rec.dir = rec.sample = rec.blend = splat3(0.0f);
        if (valid) {
            rec = load_rec(in.recs + (size_t(q) * in.shard_capacity + entry) * 4u);
            Rng rng;
            rng.noise = a.noise;
            rng.index = rec.rng_index;
            // Path segments first_bounce .. last_bounce run in this launch (lanes whose path ends simply idle);
            // a path that is still alive after segment last_bounce goes to the next queue.
            for (int bounce = first_bounce;; bounce++) {
                const f3 n = mk3(unpack_axis(rec.normal_ambient & 3u), unpack_axis((rec.normal_ambient >> 2) & 3u), unpack_axis((rec.normal_ambient >> 4) & 3u));
                uint32_t ambient_rays = rec.normal_ambient >> 8;
                f3 sample = rec.sample, blend = rec.blend;
                const f3 color = bounce == 0 ? splat3(1.0f) : node_color(rec.node);          // voxels.comp:317
                const f3 emit = node_emittance(rec.node, a.emit_strength);
                const f3 o = rec.hit_pos + 1e-5f * n;                                       // voxels.comp:333,353,370
                f3 d;
                if (rng.next() < a.specularity) {  // specular                              voxels.comp:326-334
                    d = norm3(reflect3(rec.dir, n));
                    sample = sample + emit * blend;
                    blend = blend * ((2.0f * color) * dot3(d, n));
                } else if (a.sun_strength > 0.0f) {  // diffuse + sun sample                  voxels.comp:339-371
                    float r0 = rng.next(), r1 = rng.next(), r2 = rng.next();
                    f3 up_dir = norm3(cross3(mk3(r0, r1, r2), sun_dir));
                    f3 right_dir = norm3(cross3(sun_dir, up_dir));
                    float dx = 2.0f * rng.next() - 1.0f;
                    float dy = 2.0f * rng.next() - 1.0f;
                    f3 light_dir = ld3(a.sun_dir_n) + (dx * right_dir + dy * up_dir) * a.sun_size;
                    f3 to_light = norm3(-light_dir);
                    ambient_rays++;
                    RayHit sun_hit;
                    rays++;
                    if (!cast_ray(sc, o, to_light, kAlmostInfinity, stack, sun_hit))
                        sample = sample + ((sun_color * color) * blend) * vx_max(0.0f, dot3(n, to_light));
                    d = random_hemisphere(n, rng);
                    sample = sample + emit * blend;
                    blend = blend * (color * dot3(n, d));
                } else {  // diffuse, sun switched off
                    d = random_hemisphere(n, rng);
                    sample = sample + emit * blend;
                    blend = blend * (color * dot3(n, d));
                }

                bool finished = true;
                if (bounce + 1 < a.max_bounces) {  // next path segment                        voxels.comp:309-313
                    RayHit hit;
                    rays++;
                    if (cast_ray(sc, o, d, kAlmostInfinity, stack, hit)) {
                        const f3 hn = hit.normal;
                        rec.hit_pos = o + d * hit.time;
                        rec.node = hit.node;
                        rec.dir = d;
                        rec.normal_ambient = pack_axis(hn.x) | pack_axis(hn.y) << 2 | pack_axis(hn.z) << 4 | ambient_rays << 8;
                        rec.sample = sample;
                        rec.blend = blend;
                        finished = false;
                    } else {
                        sample = sample + sky * blend;                                        // voxels.comp:384
                    }
                }
                if (finished) {
                    f3 outc = sample / float(ambient_rays);                                   // voxels.comp:391
                    a.out_color[rec.pix] = make_float4(outc.x, outc.y, outc.z, 1.0f);
                    break;
                }
                if (bounce == last_bounce) {
                    rec.rng_index = rng.index;
                    keep = true;
                    break;
                }
            }
        }
        queue_append(out, c % kShards, keep, rec, lane);
    }
    count_rays(a.ray_counter, rays, lane);
}

}  // namespace

hipError_t launch_trace(const TraceArgs& a, hipStream_t s) {
    dim3 grid(unsigned((a.band.width + 15) / 16) * unsigned((a.band.local_rows + 15) / 16));
    size_t lds = size_t(a.stack_levels) * kBlock * sizeof(uint2);
    hipLaunchKernelGGL(trace_kernel, grid, dim3(kBlock), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_tile_order(uint32_t* cost, uint32_t* order, uint32_t* last_cost, unsigned tiles, hipStream_t s) {
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, s, cost, order, last_cost, tiles);
    return hipGetLastError();
}

hipError_t launch_trace_wavefront(const TraceArgs& a, const PathQueue queues[2], unsigned* count_sets[3], unsigned* launch_counter,
                                  int blocks, unsigned split_mask, hipStream_t s) {
    dim3 grid((a.band.width + 15) / 16, (a.band.local_rows + 15) / 16);
    size_t lds = size_t(a.stack_levels) * kBlock * sizeof(uint2);
    // launch J reads count set J%3, writes (J+1)%3 and clears (J+2)%3 (the set launch J-1 consumed)
    unsigned J = *launch_counter;
    PathQueue out = queues[0];
    out.counts = count_sets[(J + 1) % 3];
    hipLaunchKernelGGL(primary_kernel, grid, dim3(kBlock), lds, s, a, out, count_sets[(J + 2) % 3]);
    J++;
    // bit k of split_mask set: a new launch (with compaction of the live paths) starts at path segment k
    int stage = 0;
    for (int first = 0; first < a.max_bounces;) {
        int last = first;
        while (last + 1 < a.max_bounces && !((split_mask >> (last + 1)) & 1u)) last++;
        PathQueue in = queues[stage & 1];
        in.counts = count_sets[J % 3];
        out = queues[(stage & 1) ^ 1];
        out.counts = count_sets[(J + 1) % 3];
        hipLaunchKernelGGL(bounce_kernel, dim3(blocks), dim3(kBlock), lds, s, a, in, out, count_sets[(J + 2) % 3], first, last);
        J++;
        stage++;
        first = last + 1;
    }
    *launch_counter = J;
    return hipGetLastError();
}

}  // namespace vxrt
