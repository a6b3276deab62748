// trace_block.h — the body of trace_kernel (trace.hip): one block = one 8 x 8 pixel tile of one frame (or 8 / kF rows of it in kF frames),
// from the primary ray to the hand-over of the paths still alive at their second hit.  A header of its own since round 6, so that the
// variants build's fused_kernel (trace_fused.hip) — which runs the same block as one item of its persistent waves — lives in a file of
// its own instead of inside the product's.  Everything is in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#ifndef VXRT_TRACE_BLOCK
#define VXRT_TRACE_BLOCK 64   // threads per block of trace_kernel: 64 = one wave, an 8x8 pixel tile (128: 16x8, 256: 16x16).
// One wave per block: a wave's slot is free for the next tile the moment it ends, instead of idling until the slowest of a
// block's four waves has finished, and the longest-tile-first schedule works at 8x8 granularity (measured: 256 -> 128 -> 64
// threads: 23.7 -> 24.5 -> 25.0 Gray/s).
#endif
#define VXRT_STACK_STRIDE VXRT_TRACE_BLOCK
#include "kernels.h"
#include "vx_vec.h"

#include "trace_common.h"
#include "trace_tail_body.h"


namespace vxrt {
namespace {


// What is kept while a hit's sun ray is out.  0: the shader's order — everything about the hit is computed before the sun ray
// (pend_sun, pend_emit, next_dir: 9 registers alive during the cast).  1: only the packed normal and the leaf word are kept and the
// rest of the shading runs after the cast (87 instead of 93 VGPRs) — measured 5 % SLOWER at 5 waves per SIMD: the colour table
// loads and the hemisphere's noise loads then sit in a second wait between two casts.  2: as 1 with the colour carried.
#ifndef VXRT_DEFER_SHADING
#define VXRT_DEFER_SHADING 0
#endif
#ifndef VXRT_TRACE_WAVES
#define VXRT_TRACE_WAVES 5   // waves per SIMD the register allocation aims for (96 VGPRs)
#endif
#ifndef VXRT_TRACE_WAVES_HBM
#define VXRT_TRACE_WAVES_HBM 6   // the same for a scene beyond the Infinity Cache, whose walk waits on HBM (80 VGPRs, with spills): see launch_trace
#endif
// Sky cull.  True only when the primary ray (o, d) PROVABLY makes cast_bounded_ray return false, decided without walking:
//  * the ray is regular (every component of 1 / d finite and non-zero) and, by a slab test in plain binary32, misses the box
//    TraceArgs::cull_min/max — the smallest box of cells of tree level L = min(depth, 7) that holds every voxel, grown by a margin
//    m >= 32 x the largest rounding error of the walk's plane times (api_trace.hip).  Comparisons with NaN are false: no cull.
//  * then the walk can visit (descend into) no node of level >= L — every such node that exists lies inside the box, and the walk
//    only enters cells the ray passes within rounding distance of — hence no leaf: it cannot return a hit from a leaf;
//  * and it cannot return the iteration cap's "hit" (voxels.comp:166-169) either: each trip of the loop handles one (node, octant)
//    pair, a node's octants are left along each axis at most once (a sibling step needs (directional & transition) == 0), so a
//    node costs at most 4 trips; the nodes visited at level l form a path that is monotone along each axis of a 2^l grid, at most
//    3 * 2^l - 2 cells; summed over the levels 0 .. L - 1 <= 6 that can be visited: at most 367 nodes, 1468 trips < 2048.
// So the walk ends in a miss (or the root test fails first), and voxels.comp:373-388 / :292-294 with bounce == 0 give the pixel's
// outputs from d alone.  tests/test_gpu_trace.py::test_sky_cull* compare culled and walked frames value for value.
__device__ __forceinline__ bool primary_miss_is_certain(const TraceArgs& a, f3 o, f3 d) {
    if (!a.cull) return false;
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    if (!ray_is_regular(inv)) return false;
    const f3 t0 = (ld3(a.cull_min) - o) * inv, t1 = (ld3(a.cull_max) - o) * inv;
    const float t_in = vx_max3(__builtin_fminf(t0.x, t1.x), __builtin_fminf(t0.y, t1.y), __builtin_fminf(t0.z, t1.z));
    const float t_out = vx_min3(__builtin_fmaxf(t0.x, t1.x), __builtin_fmaxf(t0.y, t1.y), __builtin_fmaxf(t0.z, t1.z));
    return t_in > t_out || t_out < 0.0f;
}

// voxels.comp:373-388 for a primary ray that misses (bounce == 0), :292-294 and :391-396: the three outputs of such a pixel
__device__ __forceinline__ void store_primary_miss(const TraceArgs& a, const FrameOut& fo, size_t pix, f3 d, bool gbuf) {
    const f3 sun_color = ld3(a.sun_color), sky = ld3(a.sky_color);
    const float sun_power = sun_power_of(a, d);
    const f3 sample = splat3(0.0f) + (sky + sun_color * sun_power) * splat3(1.0f);
    const f3 out = sample / float(1u);
    if (gbuf) {
        store_out(fo.nd + pix, make_float4(kAlmostInfinity, kAlmostInfinity, kAlmostInfinity, -1.0f));
        store_out(fo.albedo + pix, make_float4(1.0f, 1.0f, 1.0f, __int_as_float(0xffffff)));
    }
    store_out(fo.color + pix, make_float4(out.x, out.y, out.z, 1.0f));
}

constexpr int kTB = VXRT_TRACE_BLOCK;
constexpr uint32_t kLightCost = 1u;   // cost-map entry of a tile none of whose pixels walked (any walking wave records its duration, >= 4)
constexpr int kTileW = kTB == 64 ? 8 : 16, kTileH = kTB == 256 ? 16 : 8;

// kF: which 64 (pixel, frame) pairs share a wave.  1: an 8 x 8 pixel tile of ONE frame of the launch.  8 or 4 (the frames of the launch
// share one camera and come in whole groups of kF): 8 / kF ROWS of a tile in kF consecutive frames, lane = (frame, row, column) —
// a pixel's primary ray is the same in every frame and its first sun rays nearly so, so the lanes of a wave leave the walk's lock-step
// rounds closer together (priced on the oracle's step counts, tests/sim_schedule.py: lane_mappings: - 12 % wave-instructions in this
// kernel for kF = 8, - 10 % for 4; measured + 7 % on the bench view); each frame's stores stay whole 128-byte row segments (4 x 2
// pixels x 8 frames, 64-byte segments, is priced 3 % better and measured 6 % worse).  The per-pixel operations are the same either way.
// kFused: the block is one item of fused_kernel's head phase — `bid` comes from its work cursor, and a path that is handed over is
// stored for a consumer that may be polling the record already (queue_store_fused).
template <bool kWide, int kF, bool kFused>
__device__ __forceinline__ void trace_block(const TraceArgs& a, const unsigned bid, uint4* lds_stack, const uint32_t stamp) {
    static_assert(kF == 1 || ((kF == 4 || kF == 8) && kTB == 64), "frame lanes: one wave per block");
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    // One kTileW x kTileH pixel tile per block, an 8x8 sub-tile per wave.  Blocks take tiles in the order of tile_order
    // (longest tile of the previous frames first): a frame's cost is concentrated in the tiles that see
    // geometry, and started last they would leave the chip idling behind a few long waves.
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    const unsigned tiles_x = unsigned(a.band.width + kTileW - 1) / unsigned(kTileW);
    // a launch covers `batch` consecutive frames (same camera, frame numbers frame_number ..): the `batch` blocks of one tile
    // position are neighbours in launch order, so longest-first holds across the whole batch.  block = (tile, part of the tile,
    // group of kF frames): kF parts of 8 / kF rows, batch / kF groups
    const unsigned batch = unsigned(a.batch);
    constexpr unsigned kRows = 8u / unsigned(kF);
    const unsigned groups = batch / unsigned(kF);
    const unsigned fb = (bid % groups) * unsigned(kF) + unsigned(lane >> 3) / kRows;
    const unsigned row_in_tile = ((bid / groups) % unsigned(kF)) * kRows + unsigned(lane >> 3) % kRows;
    const unsigned ord = bid / batch;
    const bool gbuf = ((a.gbuf_frames >> fb) & 1u) != 0u;   // this frame's normal/depth and albedo/node images are wanted
    const unsigned tile = a.tile_order ? a.tile_order[ord] : ord;
    const int x = int(tile % tiles_x) * kTileW + (wave & 1) * 8 + (lane & 7);
    const int lrow = int(tile / tiles_x) * kTileH + (wave >> 1) * 8 + int(row_in_tile);
    const int y = frame_row(a.band, lrow);
    const bool active = x < a.band.width && lrow < a.band.local_rows && y < a.band.height && fb < batch;
    const unsigned cam_index = fb;   // per lane when kF > 1: the frames of a wave may have cameras of their own (vxrt_render_path)

    uint32_t rays = 0;
    // the shard of the tail queue this wave appends to: the top bits of a multiplicative hash of the wave's index, because the waves
    // that append (tiles that see geometry) can sit at regular distances in the launch order (tile_scatter_kernel)
    static_assert(kShards == 64, "6 hash bits");
#ifndef VXRT_TAIL_SHARD
#define VXRT_TAIL_SHARD 0   // 1: by tile (all frames of a tile to one shard: chunks of neighbours in space); 2: by tile, hashed.  A/B only (round 5)
#endif
#if VXRT_TAIL_SHARD == 1
    const unsigned tail_shard = ord % kShards;
#elif VXRT_TAIL_SHARD == 2
    const unsigned tail_shard = (ord * 0x9E3779B1u) >> 26;
#else
    const unsigned tail_shard = ((bid * unsigned(kTB / 64) + unsigned(wave)) * 0x9E3779B1u) >> 26;
#endif
    if (!kFused && a.tail.recs != nullptr) zero_counts(a.tail_zero, tid);
    bool walk = active;
    if (active) {   // the sky cull: a pixel whose primary ray certainly misses needs no walk
        const Cam& cam = a.cams[cam_index];
        const f3 o = ld3(cam.o);
        const f3 d = norm3((float(x) * ld3(cam.r) - float(y) * ld3(cam.u)) + ld3(cam.f));  // voxels.comp:299-303
        if (primary_miss_is_certain(a, o, d)) {
            store_primary_miss(a, a.out[fb], size_t(lrow) * a.band.width + x, d, gbuf);
            rays = 1;
            walk = false;
        }
    }
    const bool light_wave = __ballot(walk) == 0ull;   // nobody walks: sky (or beyond the frame's edge) — see tile_scatter_kernel
    if (walk) {
        bool handed_over = false;  // this lane's path continues in bounce_kernel (TraceArgs::tail)
        const Caster<kWide> caster(a, lds_stack, tid);
        const size_t pix = size_t(lrow) * a.band.width + x;

        Rng rng;
        rng.noise = a.noise;
        rng.index = uint32_t(x) % 128u + (uint32_t(y) % 128u) * 128u + ((a.frame_number + fb) % 512u) * kNoiseLayer;

        const f3 sun_dir = ld3(a.sun_dir), sun_color = ld3(a.sun_color), sky = ld3(a.sky_color);

        const Cam& cam = a.cams[cam_index];
        f3 o = ld3(cam.o);
        f3 d = norm3((float(x) * ld3(cam.r) - float(y) * ld3(cam.u)) + ld3(cam.f));  // voxels.comp:299-303

        f3 sample = splat3(0.0f), blend = splat3(1.0f);
        uint32_t ambient_rays = 1;
        int bounce = 0;
        bool sun_phase = false;
#if VXRT_DEFER_SHADING
        uint32_t held = 0;        // while a sun ray is out: the hit's normal, 2 bits per axis (pack_axis)
        int32_t held_node = 0;    // ... and its leaf word
#endif
#if VXRT_DEFER_SHADING == 2
        f3 held_color = splat3(1.0f);
#endif
#if !VXRT_DEFER_SHADING
        f3 pend_sun = splat3(0.0f), pend_emit = splat3(0.0f), next_dir = splat3(0.0f);   // what the deferral replaces
#endif

        for (;;) {
            RayHit hit;
            rays++;
            const bool is_hit = caster.cast(o, d, hit);

#if !VXRT_DEFER_SHADING
            if (sun_phase) {  // back from the sun shadow ray                 voxels.comp:357-371
                if (!is_hit) sample = sample + pend_sun;
                sample = sample + pend_emit;
                d = next_dir;
                sun_phase = false;
                if (++bounce >= a.max_bounces) break;
                continue;
            }
#else
            if (sun_phase) {  // back from the sun shadow ray: the rest of the hit's shading       voxels.comp:357-371
                // (what the shader computes before the cast is computed here, after it, from the packed normal and the leaf word:
                // the same operations on the same operands, but seven registers fewer are alive while the sun ray walks)
                const f3 n = mk3(unpack_axis(held & 3u), unpack_axis((held >> 2) & 3u), unpack_axis((held >> 4) & 3u));
#if VXRT_DEFER_SHADING == 2
                const f3 color = held_color;   // carried: no table loads on the way from the sun ray to the bounce ray
#else
                const f3 color = bounce == 0 ? splat3(1.0f) : node_color(held_node);
#endif
                const f3 emit = node_emittance(held_node, a.emit_strength);
                if (!is_hit) sample = sample + ((sun_color * color) * blend) * vx_max(0.0f, dot3(n, d));   // d is still the sun ray
                const f3 refl = random_hemisphere(n, rng);
                sample = sample + emit * blend;
                blend = blend * (color * dot3(n, refl));
                d = refl;
                sun_phase = false;
                if (++bounce >= a.max_bounces) break;
                continue;
            }
#endif

            if (!is_hit) {  // sky                                              voxels.comp:373-388
                if (bounce == 0) {
                    blend = splat3(1.0f);
                    float sun_power = sun_power_of(a, d);
                    sample = sample + (sky + sun_color * sun_power) * blend;
                    if (gbuf) {
                        store_out(a.out[fb].nd + pix, make_float4(kAlmostInfinity, kAlmostInfinity, kAlmostInfinity, -1.0f));
                        store_out(a.out[fb].albedo + pix, make_float4(1.0f, 1.0f, 1.0f, __int_as_float(0xffffff)));
                    }
                } else {
                    sample = sample + sky * blend;
                }
                break;
            }

            const f3 n = hit.normal;
            const f3 hit_pos = o + d * hit.time;
            const f3 color = bounce == 0 ? splat3(1.0f) : node_color(hit.node);
            const f3 emit = node_emittance(hit.node, a.emit_strength);
            if (bounce == 0 && gbuf) {  // first-hit G-buffer                   voxels.comp:320-324,392-396
                store_out(a.out[fb].nd + pix, make_float4(n.x, n.y, n.z, hit.time));
                f3 alb = (hit.node & kEmitBit) == 0 ? node_color(hit.node) : splat3(1.0f);
                store_out(a.out[fb].albedo + pix, make_float4(alb.x, alb.y, alb.z, __int_as_float(hit.node)));
            }
            // Hand the path over, in the state bounce_kernel resumes from — unless the queue is full (it is sized from what earlier
            // launches queued, not for the worst case): then this lane goes on as in the all-in-one kernel.  The record is stored
            // right here, so that no register holds it while the wave's other lanes go on looping.
            if (a.tail.recs != nullptr && bounce == a.tail_from) {
                const uint32_t slot = queue_reserve(a.tail, tail_shard);
                if (slot != kNoSlot) {
                    PathRec rec;
                    rec.hit_pos = hit_pos;
                    rec.node = hit.node;
                    rec.dir = d;
                    rec.normal_ambient = pack_axis(n.x) | pack_axis(n.y) << 2 | pack_axis(n.z) << 4 | ambient_rays << 8;
                    rec.sample = sample;
                    rec.blend = blend;
                    rec.rng_index = rng.index;
                    rec.pix = uint32_t(pix) | fb << kPixBits;
                    if (kFused) queue_store_fused(a.tail, tail_shard, slot, rec, stamp);
                    else queue_store(a.tail, tail_shard, slot, rec);
                    handed_over = true;
                    break;
                }
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
#if VXRT_DEFER_SHADING
                held = pack_axis(n.x) | pack_axis(n.y) << 2 | pack_axis(n.z) << 4;
                held_node = hit.node;
#if VXRT_DEFER_SHADING == 2
                held_color = color;
#endif
#else
                pend_sun = ((sun_color * color) * blend) * vx_max(0.0f, dot3(n, to_light));
                f3 refl = random_hemisphere(n, rng);
                pend_emit = emit * blend;
                blend = blend * (color * dot3(n, refl));
                next_dir = refl;
#endif
                o = hit_pos + 1e-5f * n;
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

        if (!handed_over) {
            f3 out = sample / float(ambient_rays);  // voxels.comp:391
            store_out(a.out[fb].color + pix, make_float4(out.x, out.y, out.z, 1.0f));
        }
    }

    count_rays(a.ray_counter, rays, lane);
    if (a.tile_cost && lane == 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_start;
        atomicMax(a.tile_cost + tile, light_wave ? kLightCost : (dt > 0xffffffffull ? 0xffffffffu : (dt < 4ull ? 4u : uint32_t(dt))));
    }
}

}  // namespace
}  // namespace vxrt
