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
//  * a wave covers an 8x8 pixel tile (coherent primary rays, 128-byte output segments) and is a block of its own;
//  * the kernel follows a path up to its second hit (TraceArgs::tail_from); the paths still alive there are queued for
//    bounce_kernel (trace_tail.hip); a launch covers up to 32 consecutive frames (TraceArgs::batch).
//
// Arithmetic follows include/vxrt_detmath.h: every float operation that can change a result is the
// shader's operation, in the shader's order, never contracted.
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

template <bool kWide, int kWaves, int kF>
__global__ __launch_bounds__(kTB, kWaves) void trace_kernel(const TraceArgs a) {
    extern __shared__ uint4 lds_stack[];  // the threads' frames: Caster<kWide>
    // a launch may come as two grids (the longest tiles apart: VXRT_OPT_LONG_TILES)
    trace_block<kWide, kF, false>(a, blockIdx.x + a.block_first, lds_stack, 0u);
}


#if VXRT_VARIANTS
// ---- fused_kernel: head and compacted tail of a launch in ONE grid of persistent waves (VXRT_OPT_FUSED_TAIL; -DVXRT_VARIANTS=1 only) --------
// MEASURED SLOWER (round 5; HISTORY.md section 11): bit-identical images, 0.50 ms against 0.41 (two kernels) and 0.375 (all-in-one) for a
// rank of 8's 20-frame block.  A software scheduler pays for every decision with round trips through device-scope memory (2-5 us each:
// a claim, a look at the shard counters, a flag) where the hardware's dispatcher starts the next wave for nothing, and it holds both
// kernels' bodies (4 waves per SIMD) in waves that sleep in their slots when they have nothing to do.  Kept beside tracers 2, 3 and 5
// as the record of the attempt, with its parity cases.
// A launch that is little more than its longest chains — a rank's share of a short block on many GPUs: 20 frames of an eighth of the
// rows — spends its time DRAINING: trace_kernel ends when its longest wave ends (the chip two thirds idle by then), and only then may
// bounce_kernel start, which drains again.  Two chains end to end, at 83 % and 64 % of the instruction rate the same kernels reach in
// the steady state (round 5: profiles/r05/short_block_timelines.txt).  Here the waves are persistent: each takes the launch's blocks
// (tile x frame group, in the launch order: longest first) from a cursor, and when those run out it takes CHUNKS of 64 queued paths,
// as soon as a chunk is complete — the tail of the paths handed over early runs beside the long head chains, and nothing waits for a
// kernel boundary.  Same records, same per-path operations (trace_block, bounce_path): same image.
//
// Hot words.  One memory channel takes ~90 atomics (or agent-scope loads) per microsecond — the ray counters taught that in round 1, and
// the first version of this kernel, with one cursor block of 1 KB and every idle wave polling the same three words, spent 76 % of its
// wave-cycles in s_waitcnt (profiles/r05/fused_kernel_counters.txt).  So: kCursors head cursors 256 bytes apart, cursor k over the
// blocks k, k + kCursors, ... (the launch order survives in each); a wave starts at its own and goes round when that is dry; ONE block
// per claim among the tiles that walk (the front of the order), four among the tiles of sky.  A wave that leaves the head phase adds
// its blocks to one counter (one atomic per wave); the wave that completes the total raises kFlags copies of a flag, and an idle
// wave looks only at its own copy, at the four shard counters of its own group and at their four chunk cursors, every ~14 us.
//
// Hand-over.  A head lane reserves its record slot with the shard's counter as before and stores the record with queue_store_fused:
// seven 8-byte stores at agent scope (write-through, visible to every XCD), s_waitcnt, then the eighth, which holds the launch's
// 16-bit stamp in the unused top of normal_ambient.  A consumer lane polls that word (agent scope) until the stamp is there, reads
// the rest and writes the word back with stamp 0.  Before the heads are done only chunks whose 64 slots are all reserved are taken;
// afterwards every shard counter is final and the part-filled last chunks go too.
//
// Waiting is bounded in three ways: a wave only waits for work that resident, running waves are producing (a block is claimed by a
// wave that is already running, so no wave ever waits for one that has no slot); every spin sleeps; a spin that lasts longer than
// any legitimate wait (tens of milliseconds) sets ctl->error and gives up — the frame is then wrong and vxrt_sync reports it, but
// the grid drains.
constexpr unsigned kCursors = 64, kFlags = 64, kHotStride = 64;   // uints: 256 bytes between hot words
struct FusedCtl {
    unsigned cursor[kCursors * kHotStride];        // head work cursors (local block index of cursor k)
    unsigned done_flag[kFlags * kHotStride];       // copies of "every head block is finished"
    unsigned heavy_flag[kFlags * kHotStride];      // copies of "every block of a tile that walked last time is finished": part-filled chunks may be closed
    unsigned blocks_done[kHotStride];              // head blocks finished (their records stored and stamped), added wave by wave
    unsigned heavy_done[kHotStride];               // ... of those, blocks before heavy_blocks
    unsigned error[kHotStride];                    // a bounded wait ran out
    unsigned next_chunk[kShards * kCountStride];   // per shard: the next chunk nobody has claimed yet
    // diagnostics (vxrt_debug_fused_profile): shader clock ~earliest wave start (stored inverted), the clock when the done flags went
    // up, the latest wave end; chunks taken before / after the flags; idle sleeps; stamp polls; head claims
    unsigned long long prof[8];
};

__device__ __forceinline__ unsigned agent_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ... and one that has RETURNED before anything after it is issued: two loads sent off back to back may be sampled in either order, and
// "the flag is up, so the counter I read is final" needs the flag sampled first
__device__ __forceinline__ unsigned agent_load_first(const unsigned* p) {
    const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return v;
}

#ifndef VXRT_FUSED_WAVES
#define VXRT_FUSED_WAVES 4   // waves per SIMD of fused_kernel: 128 VGPRs — at 5 (96) the persistent state spills into the walk's loops
#endif
template <int kF>
__global__ __launch_bounds__(kTB, VXRT_FUSED_WAVES) void fused_kernel(const TraceArgs a, FusedCtl* ctl, const unsigned total_blocks, const uint32_t* sort_info,
                                                                      const uint32_t stamp, const int first_bounce) {
    static_assert(kTB == 64 && kCursors == 64, "one wave per block; lane c reads cursor c");
    extern __shared__ uint4 lds_stack[];
    const int lane = threadIdx.x & 63;
    zero_counts(a.tail_zero, threadIdx.x);
    const bool prof_wave = lane == 0 && blockIdx.x % 64u == 0u;    // every 64th wave reports (the counter line takes ~90 atomics per us)
    if (prof_wave) atomicMax(&ctl->prof[0], ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
    unsigned prof_claims = 0, prof_idle = 0, prof_polls = 0, prof_before = 0, prof_after = 0;
    // where the tiles that walk end in the launch order (tile_scan_kernel left the count and the spread beside the sort's histogram;
    // spread_position puts them at the front, each followed by k tiles of sky): unknown (the stream's first launch) -> 0
    unsigned heavy_blocks = 0;
    if (sort_info != nullptr && a.tile_order != nullptr) {
        const unsigned tiles = total_blocks / unsigned(a.batch), nh = sort_info[0], nl = tiles - (nh < tiles ? nh : tiles);
        const unsigned k = nh ? unsigned((unsigned long long)nl * sort_info[1] / 256u / nh) & ~1u : 0u;
        const unsigned long long hb = (unsigned long long)(nh < tiles ? nh : tiles) * (k + 1u) * unsigned(a.batch);
        heavy_blocks = hb > total_blocks ? total_blocks : unsigned(hb);
    }
    // ---- one loop, three kinds of work, in this order of preference:
    //   1. a block of a tile that WALKS (the front of the launch order): the launch's critical path — claimed one at a time, run at
    //      the highest priority;
    //   2. a chunk of 64 queued paths that is complete (its tail is the second half of the critical path);
    //   3. blocks of SKY, four per claim: filler, 5 us each, which nothing waits for.
    // (The first version ran all of 1 and 3 before any of 2: the blocks of sky at the end of the order kept every wave in the head
    // phase until it was over, and the "fused" launch was two phases again: 324 us to the last head block, 230 us of tail behind it.)
    const Caster<false> caster(a, lds_stack, int(threadIdx.x));
    const f3 sun_dir = ld3(a.sun_dir), sun_color = ld3(a.sun_color), sky = ld3(a.sky_color);
    const PathQueue none{nullptr, nullptr, 0u};
    uint32_t rays = 0;
    const unsigned group = blockIdx.x % 16u;       // before part-filled chunks may be taken this wave serves the shards group, group + 16, + 32, + 48
    const unsigned home = blockIdx.x % kShards;
    const unsigned* my_flag = &ctl->done_flag[(blockIdx.x % kFlags) * kHotStride];
    const unsigned* my_heavy_flag = &ctl->heavy_flag[(blockIdx.x % kFlags) * kHotStride];
    unsigned my_blocks = 0, my_heavy = 0, idle = 0;
    unsigned k = blockIdx.x % kCursors, last = 0;  // the head cursor this wave claims from, and where it stood at the wave's last claim
    bool head_dry = total_blocks == 0u;
    auto report_heavy = [&]() {   // this wave's finished blocks of walking tiles -> the counter; the wave that completes it raises the flags
        if (my_heavy == 0u) return;
        __builtin_amdgcn_s_waitcnt(0);             // their records are written through
        if (lane == 0) {
            const unsigned before = atomicAdd(&ctl->heavy_done[0], my_heavy);
            if (before + my_heavy >= heavy_blocks)
                for (unsigned f = 0; f < kFlags; f++) __hip_atomic_store(&ctl->heavy_flag[f * kHotStride], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        my_heavy = 0;
    };
    auto report_blocks = [&]() {  // ... and all its finished blocks, once, when the cursors are dry
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0 && my_blocks != 0u) {
            const unsigned before = atomicAdd(&ctl->blocks_done[0], my_blocks);
            if (before + my_blocks >= total_blocks) {  // the last head wave: every record is stored, every shard counter final
                for (unsigned f = 0; f < kFlags; f++) __hip_atomic_store(&ctl->done_flag[f * kHotStride], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ctl->prof[1] = __builtin_amdgcn_s_memrealtime();
            }
        }
        my_blocks = 0;
    };
    // claims `step` blocks of cursor k; false: k is dry — then k moves to a cursor that is not, or head_dry is set
    unsigned first_j = 0, count_j = 0;
    // what was asked for before the last blocks of sky ran: a claim of four more (lane 0 holds the answer), and a look at the chunks
    bool sky_pending = false, pf = false, pf_mine = false, closing_seen = false;
    // a chunk that was claimed part-filled while a claim of sky is in hand: waiting for it may mean waiting for the done flag, i.e. for
    // this wave's own unrun blocks — so it is set aside until they have run
    bool held = false;
    unsigned held_shard = 0, held_chunk = 0, held_ns = 0;
    unsigned pf_sky_i = 0, pf_k = 0, pf_done = 0, pf_hv = 0, pf_n = 0, pf_next = 0;
    auto claim_head = [&](unsigned step) -> bool {
        for (;;) {
            const unsigned per_cursor = total_blocks > k ? (total_blocks - k + kCursors - 1u) / kCursors : 0u;   // blocks k, k + kCursors, ... < total_blocks
            unsigned i = 0;
            if (lane == 0) i = atomicAdd(&ctl->cursor[k * kHotStride], step);
            i = __builtin_amdgcn_readfirstlane(i);
            prof_claims++;
            if (i < per_cursor) { first_j = i; count_j = i + step <= per_cursor ? step : per_cursor - i; last = i + step; return true; }
            // this cursor is dry for good.  ONE look at all of them (lane c reads cursor c) instead of 63 more failed claims — which is
            // what every wave did at first, 330 000 atomics per launch on 64 lines — then on to the first after this one that is not
            const unsigned c = unsigned(lane) % kCursors;
            const unsigned per_c = total_blocks > c ? (total_blocks - c + kCursors - 1u) / kCursors : 0u;
            unsigned long long m = __ballot(agent_load(&ctl->cursor[c * kHotStride]) < per_c);
            if (m == 0ull) { head_dry = true; return false; }                                // every block is taken
            const unsigned r = (k + 1u) % kCursors;
            m = r == 0u ? m : (m >> r | m << (64u - r));
            k = (r + unsigned(__ffsll((long long)m) - 1)) % kCursors;
            last = 0xffffffffu / kCursors;                                                   // its front is gone by now
            step = 4u;
        }
    };
    for (;;) {
        // ---- what next?
        bool do_head = false, do_chunk = false;
        unsigned shard = 0, chunk = 0, n_s = 0;
        bool heads_done = false;
        if (head_dry && my_blocks != 0u && !sky_pending) { report_heavy(); report_blocks(); }   // the cursors are dry: this wave's blocks count now
        if (!head_dry && !sky_pending && last * kCursors + k < heavy_blocks) do_head = claim_head(1u);   // 1. a walking tile
        // 2. a chunk — from the look that was sent off BEFORE the last blocks of sky ran (pf), or from a look made now.  (With a claim
        // of sky in hand and no look, the blocks go first: they send the next look off.)
        if (!do_head && held && !sky_pending) { do_chunk = true; shard = held_shard; chunk = held_chunk; n_s = held_ns; held = false; pf = false; }
        if (!do_head && !do_chunk && !held && (pf || !sky_pending)) {
            const unsigned q = unsigned(lane);      // lane q looks at shard q — before `closing` only the four lanes of this wave's group do
            // The look that travelled while the blocks ran is good for ONE thing: "a whole chunk is reserved" (a counter only grows,
            // and the claim below is checked against it).  Its flags and its counters were sampled in no particular order, so
            // whatever depends on the flags — taking a part-filled chunk, deciding that nothing is left — is decided by a look made
            // now, flag first.
            for (bool fresh = !pf;; fresh = true) {
                unsigned n = 0, next = 0;
                bool mine, closing = false;
                if (!fresh) {
                    mine = pf_mine; n = pf_n; next = pf_next;
                    pf = false;
                } else {
                    heads_done = agent_load_first(my_flag) != 0u;                            // sampled BEFORE the counters: then they are final
                    // once the tiles that walked last time are through, a part-filled chunk is worth taking: whoever takes it closes it
                    closing = closing_seen = heads_done || (heavy_blocks != 0u && agent_load(my_heavy_flag) != 0u);
                    mine = closing || (q % 16u) == group;
                    if (mine) {
                        n = agent_load(a.tail.counts + q * kCountStride);
                        next = agent_load(&ctl->next_chunk[q * kCountStride]);
                    }
                }
                n = n < a.tail.shard_capacity ? n : a.tail.shard_capacity;
                const bool open = mine && ((next + 1u) * 64u <= n || (closing && next * 64u < n));
                unsigned long long m = __ballot(open);
                m = home == 0u ? m : (m >> home | m << (64u - home));                       // rotate: bit 0 = the wave's own shard
                if (m != 0ull) {
                    shard = (home + unsigned(__ffsll((long long)m) - 1)) % kShards;
                    n_s = __shfl(n, int(shard), 64);
                    if (lane == 0) chunk = atomicAdd(&ctl->next_chunk[shard * kCountStride], 1u);
                    chunk = __builtin_amdgcn_readfirstlane(chunk);
                    do_chunk = !(heads_done && chunk * 64u >= n_s);                          // (somebody else took the shard's last chunk)
                    break;
                }
                if (fresh || (pf_done | pf_hv) == 0u) break;                                 // nothing; or the old look's flags say: look properly
            }
        }
        if (!do_head && !do_chunk && sky_pending) {                                          // 3. sky: the claim sent off before the last blocks ran
            sky_pending = false;
            const unsigned i = __builtin_amdgcn_readfirstlane(pf_sky_i);
            const unsigned per_cursor = total_blocks > pf_k ? (total_blocks - pf_k + kCursors - 1u) / kCursors : 0u;
            if (i < per_cursor) { k = pf_k; first_j = i; count_j = i + 4u <= per_cursor ? 4u : per_cursor - i; last = i + 4u; do_head = true; }
        }
        if (!do_head && !do_chunk && !head_dry) do_head = claim_head(4u);                    //    ... or a claim made now
        if (!do_head && !do_chunk) {
            if (heads_done) break;                 // every counter is final and every chunk is claimed: done
            idle++;
            prof_idle++;
            for (unsigned z = 0; z < 4u; z++) __builtin_amdgcn_s_sleep(127);               // ~14 us
            if (idle > (1u << 13)) { if (lane == 0) atomicOr(&ctl->error[0], 1u); break; }  // > 0.1 s of nothing: give up
            continue;
        }
        idle = 0;
        if (do_head) {
            if (first_j * kCursors + k >= heavy_blocks && !head_dry && !held) {
                // Blocks of sky: what the wave will want to know when they are done is asked for NOW — the next claim of four and a
                // look at the chunks — so that the answers travel while the blocks run.  (Every decision of this scheduler is a round
                // trip through device-scope memory, 2-5 us; made one after the other they cost more than the 20 us of sky between them.)
                // Never among the walking tiles: a wave that held two of the longest chains would run them one after the other.
                pf_k = k;
                pf_sky_i = 0;
                if (lane == 0) pf_sky_i = atomicAdd(&ctl->cursor[k * kHotStride], 4u);
                prof_claims++;
                sky_pending = true;
                pf_done = agent_load(my_flag);
                pf_hv = agent_load(my_heavy_flag);
                pf_mine = closing_seen || (unsigned(lane) % 16u) == group;
                pf_n = 0; pf_next = 0;
                if (pf_mine) {
                    pf_n = agent_load(a.tail.counts + unsigned(lane) * kCountStride);
                    pf_next = agent_load(&ctl->next_chunk[unsigned(lane) * kCountStride]);
                }
                pf = true;
            }
            for (unsigned j = first_j; j < first_j + count_j; j++) {
                const unsigned b = j * kCursors + k;
                // a walking tile's chain is the launch's critical path, and here it shares its SIMD with chunks and blocks of sky
                if (b < heavy_blocks) __builtin_amdgcn_s_setprio(3);
                trace_block<false, kF, true>(a, b, lds_stack, stamp);
                __builtin_amdgcn_s_setprio(0);
                my_blocks++;
                if (b < heavy_blocks) my_heavy++;
            }
            if (my_heavy != 0u && last * kCursors + k >= heavy_blocks) report_heavy();       // this wave has left the front of the order
            continue;
        }
        // ---- a chunk.  Which of its 64 slots hold (or will hold) a record: all of them if the chunk is full; for a part-filled chunk,
        // wait until it fills, or — once part-filled chunks may be taken — CLOSE it: the shard's counter jumps from n to the chunk's
        // end, so that later records start the next chunk, and the slots from n on stay empty.  Wave-uniform.
        __builtin_amdgcn_s_setprio(1);             // a chunk is a chain too: ahead of the blocks of sky, behind the walking tiles
        const unsigned chunk_end = (chunk + 1u) * 64u < a.tail.shard_capacity ? (chunk + 1u) * 64u : a.tail.shard_capacity;
        unsigned limit = chunk_end;
        if (n_s < chunk_end && sky_pending) {      // (see `held`)
            held = true; held_shard = shard; held_chunk = chunk; held_ns = n_s;
            __builtin_amdgcn_s_setprio(0);
            continue;
        }
        if (n_s < chunk_end) {
            report_heavy();                        // this wait may depend on the flags: nothing this wave has finished may be missing from them
            report_blocks();
            for (unsigned tries = 0;; tries++) {
                const bool done_now = agent_load_first(my_flag) != 0u;                      // sampled before the counter: then it is final
                const unsigned cnt = agent_load(a.tail.counts + shard * kCountStride);
                if (cnt >= chunk_end) break;                                                // filled meanwhile
                if (done_now) { limit = cnt > chunk * 64u ? cnt : chunk * 64u; break; }     // final: what is there is all there will be
                // (only the chunk at the shard's fill front is closed — cnt > chunk * 64: a jump over an earlier chunk's free slots would
                // make its owner wait for records that never come)
                if (heavy_blocks != 0u && cnt > chunk * 64u && agent_load(my_heavy_flag) != 0u && chunk_end == (chunk + 1u) * 64u) {
                    unsigned old = 0;
                    if (lane == 0) old = atomicCAS(a.tail.counts + shard * kCountStride, cnt, chunk_end);
                    old = __builtin_amdgcn_readfirstlane(old);
                    if (old == cnt) { limit = cnt; break; }                                  // closed with cnt records
                    continue;                                                               // the counter moved: look again
                }
                __builtin_amdgcn_s_sleep(100);
                if (tries > (1u << 15)) { if (lane == 0) atomicOr(&ctl->error[0], 8u); limit = chunk * 64u; break; }
            }
        }
        if (heads_done) prof_after++; else prof_before++;
        const unsigned entry = chunk * 64u + unsigned(lane);
        const bool in_queue = entry < limit;
        float4* slot = a.tail.recs + (size_t(shard) * a.tail.shard_capacity + (in_queue ? entry : 0u)) * 4u;
        unsigned long long* word = reinterpret_cast<unsigned long long*>(slot) + 3;          // dir.z | normal_ambient, stamp on top
        bool valid = false;
        unsigned long long w3 = 0ull;
        if (in_queue) {   // reserved: the record is there or on its way
            for (unsigned polls = 0;; polls++) {
                w3 = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (unsigned(w3 >> 48) == stamp) { valid = true; break; }
                if (lane == 0) prof_polls++;
                if (polls < 16u) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(64);
                if (polls > (1u << 14)) { atomicOr(&ctl->error[0], 2u); break; }
            }
        }
        if (valid) {
            const PathRec rec = load_rec_fused(slot, w3);
            __hip_atomic_store(word, w3 & 0x0000ffffffffffffull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // stamp 0: the slot is free for the next launch
            bounce_path<false>(a, caster, rec, none, shard, first_bounce, a.max_bounces, sun_dir, sun_color, sky, rays);
        }
        __builtin_amdgcn_s_setprio(0);
    }
    count_rays(a.ray_counter, rays, lane);
    if (prof_wave) {
        atomicMax(&ctl->prof[2], (unsigned long long)__builtin_amdgcn_s_memrealtime());
        atomicAdd(&ctl->prof[3], (unsigned long long)prof_before);
        atomicAdd(&ctl->prof[4], (unsigned long long)prof_after);
        atomicAdd(&ctl->prof[5], (unsigned long long)prof_idle);
        atomicAdd(&ctl->prof[6], (unsigned long long)prof_polls);
        atomicAdd(&ctl->prof[7], (unsigned long long)prof_claims);
    }
}

#endif  // VXRT_VARIANTS

// Counting sort of the tiles by descending cost (key = log2 of the cost with two mantissa bits: 128 bins) in three small launches:
// per-block histograms over contiguous tile ranges (coalesced reads) -> one block turns them into (key, block) offsets -> each block
// scatters its tiles.  (One 1024-thread block doing all of it took 0.46 ms for the 129 600 tiles of a 4K frame — on the trace
// stream's critical path whenever a single frame is rendered per call.)
constexpr unsigned kSortBlocks = 64, kSortBins = 128;   // block_hist is [bin][block]

__device__ __forceinline__ unsigned tile_key(uint32_t c) {
    if (c < 4u) return c;
    const unsigned e = 31u - unsigned(__clz(int(c)));
    return (e << 2 | ((c >> (e - 2u)) & 3u)) - 4u;   // 4..127, monotone in c
}

__global__ __launch_bounds__(256) void tile_hist_kernel(const uint32_t* cost, uint32_t* block_hist, unsigned tiles, unsigned per_block) {
    __shared__ unsigned hist[kSortBins];
    if (threadIdx.x < kSortBins) hist[threadIdx.x] = 0;
    __syncthreads();
    const unsigned t0 = blockIdx.x * per_block, t1 = t0 + per_block < tiles ? t0 + per_block : tiles;
    for (unsigned t = t0 + threadIdx.x; t < t1; t += 256u) atomicAdd(&hist[kSortBins - 1u - tile_key(cost[t])], 1u);
    __syncthreads();
    if (threadIdx.x < kSortBins) block_hist[threadIdx.x * gridDim.x + blockIdx.x] = hist[threadIdx.x];
}

// block_hist[k][b] -> the position in `order` where block b's tiles of bin k start (bins in descending cost, blocks in order).
// 4 waves x 32 bins; lane = block (blocks <= 64): an exclusive wave scan per bin, then the bins' bases.
// lower bound of the costs with key k (the inverse of tile_key)
__device__ __forceinline__ unsigned long long key_cost(unsigned k) {
    if (k < 4u) return k;
    const unsigned e = (k + 4u) >> 2, m = (k + 4u) & 3u;
    return (unsigned long long)(4u + m) << (e - 2u);
}

// ... and decides how far the tiles that walk are spread over the launch (tile_scatter_kernel), in 1/256: the launch lasts about
// T = (sum of the tiles' costs) x waves_x_launches / wave_slots (waves per tile x launches that share the chip; a tile's cost is its
// longest wave's duration), its longest chain L = the largest cost; the chains must have started by T - L, and they get shorter down
// the order, so the tiles that walk are spread over the first 1 - 2 L / T of the launch — not at all when the launch is little more
// than its chains (a rank's share of a frame at 4 or 8 ranks: there longest-first is 3 % faster, measured; with a whole frame per
// rank spreading is 1-3 % faster and steadier)
__global__ __launch_bounds__(256) void tile_scan_kernel(uint32_t* block_hist, unsigned blocks, unsigned waves_x_launches, unsigned wave_slots,
                                                        int spread_override) {
    __shared__ unsigned total[kSortBins], base[kSortBins];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (unsigned k = wave * 32u; k < wave * 32u + 32u; k++) {
        const unsigned v = lane < blocks ? block_hist[k * blocks + lane] : 0u;
        unsigned x = v;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned y = __shfl_up(x, off, 64);
            if (int(lane) >= off) x += y;
        }
        if (lane < blocks) block_hist[k * blocks + lane] = x - v;
        if (lane == 63u) total[k] = x;
    }
    __shared__ unsigned long long all_cost;
    __shared__ unsigned first_bin, half_total;
    if (threadIdx.x == 0) { all_cost = 0ull; first_bin = kSortBins; }
    __syncthreads();
    if (threadIdx.x < kSortBins) {   // bins' bases (an exclusive scan over the 128 totals, 64 per wave), the summed cost, the first bin in use
        const unsigned i = threadIdx.x, v = total[i];
        unsigned x = v;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned y = __shfl_up(x, off, 64);
            if (int(lane) >= off) x += y;
        }
        base[i] = x - v;
        if (i == 63u) half_total = x;
        if (v != 0u) {
            atomicAdd(&all_cost, key_cost(kSortBins - 1u - i) * v);
            atomicMin(&first_bin, i);
        }
    }
    __syncthreads();
    if (threadIdx.x >= 64u && threadIdx.x < kSortBins) base[threadIdx.x] += half_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        block_hist[kSortBins * kSortBlocks] = base[kSortBins - 1u - 3u];   // tiles with cost >= 4: somebody walked (bins are in descending cost)
        const unsigned long long longest = first_bin < kSortBins ? key_cost(kSortBins - 1u - first_bin) : 0ull;
        const unsigned long long T = all_cost * waves_x_launches / (wave_slots ? wave_slots : 1u);
        unsigned spread = T > 2u * longest ? unsigned((T - 2u * longest) * 256u / T) : 0u;
        if (spread_override >= 0) spread = unsigned(spread_override);
        block_hist[kSortBins * kSortBlocks + 1u] = spread > 256u ? 256u : spread;     // spread_position needs <= 256 (k <= nl / nh)
    }
    __syncthreads();
    for (unsigned k = wave * 32u; k < wave * 32u + 32u; k++)
        if (lane < blocks) block_hist[k * blocks + lane] += base[k];
}

// Where the tile of sorted rank r (descending cost) goes in the launch order.  Plain longest-first puts every tile that walks ahead
// of every tile of sky: the chip then runs a VALU-bound phase followed by a store-bound one, and the trace stage's rate came to
// depend on the noise of the measured costs (100.7 .. 106.1 ms per 960 bench frames from process to process, against a steady 100.9
// in plain row-major order, where sky and geometry tiles alternate by themselves).  So: the nh tiles that walk keep their
// descending order — the longest chains still start first — but are SPREAD over the launch, each followed by k = nl / nh of the nl
// tiles that only store (kLightCost); what is left of those comes last.  With more walking than light tiles the order stays
// longest-first (a view without sky is bound by its chains).
__device__ __forceinline__ unsigned spread_position(unsigned r, unsigned n, unsigned nh, unsigned spread256) {
    const unsigned nl = n - nh;
    if (nh == 0u) return r;
    // k even: an odd period.  Blocks reach the CUs in a fixed rotation of their index, and with an even period the tiles that walk
    // met the same half (period 2) or quarter (4) of the CUs launch after launch: 22 instead of 31 Gray/s on the bench view
    const unsigned k = unsigned((unsigned long long)nl * spread256 / 256u / nh) & ~1u, period = k + 1u;
    if (k == 0u) return r;
    if (r < nh) return r * period;
    const unsigned q = r - nh, turn = q / k;
    return turn < nh ? turn * period + 1u + (q - turn * k) : nh * period + (q - nh * k);
}

__global__ __launch_bounds__(256) void tile_scatter_kernel(uint32_t* cost, uint32_t* order, uint32_t* last_cost, const uint32_t* block_offs,
                                                          unsigned tiles, unsigned per_block) {
    __shared__ unsigned offs[kSortBins];
    if (threadIdx.x < kSortBins) offs[threadIdx.x] = block_offs[threadIdx.x * gridDim.x + blockIdx.x];
    __syncthreads();
    const unsigned heavy = block_offs[kSortBins * kSortBlocks], spread256 = block_offs[kSortBins * kSortBlocks + 1u];
    const unsigned t0 = blockIdx.x * per_block, t1 = t0 + per_block < tiles ? t0 + per_block : tiles;
    for (unsigned t = t0 + threadIdx.x; t < t1; t += 256u) {
        const uint32_t c = cost[t];
        order[spread_position(atomicAdd(&offs[kSortBins - 1u - tile_key(c)], 1u), tiles, heavy, spread256)] = t;
        last_cost[t] = c;
        cost[t] = 0u;
    }
}

}  // namespace

namespace {
// Test hook (vxrt_debug_cast_rays): cast_ray — the walk every tracer uses — for caller-given rays, one lane per ray.
// out: 8 floats per ray = hit flag, time, bits(leaf word), normal xyz, 0, 0.
template <bool kWide>
__global__ __launch_bounds__(kTB) void cast_probe_kernel(const TraceArgs a, const float* origins, const float* dirs, float* out, unsigned n) {
    extern __shared__ uint4 lds_stack[];
    const unsigned i = blockIdx.x * kTB + threadIdx.x;
    if (i >= n) return;
    const Caster<kWide> caster(a, lds_stack, int(threadIdx.x));
    RayHit hit;
    hit.time = 0.0f; hit.node = 0; hit.normal = splat3(0.0f);
    const bool ok = caster.cast(ld3(origins + 3 * i), ld3(dirs + 3 * i), hit);
    float* o = out + 8 * size_t(i);
    o[0] = ok ? 1.0f : 0.0f; o[1] = hit.time; o[2] = __int_as_float(hit.node);
    o[3] = hit.normal.x; o[4] = hit.normal.y; o[5] = hit.normal.z; o[6] = 0.0f; o[7] = 0.0f;
}
// Test hook (vxrt_debug_path_log): one pixel's path with every cast logged — cast_ray and shade_hit, the code of all tracers, in one lane.
template <bool kWide>
__global__ __launch_bounds__(kTB) void path_log_kernel(const TraceArgs a, int x, int y, float* log) {
    extern __shared__ uint4 lds_stack[];
    if (threadIdx.x != 0) return;
    const Caster<kWide> caster(a, lds_stack, 0);
    Rng rng;
    rng.noise = a.noise;
    rng.index = uint32_t(x) % 128u + (uint32_t(y) % 128u) * 128u + (a.frame_number % 512u) * kNoiseLayer;
    const f3 sun_dir = ld3(a.sun_dir), sun_color = ld3(a.sun_color);
    f3 o = ld3(a.cam.o);
    f3 d = norm3((float(x) * ld3(a.cam.r) - float(y) * ld3(a.cam.u)) + ld3(a.cam.f));
    f3 sample = splat3(0.0f), blend = splat3(1.0f);
    uint32_t ambient_rays = 1;
    int casts = 0;
    auto cast = [&](f3 ro, f3 rd, RayHit& hit) {
        hit.time = 0.0f; hit.node = 0; hit.normal = splat3(0.0f);
        const bool ok = caster.cast(ro, rd, hit);
        if (casts < 32) {
            float* r = log + 12 * casts++;
            r[0] = ro.x; r[1] = ro.y; r[2] = ro.z; r[3] = rd.x; r[4] = rd.y; r[5] = rd.z; r[6] = ok ? 1.0f : 0.0f; r[7] = hit.time;
            r[8] = __int_as_float(hit.node); r[9] = hit.normal.x; r[10] = hit.normal.y; r[11] = hit.normal.z;
        }
        return ok;
    };
    for (int bounce = 0; bounce < a.max_bounces; bounce++) {
        RayHit hit;
        if (!cast(o, d, hit)) break;
        const Shaded s = shade_hit(a, bounce, o + d * hit.time, d, hit.normal, hit.node, sample, blend, ambient_rays, rng, sun_dir, sun_color);
        sample = s.sample; blend = s.blend; ambient_rays = s.ambient_rays;
        if (s.flags & kFlagSun) {
            RayHit sh;
            cast(s.origin, s.sun_dir, sh);
        }
        o = s.origin;
        d = s.bounce_dir;
    }
    log[12 * 32] = float(casts);
}
}  // namespace

hipError_t launch_cast_probe(const TraceArgs& a, bool wide, const float* origins, const float* dirs, float* out, unsigned n, hipStream_t s) {
    const size_t lds = caster_lds_bytes(a, wide, kTB);
#if VXRT_VARIANTS
    if (wide) {
        hipLaunchKernelGGL(cast_probe_kernel<true>, dim3((n + kTB - 1) / kTB), dim3(kTB), lds, s, a, origins, dirs, out, n);
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL(cast_probe_kernel<false>, dim3((n + kTB - 1) / kTB), dim3(kTB), lds, s, a, origins, dirs, out, n);
    return hipGetLastError();
}

hipError_t launch_path_log(const TraceArgs& a, bool wide, int x, int y, float* log, hipStream_t s) {
    const size_t lds = caster_lds_bytes(a, wide, kTB);
#if VXRT_VARIANTS
    if (wide) {
        hipLaunchKernelGGL(path_log_kernel<true>, dim3(1), dim3(kTB), lds, s, a, x, y, log);
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL(path_log_kernel<false>, dim3(1), dim3(kTB), lds, s, a, x, y, log);
    return hipGetLastError();
}

// Diagnostics (vxrt_debug_culled_pixels): how many of this rank's pixels the sky cull decides without a walk for the camera a.cam —
// the same test on the same ray as trace_kernel's, one lane per pixel, one atomic per wave.
__global__ __launch_bounds__(256) void count_culled_kernel(TraceArgs a, unsigned long long* count) {
    const int x = int(blockIdx.x * 64u + (threadIdx.x & 63u)), lrow = int(blockIdx.y * 4u + (threadIdx.x >> 6));
    bool culled = false;
    if (x < a.band.width && lrow < a.band.local_rows) {
        const int y = frame_row(a.band, lrow);
        const f3 o = ld3(a.cam.o);
        const f3 d = norm3((float(x) * ld3(a.cam.r) - float(y) * ld3(a.cam.u)) + ld3(a.cam.f));  // voxels.comp:299-303
        culled = y < a.band.height && primary_miss_is_certain(a, o, d);
    }
    const unsigned long long m = __ballot(culled);
    if ((threadIdx.x & 63u) == 0u && m != 0ull) atomicAdd(count, (unsigned long long)__popcll(m));
}

hipError_t launch_count_culled(const TraceArgs& a, unsigned long long* count, hipStream_t s) {
    if (a.band.local_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(count_culled_kernel, dim3(unsigned(a.band.width + 63) / 64u, unsigned(a.band.local_rows + 3) / 4u), dim3(256), 0, s, a, count);
    return hipGetLastError();
}

// tiles (= blocks per frame) of trace_kernel: the unit of the longest-tile-first schedule
unsigned trace_tile_count(int width, int local_rows) {
    return unsigned((width + kTileW - 1) / kTileW) * unsigned((local_rows + kTileH - 1) / kTileH);
}

void trace_tile_dims(int* w, int* h) { *w = kTileW; *h = kTileH; }

// hbm_scene: the scene does not fit the Infinity Cache (BASELINE config 5: 5.6 GB), so a descend waits for HBM and one more wave per
// SIMD hides more of that than its spilled registers cost: 2.25 -> 2.08 ms (outside view), 15.5 -> 14.1 ms (tunnel) at 4K, 8 bounces;
// 8 waves: 3.79 / 26.1 ms.  On a cache-resident scene the same change loses 3-5 %.
hipError_t launch_trace(const TraceArgs& args, bool wide, bool hbm_scene, hipStream_t s, unsigned block_first, unsigned block_count) {
    TraceArgs a = args;
    a.block_first = block_first;
    // those kernels exist with one frame per wave only (for a scene in HBM, config 5 at 16 spp, frame lanes measured 7 % slower: 14.9
    // against 13.9 ms per displayed frame — a wave's 64 pixels are neighbours in the tree, its 8 frames of 8 pixels less so)
    if (hbm_scene || wide || kTB != 64) a.frame_lanes = 0;
    const unsigned all_blocks = trace_tile_count(a.band.width, a.band.local_rows) * unsigned(a.batch);   // kF parts x batch / kF groups per tile
    if (block_first >= all_blocks) return hipSuccess;
    dim3 grid(block_count == 0u || block_count > all_blocks - block_first ? all_blocks - block_first : block_count);
    const size_t lds = caster_lds_bytes(a, wide, kTB);
#if VXRT_VARIANTS
    if (wide) {
        hipLaunchKernelGGL((trace_kernel<true, VXRT_TRACE_WAVES, 1>), grid, dim3(kTB), lds, s, a);
        return hipGetLastError();
    }
#endif
    constexpr int kF8 = kTB == 64 ? 8 : 1, kF4 = kTB == 64 ? 4 : 1;
    if (hbm_scene) hipLaunchKernelGGL((trace_kernel<false, VXRT_TRACE_WAVES_HBM, 1>), grid, dim3(kTB), lds, s, a);
    else if (a.frame_lanes == 8) hipLaunchKernelGGL((trace_kernel<false, VXRT_TRACE_WAVES, kF8>), grid, dim3(kTB), lds, s, a);
    else if (a.frame_lanes == 4) hipLaunchKernelGGL((trace_kernel<false, VXRT_TRACE_WAVES, kF4>), grid, dim3(kTB), lds, s, a);
    else hipLaunchKernelGGL((trace_kernel<false, VXRT_TRACE_WAVES, 1>), grid, dim3(kTB), lds, s, a);
    return hipGetLastError();
}


#if VXRT_VARIANTS
size_t fused_ctl_bytes() { return sizeof(FusedCtl); }

// One grid of persistent one-wave blocks for the whole launch (fused_kernel).  ctl: device memory of fused_ctl_bytes(), zeroed on the
// stream before this call.  sort_scratch: the tile sort's scratch of the order in a.tile_order (null: none yet — every block is claimed in fours).
// stamp: 1 .. 65535, different from the previous launch's on the same queue.
hipError_t launch_fused(const TraceArgs& args, void* ctl, unsigned waves, const uint32_t* sort_scratch, uint32_t stamp, hipStream_t s) {
    TraceArgs a = args;
    a.block_first = 0;
    if (kTB != 64) return hipErrorInvalidValue;
    const unsigned all_blocks = trace_tile_count(a.band.width, a.band.local_rows) * unsigned(a.batch);
    const uint32_t* sort_info = sort_scratch ? sort_scratch + kSortBins * kSortBlocks : nullptr;
    const size_t lds = caster_lds_bytes(a, false, kTB);
    dim3 grid(waves < 1u ? 1u : waves);
    FusedCtl* fc = static_cast<FusedCtl*>(ctl);
    constexpr int kF8 = kTB == 64 ? 8 : 1, kF4 = kTB == 64 ? 4 : 1;
    if (a.frame_lanes == 8) hipLaunchKernelGGL((fused_kernel<kF8>), grid, dim3(kTB), lds, s, a, fc, all_blocks, sort_info, stamp, a.tail_from);
    else if (a.frame_lanes == 4) hipLaunchKernelGGL((fused_kernel<kF4>), grid, dim3(kTB), lds, s, a, fc, all_blocks, sort_info, stamp, a.tail_from);
    else hipLaunchKernelGGL((fused_kernel<1>), grid, dim3(kTB), lds, s, a, fc, all_blocks, sort_info, stamp, a.tail_from);
    return hipGetLastError();
}

// byte offset of the error word in the control block (the host copies the block's head back after a launch)
size_t fused_ctl_error_offset() { return offsetof(FusedCtl, error); }
size_t fused_ctl_profile_offset() { return offsetof(FusedCtl, prof); }

#endif  // VXRT_VARIANTS

hipError_t launch_tile_order(uint32_t* cost, uint32_t* order, uint32_t* last_cost, uint32_t* scratch, unsigned tiles, unsigned waves_x_launches,
                             unsigned wave_slots, int spread_override, hipStream_t s) {
    const unsigned blocks = (tiles + 255u) / 256u < kSortBlocks ? (tiles + 255u) / 256u : kSortBlocks;
    const unsigned per_block = (tiles + blocks - 1u) / blocks;
    hipLaunchKernelGGL(tile_hist_kernel, dim3(blocks), dim3(256), 0, s, cost, scratch, tiles, per_block);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(256), 0, s, scratch, blocks, waves_x_launches, wave_slots, spread_override);
    hipLaunchKernelGGL(tile_scatter_kernel, dim3(blocks), dim3(256), 0, s, cost, order, last_cost, scratch, tiles, per_block);
    return hipGetLastError();
}
}  // namespace vxrt
