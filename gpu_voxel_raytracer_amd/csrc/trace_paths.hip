// trace_paths.hip — path_kernel: the tail of tracer 5, persistent waves that REFILL EACH LANE with a new path (gfx950).
//
// bounce_kernel (trace_wavefront.hip) finishes the queued paths one lane per path in lock step: every cast_ray ends when
// the wave's longest ray ends, and a lane whose path has died idles until the whole chunk is done.  Live paths are dense
// there, yet rocprofv3 shows 28 % lane utilisation: ray lengths are heavy-tailed (mean 8 octree steps, p99 50-70) and
// about a fifth of the paths die at every segment.  Here a wave owns a contiguous range of the queue's 64-entry chunks
// and its lanes run free:
//
//   * a lane walks its current ray one octree step per trip (walkf_step, the regular-ray walk of trace_common.h);
//   * a lane whose ray has ended, or whose path has ended while the wave's range still holds paths, WAITS;
//   * when kGate lanes wait (or nobody walks) the wave runs the ADVANCE block once for all of them: refill idle lanes
//     from the range (entries of the current chunk, handed out by mbcnt over the idle lanes — no atomics), resolve the
//     rays that ended (leaf word, normal: voxels.comp:177-189), apply sun results, shade new hits (shade_hit:
//     voxels.comp:314-371), finish paths (voxels.comp:384-391), start the next ray (voxels.comp:138-160).
//
// The shading code exists once and runs for >= kGate lanes at a time; the walk never waits for the longest ray of a round.
// tests/sim_schedule.py::tail_refill prices this on the oracle's per-ray step counts: 17-21 M wave-instructions for the
// bench frame's tail against 31.6 M in lock step.  Per path the operation order is voxels.comp's; results are bit-identical.
#include "trace_common.h"

namespace vxrt {
namespace {

#ifndef VXRT_PATH_GATE
#define VXRT_PATH_GATE 24
#endif
#ifndef VXRT_PATH_WAVES
#define VXRT_PATH_WAVES 4
#endif
constexpr int kGate = VXRT_PATH_GATE;
#ifndef VXRT_PATH_MIN_CHUNKS
#define VXRT_PATH_MIN_CHUNKS 8
#endif
constexpr unsigned kMinChunksPerWave = VXRT_PATH_MIN_CHUNKS;  // a wave's range: at least 512 paths, so that refilling has something to draw on

enum : int { kIdle = 0, kWalkSun = 1, kWalkBounce = 2, kDoneSun = 3, kDoneBounce = 4, kFresh = 5 };

__global__ __launch_bounds__(kBlock, VXRT_PATH_WAVES) void path_kernel(const TraceArgs a, const PathQueue in, unsigned* zero, int first_bounce) {
    extern __shared__ uint2 lds_stack[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    zero_counts(zero, tid);
    const SceneView sc = make_scene(a);
    const f3 sun_dir = ld3(a.sun_dir), sun_color = ld3(a.sun_color), sky = ld3(a.sky_color);
    uint2* stack = lds_stack + tid;

    // chunk table (as in bounce_kernel): lane q owns shard q; chunk c lives in the shard whose inclusive chunk count exceeds c
    const unsigned my_count = queue_count(in, unsigned(lane));
    const unsigned my_chunks = (my_count + 63u) / 64u;
    unsigned incl = my_chunks;
    for (int off = 1; off < 64; off <<= 1) {
        unsigned v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    const unsigned total_chunks = __shfl(incl, 63, 64);
    const unsigned total_waves = gridDim.x * 4u;
    unsigned per_wave = (total_chunks + total_waves - 1u) / total_waves;
    per_wave = per_wave < kMinChunksPerWave ? kMinChunksPerWave : per_wave;
    const unsigned wave_id = blockIdx.x * 4u + unsigned(wave);
    unsigned chunk = wave_id * per_wave;                                   // next chunk of this wave's range ...
    const unsigned chunk_end = chunk + per_wave < total_chunks ? chunk + per_wave : total_chunks;
    unsigned cursor = 0, chunk_count = 0, chunk_shard = 0;                 // ... and the entries left in the one being handed out
    bool chunk_open = false;

    // lane state: the path, its pending shading results, the ray being walked
    int phase = kIdle, bounce = 0, status = kWalkMiss;
    f3 hit_pos = splat3(0.0f), dir = splat3(0.0f), n = splat3(0.0f), sample = splat3(0.0f), blend = splat3(0.0f);
    f3 pend_sun = splat3(0.0f), pend_emit = splat3(0.0f), bounce_dir = splat3(0.0f);
    int32_t node = 0;
    uint32_t ambient_rays = 0, pix = 0, flags = 0, rays = 0;
    Rng rng;
    rng.noise = a.noise;
    rng.index = 0;
    WalkF w;
    w.o = w.d = w.inv = w.center = w.en = w.ex = splat3(0.0f);
    w.time = w.exit = 0.0f;
    w.ix = w.iy = w.iz = w.lvl = w.has_next_mask = w.octant = w.dir_mask = 0;
    w.rec.masks = w.rec.base = 0;
    w.iterations = 0;

    for (;;) {
        if (phase == kWalkSun || phase == kWalkBounce) {  // one trip of voxels.comp:163-246
            status = walkf_step(w, sc, stack);
            if (status != kWalkOn) phase += 2;             // kWalk* -> kDone*
        }
        const bool range_left = chunk_open ? cursor < chunk_count || chunk < chunk_end : chunk < chunk_end;
        const bool waits = phase >= kDoneSun || (phase == kIdle && range_left);
        const unsigned long long waiting = __ballot(waits);
        const unsigned long long walking = __ballot(phase == kWalkSun || phase == kWalkBounce);
        if (waiting == 0ull && walking == 0ull) break;     // the range is empty and every path has ended
        if (__popcll(waiting) < kGate && walking != 0ull) continue;

        // ---------------------------------------- ADVANCE (wave-uniform entry) ----------------------------------------
        // 1. refill: idle lanes take the next entries of the current chunk
        if (!chunk_open && chunk < chunk_end) {
            const unsigned long long above = __ballot(incl > chunk);
            chunk_shard = unsigned(__ffsll((long long)above) - 1);
            const unsigned first = __shfl(incl - my_chunks, int(chunk_shard), 64);
            const unsigned count_q = __shfl(my_count, int(chunk_shard), 64);
            const unsigned base = (chunk - first) * 64u;
            chunk_count = count_q - base < 64u ? count_q - base : 64u;     // entries in this chunk
            cursor = 0;
            chunk_open = true;
            // entry index of the chunk's first record inside its shard is kept in chunk_base below
            chunk++;
            // (re-derive base when loading: (chunk - 1 - first) * 64)
        }
        if (chunk_open) {
            const unsigned long long idle = __ballot(phase == kIdle);
            const unsigned rank = __builtin_amdgcn_mbcnt_hi(unsigned(idle >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(idle), 0u));
            const unsigned first = __shfl(incl - my_chunks, int(chunk_shard), 64);
            if (phase == kIdle && cursor + rank < chunk_count) {
                const unsigned entry = (chunk - 1u - first) * 64u + cursor + rank;
                const PathRec rec = load_rec(in.recs + (size_t(chunk_shard) * in.shard_capacity + entry) * 4u);
                hit_pos = rec.hit_pos; node = rec.node; dir = rec.dir;
                n = mk3(unpack_axis(rec.normal_ambient & 3u), unpack_axis((rec.normal_ambient >> 2) & 3u), unpack_axis((rec.normal_ambient >> 4) & 3u));
                ambient_rays = rec.normal_ambient >> 8;
                sample = rec.sample; blend = rec.blend; rng.index = rec.rng_index; pix = rec.pix;
                bounce = first_bounce;
                phase = kFresh;
            }
            const unsigned taken = unsigned(__popcll(idle));
            cursor = cursor + taken < chunk_count ? cursor + taken : chunk_count;
            if (cursor >= chunk_count) chunk_open = false;
        }

        // 2. rays that ended
        bool finish = false;
        if (phase == kDoneSun) {                            // voxels.comp:357-371
            if (status == kWalkMiss) sample = sample + pend_sun;
            sample = sample + pend_emit;
            if (!(flags & kFlagBounce)) finish = true;
        } else if (phase == kDoneBounce) {
            if (status == kWalkMiss) {                      // voxels.comp:384
                sample = sample + sky * blend;
                finish = true;
            } else {                                        // a hit: leaf word and normal, voxels.comp:177-189
                const f3 o = w.o, d = w.d;
                RayHit hit;
                const uint32_t bit = 1u << w.octant;
                finish_ray(sc, status, o, d, w.time, w.center, w.lvl, w.octant, w.rec.base + __popc((w.rec.masks >> 8) & (bit - 1u)), hit);
                hit_pos = o + d * hit.time;
                node = hit.node;
                dir = d;
                n = hit.normal;
                bounce++;
                phase = kFresh;
            }
        }
        // 3. shade new hits: voxels.comp:314-371
        if (phase == kFresh) {
            const Shaded s = shade_hit(a, bounce, hit_pos, dir, n, node, sample, blend, ambient_rays, rng, sun_dir, sun_color);
            sample = s.sample; blend = s.blend; pend_sun = s.pend_sun; pend_emit = s.pend_emit;
            ambient_rays = s.ambient_rays; flags = s.flags;
            hit_pos = s.origin;                             // the origin of this hit's rays
            bounce_dir = s.bounce_dir;
            dir = s.sun_dir;
            if (!(flags & (kFlagSun | kFlagBounce))) finish = true;
        }
        // 4. finish paths: voxels.comp:391
        if (finish) {
            const f3 outc = sample / float(ambient_rays);
            store_out(a.out[pix >> kPixBits].color + (pix & ((1u << kPixBits) - 1u)), make_float4(outc.x, outc.y, outc.z, 1.0f));
            phase = kIdle;
        }
        // 5. start the next ray: the sun ray of a fresh hit, else the bounce ray
        if (phase == kFresh || phase == kDoneSun) {
            const bool sun = phase == kFresh && (flags & kFlagSun) != 0u;
            const f3 d = sun ? dir : bounce_dir;
            const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            rays++;
            phase = sun ? kWalkSun : kWalkBounce;
            if (ray_is_regular(inv)) {
                if (!walkf_begin(w, sc, hit_pos, d, inv)) { status = kWalkMiss; phase += 2; }
            } else {  // a direction component is 0 (or NaN), ~1e-7 of the rays: the shader's text, run to its end right here
                Walk g;
                status = kWalkMiss;
                if (walk_begin(g, sc, hit_pos, d)) {
                    do { status = walk_step(g, sc, kAlmostInfinity, stack); } while (status == kWalkOn);
                    w.o = hit_pos; w.d = d; w.time = g.time; w.center = g.center; w.lvl = g.lvl; w.octant = g.octant; w.rec = g.rec;
                }
                phase += 2;
            }
        }
    }
    count_rays(a.ray_counter, rays, lane);
}

}  // namespace

// tracer 5: the paths queued by trace_kernel (TraceArgs::tail), followed to their end by path_kernel.  `in.counts` is count set
// J % 3 of the rotation described at launch_bounces; this launch clears set (J + 2) % 3 and appends nothing.
hipError_t launch_paths(const TraceArgs& a, const PathQueue& in, unsigned* zero, int first_bounce, int blocks, hipStream_t s) {
    size_t lds = size_t(a.stack_levels) * kBlock * sizeof(uint2);
    hipLaunchKernelGGL(path_kernel, dim3(blocks), dim3(kBlock), lds, s, a, in, zero, first_bounce);
    return hipGetLastError();
}

}  // namespace vxrt
