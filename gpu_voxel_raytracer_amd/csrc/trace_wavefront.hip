// trace_wavefront.hip — queue-based variants of the tracer (gfx950): one launch per path segment.
#include "trace_common.h"

namespace vxrt {
namespace {

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
