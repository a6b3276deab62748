// trace_tail.hip — bounce_kernel: the compacted tail of the tracer (gfx950).  One lane per queued path, lock step: shade the hit the
// path arrived with, cast its sun ray, cast its bounce ray, and so on for the launch's range of path segments; survivors go to the
// next queue.  Used by tracer 4 (after trace_kernel) and tracer 2 (after primary_kernel).  Queues: trace_common.h.
#ifndef VXRT_TAIL_BLOCK
#define VXRT_TAIL_BLOCK 64   // threads per block: one wave — its slot is re-usable the moment its last chunk is done
#endif
#define VXRT_STACK_STRIDE VXRT_TAIL_BLOCK
#include "trace_common.h"

namespace vxrt {
namespace {

constexpr int kTailBlock = VXRT_TAIL_BLOCK;

#ifndef VXRT_BOUNCE_WAVES
#define VXRT_BOUNCE_WAVES 5
#endif
template <bool kWide>
__global__ __launch_bounds__(kTailBlock, VXRT_BOUNCE_WAVES) void bounce_kernel(const TraceArgs a, const PathQueue in, const PathQueue out, unsigned* zero,
                                                        int first_bounce, int last_bounce) {
    extern __shared__ uint4 lds_stack[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    zero_counts(zero, tid);
    const Caster<kWide> caster(a, lds_stack, tid);
    const f3 sun_dir = ld3(a.sun_dir), sun_color = ld3(a.sun_color), sky = ld3(a.sky_color);

    // chunk table: lane q owns shard q
    const unsigned my_count = queue_count(in, unsigned(lane));
    const unsigned my_chunks = (my_count + 63u) / 64u;
    unsigned incl = my_chunks;
    for (int off = 1; off < 64; off <<= 1) {
        unsigned v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    const unsigned total_chunks = __shfl(incl, 63, 64);
    const unsigned total_waves = gridDim.x * unsigned(kTailBlock / 64);
    uint32_t rays = 0;

    for (unsigned c = blockIdx.x * unsigned(kTailBlock / 64) + unsigned(wave); c < total_chunks; c += total_waves) {
        const unsigned long long above = __ballot(incl > c);
        const int q = __ffsll((long long)above) - 1;                       // shard that holds chunk c
        const unsigned first = __shfl(incl - my_chunks, q, 64);           // chunks before shard q
        const unsigned count_q = __shfl(my_count, q, 64);
        const unsigned entry = (c - first) * 64u + unsigned(lane);
        const bool valid = entry < count_q;

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
                    const float facing = vx_max(0.0f, dot3(n, to_light));
                    RayHit sun_hit;
                    rays++;
                    const bool lit = !caster.cast(o, to_light, sun_hit);
                    // normal, colour and emittance are derived again from the record instead of being kept alive across the cast
                    // (the same operations on the same operands; ten registers fewer while the sun ray walks)
#ifndef VXRT_TAIL_REMAT
#define VXRT_TAIL_REMAT 1
#endif
#if VXRT_TAIL_REMAT
                    uint32_t packed = rec.normal_ambient;
                    int32_t word = rec.node;
                    asm volatile("" : "+v"(packed), "+v"(word));   // new values to the compiler: no common subexpression with n / color / emit above
                    const f3 n2 = mk3(unpack_axis(packed & 3u), unpack_axis((packed >> 2) & 3u), unpack_axis((packed >> 4) & 3u));
                    const f3 color2 = bounce == 0 ? splat3(1.0f) : node_color(word);
                    const f3 emit2 = node_emittance(word, a.emit_strength);
#else
                    const f3 n2 = n, color2 = color, emit2 = emit;
#endif
                    if (lit) sample = sample + ((sun_color * color2) * blend) * facing;
                    d = random_hemisphere(n2, rng);
                    sample = sample + emit2 * blend;
                    blend = blend * (color2 * dot3(n2, d));
                } else {  // diffuse, sun switched off
                    d = random_hemisphere(n, rng);
                    sample = sample + emit * blend;
                    blend = blend * (color * dot3(n, d));
                }

                bool finished = true;
                if (bounce + 1 < a.max_bounces) {  // next path segment                        voxels.comp:309-313
                    RayHit hit;
                    rays++;
                    if (caster.cast(o, d, hit)) {
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
                    store_out(a.out[rec.pix >> kPixBits].color + (rec.pix & ((1u << kPixBits) - 1u)), make_float4(outc.x, outc.y, outc.z, 1.0f));
                    break;
                }
                // compact again: to the next queue — unless that is full, then this lane simply goes on
                if (bounce == last_bounce) {
                    const uint32_t slot = queue_reserve(out, c % kShards);
                    if (slot != kNoSlot) {
                        rec.rng_index = rng.index;
                        queue_store(out, c % kShards, slot, rec);
                        break;
                    }
                }
            }
        }
    }
    count_rays(a.ray_counter, rays, lane);
}


}  // namespace

// bounce_kernel launches for the path segments from..max_bounces-1 of the paths in queues[0] (written by launch J-1 with
// count set J%3).  Bit k of split_mask set: a new launch (with compaction of the live paths) starts at path segment k.
// Launch J reads count set J%3, writes (J+1)%3 and clears (J+2)%3 (the set launch J-1 consumed).
hipError_t launch_bounces(const TraceArgs& a, bool wide, const PathQueue queues[2], unsigned* count_sets[3], unsigned* launch_counter, int blocks,
                          unsigned split_mask, int from, hipStream_t s) {
    const size_t lds = caster_lds_bytes(a, wide, kTailBlock);
    blocks *= kBlock / kTailBlock;  // `blocks` counts 4-wave blocks
    unsigned J = *launch_counter;
    int stage = 0;
    for (int first = from; first < a.max_bounces;) {
        int last = first;
        while (last + 1 < a.max_bounces && !((split_mask >> (last + 1)) & 1u)) last++;
        PathQueue in = queues[stage & 1];
        in.counts = count_sets[J % 3];
        PathQueue out = queues[(stage & 1) ^ 1];
        out.counts = count_sets[(J + 1) % 3];
#if VXRT_VARIANTS
        if (wide)
            hipLaunchKernelGGL(bounce_kernel<true>, dim3(blocks), dim3(kTailBlock), lds, s, a, in, out, count_sets[(J + 2) % 3], first, last);
        else
#endif
            hipLaunchKernelGGL(bounce_kernel<false>, dim3(blocks), dim3(kTailBlock), lds, s, a, in, out, count_sets[(J + 2) % 3], first, last);
        J++;
        stage++;
        first = last + 1;
    }
    *launch_counter = J;
    return hipGetLastError();
}

}  // namespace vxrt
