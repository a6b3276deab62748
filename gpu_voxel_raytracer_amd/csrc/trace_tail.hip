// trace_tail.hip — bounce_kernel: the compacted tail of the tracer (gfx950).  One lane per queued path, lock step: shade the hit the
// path arrived with, cast its sun ray, cast its bounce ray, and so on for the launch's range of path segments; survivors go to the
// next queue.  Used by tracer 4 (after trace_kernel) and tracer 2 (after primary_kernel).  Queues: trace_common.h.
#ifndef VXRT_TAIL_BLOCK
#define VXRT_TAIL_BLOCK 64   // threads per block: one wave — its slot is re-usable the moment its last chunk is done
#endif
#define VXRT_STACK_STRIDE VXRT_TAIL_BLOCK
#include "trace_common.h"
#include "trace_tail_body.h"

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

        if (valid) {
            const PathRec rec = load_rec(in.recs + (size_t(q) * in.shard_capacity + entry) * 4u);
            bounce_path<kWide>(a, caster, rec, out, c % kShards, first_bounce, last_bounce, sun_dir, sun_color, sky, rays);
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
