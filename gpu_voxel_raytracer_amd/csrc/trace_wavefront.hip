// trace_wavefront.hip — queue-based variants of the tracer (gfx950): one launch per path segment.
#include "trace_common.h"
#include "ray_queue.h"

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
    const int y = frame_row(a.band, lrow);
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
            float sun_power = sun_power_of(a, d);
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

// ------------------------------------------------------------------------------------------------------
// Ray-queue variant: shading and traversal in separate launches, rays traced by persistent waves that
// REFILL EACH LANE as soon as its ray ends.
//
// What the monolithic kernel loses (oracle step statistics, menger bench frame): after the primary phase a
// wave's lanes trace rays of wildly different length (mean 8 octree steps, p99 50-70) in lockstep phases, so
// the six later ray phases keep only 13-27 % of the lanes busy.  Compacting dead paths does not help much —
// the variance between LIVE rays is the loss.  Here:
//
//   primary_kernel     (above)  primary rays, G-buffer, hits -> sharded PathRec queue
//   shade_kernel<true>           hit 0: RNG, sun sample, BRDF sample -> path state + (sun ray, bounce ray)
//   trace_rays_kernel            persistent waves; each wave owns a contiguous range of the stage's rays and
//                                hands a new ray to every lane that goes idle (wave64 ballot + mbcnt prefix
//                                over the idle lanes, no atomics); a ray's end state (time, leaf index, voxel
//                                coordinates) is stored raw — no shading code inside the traversal loop
//   shade_kernel<false>          per path: apply the sun result, resolve the bounce hit (leaf word, normal),
//                                shade it, emit the next two rays — or finish the pixel
//
// Paths are kept dense: a block appends its survivors with one atomicAdd on the counter of queue segment
// blockIdx % 8 (block-wide prefix through LDS), so trace waves can split the rays evenly without a scan.
// The per-path operation order is that of voxels.comp, the results are bit-identical to trace_kernel's.
// ------------------------------------------------------------------------------------------------------

// Block-wide dense append: returns this thread's slot in segment blockIdx % 8 of stage `stage` (or ~0u).
__device__ __forceinline__ unsigned dense_append(const RayQueue& q, int stage, bool keep, unsigned* lds_counts, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    const unsigned long long m = __ballot(keep);
    if (lane == 0) lds_counts[wave] = unsigned(__popcll(m));
    __syncthreads();
    const unsigned c0 = lds_counts[0], c1 = lds_counts[1], c2 = lds_counts[2], c3 = lds_counts[3];
    const unsigned seg = blockIdx.x % kSegments;
    if (tid == 0) {
        const unsigned total = c0 + c1 + c2 + c3;
        lds_counts[4] = total ? atomicAdd(q.counts + (unsigned(stage) * kSegments + seg) * kCountStride, total) : 0u;
    }
    __syncthreads();
    const unsigned base = lds_counts[4];
    __syncthreads();  // lds_counts is rewritten by the next call
    if (!keep) return ~0u;
    const unsigned before = wave == 0 ? 0u : (wave == 1 ? c0 : (wave == 2 ? c0 + c1 : c0 + c1 + c2));
    const unsigned rank = __builtin_amdgcn_mbcnt_hi(unsigned(m >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(m), 0u));
    return seg * q.seg_capacity + base + before + rank;
}

// kFirst: paths come from primary_kernel's sharded hit queue (hit already resolved); otherwise from stage-1.
template <bool kFirst>
__global__ __launch_bounds__(kBlock) void shade_kernel(const TraceArgs a, const PathQueue hits, const RayQueue q, unsigned* zero, int stage) {
    extern __shared__ uint2 lds_stack[];   // for the literal walk of an irregular ray
    __shared__ unsigned lds_counts[8];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    if (kFirst) zero_counts(zero, tid);
    const SceneView sc = make_scene(a);
    const f3 sun_dir = ld3(a.sun_dir), sun_color = ld3(a.sun_color), sky = ld3(a.sky_color);

    // work list: kFirst -> 64-entry chunks of the sharded hit queue; else dense path indices of stage-1
    unsigned my_count = 0, my_chunks = 0, incl = 0, total_items = 0;
    SegTable seg{};
    if (kFirst) {
        my_count = hits.counts[lane * kCountStride];
        my_chunks = (my_count + 63u) / 64u;
        incl = my_chunks;
        for (int off = 1; off < 64; off <<= 1) {
            unsigned v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        total_items = __shfl(incl, 63, 64) * 64u;  // padded: one item = one lane of a chunk
    } else {
        seg = load_segments(q.counts, stage - 1);
        total_items = seg.pre[kSegments];
    }
    const unsigned per_trip = gridDim.x * unsigned(kBlock);
    const unsigned trips = (total_items + per_trip - 1u) / per_trip;  // same for every block: barriers inside
    const float4* state_in = q.state[(stage + 1) & 1];
    const float4* rays_in = q.rays[(stage + 1) & 1];
    float4* state_out = q.state[stage & 1];
    float4* rays_out = q.rays[stage & 1];
    const uint4* results_in = q.results[(stage + 1) & 1];
    uint4* results_out = q.results[stage & 1];
    uint32_t rays_cast = 0;

    for (unsigned trip = 0; trip < trips; trip++) {
        const unsigned item = trip * per_trip + blockIdx.x * unsigned(kBlock) + unsigned(tid);
        bool have = false, keep = false;
        f3 hit_pos = splat3(0.0f), dir = splat3(0.0f), n = splat3(0.0f), sample = splat3(0.0f), blend = splat3(0.0f);
        int32_t node = 0;
        uint32_t ambient_rays = 1, pix = 0;
        Rng rng;
        rng.noise = a.noise;
        rng.index = 0;
        if (kFirst) {
            const unsigned c = item / 64u;  // wave-uniform
            const unsigned long long above = __ballot(incl > c);
            if (above != 0ull) {
                const int sh = __ffsll((long long)above) - 1;
                const unsigned first = __shfl(incl - my_chunks, sh, 64);
                const unsigned count_q = __shfl(my_count, sh, 64);
                const unsigned entry = (c - first) * 64u + unsigned(lane);
                if (entry < count_q) {
                    const PathRec rec = load_rec(hits.recs + (size_t(sh) * hits.shard_capacity + entry) * 4u);
                    hit_pos = rec.hit_pos; dir = rec.dir; node = rec.node;
                    n = mk3(unpack_axis(rec.normal_ambient & 3u), unpack_axis((rec.normal_ambient >> 2) & 3u), unpack_axis((rec.normal_ambient >> 4) & 3u));
                    ambient_rays = rec.normal_ambient >> 8;
                    sample = rec.sample; blend = rec.blend; rng.index = rec.rng_index; pix = rec.pix;
                    have = true;
                }
            }
        } else if (item < total_items) {
            const unsigned slot = segment_slot(seg, item, q.seg_capacity);
            const float4 s0 = state_in[size_t(slot) * 4], s1 = state_in[size_t(slot) * 4 + 1], s2 = state_in[size_t(slot) * 4 + 2],
                         s3 = state_in[size_t(slot) * 4 + 3];
            const float4 r0 = rays_in[size_t(slot) * 4];
            sample = xyz4(s0); rng.index = __float_as_uint(s0.w);
            blend = xyz4(s1); pix = __float_as_uint(s1.w);
            ambient_rays = __float_as_uint(s2.w);
            const uint32_t flags = __float_as_uint(r0.w);
            const f3 o = xyz4(r0);
            if (flags & kFlagSun) {  // voxels.comp:357-367: the sun sample counts unless something is in the way
                const uint4 rs = results_in[size_t(slot) * 2];
                if ((rs.w >> 16 & 3u) == unsigned(kWalkMiss)) sample = sample + xyz4(s2);
                sample = sample + xyz4(s3);
            }
            bool finished = true;
            if (flags & kFlagBounce) {
                const uint4 rb = results_in[size_t(slot) * 2 + 1];
                const unsigned status = rb.w >> 16 & 3u;
                const float4 r2 = rays_in[size_t(slot) * 4 + 2];
                const f3 d = mk3(r2.z, r2.w, rays_in[size_t(slot) * 4 + 3].x);
                if (status == unsigned(kWalkMiss)) {
                    sample = sample + sky * blend;                                        // voxels.comp:384
                } else {
                    const float time = __uint_as_float(rb.x);
                    hit_pos = o + d * time;
                    dir = d;
                    if (status == unsigned(kWalkCap)) {
                        node = kLeafBit;
                        n = splat3(0.0f);
                    } else {
                        node = sc.leaves[rb.y];
                        const unsigned lvl = rb.w >> 20 & 15u;  // level of the node the leaf sits in
                        const float voxel = __builtin_ldexpf(sc.root_size, -int(lvl) - 1);
                        const f3 oc = sc.root_min + mk3(float(rb.z & 0xffffu) + 0.5f, float(rb.z >> 16) + 0.5f, float(rb.w & 0xffffu) + 0.5f) * voxel;
                        n = hit_normal(o, d, time, oc);
                    }
                    finished = false;
                    have = true;
                }
            }
            if (finished) {
                f3 outc = sample / float(ambient_rays);                                   // voxels.comp:391
                a.out_color[pix] = make_float4(outc.x, outc.y, outc.z, 1.0f);
            }
        }

        Shaded sh{};
        if (have) {
            sh = shade_hit(a, stage, hit_pos, dir, n, node, sample, blend, ambient_rays, rng, sun_dir, sun_color);
            if (sh.flags == 0u) {  // nothing left to trace (specular hit at the last segment): the pixel is done
                f3 outc = sh.sample / float(sh.ambient_rays);
                a.out_color[pix] = make_float4(outc.x, outc.y, outc.z, 1.0f);
            } else {
                keep = true;
            }
        }
        const unsigned slot = dense_append(q, stage, keep, lds_counts, tid);
        if (keep) {
            float4* so = state_out + size_t(slot) * 4;
            so[0] = make_float4(sh.sample.x, sh.sample.y, sh.sample.z, __uint_as_float(rng.index));
            so[1] = make_float4(sh.blend.x, sh.blend.y, sh.blend.z, __uint_as_float(pix));
            so[2] = make_float4(sh.pend_sun.x, sh.pend_sun.y, sh.pend_sun.z, __uint_as_float(sh.ambient_rays));
            so[3] = make_float4(sh.pend_emit.x, sh.pend_emit.y, sh.pend_emit.z, 0.0f);
            // 1 / d is made here, where every lane has a ray; a ray with a zero or NaN direction component (~1e-7 of them) is
            // walked on the spot by the shader's literal text, so that the ray pool holds regular rays only
            unsigned flags = sh.flags;
            f3 si = splat3(0.0f), bi = splat3(0.0f);
#pragma unroll
            for (unsigned which = 0; which < 2u; which++) {
                if ((flags & (which ? kFlagBounce : kFlagSun)) == 0u) continue;
                const f3 d = which ? sh.bounce_dir : sh.sun_dir;
                const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                if (which) bi = inv; else si = inv;
                if (!ray_is_regular(inv)) {
                    Walk g;
                    int status = kWalkMiss;
                    uint4 r = make_uint4(0u, 0u, 0u, unsigned(kWalkMiss) << 16);
                    if (walk_begin(g, sc, sh.origin, d)) {
                        do { status = walk_step(g, sc, kAlmostInfinity, lds_stack + tid); } while (status == kWalkOn);
                        const unsigned vx = (g.ix << 1) | ((g.octant >> 2) & 1u), vy = (g.iy << 1) | ((g.octant >> 1) & 1u), vz = (g.iz << 1) | (g.octant & 1u);
                        r = make_uint4(__float_as_uint(g.time), status == kWalkLeaf ? walk_leaf_index(g) : 0u, vx | vy << 16,
                                       vz | unsigned(status) << 16 | g.lvl << 20);
                    }
                    results_out[size_t(slot) * 2 + which] = r;
                    flags |= which ? kFlagBounceTraced : kFlagSunTraced;
                    rays_cast++;
                }
            }
            float4* ro = rays_out + size_t(slot) * 4;
            ro[0] = make_float4(sh.origin.x, sh.origin.y, sh.origin.z, __uint_as_float(flags));
            ro[1] = make_float4(sh.sun_dir.x, sh.sun_dir.y, sh.sun_dir.z, si.x);
            ro[2] = make_float4(si.y, si.z, sh.bounce_dir.x, sh.bounce_dir.y);
            ro[3] = make_float4(sh.bounce_dir.z, bi.x, bi.y, bi.z);
        }
    }
    count_rays(a.ray_counter, rays_cast, lane);
}

}  // namespace

hipError_t launch_trace_wavefront(const TraceArgs& a, const PathQueue queues[2], unsigned* count_sets[3], unsigned* launch_counter,
                                  int blocks, unsigned split_mask, hipStream_t s) {
    dim3 grid((a.band.width + 15) / 16, (a.band.local_rows + 15) / 16);
    size_t lds = size_t(a.stack_levels) * kBlock * sizeof(uint2);
    unsigned J = *launch_counter;
    PathQueue out = queues[0];
    out.counts = count_sets[(J + 1) % 3];
    hipLaunchKernelGGL(primary_kernel, grid, dim3(kBlock), lds, s, a, out, count_sets[(J + 2) % 3]);
    *launch_counter = J + 1;
    return launch_bounces(a, false, queues, count_sets, launch_counter, blocks, split_mask, 0, s);
}


hipError_t launch_trace_rayqueue(const TraceArgs& a, const PathQueue& hits_in, unsigned* count_sets[3], unsigned* launch_counter,
                                 const RayQueue& q, int shade_blocks, int trace_blocks, unsigned min_rays_per_wave, hipStream_t s) {
    dim3 grid((a.band.width + 15) / 16, (a.band.local_rows + 15) / 16);
    const size_t lds = size_t(a.stack_levels) * kBlock * sizeof(uint2);
    hipError_t e = hipMemsetAsync(q.counts, 0, size_t(a.max_bounces + 1) * (kSegments + 1) * kCountStride * sizeof(unsigned), s);
    if (e != hipSuccess) return e;
    // the sharded hit queue's counter sets rotate as in launch_trace_wavefront: launch J writes set (J+1)%3,
    // the consumer (launch J+1) reads it and clears set J%3
    unsigned J = *launch_counter;
    PathQueue hits = hits_in;
    hits.counts = count_sets[(J + 1) % 3];
    hipLaunchKernelGGL(primary_kernel, grid, dim3(kBlock), lds, s, a, hits, count_sets[(J + 2) % 3]);
    hipLaunchKernelGGL(shade_kernel<true>, dim3(shade_blocks), dim3(kBlock), lds, s, a, hits, q, count_sets[J % 3], 0);
    *launch_counter = J + 2;
    for (int stage = 0; stage < a.max_bounces; stage++) {
        if (hipError_t pe = launch_pool_rays(a, q, stage, trace_blocks * 4, s); pe != hipSuccess) return pe;
        hipLaunchKernelGGL(shade_kernel<false>, dim3(shade_blocks), dim3(kBlock), lds, s, a, hits, q, nullptr, stage + 1);
    }
    return hipGetLastError();
}
}  // namespace vxrt
