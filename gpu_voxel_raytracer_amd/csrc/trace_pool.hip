// trace_pool.hip — the ray pool of the ray-queue tracer (gfx950): persistent waves that trace the rays of one stage and hand
// every lane a new ray when its own has ended.  One wave per block, a 64-column LDS stack.
#define VXRT_STACK_STRIDE 64
#include "trace_common.h"
#include "ray_queue.h"

namespace vxrt {
namespace {

#ifndef VXRT_POOL_WAVES
#define VXRT_POOL_WAVES 7     // waves per SIMD the kernel is compiled for (72 VGPRs: no spill)
#endif
#ifndef VXRT_POOL_GATE
#define VXRT_POOL_GATE 24     // refill when this many lanes of the wave are idle
#endif
constexpr int kGate = VXRT_POOL_GATE;

// What a walk ended with:  x = bits(time), y = leaf index, z = voxel x | y << 16, w = voxel z | status << 16 | node level << 20.
__device__ __forceinline__ uint4 pack_result(float time, uint32_t leaf, uint32_t ix, uint32_t iy, uint32_t iz, uint32_t octant, uint32_t lvl,
                                             int status) {
    const unsigned vx = (ix << 1) | ((octant >> 2) & 1u), vy = (iy << 1) | ((octant >> 1) & 1u), vz = (iz << 1) | (octant & 1u);
    return make_uint4(__float_as_uint(time), status == kWalkLeaf ? leaf : 0u, vx | vy << 16, vz | unsigned(status) << 16 | lvl << 20);
}

// Rays [0, N) of a stage are its paths' sun rays (all towards the sun, origins in tile order: coherent), rays [N, 2N) their bounce
// rays (random directions).  A wave takes `grab` consecutive rays at a time from the stage's cursor (one atomic), hands them to its
// lanes as they go idle — when kGate lanes wait, or nothing else is left to do — and every lane walks its ray one octree step per trip
// (walkf_step: the regular-ray walk).  A finished lane keeps its end state in its registers; the result is stored when the lane is
// refilled, so that the store is not a divergent region of every trip.
__global__ __launch_bounds__(64, VXRT_POOL_WAVES) void pool_rays_kernel(const TraceArgs a, const RayQueue q, int stage) {
    extern __shared__ uint2 lds_stack[];
    const int lane = threadIdx.x;
    const SceneView sc = make_scene(a);
    uint2* stack = lds_stack + lane;
    const SegTable seg = load_segments(q.counts, stage);
    const unsigned n_paths = seg.pre[kSegments];
    const unsigned total_rays = n_paths * 2u;
    unsigned grab = (total_rays / (gridDim.x * 4u) + 63u) & ~63u;
    grab = grab < 64u ? 64u : (grab > 1024u ? 1024u : grab);
    unsigned* cursor = stage_cursor(q, a.max_bounces, stage);
    const float4* rays = q.rays[stage & 1];
    uint4* results = q.results[stage & 1];

    WalkF w;
    bool active = false, pending = false, more = true;
    int end_status = kWalkMiss;
    unsigned res_slot = 0, next = 0, end = 0;
    uint32_t rays_cast = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        const int n_idle = __popcll(idle);
        if ((more && n_idle >= kGate) || n_idle == 64) {
            if (pending) {
                const uint32_t bit = 1u << w.octant;
                results[res_slot] = pack_result(w.time, w.rec.base + __popc((w.rec.masks >> 8) & (bit - 1u)), w.ix, w.iy, w.iz, w.octant, w.lvl, end_status);
                pending = false;
            }
            if (!more) break;
            if (next == end) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(cursor, grab);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= total_rays) { more = false; continue; }
                next = base;
                end = base + grab < total_rays ? base + grab : total_rays;
            }
            const unsigned rank = __builtin_amdgcn_mbcnt_hi(unsigned(idle >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(idle), 0u));
            const unsigned left = end - next;
            const unsigned take = unsigned(n_idle) < left ? unsigned(n_idle) : left;
            if (!active && rank < take) {
                const unsigned g = next + rank;
                const unsigned which = g >= n_paths ? 1u : 0u;
                const unsigned slot = segment_slot(seg, which ? g - n_paths : g, q.seg_capacity);
                const float4* ray = rays + size_t(slot) * 4;
                const float4 r0 = ray[0], r2 = ray[2];
                const unsigned flags = __float_as_uint(r0.w);
                res_slot = slot * 2u + which;
                if ((flags & (which ? kFlagBounce : kFlagSun)) != 0u && (flags & (which ? kFlagBounceTraced : kFlagSunTraced)) == 0u) {
                    f3 d, inv;
                    if (which) {
                        const float4 r3 = ray[3];
                        d = mk3(r2.z, r2.w, r3.x);
                        inv = mk3(r3.y, r3.z, r3.w);
                    } else {
                        const float4 r1 = ray[1];
                        d = xyz4(r1);
                        inv = mk3(r1.w, r2.x, r2.y);
                    }
                    rays_cast++;
                    if (walkf_begin(w, sc, xyz4(r0), d, inv)) active = true;
                    else results[res_slot] = make_uint4(0u, 0u, 0u, unsigned(kWalkMiss) << 16);
                }
            }
            next += take;
        }
        if (active) {
            const int status = walkf_step(w, sc, stack);
            if (status != kWalkOn) {
                end_status = status;
                active = false;
                pending = true;
            }
        }
    }
    count_rays(a.ray_counter, rays_cast, lane);
}

}  // namespace

hipError_t launch_pool_rays(const TraceArgs& a, const RayQueue& q, int stage, int waves, hipStream_t s) {
    const size_t lds = size_t(a.stack_levels) * 64 * sizeof(uint2);
    hipLaunchKernelGGL(pool_rays_kernel, dim3(waves), dim3(64), lds, s, a, q, stage);
    return hipGetLastError();
}
}  // namespace vxrt
