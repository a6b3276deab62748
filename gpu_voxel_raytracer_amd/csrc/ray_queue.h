// ray_queue.h — addressing of the ray-queue tracer's dense, segmented arrays (trace_wavefront.hip: shade kernels; trace_pool.hip: the
// ray pool).  Included after trace_common.h.
#pragma once
namespace vxrt {
namespace {

struct SegTable {  // the 8 segment counts of one stage as exclusive prefix sums
    unsigned pre[kSegments + 1];
};
__device__ __forceinline__ SegTable load_segments(const unsigned* counts, int stage) {
    SegTable t;
    t.pre[0] = 0;
#pragma unroll
    for (unsigned s = 0; s < kSegments; s++) t.pre[s + 1] = t.pre[s] + counts[(unsigned(stage) * kSegments + s) * kCountStride];
    return t;
}
// dense path index -> slot in the segmented arrays
__device__ __forceinline__ unsigned segment_slot(const SegTable& t, unsigned j, unsigned cap) {
    unsigned s = 0, first = 0;  // select chain with compile-time indices: no runtime-indexed array (that would go to scratch)
#pragma unroll
    for (unsigned k = 1; k < kSegments; k++)
        if (j >= t.pre[k]) { s = k; first = t.pre[k]; }
    return s * cap + (j - first);
}

// the ray pool's cursor of a stage: one word on its own line behind the segment counters (zeroed with them)
__device__ __forceinline__ unsigned* stage_cursor(const RayQueue& q, int max_bounces, int stage) {
    return q.counts + (unsigned(max_bounces + 1) * kSegments + unsigned(stage)) * kCountStride;
}

}  // namespace
}  // namespace vxrt
