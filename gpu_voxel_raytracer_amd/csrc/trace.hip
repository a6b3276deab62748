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
#include "trace_block.h"

namespace vxrt {
namespace {


template <bool kWide, int kWaves, int kF>
__global__ __launch_bounds__(kTB, kWaves) void trace_kernel(const TraceArgs a) {
    extern __shared__ uint4 lds_stack[];  // the threads' frames: Caster<kWide>
    // a launch may come as two grids (the longest tiles apart: VXRT_OPT_LONG_TILES)
    trace_block<kWide, kF, false>(a, blockIdx.x + a.block_first, lds_stack, 0u);
}



// Counting sort of the tiles by descending cost (key = log2 of the cost with two mantissa bits: 128 bins) in three small launches:
// per-block histograms over contiguous tile ranges (coalesced reads) -> one block turns them into (key, block) offsets -> each block
// scatters its tiles.  (One 1024-thread block doing all of it took 0.46 ms for the 129 600 tiles of a 4K frame — on the trace
// stream's critical path whenever a single frame is rendered per call.)
// (kSortBlocks = 64, kSortBins = 128: kernels.h; block_hist is [bin][block])

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
