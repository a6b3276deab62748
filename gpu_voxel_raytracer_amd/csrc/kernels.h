// kernels.h — device-side argument blocks and launchers of libvxrt (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// The default library holds the two schedules that won on MI355X — tracer 1 (all-in-one kernel) and tracer 4 (head + compacted tail) —
// over the 8-byte scene records.  -DVXRT_VARIANTS=1 also builds what was measured slower and is kept for comparison:
// tracers 2 (wavefront), 3 (ray queues), 5 (per-lane path refill) and the wide scene records (two tree levels per 16-byte record).
// Every variant gives bit-identical images (scripts/test_variants.sh runs the parity tests over that build).
#ifndef VXRT_VARIANTS
#define VXRT_VARIANTS 0
#endif

namespace vxrt {

// Compact sparse voxel octree, the device scene format ("SVO record"): the same tree as the
// reference's 8-ints-per-node buffer (shaders/voxels.comp:58-63, built by src/context.rs:710-773) but
// 8 bytes per node instead of 32, in breadth-first order with the children of a node contiguous:
//   .x = child_mask (bits 0-7: slot holds a child node) | leaf_mask << 8 (bits 8-15: slot holds a leaf)
//   .y = index of the node's first child record (or of its first leaf word, for a node of leaves)
// Slot s of a node is child  base + popcount(child_mask & ((1<<s)-1)).  A traversal step that stays
// inside a node (sibling advance) needs no memory access at all; a descend is one 8-byte load.
struct SvoRecord {
    uint32_t masks;
    uint32_t base;
};

// The same tree, TWO LEVELS PER RECORD ("wide" records, 16 bytes): a node of an even-numbered pairing level (the record's "top")
// together with its up to eight children (its "subs").  Levels are paired from the bottom, so the leaf parents are always subs —
// the most numerous nodes of a tree never exist as records of their own — and when the number of node levels is odd the root is
// the only sub of a virtual top above it.
//   .mlo/.mhi  64 bits: byte s = the 8-bit occupancy mask of sub s (children, or leaves when the subs are the leaf parents);
//              byte s == 0 <=> slot s of the top is empty
//   .base      index of the record's first grandchild (a wide record two levels down), or of its first leaf word
//   .top       bits 0-7: slot s of the top is occupied (= byte s != 0), kept so that no lane has to derive it
// Grandchild (s, o) is  base + popcount(mask64 & ((1 << (8 s + o)) - 1)).  A descent top -> sub reads no memory at all and a
// descent sub -> child one 16-byte record: half the dependent loads of the 8-byte format, and for a scene that lives in HBM
// (BASELINE config 5) a third of the bytes — the leaf parents' records, 85 % of that tree's nodes, are gone.
struct WideRec {
    uint32_t mlo, mhi, base, top;
};

constexpr unsigned kSortBlocks = 64, kSortBins = 128;   // the tile sort (trace.hip): its scratch is [bin][block] + {walking tiles, spread}
constexpr unsigned kRaySlots = 2048;  // ray counters, one per 64-byte line (TraceArgs::ray_counter)

struct Cam {  // first 64 bytes of Uniforms, without padding
    float o[3], r[3], u[3], f[3];
};

// Which rows of the frame this context owns (vxrt_config.rank / nranks / band_rows).  The rows are dealt to the ranks in interleaved
// bands, band gb -> rank gb % nranks.  Whole ROUNDS of nranks bands are band_rows rows high (`full_bands` bands, rows [0, tail_y0)).
// When the frame is not a whole number of rounds, the LAST round takes the remainder as well: its bands are `tail_rows` rows high, the
// smallest multiple of the tile height (16 rows when band_rows is a multiple of 16, else 8) that covers the last round + remainder in
// nranks bands — band_rows <= tail_rows < 2 band_rows + tile.  The BUSIEST rank's row count is within one tile row of an even share
// (height / nranks); the last rank(s) of that taller round, clipped by the frame's edge, may own up to tail_rows fewer.  No
// band but the frame's very last (clipped by the frame's edge) is lower than band_rows, so a halo can be up to band_rows rows deep at
// every band edge (round 5; round 4 put the remainder into an EXTRA round of LOWER bands, same balance — 272 / 256 rows at 2160 rows on
// 8 ranks with 64-row bands either way — but the halo of the whole frame was capped at those bands' 16 rows: ADVICE r4).  Only a frame
// lower than one round (height < nranks * band_rows) has nothing to fold into: its single round's bands are lower than band_rows.
struct BandMap {
    int width, height, local_rows, band_rows, rank, nranks;
    int full_bands;   // bands of band_rows rows (a multiple of nranks)
    int tail_y0;      // = full_bands * band_rows: where the last, taller round starts (== height: every round is band_rows high)
    int tail_rows;    // height of that round's bands, a multiple of the tile height (>= band_rows unless it is the frame's only round)
};
#define VX_BAND_FN __host__ __device__ __forceinline__
VX_BAND_FN int band_of_row(const BandMap& b, int y) { return y < b.tail_y0 ? y / b.band_rows : b.full_bands + (y - b.tail_y0) / b.tail_rows; }
VX_BAND_FN int band_first_row(const BandMap& b, int gb) { return gb < b.full_bands ? gb * b.band_rows : b.tail_y0 + (gb - b.full_bands) * b.tail_rows; }
VX_BAND_FN int band_nominal_rows(const BandMap& b, int gb) { return gb < b.full_bands ? b.band_rows : b.tail_rows; }    // before the frame's edge clips it
VX_BAND_FN int band_count(const BandMap& b) { return b.full_bands + (b.height - b.tail_y0 + b.tail_rows - 1) / b.tail_rows; }
// local band lb of a rank (its lb-th band): the first local row; and the local band a local row lies in
VX_BAND_FN int local_band_first_row(const BandMap& b, int lb) { const int r = b.full_bands / b.nranks; return lb < r ? lb * b.band_rows : r * b.band_rows; }
VX_BAND_FN int local_band_of(const BandMap& b, int lrow) { const int r = b.full_bands / b.nranks; return lrow < r * b.band_rows ? lrow / b.band_rows : r; }
// frame row -> local row of this context, or -1 when another rank owns it; and back
VX_BAND_FN int local_row(const BandMap& b, int y) {
    if (b.nranks == 1) return y;               // one context: the identity (a uniform branch past two integer divisions)
    const int gb = band_of_row(b, y);
    if (gb % b.nranks != b.rank) return -1;
    return local_band_first_row(b, gb / b.nranks) + (y - band_first_row(b, gb));
}
VX_BAND_FN int frame_row(const BandMap& b, int lrow) {
    if (b.nranks == 1) return lrow;
    const int lb = local_band_of(b, lrow);
    return band_first_row(b, lb * b.nranks + b.rank) + (lrow - local_band_first_row(b, lb));
}

// A queue of 64-byte path records in 64 shards (trace_common.h: PathRec, queue_append).
struct PathQueue {
    float4* recs;             // [64 shards][shard_capacity][4 float4]
    unsigned* counts;         // 64 counters, 16 uints (one 64-byte line) apart
    unsigned shard_capacity;  // records per shard
};

constexpr int kMaxBatch = 32;      // frames one trace launch can cover (vxrt_config.frames_per_launch)
constexpr uint32_t kPixBits = 27;  // PathRec::pix = local pixel index (< 2^27) | frame-in-batch << 27
struct FrameOut {                  // the three voxels.comp outputs of one frame (a ring slot of the context)
    float4* color;
    float4* nd;
    float4* albedo;
};

struct TraceArgs {
    const WideRec* wide;   // the scene as wide records (null: only the 8-byte format was built) — trace_kernel<true> / bounce_kernel<true>
    WideRec wide_root;     // wide[0]
    int node_levels;       // node levels of the tree (octree depth + 1): the root is level 0, the leaf parents level node_levels - 1
    const SvoRecord* svo;
    SvoRecord root_rec;    // svo[0]: every cast begins with it, so it travels with the kernel arguments instead of being loaded
    const int32_t* leaves;
    const float* noise;
    float4* out_color;   // = out[0].*: the single-frame kernels (tracer 2 / 3) use these
    float4* out_nd;
    float4* out_albedo;
    FrameOut out[kMaxBatch];  // trace_kernel / bounce_kernel: frame f of the launch's batch (frame_number + f) writes out[f]
    int batch;                // frames in this launch, 1..kMaxBatch: block b renders frame b % batch of tile order[b / batch]
    unsigned long long* ray_counter;  // kRaySlots counters, 8 words apart
    const uint32_t* tile_order;       // monolithic kernel: block b renders 16x16 tile tile_order[b] (null: b)
    uint32_t* tile_cost;              // ... and records how long the tile took (shader clocks, max over its waves)
    float root_center[3];
    float root_size;
    BandMap band;
    int max_bounces;
    uint32_t frame_number;
    uint32_t launch_index;  // counts trace launches of the context (persistent kernel: which tile counter to use)
    uint32_t block_first;   // trace_kernel: this grid's block 0 is block `block_first` of the launch (a launch in two grids: VXRT_OPT_LONG_TILES)
    int stack_levels;  // LDS stack entries per thread (= octree depth, >= 1)
    Cam cam;                 // = cams[0]: the single-frame kernels use this
    Cam cams[kMaxBatch];     // trace_kernel: frame f of the launch is seen through cams[f] (all equal for a camera at rest)
    // per-frame constants hoisted from voxels.comp main() (identical for every pixel)
    float sun_dir[3];        // voxels.comp:296
    float sun_dir_n[3];      // normalize(sun_dir)      voxels.comp:347
    float neg_sun_dir_n[3];  // normalize(-sun_dir)     voxels.comp:379
    float sun_color[3];      // SUN_COLOR               voxels.comp:6
    float sky_color[3];
    float sun_exponent;      // 1.0 / pow(sun_size, 2)  voxels.comp:380
    float sun_zero_below;    // pow(x, sun_exponent) is exactly +0 for 0 <= x < this (0: unknown; see sun_power_of)
    float sun_size, sun_strength, emit_strength, specularity;
    // monolithic kernel with a compacted tail (tracer 4): a path that is still alive when it reaches hit number
    // `tail_from` (0 = the first hit; tail.recs == nullptr: off) is appended to `tail` instead of being followed; bounce_kernel finishes those paths
    PathQueue tail;
    unsigned* tail_zero;   // counter set to clear for a later launch (see launch_trace_wavefront)
    int tail_from;
    uint32_t gbuf_frames;  // bit k: frame k of the launch writes its normal/depth and albedo/node images (all set, except for the samples of one
                           // displayed frame, vxrt_render_spp, whose first hits are identical: only the frame that is kept writes them)
    // Sky cull (trace.hip: primary_miss_is_certain; DESIGN.md "sky cull"): a box that holds every occupied cell of the tree at level
    // min(depth, 7), grown by a margin far above the walk's rounding.  A regular primary ray that misses it provably makes the walk
    // return "miss" (no leaf is near it and fewer than 1536 of the 2048 trips are possible), so its pixel takes the miss outputs of
    // voxels.comp:373-388 / :292-294 without walking.  cull = 0: off (no box, odd camera, VXRT_OPT_SKY_CULL 0).
    int frame_lanes;       // trace_kernel: 0, or the frames a wave holds: 8 (of a row of 8 pixels) or 4 (of two rows) — one camera for the launch, whole groups
    int cull;
    float cull_min[3], cull_max[3];
#if VXRT_VARIANTS
    // Touch map (vxrt_debug_touch_map; -DVXRT_VARIANTS=1 only): when set, every scene load of the walk marks the 64-byte line it reads —
    // bit (index * record bytes) >> 6 of touch_nodes for a node record (8-byte or wide), of touch_leaves for a leaf word — so that the
    // unique bytes a frame touches ("bricks actually touched") can be counted.  null: off (every timed run).
    uint32_t* touch_nodes;
    uint32_t* touch_leaves;
#endif
};

// The rows of the neighbouring ranks that this rank can see (multi-rank: api_halo.hip, halo.hip).  Two messages are kept, side 0 =
// the `rows` rows ABOVE each local band (sent by the previous rank), side 1 = the `rows` rows BELOW it (sent by the next rank); a
// message is three planes over (slot = local band, row k, x):  A = (r, g, b, depth)  B = (nx, ny, nz, bits(material id))  C = blending
// factor of the accumulated colour — 36 bytes per pixel: what denoise.comp:51-57's window reads of a neighbour's pixel (colour,
// normal, depth, material) plus what temporal.comp:85-113's reprojection reads of it (colour + blending factor, depth).
// Row k of side 0 is frame row band_y0 - rows + k, row k of side 1 is frame row band_end + k.
struct HaloView {
    const float4* base;   // null: no halo
    int rows, slots;      // rows per band edge; bands a message has room for
    size_t plane;         // float4 per A / B plane = slots * rows * width
    size_t message;       // float4 per message (>= 2 * plane + plane / 4, rounded up)
};

struct TemporalArgs {
    const float4* sampled_color;
    const float4* new_nd;
    const float4* old_color;
    const float4* old_nd;
    float4* new_color;
    BandMap band;
    Cam cam, old_cam;
    float inv[12];  // affine inverse of the old camera matrix (temporal.comp:75-82), rows + translation
    float sample_blending, maximum_blending, blending_distance_cutoff;
    int has_history;
    // multi-rank: the halo unpacked for the previous frame holds the neighbours' rows of exactly this history
    // (accumulated colour and depth, halo.rows rows beyond each band edge); base null: such rows are a disocclusion
    HaloView halo;
    // fused denoise for radius 0 (the reference's default): out = mix(c, albedo * c, albedo_factor) of the blended colour
    const float4* albedo;   // null: no fusion
    float4* denoised;
    float albedo_factor;
};

struct DenoiseArgs {
    const float4* colors;
    const float4* nd;
    const float4* albedo;
    float4* output;
    // rows just outside this context's bands, received from the neighbouring ranks (base may be null)
    HaloView halo;
    // which rows of 16x16 tiles to denoise: null = all of them (blockIdx.y is the tile row); else tile row = tile_rows[blockIdx.y]
    // (api_halo.hip: the tiles whose window stays inside this rank's rows, and the tiles that need the halo)
    const uint16_t* tile_rows;
    uint32_t tile_row_count;
    BandMap band;
    Cam cam;
    uint32_t radius;
    float sigma_distance_2, sigma_range_2, albedo_factor;
    int mode;   // bit 0: 0 exact (bit-identical to the oracle), 1 tolerant (vxrt_set_option VXRT_OPT_DENOISE_MODE); bit 1: the generic kernel
    float wdist[17 * 17];   // filled by launch_denoise: factor_distance of denoise.comp:79 per window offset (dy + r) * (2 r + 1) + (dx + r)
};

// Queue of live paths between two launches of the wavefront tracer (trace.hip): 64-byte records in 64 shards.
// Ray-queue variant (trace_wavefront.hip): paths and rays of one frame between its shade / trace launches.
constexpr unsigned kSegments = 8;  // dense queue segments; a block appends to segment blockIdx % 8 with one atomic
struct RayQueue {
    float4* state[2];        // [kSegments][seg_capacity][4 float4]  path state, ping-pong between segments of a path
    float4* rays[2];         // [kSegments][seg_capacity][4 float4]  (origin, flags) (sun d, 1/d.x) (1/d.yz, bounce d.xy) (d.z, 1/d)
    uint4* results[2];       // [kSegments][seg_capacity][2]         what the walk of each ray ended with, ping-pong like the rays
    unsigned* counts;        // [max_bounces + 1][kSegments] counters, 16 uints (one 64-byte line) apart, zero at frame start
    unsigned seg_capacity;
};

// `wide`: walk the wide records (TraceArgs::wide) instead of the 8-byte ones; same results
// hbm_scene: the kernel compiled for one more wave per SIMD (a scene beyond the Infinity Cache)
// head and compacted tail of a launch as one grid of `waves` persistent waves (trace.hip: fused_kernel); ctl: fused_ctl_bytes() of zeros
size_t fused_ctl_bytes();
// trace_dda.hip (variants build): the DDA prototype's grid, built on the device from the 8-byte records, and its probe kernel
void dda_grid_sizes(int levels, size_t* brick_bytes, size_t* brick_bit_bytes, size_t* super_bit_bytes, size_t* first_leaf_bytes);
hipError_t launch_dda_build(const TraceArgs& a, int levels, void* bricks, void* brick_bits, void* super_bits, void* first_leaf, hipStream_t s);
hipError_t launch_dda_probe(const TraceArgs& a, const void* bricks, const void* brick_bits, const void* super_bits, const void* first_leaf, int levels,
                            const float* origins, const float* dirs, float* out, unsigned n, int certify, float margin_scale, unsigned max_steps, int lds_top,
                            hipStream_t s);
size_t fused_ctl_error_offset();
size_t fused_ctl_profile_offset();
hipError_t launch_fused(const TraceArgs& a, void* ctl, unsigned waves, const uint32_t* sort_scratch, uint32_t stamp, hipStream_t s);
// blocks [block_first, block_first + block_count) of the launch's tiles x frames; block_count 0: all of them
hipError_t launch_trace(const TraceArgs& a, bool wide, bool hbm_scene, hipStream_t s, unsigned block_first = 0, unsigned block_count = 0);
// test hook: the walk (cast_ray) for caller-given rays; out = 8 floats per ray (hit, time, bits(leaf word), normal, 0, 0)
hipError_t launch_path_log(const TraceArgs& a, bool wide, int x, int y, float* log, hipStream_t s);
hipError_t launch_count_culled(const TraceArgs& a, unsigned long long* count, hipStream_t s);   // diagnostics: pixels the sky cull decides
hipError_t launch_cast_probe(const TraceArgs& a, bool wide, const float* origins, const float* dirs, float* out, unsigned n, hipStream_t s);
unsigned trace_tile_count(int width, int local_rows);  // blocks per frame of trace_kernel = entries of a tile schedule
void trace_tile_dims(int* w, int* h);                  // pixels per block  // monolithic: one pixel per lane, all bounces
#if VXRT_VARIANTS
// ray queues: primary_kernel, then shade / trace-rays launches per path segment with per-lane ray refill
// trace_pool.hip: the rays of one stage of the ray queue, `waves` persistent one-wave blocks with per-lane refill
hipError_t launch_pool_rays(const TraceArgs& a, const RayQueue& q, int stage, int waves, hipStream_t s);
hipError_t launch_trace_rayqueue(const TraceArgs& a, const PathQueue& hits, unsigned* count_sets[3], unsigned* launch_counter,
                                 const RayQueue& q, int shade_blocks, int trace_blocks, unsigned min_rays_per_wave, hipStream_t s);
#endif
// longest-tile-first schedule for the next launch: order[] = tiles sorted by descending cost; cost[] is cleared
// scratch: 256 * 128 uint32
// waves_x_launches: waves per tile of a launch x launches that share the chip; wave_slots: waves the chip holds at once;
// spread_override: -1 = the sort decides how far the tiles that walk are spread over the launch, 0 .. 256 = that many 256ths
hipError_t launch_tile_order(uint32_t* cost, uint32_t* order, uint32_t* last_cost, uint32_t* scratch, unsigned tiles, unsigned waves_x_launches,
                             unsigned wave_slots, int spread_override, hipStream_t s);
#if VXRT_VARIANTS
// wavefront: primary_kernel + max_bounces x bounce_kernel; queues[2] ping-pong, count_sets[3] rotate
hipError_t launch_trace_wavefront(const TraceArgs& a, const PathQueue queues[2], unsigned* count_sets[3], unsigned* launch_counter,
                                  int blocks, unsigned split_mask, hipStream_t s);
// tracer 5: path_kernel (trace_paths.hip) follows the paths queued in `in` to their end, refilling each lane with a new path
hipError_t launch_paths(const TraceArgs& a, const PathQueue& in, unsigned* zero, int first_bounce, int blocks, hipStream_t s);
#endif
// bounce_kernel launches for path segments from.. of the paths queued in queues[0] (tracer 2 and the tail of tracer 4)
hipError_t launch_bounces(const TraceArgs& a, bool wide, const PathQueue queues[2], unsigned* count_sets[3], unsigned* launch_counter, int blocks,
                          unsigned split_mask, int from, hipStream_t s);
// "N samples per pixel" (SURVEY.md 8d): sum[p] (+)= frames[0][p] + ... + frames[count-1][p], added in that order;
// first: sum starts from frames[0]; last: out[p] = sum[p] / float(total) is written as well.  One streaming pass.
struct SppArgs {
    const float4* frames[kMaxBatch];
    float4* sum;
    float4* out;
    size_t pixels;
    int count, first, last, total;
};
hipError_t launch_spp_accumulate(const SppArgs& a, hipStream_t s);
// multi-rank halo (halo.hip): ONE pack launch fills both outgoing messages from this rank's band-edge rows, ONE unpack launch copies
// both incoming messages into the context's halo store
struct HaloPackArgs {
    const float4* color;    // accumulated colour (rgb + blending factor)
    const float4* nd;       // normal / depth
    const float4* albedo;   // .w = bits(leaf word): material id = bits >> 24
    float4* to_prev;        // message for rank - 1: this rank's top rows = the rows below that rank's bands
    float4* to_next;        // message for rank + 1: this rank's bottom rows = the rows above that rank's bands
    BandMap band;
    int rows, slots, local_bands;
    size_t plane;
};
hipError_t launch_halo_pack(const HaloPackArgs& a, hipStream_t s);
hipError_t launch_halo_unpack(float4* store, const float4* from_prev, const float4* from_next, size_t message_f4, hipStream_t s);
hipError_t launch_temporal(const TemporalArgs& a, hipStream_t s);
hipError_t launch_denoise(const DenoiseArgs& a, hipStream_t s);
hipError_t launch_noise_fill(float* dst, uint32_t seed, size_t n, hipStream_t s);
// void-and-cluster blue noise (include/vxrt_bluenoise.h): `layers` layers of size x size floats, one block per layer
hipError_t launch_blue_noise(float* dst, uint32_t seed, uint32_t first_layer, uint32_t layers, int size, hipStream_t s);
// detmath probe for the device-vs-host bit-equality test (tests/test_detmath_gpu.py)
hipError_t launch_detmath_probe(int fn, const float* x, const float* y, float* out, size_t n, hipStream_t s);

}  // namespace vxrt
