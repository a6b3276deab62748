/*
 * vxrt_debug.h — test hooks, diagnostics and the scheduling options of experiments of libvxrt.so.  NOT part of the contract
 * (vxrt.h): nothing here is needed to render, none of it changes what a frame means, and a binding for the reference's host can
 * leave this header out.  The tests, bench.py's probes and the A/B scripts are its users.
 */
#ifndef VXRT_DEBUG_H
#define VXRT_DEBUG_H

#include "vxrt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Scheduling options for experiments and tests, continuing vxrt_option from 7 (same image whatever they say; the library reads NO
 * environment variable — this is the only way in).  "create": accepted by vxrt_create_tuned only, because a launch in flight must
 * not see them change or because they size what the context allocates.
 *   VXRT_OPT_TILE_ORDER    1 (default): the tiles of a trace launch start longest first (csrc/trace.hip); 0: raster order.
 *   VXRT_OPT_TILE_SPREAD   how far the tiles that walk are spread between the sky tiles of a launch: 0 .. 256 = that many 256ths of
 *                          the launch, VXRT_TILE_SPREAD_AUTO (default) = decided on the device from the cost histogram.
 *   VXRT_OPT_TRACE_BLOCKS  blocks of the compacted tail's launches (default 2048).
 *   VXRT_OPT_TAIL_FROM     create: the hit at which a path moves from trace_kernel to the compacted tail (default 1 = the second hit).
 *   VXRT_OPT_TAIL_SPLIT    create: bit k = the tail compacts once more at path segment k (default: 0x2c — segments 2, 3 and 5 — from 6 bounces on).
 *   VXRT_OPT_HOST_SCENE_BUILD  1: vxrt_set_menger builds the scene on the host also where the device builder could (cross-check).
 *   VXRT_OPT_NODE_ORDER    the order of the scene's 8-byte records in memory, chosen before the scene is set: 0 (default) breadth-first,
 *                          level after level; 2 / 3: the last two / three node levels as depth-first treelets — below every node of
 *                          level depth - 2 / depth - 3 its children as one block, then child by child their children's blocks
 *                          (<= 576 bytes / <= 4.6 KB) — so that the end of a descent stays in one neighbourhood of memory (BASELINE
 *                          config 5's scene lives in HBM).  Node indices never reach an output and the children of a node stay
 *                          contiguous: same image, same walk code.  vxrt_stats.node_order reads what the scene in place has.
 *   VXRT_OPT_HEAD_STAGGER  1: with several trace launches in flight (one stream each), a launch's trace_kernel starts only when the previous
 *                          launch's trace_kernel has finished, so that it runs beside that launch's bounce_kernel (a short block of few
 *                          launches: the heads do not drain together).  Measured slower (-DVXRT_VARIANTS=1 builds only).  0 (default).
 *   VXRT_OPT_FUSED_TAIL    1: head and compacted tail of a trace launch run as ONE grid of persistent waves that take the launch's tiles
 *                          from a cursor and then its queued paths, chunk by chunk as they become complete — for launches that are
 *                          little more than their longest chains (a rank's share of a short block on many GPUs); 4-bounce tails,
 *                          8-byte records, scenes in cache.  Same image — and measured SLOWER (a scheduler in software pays for every
 *                          decision with device-scope memory round trips): -DVXRT_VARIANTS=1 builds only.  0 (default): trace_kernel,
 *                          then bounce_kernel.
 *   VXRT_OPT_LONG_TILES    per mille (0 = off .. 500) of a trace launch's tiles — the first of its longest-first order — that run as an
 *                          all-in-one grid on a second stream (a path's whole chain in one wave, begun when the launch begins) beside
 *                          the head + compacted tail of the others: for launches that are little more than their longest chains
 *                          (a rank's share of a short block on many GPUs).  Same image; measured slower (-DVXRT_VARIANTS=1 builds only).
 *   VXRT_OPT_TRACER_OVERRIDE   create: the internal schedule past vxrt_config.tracer's automatic choice: 0 all-in-one kernel, 4 head +
 *                          compacted tail; 2 / 3 / 5 (-DVXRT_VARIANTS=1 builds) wavefront, ray queues, path kernel.
 *   VXRT_OPT_TRACE_SPLIT / PATH_BLOCKS / SHADE_BLOCKS / RAYS_PER_WAVE   create: launch shapes of tracers 2, 3 and 5.
 *   VXRT_OPT_TRACE_PRIORITY  create: 1 = the trace streams are made with the device's HIGHEST stream priority and a launch's tiles go out
 *                          as TWO grids — the tiles that walk on the high-priority stream, the tiles that only store sky on a
 *                          low-priority stream of their own — so that the dispatcher prefers the long chains whenever both have blocks
 *                          waiting (round 6's experiment for a rank's share of a short block on many GPUs; needs a tile order, i.e. the
 *                          second launch of a stream on; head + compacted tail or the all-in-one kernel, 8-byte records).  Measured 3 % slower:
 *                          -DVXRT_VARIANTS=1 builds only.  0 (default).
 *   VXRT_OPT_XCD_AFFINITY  S = 1 .. 64: for a scene beyond the Infinity Cache (BASELINE config 5) and one frame per launch, the launch order
 *                          deals the tiles that walk to the 8 XCDs by screen region (super-tiles of S x S tiles, balanced by measured
 *                          cost, longest first inside each XCD's list) instead of round robin, so that an XCD's L2 serves one region's
 *                          part of the tree (csrc/api_trace.hip: xcd_affine_order; made on the host: an experiment; fetch - 20 %, time - 2 %:
 *                          -DVXRT_VARIANTS=1 builds only).  0 (default): off.  */
#define VXRT_OPT_TILE_ORDER ((vxrt_option)7)
#define VXRT_OPT_TILE_SPREAD ((vxrt_option)8)
#define VXRT_OPT_TRACE_BLOCKS ((vxrt_option)9)
#define VXRT_OPT_TAIL_FROM ((vxrt_option)10)
#define VXRT_OPT_TAIL_SPLIT ((vxrt_option)11)
#define VXRT_OPT_HOST_SCENE_BUILD ((vxrt_option)12)
#define VXRT_OPT_TRACER_OVERRIDE ((vxrt_option)13)
#define VXRT_OPT_TRACE_SPLIT ((vxrt_option)14)
#define VXRT_OPT_PATH_BLOCKS ((vxrt_option)15)
#define VXRT_OPT_SHADE_BLOCKS ((vxrt_option)16)
#define VXRT_OPT_RAYS_PER_WAVE ((vxrt_option)17)
#define VXRT_OPT_NODE_ORDER ((vxrt_option)18)
#define VXRT_OPT_HEAD_STAGGER ((vxrt_option)19)
#define VXRT_OPT_LONG_TILES ((vxrt_option)20)
#define VXRT_OPT_FUSED_TAIL ((vxrt_option)21)
#define VXRT_OPT_TRACE_PRIORITY ((vxrt_option)22)
#define VXRT_OPT_XCD_AFFINITY ((vxrt_option)23)
#define VXRT_TILE_SPREAD_AUTO 0xffffffffu
/* An (option, value) pair for vxrt_create_tuned. */
typedef struct vxrt_tuning { uint32_t option; uint32_t value; } vxrt_tuning;
/* vxrt_create with `count` options applied before anything is allocated (vxrt_create = none).  The reference has no counterpart: its
 * tuning is compile-time constants in the shaders; this is where an A/B script or a test says what it wants instead of the process
 * environment. */
int vxrt_create_tuned(const vxrt_config* cfg, const vxrt_tuning* tuning, size_t count, vxrt_ctx** out);

/* Test hook: the next render uses frame_number + 1 (the library owns the counter, src/context.rs:2152; tests replay a given frame). */
int vxrt_set_frame_number(vxrt_ctx* ctx, uint32_t frame_number);

/* = vxrt_halo_info.message_bytes (kept from ABI 3). */
int vxrt_halo_bytes(vxrt_ctx* ctx, size_t* bytes_per_neighbour);
/* Synchronous forms of vxrt_halo_pack / vxrt_halo_unpack: the launch, and a wait for it (the buffers are borrowed for the call only).
 * What single-process tests use to hand halos between contexts by pointer. */
int vxrt_halo_export(vxrt_ctx* ctx, void* dev_to_prev, void* dev_to_next);
int vxrt_halo_import(vxrt_ctx* ctx, const void* dev_from_prev, const void* dev_from_next);

/* Test hook: the scene as the device holds it right now (8-byte records, 2 words each; leaf words).  Null arrays: sizes only. */
int vxrt_debug_read_scene(vxrt_ctx* ctx, uint32_t* svo, size_t svo_cap, size_t* n_svo, int32_t* leaves, size_t leaf_cap, size_t* n_leaves);
/* The device scene formats for a voxel list (csrc/kernels.h: SvoRecord = 2 words, WideRec = 4 words per record; leaf words as in
 * src/context.rs:732-735), built on the host exactly as vxrt_set_voxels builds them: for tools and tests.  Null arrays: sizes only. */
int vxrt_build_records(const int16_t (*pos)[3], const uint8_t (*mrgb)[4], size_t n, uint32_t* svo, size_t svo_cap, size_t* n_svo,
                       uint32_t* wide, size_t wide_cap, size_t* n_wide, int32_t* leaves, size_t leaf_cap, size_t* n_leaves, uint32_t* depth);

/* Test hook: evaluates function `fn` of include/vxrt_detmath.h on the device for host arrays x, y
 * (0 sin, 1 cos, 2 exp, 3 log, 4 pow, 5 sqrt, 6 div, 7 tan, 8 normalize/cross/dot chain) so that the
 * host-vs-device bit equality the numeric contract promises can be checked. */
int vxrt_detmath_probe(int32_t device, int32_t fn, const float* x, const float* y, float* out, size_t n);

/* Test hook: cast_bounded_ray (shaders/voxels.comp:134-247, max_distance 2^30) as the kernels implement it, for n caller-given
 * rays (origins, dirs: 3 floats each) through the current scene: hit flag, time, leaf word, normal (3 floats) per ray. */
int vxrt_debug_cast_rays(vxrt_ctx* ctx, const float* origins, const float* dirs, size_t n, uint8_t* hit, float* time, int32_t* node,
                         float* normal);

/* Test hook: the path of pixel (x, y) of the NEXT frame (frame_number + 1, camera as set), cast by cast, as the kernels compute it.
 * log: room for 32 casts of 12 floats = origin, direction, hit flag, time, bits(leaf word), normal; *casts = how many were made.
 * Renders nothing and leaves the context as it was. */
int vxrt_debug_path_log(vxrt_ctx* ctx, int32_t x, int32_t y, float* log, int32_t* casts);

/* Diagnostics: duration (shader clocks) of each 16x16 tile of the last traced frame, row-major over
 * ceil(width/16) x ceil(local_rows/16) tiles — the data the longest-tile-first scheduler works from.  A tile none of
 * whose pixels walked the octree (sky, culled) reads 1; a tile that walked, its longest wave's duration (>= 4). */
int vxrt_debug_tile_costs(vxrt_ctx* ctx, uint32_t* out, size_t n);

/* Diagnostics: the launch order of the tracer's 8x8-pixel tiles as the last sort made it (order[k] = tile index, row-major over
 * ceil(width/8) x ceil(local_rows/8)), the per-tile costs it was made from, the number of tiles that walked the octree and how far
 * they were spread over the launch, in 1/256 (0 = plain longest-first; csrc/trace.hip: tile_scatter_kernel).  order and cost may
 * be null; n = the tile count.  VXRT_E_INVALID before a stream's first sort. */
int vxrt_debug_tile_order(vxrt_ctx* ctx, uint32_t* order, uint32_t* cost, size_t n, uint32_t* walking_tiles, uint32_t* spread_256);

/* Diagnostics: of this rank's pixels, how many primary rays of the next frame (camera as set) VXRT_OPT_SKY_CULL decides by its box
 * test instead of walking the octree.  They are counted in vxrt_stats.rays (each is one cast_bounded_ray of voxels.comp:134-247,
 * answered without a walk); for a camera at rest, rays per frame minus this = the rays that walked. */
int vxrt_debug_culled_pixels(vxrt_ctx* ctx, uint64_t* count);
/* Diagnostics of trace stream 0's last fused launch (VXRT_OPT_FUSED_TAIL): shader clocks to "every head block finished" and to the last
 * wave's end, chunks taken before / after that moment, idle sleeps, stamp polls, head claims, 0. */
int vxrt_debug_fused_profile(vxrt_ctx* ctx, uint64_t out[8]);

/* Touch map (-DVXRT_VARIANTS=1 builds only; VXRT_E_INVALID otherwise): which bytes of the scene does a frame touch?  enable = 1
 * allocates (for the scene in place) and clears two bitmaps, one bit per 64-byte line of the node records the tracer walks (8-byte
 * or wide) and of the leaf words; from then on every scene load of trace_kernel / bounce_kernel marks its line (slower: never time
 * such a frame).  enable = 0 frees them.  vxrt_debug_touch_count: lines marked so far — out[0] node lines of 64 bytes, out[1] leaf
 * lines of 64 bytes, out[2] / out[3] the same at 128-byte granularity (the L2's line), out[4] / out[5] the lines the two arrays
 * have (64-byte) — and, with reset != 0, clears the maps.  SURVEY 8d's "bricks actually touched": unique scene bytes per frame. */
int vxrt_debug_touch_map(vxrt_ctx* ctx, uint32_t enable);
int vxrt_debug_touch_count(vxrt_ctx* ctx, uint64_t out[6], uint32_t reset);

/* What this build of the library contains: VXRT_FEATURE_VARIANTS = it was compiled with -DVXRT_VARIANTS=1 and also holds the
 * schedules and the scene format that measured slower on MI355X and are kept for comparison (tracers 2, 3, 5; the wide scene
 * records).  The default build holds tracers 1 and 4 over the 8-byte records and refuses the others with VXRT_E_INVALID. */
enum { VXRT_FEATURE_VARIANTS = 1 };
uint32_t vxrt_build_features(void);

#ifdef __cplusplus
}
#endif
#endif /* VXRT_DEBUG_H */
