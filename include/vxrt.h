/*
 * vxrt.h — C ABI of the MI355X-native voxel path tracer (libvxrt.so).
 *
 * This is the drop-in boundary for the ONE hot path of nolanderc/gpu-voxel-raytracer:
 *   shaders/voxels.comp -> shaders/temporal.comp -> shaders/denoise.comp,
 * plus the scene/camera preparation that feeds them.  The reference has no FFI for this path; its
 * boundary is the set of wgpu calls `Context` makes (src/context.rs).  Each entry point below names
 * the reference call site it replaces, so that a host (the reference's Rust `Context`, or any other)
 * can bind it one-for-one — see INTEGRATION.md for the `extern "C"` block a Rust maintainer would add.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 (VXRT_OK) or a negative
 * vxrt_status; nothing throws or aborts across the boundary; host pointers are borrowed for the
 * duration of the call only; a context owns all of its device memory.  A context is bound to one GPU
 * and is not thread-safe; distinct contexts are independent (the reference renders from a single
 * thread, src/main.rs:34-38).  Multi-GPU = one process and one context per GPU, each rendering its
 * share of the frame's rows (vxrt_config.rank / nranks / band_rows).
 */
#ifndef VXRT_H
#define VXRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vxrt_ctx vxrt_ctx;

typedef enum vxrt_status {
    VXRT_OK = 0,
    VXRT_E_INVALID = -1,      /* bad argument (null pointer, zero size, bad enum, size mismatch)      */
    VXRT_E_DEVICE = -2,       /* HIP runtime error (no device, allocation, launch); see vxrt_last_error */
    VXRT_E_VOX_MAGIC = -10,   /* "invalid magic number"               src/vox.rs:12-14                */
    VXRT_E_VOX_VERSION = -11, /* "unsupported VOX-format: version N"  src/vox.rs:17-19                */
    VXRT_E_VOX_NOMAIN = -12,  /* "missing MAIN chunk"                 src/vox.rs:21                   */
    VXRT_E_VOX_EOF = -13,     /* "unexpected end of file"             src/vox.rs:262-283              */
    VXRT_E_VOX_CHUNK = -14,   /* "expected chunk X, found chunk Y"    src/vox.rs:230-241              */
    VXRT_E_VOX_MATERIAL = -15,/* unsupported _type / unparsable _flux src/vox.rs:82-96                */
    VXRT_E_VOX_NOMATL = -16,  /* colour index without a MATL entry (the reference panics) src/context.rs:919 */
    VXRT_E_VOX_NOMODEL = -17, /* file holds no model                  src/context.rs:916              */
    VXRT_E_IO = -18,          /* "failed to read file"                src/vox.rs:7                    */
    VXRT_E_SCENE = -20,       /* voxel list cannot be represented (leaf split, src/context.rs:746; depth > 15;
                                 too many nodes)                                                       */
    VXRT_E_NOSCENE = -21,     /* render called before any scene was set                               */
    VXRT_E_NOISE = -30        /* "failed to load blue noise"          src/context.rs:1024,1042-1085   */
} vxrt_status;

/* Uniforms — src/context.rs:425-469, std140 layout of shaders/voxels.comp:28-49 (148 bytes, same
 * offsets: a Rust `Uniforms` can be passed as is).  Fields the kernels never read (light,
 * global_time, still_sample) are carried for layout compatibility only.  The camera_* fields and
 * frame_number are overwritten by the library (vxrt_set_camera / vxrt_render), exactly as
 * Context::update_bindings does (src/context.rs:2145-2154). */
typedef struct vxrt_uniforms {
    float camera_origin[4];
    float camera_right[4];
    float camera_up[4];
    float camera_forward[4];
    float light[4];
    float global_time;
    uint32_t still_sample;
    uint32_t frame_number;
    float emit_strength;
    float sun_strength;
    float sun_size;
    float sun_yaw;
    float sun_pitch;
    float sun_color[4];
    float sky_color[4];
    float specularity;
} vxrt_uniforms;

/* TemporalUniforms — src/context.rs:502-515.  Defaults {0.5, 0.98, 1e-2} (:517-525). */
typedef struct vxrt_temporal {
    float sample_blending;
    float maximum_blending;
    float blending_distance_cutoff;
} vxrt_temporal;

/* DenoiseUniforms — src/context.rs:304-314.  Defaults {0, 2.0, 1.5, 1.0} (:316-325); radius 0..8
 * (GUI range, src/context.rs:1791). */
typedef struct vxrt_denoise {
    uint32_t radius;
    float sigma_distance;
    float sigma_range;
    float albedo_factor;
} vxrt_denoise;

typedef struct vxrt_config {
    uint32_t width, height;   /* full frame size in pixels                                              */
    int32_t device;           /* HIP device ordinal                                                     */
    uint32_t max_bounces;     /* MAX_BOUNCES of shaders/voxels.comp:4 (reference: 3); 1..16             */
    uint32_t noise_seed;      /* seed of the generated noise table when `noise` is NULL                 */
    const float* noise;       /* optional 512*128*128 floats in [0,1) (layout of shaders/voxels.comp:65-71) */
    uint32_t rank, nranks;    /* this context renders the row bands b with b % nranks == rank ...       */
    uint32_t band_rows;       /* ... where band b = rows [b*band_rows, (b+1)*band_rows); 0 -> 16.  2, 4 or a multiple of 8
                               * (8 = the tracer's tile height: what launches of single frames want; 2 and 4 suit launches of
                               * frame groups, whose waves hold 1 or 2 rows — the finer the interleave the more alike the ranks'
                               * shares); a multiple of 16 if the denoise stage runs with radius > 0 (its tiles are 16 rows).
                               * That holds for the whole ROUNDS of nranks bands; when the frame is not a whole number of
                               * rounds the LAST round takes the remainder as well, in taller bands — the smallest multiple
                               * of the tile height (16, 8 or band_rows) that covers it in nranks bands — so that every rank
                               * owns within one tile row of height / nranks rows and no band with a band below it is lower
                               * than band_rows.  vxrt_local_rows lists a context's rows; distributed.BandLayout states the rule. */
                              /* nranks = 0 or 1 -> the whole frame                                     */
    uint32_t frames_in_flight;/* 0/1: every stage of a frame runs in submission order on one stream (the
                                 reference's single queue).  F = 2..16: the TRACE stage of up to F consecutive
                                 frames may run concurrently (one HIP stream each, ring of F+2 G-buffer slots);
                                 temporal/denoise still run in frame order.  Results are identical.  Beyond 3 the
                                 streams outnumber the 4 hardware queues a process gets by default (GPU_MAX_HW_QUEUES):
                                 launches of streams that share a queue serialise.                          */
    uint32_t tracer;          /* scheduling of the trace stage; every choice gives bit-identical images:
                                 0 auto (4 when max_bounces >= 2, else 1), 1 monolithic kernel (one pixel per lane,
                                 all bounces, longest-tile-first), 2 wavefront (one launch per path segment, live
                                 paths compacted in between), 3 ray queues (shade / trace launches, lanes refilled
                                 ray by ray), 4 monolithic head + compacted tail (the monolithic kernel follows a
                                 path up to its second hit; the paths still alive there — about a third of a
                                 geometry tile's lanes — are queued and finished by a dense launch), 5 as 4 with
                                 a tail whose lanes are refilled path by path (slower; kept for comparison).  */
    uint32_t frames_per_launch;/* 0/1: one trace launch per frame.  B = 2..32: vxrt_render_frames (parameters at rest)
                                 traces up to B consecutive frames with ONE launch of the tracer (tracers 1 and 4;
                                 B ring slots, frame numbers n+1..n+B, longest tile first across the whole batch);
                                 temporal / denoise still run per frame in frame order.  Results are identical.
                                 Up to frames_in_flight such launches overlap (one stream each).             */
} vxrt_config;

typedef enum vxrt_image {
    VXRT_SAMPLED_COLOR = 0,   /* voxels.comp binding 0: (rgb, 1)                                        */
    VXRT_NORMAL_DEPTH = 1,    /* voxels.comp binding 1: (normal, t); miss = (2^30,2^30,2^30,-1)         */
    VXRT_ALBEDO_NODE = 2,     /* voxels.comp binding 2: (albedo, bits(leaf word))                       */
    VXRT_ACCUM_COLOR = 3,     /* temporal.comp binding 3: (blended rgb, next blending)                  */
    VXRT_DENOISED = 4         /* denoise.comp binding 0: (rgb, 1) — what the reference displays         */
} vxrt_image;

enum {                        /* vxrt_render flags: which of the three dispatches of                    */
    VXRT_TRACE = 1,           /* Context::render (src/context.rs:2024-2037) to run                      */
    VXRT_TEMPORAL = 2,
    VXRT_DENOISE = 4,
    VXRT_ALL = 7,
    VXRT_TIMED = 8,           /* bracket every kernel with HIP events (read back through vxrt_get_stats) */
    /* The denoise stage in two launches around a halo exchange (multi-GPU, see "halo" below): the 16x16 tiles whose
     * (2r+1)^2 window (denoise.comp:51-57) stays inside this context's rows, and the tiles that read rows of a neighbour.
     * INTERIOR needs no halo (it runs while the messages travel), EDGE needs vxrt_halo_unpack first; together they write
     * exactly what VXRT_DENOISE writes.  On a single-GPU context every tile is interior. */
    VXRT_DENOISE_INTERIOR = 16,
    VXRT_DENOISE_EDGE = 32
};

typedef struct vxrt_stats {
    uint64_t frames;          /* frames rendered since create / reset                                   */
    uint64_t rays;            /* cast_bounded_ray invocations (primary + bounce + sun) in those frames  */
    uint64_t pixels;          /* pixels traced (this context's rows) in those frames                    */
    double trace_ms;          /* summed kernel time of VXRT_TIMED frames, per stage                     */
    double temporal_ms;
    double denoise_ms;
    uint64_t timed_frames;
    uint64_t timed_launches;  /* trace launches behind trace_ms (= timed_frames unless frames_per_launch > 1)   */
    uint64_t scene_bytes;     /* device bytes of the scene (octree) and of the noise table              */
    uint64_t noise_bytes;
    uint32_t local_rows;      /* rows owned by this context                                             */
    uint32_t octree_depth;
    uint64_t octree_nodes;
    uint64_t wide_nodes;      /* records of the scene's two-levels-per-record form (16 bytes each); 0 unless VXRT_OPT_SCENE_FORMAT 1 */
    uint32_t scene_format;    /* which of the two the default tracer walks right now: 0 8-byte records, 1 wide records      */
    uint32_t node_order;      /* 2 / 3: the scene in place has its last 2 / 3 node levels as treelets (VXRT_OPT_NODE_ORDER)  */
    uint64_t queue_bytes;     /* device bytes of the tracer's path queues (sized by need for tracers 4 / 5)             */
    uint64_t queue_overflow_paths; /* paths that found their queue shard full and were followed by the head kernel
                                 instead (same image; the queues grow before the stream's next launch)               */
    double halo_pack_ms;      /* summed kernel time of the halo pack / unpack launches (always timed) ...               */
    double halo_unpack_ms;
    uint64_t halo_exchanges;  /* ... and how many unpacks that was                                                      */
    uint32_t cull_box_valid;  /* the sky cull's box (VXRT_OPT_SKY_CULL): every occupied cell of tree level min(depth, 7),       */
    float cull_box_min[3];    /* world units, before the per-frame margin is added                                              */
    float cull_box_max[3];
    uint32_t frame_lane_launches; /* trace launches whose waves held 8 frames of a pixel row (VXRT_OPT_FRAME_LANES)              */
} vxrt_stats;

/* Run-time options (none of them changes what a frame means; defaults are the reference's behaviour).
 *   VXRT_OPT_DENOISE_MODE  0 (default): denoise.comp evaluated exactly as the oracle restates it (bit-identical).
 *                          1: tolerant — the per-tap weight exp(-(..)/sigma_range_2 - (..)/sigma_distance_2) (denoise.comp:64-80)
 *                             with a reciprocal multiply and the hardware's exp2; within BASELINE's RMSE <= 1e-3 of mode 0
 *                             (tests/test_gpu_pipeline.py, test_gpu_denoise.py), 1.7 - 2.6 x as fast.
 *                          + 2: the generic kernel (the full formula for every tap) instead of the fast one — a cross-check; the
 *                             library takes it by itself when sigma_range > 7 (the GUI of the reference offers 0.1 .. 5).
 *   VXRT_OPT_TAIL_CAPACITY records per shard of the compacted tail's path queue (test hook: a small value forces the
 *                          queue-full path); 0 = back to automatic sizing.
 *   VXRT_OPT_SCENE_FORMAT  which scene records tracers 1 and 4 walk: 0 (default) the 8-byte records, one tree level each; 1 the wide
 *                          records, two levels per 16-byte record (half the dependent loads of a descent, ~35 % more instructions
 *                          per step: slower on MI355X for every scene measured, kept for comparison).  Same image either way.
 *                          Must be chosen before the scene is set (the wide records are built with the scene).  Needs a
 *                          library built with -DVXRT_VARIANTS=1 (vxrt_build_features); the default build refuses 1.
 *   VXRT_OPT_HALO_ROWS     multi-GPU: the fewest rows beyond each band edge that a halo exchange carries (default 1).  The exchange
 *                          carries max(denoise radius, this) rows; temporal.comp's reprojection (:85-113) sees that many rows of the
 *                          neighbouring bands' history, so set it to the largest vertical image motion per frame, in rows, that
 *                          should keep its history across a band edge (up to band_rows); beyond it a pixel is treated as
 *                          disoccluded, the reference's rule for a reprojection that leaves the screen (temporal.comp:92).
 *   VXRT_OPT_SKY_CULL      1 (default): a pixel whose primary ray provably misses the scene — it misses, with a margin, the box of the
 *                          tree's occupied cells at level min(depth, 7) — gets voxels.comp's miss outputs without walking the octree
 *                          (same values: csrc/trace.hip, primary_miss_is_certain states the proof).  0: every primary ray walks.
 *   VXRT_OPT_FRAME_LANES   1 (default): a trace launch of a multiple of 8 (of 4) frames of one camera (vxrt_render_frames / vxrt_render_spp
 *                          with such a frames_per_launch), or of a camera path that moves the image by less than ~16 pixels across
 *                          8 (4) frames (vxrt_render_path), gives each wave one row (two rows) of 8 pixels in 8 (4) consecutive frames
 *                          instead of an 8 x 8 tile of one frame — the same per-pixel operations, more coherent waves (a pixel's primary
 *                          ray is the same in every frame).  0: always one frame per wave.  vxrt_stats.frame_lane_launches counts the
 *                          former.  Not used for scenes beyond the Infinity Cache (measured slower there).
 *
 * Scheduling options for experiments and tests (same image whatever they say; the library reads NO environment variable — this is
 * the only way in).  "create": accepted by vxrt_create_tuned only, because a launch in flight must not see them change or because
 * they size what the context allocates.
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
 *   VXRT_OPT_TRACE_SPLIT / PATH_BLOCKS / SHADE_BLOCKS / RAYS_PER_WAVE   create: launch shapes of tracers 2, 3 and 5.                    */
typedef enum vxrt_option { VXRT_OPT_DENOISE_MODE = 1, VXRT_OPT_TAIL_CAPACITY = 2, VXRT_OPT_SCENE_FORMAT = 3, VXRT_OPT_HALO_ROWS = 4,
                           VXRT_OPT_SKY_CULL = 5, VXRT_OPT_FRAME_LANES = 6, VXRT_OPT_TILE_ORDER = 7, VXRT_OPT_TILE_SPREAD = 8,
                           VXRT_OPT_TRACE_BLOCKS = 9, VXRT_OPT_TAIL_FROM = 10, VXRT_OPT_TAIL_SPLIT = 11, VXRT_OPT_HOST_SCENE_BUILD = 12,
                           VXRT_OPT_TRACER_OVERRIDE = 13, VXRT_OPT_TRACE_SPLIT = 14, VXRT_OPT_PATH_BLOCKS = 15, VXRT_OPT_SHADE_BLOCKS = 16,
                           VXRT_OPT_RAYS_PER_WAVE = 17, VXRT_OPT_NODE_ORDER = 18, VXRT_OPT_HEAD_STAGGER = 19, VXRT_OPT_LONG_TILES = 20, VXRT_OPT_FUSED_TAIL = 21 } vxrt_option;
#define VXRT_TILE_SPREAD_AUTO 0xffffffffu
int vxrt_set_option(vxrt_ctx* ctx, vxrt_option option, uint32_t value);
/* An (option, value) pair for vxrt_create_tuned. */
typedef struct vxrt_tuning { uint32_t option; uint32_t value; } vxrt_tuning;

/* ---- context: replaces Context::new / create_bindings / resize (src/context.rs:595-660, 936-1016,
 *      1430-1461).  resize drops the temporal history like the reference (:1440-1448). -------------- */
int vxrt_create(const vxrt_config* cfg, vxrt_ctx** out);
/* vxrt_create with `count` options applied before anything is allocated (vxrt_create = none).  The reference has no counterpart: its
 * tuning is compile-time constants in the shaders; this is where an A/B script or a test says what it wants instead of the process
 * environment. */
int vxrt_create_tuned(const vxrt_config* cfg, const vxrt_tuning* tuning, size_t count, vxrt_ctx** out);
int vxrt_destroy(vxrt_ctx* ctx);
int vxrt_resize(vxrt_ctx* ctx, uint32_t width, uint32_t height);

/* ---- scene: replaces Context::recreate_octree (src/context.rs:799-810); input is the reference's
 *      Vec<([i16;3],[u8;4])> (position, [material, r, g, b]).  Resets the temporal history. -------- */
int vxrt_set_voxels(vxrt_ctx* ctx, const int16_t (*pos)[3], const uint8_t (*mrgb)[4], size_t n);
/* vox::load + Context::voxels_from_vox + recreate_octree (src/context.rs:1817-1821). */
int vxrt_load_vox(vxrt_ctx* ctx, const char* path);
int vxrt_load_vox_memory(vxrt_ctx* ctx, const uint8_t* bytes, size_t len);

/* ---- per-frame parameters: replaces Context::update_bindings (src/context.rs:2136-2162). ---------- */
/* Camera{position, direction, fov} (src/camera.rs:5-9); the library applies Camera::axis_scaled. */
int vxrt_set_camera(vxrt_ctx* ctx, const float position[3], const float direction[3], float fov);
int vxrt_set_scene_params(vxrt_ctx* ctx, const vxrt_uniforms* u);
int vxrt_set_temporal(vxrt_ctx* ctx, const vxrt_temporal* t);
int vxrt_set_denoise(vxrt_ctx* ctx, const vxrt_denoise* d);
void vxrt_default_uniforms(vxrt_uniforms* u);   /* Uniforms::default(),          src/context.rs:471-498 */
void vxrt_default_temporal(vxrt_temporal* t);   /* TemporalUniforms::default(),  src/context.rs:517-525 */
void vxrt_default_denoise(vxrt_denoise* d);     /* DenoiseUniforms::default(),   src/context.rs:316-325 */

/* ---- frame: replaces Context::render (src/context.rs:2004-2075).  frame_number is incremented
 *      first (:2152), the selected stages run in the reference's order, then the G-buffer history
 *      is handed over (ping-pong instead of the copy at :2041-2043).  Asynchronous. ---------------- */
int vxrt_render(vxrt_ctx* ctx, uint32_t flags);
/* `count` consecutive frames with the parameters currently set (camera at rest): count x vxrt_render. */
int vxrt_render_frames(vxrt_ctx* ctx, uint32_t flags, uint32_t count);
/* `count` frames along a camera path: frame k = vxrt_set_camera(positions[k], directions[k], fov) + vxrt_render(flags) — the
 * fly-through Context::update drives interactively (src/context.rs:1959-2001) — with up to frames_per_launch consecutive
 * frames per trace launch (each frame through its own camera; temporal reprojects from the previous frame's). */
int vxrt_render_path(vxrt_ctx* ctx, uint32_t flags, uint32_t count, const float (*positions)[3], const float (*directions)[3], float fov);
/* One displayed frame of `spp` samples per pixel, as BASELINE's "N spp" is defined in SURVEY.md 8d: spp consecutive trace
 * frames (frame_number advances by spp, parameters at rest) averaged with equal weights — binary32 sum in frame order,
 * one division — then temporal / denoise (per flags) once on the average.  The reference itself takes one sample per
 * pixel per frame (shaders/voxels.comp:305-391) and leaves accumulation to temporal.comp. */
int vxrt_render_spp(vxrt_ctx* ctx, uint32_t flags, uint32_t spp);
int vxrt_sync(vxrt_ctx* ctx);
int vxrt_reset_history(vxrt_ctx* ctx);          /* still_sample = 0 path, src/context.rs:1424 */
int vxrt_set_frame_number(vxrt_ctx* ctx, uint32_t frame_number); /* next render uses frame_number+1 */

/* ---- outputs.  The reference only blits denoised_color (src/context.rs:1131-1136); every image is
 *      readable here.  dst receives this context's rows in ascending frame-row order, rgba32f,
 *      local_rows*width*16 bytes (vxrt_local_rows; = height for a single-GPU context). ------------ */
int vxrt_read(vxrt_ctx* ctx, vxrt_image which, float* dst, size_t bytes);
int vxrt_local_rows(const vxrt_ctx* ctx, uint32_t* count, uint32_t* rows /* optional: count entries */);
/* Device pointer of an image (local rows, rgba32f) for zero-copy consumers on the same GPU. */
int vxrt_device_image(vxrt_ctx* ctx, vxrt_image which, void** device_ptr, size_t* bytes);
int vxrt_get_stats(vxrt_ctx* ctx, vxrt_stats* out);
int vxrt_reset_stats(vxrt_ctx* ctx);

/* ---- halo for multi-GPU.  A context renders interleaved bands of rows; the (2r+1)^2 window of
 *      denoise.comp:51-57 and the reprojection of temporal.comp:85-113 read rows of the bands above and below each of its
 *      own, which live on rank - 1 and rank + 1.  After a frame's temporal stage each rank sends TWO messages (to rank - 1:
 *      the top `rows` rows of each of its bands; to rank + 1: the bottom `rows` rows) and receives two; rows = max(denoise
 *      radius, VXRT_OPT_HALO_ROWS), at most band_rows.  A message holds, per pixel, 36 bytes in three planes over
 *      (slot = the receiver's local band, row, x):  A float4 (r, g, b, depth) | B float4 (normal, material id) | C float
 *      (blending factor of the accumulated colour)  — what denoise.comp reads of a neighbour's pixel plus what temporal.comp
 *      reads of it.  Buffers are DEVICE memory of message_bytes each, owned by the caller (who moves them with RCCL
 *      send/recv).  The frame loop, with the exchange overlapped (gpu_voxel_raytracer_amd/distributed.py):
 *
 *        vxrt_render(TRACE | TEMPORAL)
 *        vxrt_halo_pack(to_prev, to_next)              one kernel on the context's stream
 *        vxrt_stream_wait_context(comm_stream)         the communication stream waits for it (event, no host wait)
 *        ... send to_prev / to_next, receive from_prev / from_next on comm_stream ...
 *        vxrt_render(VXRT_DENOISE_INTERIOR)            runs while the messages travel
 *        vxrt_context_wait_stream(comm_stream)         the context's stream waits for the receives (event)
 *        vxrt_halo_unpack(from_prev, from_next)        one kernel: both messages -> the context's halo store
 *        vxrt_render(VXRT_DENOISE_EDGE)
 *
 *      The unpacked rows also serve the NEXT frame's temporal stage as the neighbours' history.  from_prev is what the
 *      previous rank sent as its to_next, from_next what the next rank sent as its to_prev. ------------------------------ */
typedef struct vxrt_halo_info {
    uint32_t rows;              /* rows per band edge that the next exchange carries (0: single-GPU context, nothing to exchange) */
    uint32_t slots;             /* bands a message has room for: ceil(bands / nranks)                                    */
    uint32_t bytes_per_pixel;   /* 36                                                                                    */
    uint32_t interior_tile_rows, edge_tile_rows;   /* rows of 16x16 denoise tiles that need no halo / that read it      */
    uint32_t max_rows;          /* the most rows per band edge this band layout can carry: the lowest band that has a band below it —
                                 * band_rows, unless the frame is lower than one round of bands (height < nranks * band_rows).  rows =
                                 * min(max_rows, max(denoise radius, VXRT_OPT_HALO_ROWS)): a larger wish is clamped, and THIS is the
                                 * bound to pass to vxrt_halo_rows_for_motion (round 5; `reserved` until then)                  */
    uint64_t message_bytes;     /* size of each of the four buffers (>= slots * rows * width * 36, whole 256-byte lines) */
} vxrt_halo_info;
int vxrt_halo_info_get(vxrt_ctx* ctx, vxrt_halo_info* out);
int vxrt_halo_bytes(vxrt_ctx* ctx, size_t* bytes_per_neighbour);     /* = vxrt_halo_info.message_bytes */
/* Asynchronous: one launch on the context's stream each, return at once.  The buffers must stay untouched until the launch
 * has run: order the communication against it with the two calls below. */
int vxrt_halo_pack(vxrt_ctx* ctx, void* dev_to_prev, void* dev_to_next);
int vxrt_halo_unpack(vxrt_ctx* ctx, const void* dev_from_prev, const void* dev_from_next);
/* `stream`: a hipStream_t of the context's device (NULL = the legacy default stream).  vxrt_stream_wait_context: work
 * enqueued on `stream` after this call starts only when everything enqueued on the context so far has finished.
 * vxrt_context_wait_stream: the context's stages enqueued after this call start only when everything enqueued on `stream`
 * so far has finished.  Events only; the host does not wait.  (Also what a zero-copy consumer of vxrt_device_image needs.) */
int vxrt_stream_wait_context(vxrt_ctx* ctx, void* stream);
int vxrt_context_wait_stream(vxrt_ctx* ctx, void* stream);
/* Synchronous forms: pack / unpack and wait for the launch (the buffers are borrowed for the call only). */
/* Host-only helper: VXRT_OPT_HALO_ROWS for the exchange after a frame rendered from camera A when the next frame comes from camera B —
 * the largest vertical image motion (rows) of any point at distance >= near_distance between the two, + 2, at most `band_rows` — pass
 * vxrt_halo_info.max_rows there, the depth the context's band layout can carry.  While the returned value is BELOW that cap the next
 * frame's reprojection (shaders/temporal.comp:75-113) finds its history across band edges exactly as one GPU would; a result equal to
 * the cap says the motion may reach beyond a neighbour's band (rows of a third rank: treated as a disocclusion, temporal.comp:92).  The
 * reference has no counterpart (one GPU sees the whole history); a frame loop sets it one frame ahead on a camera path. */
int vxrt_halo_rows_for_motion(const float pos_a[3], const float dir_a[3], const float pos_b[3], const float dir_b[3], float fov, uint32_t width,
                              uint32_t height, float near_distance, uint32_t band_rows, uint32_t* rows);
int vxrt_halo_export(vxrt_ctx* ctx, void* dev_to_prev, void* dev_to_next);
int vxrt_halo_import(vxrt_ctx* ctx, const void* dev_from_prev, const void* dev_from_next);

/* ---- host-side scene preparation, callable without a GPU (src/vox.rs, src/context.rs:710-834,
 *      913-933, src/camera.rs).  Counts are returned through *n; nothing is written past cap. ------ */
/* Test hook: the scene as the device holds it right now (8-byte records, 2 words each; leaf words).  Null arrays: sizes only. */
int vxrt_debug_read_scene(vxrt_ctx* ctx, uint32_t* svo, size_t svo_cap, size_t* n_svo, int32_t* leaves, size_t leaf_cap, size_t* n_leaves);
/* The device scene formats for a voxel list (csrc/kernels.h: SvoRecord = 2 words, WideRec = 4 words per record; leaf words as in
 * src/context.rs:732-735), built on the host exactly as vxrt_set_voxels builds them: for tools and tests.  Null arrays: sizes only. */
int vxrt_build_records(const int16_t (*pos)[3], const uint8_t (*mrgb)[4], size_t n, uint32_t* svo, size_t svo_cap, size_t* n_svo,
                       uint32_t* wide, size_t wide_cap, size_t* n_wide, int32_t* leaves, size_t leaf_cap, size_t* n_leaves, uint32_t* depth);
int vxrt_vox_to_voxels(const uint8_t* bytes, size_t len, int16_t (*pos)[3], uint8_t (*mrgb)[4], size_t cap,
                       size_t* n, uint32_t size_xyz[3]);
int vxrt_build_octree(const int16_t (*pos)[3], const uint8_t (*mrgb)[4], size_t n, int32_t* words, size_t cap,
                      size_t* n_words, uint32_t* depth);
int vxrt_camera_axis_scaled(const float position[3], const float direction[3], float fov, uint32_t width,
                            uint32_t height, float right[3], float up[3], float forward_ray[3]);
/* Stand-in for the blue-noise table the reference does not ship (resources/blue-noise-128.zip,
 * .MISSING_LARGE_BLOBS): value i of seed s is  z = i*0x9E3779B9 + s;  z ^= z>>16; z *= 0x85EBCA6B;
 * z ^= z>>13; z *= 0xC2B2AE35; z ^= z>>16;  (z >> 8) * 2^-24. */
int vxrt_noise_table(uint32_t seed, float* out, size_t n);
/* ---- blue noise.  The reference indexes 512 layers of 128x128 blue noise loaded from
 *      resources/blue-noise-128.zip (src/context.rs:1016-1116), a file its repository does not ship.
 *      vxrt_blue_noise makes such a table on the GPU (void-and-cluster, include/vxrt_bluenoise.h): `layers`
 *      layers first_layer.. of size x size floats into the HOST buffer out (size: power of two, 16..128).
 *      The zip functions read / write the reference's archive format (entries in archive order, each
 *      BE u32 width | BE u32 height | BE f32 pixels; square, all the same size), so a generated table can be
 *      dropped into the reference as its missing resource.  vxrt_set_noise replaces a context's table
 *      (512*128*128 floats, host memory). ---------------------------------------------------------------- */
int vxrt_blue_noise(int32_t device, uint32_t seed, uint32_t size, uint32_t first_layer, uint32_t layers, float* out);
int vxrt_noise_zip_read(const char* path, float* out, size_t cap_floats, uint32_t* size, uint32_t* layers);
int vxrt_noise_zip_write(const char* path, const float* table, uint32_t size, uint32_t layers);
int vxrt_set_noise(vxrt_ctx* ctx, const float* table);

/* ---- wider scene input.  vxrt_vox_scene_to_voxels: like vxrt_vox_to_voxels, plus, per flags:
 *      every shape instance of the file's nTRN/nGRP/nSHP scene graph with its translation and axis rotation
 *      (the reference renders models[0] at its raw coordinates and skips those chunks, src/context.rs:916,
 *      src/vox.rs:61); material types other than _diffuse/_emit taken as diffuse instead of rejected
 *      (src/vox.rs:82-89); the result shifted so that its minimum corner is the origin.  bounds: inclusive voxel
 *      bounding box in renderer axes (optional).  vxrt_default_scene_voxels: Context::create_voxels
 *      (src/context.rs:838-910), the start-up scene, with a seeded generator for its random colours. ------- */
enum { VXRT_VOX_ALL_MODELS = 1, VXRT_VOX_LENIENT_MATERIALS = 2, VXRT_VOX_REBASE = 4 };
int vxrt_vox_scene_to_voxels(const uint8_t* bytes, size_t len, uint32_t flags, int16_t (*pos)[3], uint8_t (*mrgb)[4],
                             size_t cap, size_t* n, int32_t bounds_min[3], int32_t bounds_max[3]);
int vxrt_default_scene_voxels(uint32_t seed, int16_t (*pos)[3], uint8_t (*mrgb)[4], size_t cap, size_t* n);

/* Procedural level-`level` Menger sponge, side 3^level voxels at the origin (config 5 generator;
 * level 4 reproduces the voxel set of vox/menger.vox). */
int vxrt_menger_voxels(uint32_t level, const uint8_t mrgb[4], int16_t (*pos)[3], uint8_t (*out_mrgb)[4], size_t cap,
                       size_t* n);
/* The same sponge clipped to [0, clip)^3 (0 = no clip) with every voxel whose hash(x,y,z) % emissive_period == 0
 * marked emissive (0 = none): the voxel-list form of the scene vxrt_set_menger builds procedurally. */
int vxrt_menger_voxels_ex(uint32_t level, uint32_t clip, const uint8_t mrgb[4], uint32_t emissive_period, int16_t (*pos)[3],
                          uint8_t (*out_mrgb)[4], size_t cap, size_t* n);
/* BASELINE config 5: builds that scene straight into the device format without a voxel list
 * or a reference-layout octree (level 7 clipped to 2048: ~1.05e9 voxels, 1.2 GB of nodes + 4.2 GB of leaf words).
 * The tree equals the one create_octree (src/context.rs:777-796) would build from the voxel list. */
int vxrt_set_menger(vxrt_ctx* ctx, uint32_t level, uint32_t clip, const uint8_t mrgb[4], uint32_t emissive_period);

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

const char* vxrt_status_string(int status);
const char* vxrt_last_error(void);               /* thread-local detail of the last failing call */
uint32_t vxrt_abi_version(void);
/* What this build of the library contains: VXRT_FEATURE_VARIANTS = it was compiled with -DVXRT_VARIANTS=1 and also holds the
 * schedules and the scene format that measured slower on MI355X and are kept for comparison (tracers 2, 3, 5; the wide scene
 * records).  The default build holds tracers 1 and 4 over the 8-byte records and refuses the others with VXRT_E_INVALID. */
enum { VXRT_FEATURE_VARIANTS = 1 };
uint32_t vxrt_build_features(void);

#ifdef __cplusplus
}
#endif
#endif /* VXRT_H */
