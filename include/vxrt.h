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
 *
 * THIS HEADER IS THE CONTRACT: context, scene, camera and parameters, the render calls, read-back, statistics, the multi-GPU
 * halo, errors — 40 entry points.  Two more headers ship beside it and a host that only renders needs neither:
 *   vxrt_host.h   host-side helpers that need no context (the .vox decoder, the octree builder, the camera basis, the noise
 *                 generator and the reference's noise archive format) — what src/vox.rs / create_octree / camera.rs do,
 *                 callable on their own for tools and tests;
 *   vxrt_debug.h  test hooks, diagnostics and the scheduling options of experiments (vxrt_debug_*, vxrt_create_tuned,
 *                 VXRT_OPT_* from 7 on).  Nothing in there is needed to render, and none of it changes what a frame means.
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
                               * of the tile height (16, 8 or band_rows) that covers it in nranks bands — so that the BUSIEST
                               * rank owns within one tile row of height / nranks rows (the last ranks of that round, clipped by
                               * the frame's edge, may own up to a band fewer) and no band with a band below it is lower
                               * than band_rows.  vxrt_local_rows lists a context's rows; distributed.BandLayout states the rule. */
                              /* nranks = 0 or 1 -> the whole frame                                     */
    uint32_t frames_in_flight;/* 0/1: every stage of a frame runs in submission order on one stream (the
                                 reference's single queue).  F = 2..16: the TRACE stage of up to F consecutive
                                 frames may run concurrently (one HIP stream each, ring of F+2 G-buffer slots);
                                 temporal/denoise still run in frame order.  Results are identical.
                                 HARDWARE QUEUES: every launch in flight is a HIP stream, and a process's streams share
                                 GPU_MAX_HW_QUEUES hardware queues — 4 unless the HOST exports another number before its
                                 first HIP call (the HIP runtime reads it once; the library cannot set it afterwards and
                                 reads no environment variable itself).  With F >= 3, or with RCCL / a copy stream in
                                 the process, two busy streams can share a queue and their launches serialise: export
                                 GPU_MAX_HW_QUEUES=8 in the host's environment (bench.py and host.py do; measured on a
                                 rank of 8: 0.0266 ms per frame with 4 queues, 0.0178 with 8).  Nothing else differs. */
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
    uint32_t split_launches;  /* trace launches that went out as two grids of different stream priority (vxrt_debug.h: VXRT_OPT_TRACE_PRIORITY; ABI 6) */
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
 *                          library built with -DVXRT_VARIANTS=1 (vxrt_debug.h: vxrt_build_features); the default build refuses 1.
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
 */
typedef enum vxrt_option { VXRT_OPT_DENOISE_MODE = 1, VXRT_OPT_TAIL_CAPACITY = 2, VXRT_OPT_SCENE_FORMAT = 3, VXRT_OPT_HALO_ROWS = 4,
                           VXRT_OPT_SKY_CULL = 5, VXRT_OPT_FRAME_LANES = 6,
                           VXRT_OPT_LAST_ = 0x7fffffff /* vxrt_debug.h continues the list from 7 on (scheduling options of experiments) */ } vxrt_option;
int vxrt_set_option(vxrt_ctx* ctx, vxrt_option option, uint32_t value);

/* ---- context: replaces Context::new / create_bindings / resize (src/context.rs:595-660, 936-1016,
 *      1430-1461).  resize drops the temporal history like the reference (:1440-1448). -------------- */
int vxrt_create(const vxrt_config* cfg, vxrt_ctx** out);
int vxrt_destroy(vxrt_ctx* ctx);
int vxrt_resize(vxrt_ctx* ctx, uint32_t width, uint32_t height);

/* ---- scene: replaces Context::recreate_octree (src/context.rs:799-810); input is the reference's
 *      Vec<([i16;3],[u8;4])> (position, [material, r, g, b]).  Resets the temporal history. -------- */
int vxrt_set_voxels(vxrt_ctx* ctx, const int16_t (*pos)[3], const uint8_t (*mrgb)[4], size_t n);
/* vox::load + Context::voxels_from_vox + recreate_octree (src/context.rs:1817-1821). */
int vxrt_load_vox(vxrt_ctx* ctx, const char* path);
int vxrt_load_vox_memory(vxrt_ctx* ctx, const uint8_t* bytes, size_t len);
/* BASELINE config 5: the procedural level-`level` Menger sponge clipped to [0, clip)^3 (0 = no clip), every voxel
 * whose hash(x,y,z) % emissive_period == 0 emissive (0 = none), built on the device straight into the scene format without a voxel
 * list or a reference-layout octree (level 7 clipped to 2048: ~1.05e9 voxels, 1.2 GB of nodes + 4.2 GB of leaf words).  The tree
 * equals the one create_octree (src/context.rs:777-796) would build from the voxel list (vxrt_host.h: vxrt_menger_voxels_ex). */
int vxrt_set_menger(vxrt_ctx* ctx, uint32_t level, uint32_t clip, const uint8_t mrgb[4], uint32_t emissive_period);
/* Replaces a context's noise table: 512*128*128 floats in [0,1), host memory (the layers of resources/blue-noise-128.zip,
 * src/context.rs:1016-1116; vxrt_host.h makes or reads such a table). */
int vxrt_set_noise(vxrt_ctx* ctx, const float* table);


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

/* ---- outputs.  The reference only blits denoised_color (src/context.rs:1131-1136, the present at :2046-2070); every image is
 *      readable here.  dst receives this context's rows in ascending frame-row order, rgba32f,
 *      local_rows*width*16 bytes (vxrt_local_rows; = height for a single-GPU context). ------------ */
int vxrt_read(vxrt_ctx* ctx, vxrt_image which, float* dst, size_t bytes);      /* waits for the GPU, then copies: synchronous */
/* The same without stalling the render loop — what a host that shows (or encodes) EVERY frame wants in place of the reference's
 * present (src/context.rs:2046-2070).  vxrt_read_async snapshots image `which` as the stages enqueued so far leave it (a device-side
 * copy on the context's stream, so later frames may overwrite the image at once) and starts its transfer to `dst` on a copy stream of
 * the context's own; it returns without waiting.  vxrt_read_wait(slot) returns when that transfer has arrived.  Two transfers can be
 * in flight, slot 0 and slot 1: frame n + 1 renders while frame n travels, and a loop that alternates the slots costs
 * max(render, transfer) per frame instead of their sum.  dst must stay valid and untouched until the wait, and should be PINNED
 * host memory — vxrt_host_alloc, or the host's own hipHostMalloc / hipHostRegister — for the transfer to be asynchronous and at
 * PCIe rate (pageable memory works, slowly).  Re-using a slot waits for its previous transfer first. */
int vxrt_read_async(vxrt_ctx* ctx, vxrt_image which, float* dst, size_t bytes, uint32_t slot);
int vxrt_read_wait(vxrt_ctx* ctx, uint32_t slot);
/* Pinned host memory for vxrt_read_async, so that a host without HIP bindings of its own (the reference's Rust host) can have it. */
int vxrt_host_alloc(size_t bytes, void** out);
int vxrt_host_free(void* p);
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
/* Host-only helper: VXRT_OPT_HALO_ROWS for the exchange after a frame rendered from camera A when the next frame comes from camera B —
 * the largest vertical image motion (rows) of any point at distance >= near_distance between the two, + 2, at most `band_rows` — pass
 * vxrt_halo_info.max_rows there, the depth the context's band layout can carry.  While the returned value is BELOW that cap the next
 * frame's reprojection (shaders/temporal.comp:75-113) finds its history across band edges exactly as one GPU would; a result equal to
 * the cap says the motion may reach beyond a neighbour's band (rows of a third rank: treated as a disocclusion, temporal.comp:92).  The
 * reference has no counterpart (one GPU sees the whole history); a frame loop sets it one frame ahead on a camera path. */
int vxrt_halo_rows_for_motion(const float pos_a[3], const float dir_a[3], const float pos_b[3], const float dir_b[3], float fov, uint32_t width,
                              uint32_t height, float near_distance, uint32_t band_rows, uint32_t* rows);

const char* vxrt_status_string(int status);
const char* vxrt_last_error(void);               /* thread-local detail of the last failing call */
uint32_t vxrt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VXRT_H */
