// ctx.h — the context behind the C ABI of libvxrt (include/vxrt.h) and the internal helpers its translation units share.
// The ABI is split along its seams:
//   api_context.hip  create / destroy / resize, parameters, options, outputs, statistics     (Context::new, resize, update_bindings)
//   api_scene.hip    scene upload: voxel list -> octree -> device records; procedural scene  (Context::recreate_octree)
//   api_trace.hip    scheduling of the trace stage: frame slots, streams, tile order, tail queues
//   api_frame.hip    frame sequencing: vxrt_render* and the post stages                        (Context::render)
//   api_halo.hip     the multi-rank halo: layout, pack / unpack, interior / edge denoise split
//   api_host.cpp     host-only helpers (no GPU): .vox decoding, octree words, camera basis, noise archive
//   api_debug.hip    test hooks and diagnostics
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vxrt.h"
#include "../../include/vxrt_debug.h"
#include "../../include/vxrt_host.h"
#include "kernels.h"
#include "scene_host.h"

namespace vxrt {
const std::string& last_error();

constexpr size_t kNoiseCount = size_t(512) * 128 * 128;  // shaders/voxels.comp:65-67

inline int hip_fail(hipError_t e, const char* what) {
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return VXRT_E_DEVICE;
}
// Nothing may unwind across the C boundary: every int-returning entry point is a function-try-block ending in this.
#define VXRT_CATCH                                                                                                         \
    catch (const std::bad_alloc&) { vxrt::set_error("out of host memory"); return VXRT_E_INVALID; }                        \
    catch (const std::exception& e) { vxrt::set_error(std::string("internal error: ") + e.what()); return VXRT_E_INVALID; } \
    catch (...) { vxrt::set_error("internal error"); return VXRT_E_INVALID; }

#define HIP_TRY(expr)                                           \
    do {                                                        \
        hipError_t e_ = (expr);                                 \
        if (e_ != hipSuccess) return vxrt::hip_fail(e_, #expr); \
    } while (0)

// a device allocation that lives as long as the entry point that made it (released on every return path)
struct ScratchBuffer {
    void* p = nullptr;
    ~ScratchBuffer() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    int stage = 0;  // 0 trace, 1 temporal, 2 denoise, 3 halo pack, 4 halo unpack
};
}  // namespace vxrt

using namespace vxrt;

struct vxrt_ctx {
    vxrt_config cfg{};
    BandMap band{};
    hipStream_t stream = nullptr;

    // scene
    bool has_scene = false;
    SvoRecord* d_svo = nullptr;
    SvoRecord root_rec{0, 0};  // d_svo[0], passed to the kernels by value (every cast starts with it)
    WideRec* d_wide = nullptr; // the same tree as wide records (kernels.h): two levels per 16-byte record
    WideRec wide_root{0, 0, 0, 0};
    size_t wide_count = 0;
    int scene_format = 0;      // VXRT_OPT_SCENE_FORMAT: 0 the 8-byte records (default), 1 also build and walk the wide records
    int32_t* d_leaves = nullptr;
    size_t svo_count = 0, leaf_count = 0;
    float root_center[3] = {0, 0, 0};
    float root_size = 1.0f;
    // the sky cull's box (api_scene.hip: scene_box): every occupied cell of tree level min(depth, 7), world units, not yet grown
    bool box_valid = false;
    float box_min[3] = {0, 0, 0}, box_max[3] = {0, 0, 0};
    int sky_cull = 1;   // VXRT_OPT_SKY_CULL
    uint32_t depth = 0;
    float* d_noise = nullptr;

    // Images (local rows x width, rgba32f).  The trace outputs live in a ring of frame slots so that the
    // trace stage of up to `inflight` consecutive frames can be on the GPU together (one stream each) while
    // the temporal/denoise stages run in frame order on `stream`.  A slot is not re-used while it still is
    // the temporal history or while a stage that reads it is in flight (slot.last_use).
    struct Slot {
        float4* sampled_color = nullptr;
        float4* albedo = nullptr;
        float4* nd = nullptr;
        // Events.  A trace launch covers up to 32 slots and is ONE event (launch_events, per trace stream): recording two events
        // per slot cost ~8 us of host time per frame of a launch — more than a rank of 8 can afford (scripts/exp_host_submit.py).
        hipEvent_t own = nullptr;         // owned: recorded on the main stream by the stages that read the slot (temporal, denoise, halo pack, spp)
        hipEvent_t trace_done = nullptr;  // handle, not owned: the event of the launch that traced the slot
        hipEvent_t last_use = nullptr;    // handle: what must have finished before the slot is traced into again (`own` or a launch's event)
        bool last_use_recorded = false;
    };
    std::vector<Slot> ring;
    int inflight = 1;
    std::vector<hipStream_t> trace_streams;   // inflight entries; entry 0 is `stream` when inflight == 1
    char* ring_arena = nullptr;               // the ring's images: one allocation (alloc_images)
    std::vector<hipEvent_t> launch_events;    // two per trace stream, used alternately: "this launch has finished"
    std::vector<hipEvent_t> head_events;      // one per trace stream: "this stream's last trace_kernel has finished" (VXRT_OPT_HEAD_STAGGER)
    std::vector<hipStream_t> aux_streams;     // one per trace stream, made on first use: the all-in-one grid of the longest tiles (VXRT_OPT_LONG_TILES)
    std::vector<hipEvent_t> aux_fork, aux_join;
    uint32_t long_tiles_permille = 0;         // 0: off
    int fused_tail = 0;                       // VXRT_OPT_FUSED_TAIL: head and compacted tail of a launch as one grid of persistent waves
    uint64_t fused_errors = 0;                // launches whose fused_kernel gave up a bounded wait (reported by vxrt_sync)
    int head_stagger = 0;                     // 1: a launch's head waits for the previous launch's head on another stream
    int last_head_lane = -1;
    std::vector<unsigned> launch_event_turn;
    int slot = 0;        // slot of the most recently traced frame
    int hist_slot = -1;  // slot whose normal/depth pairs with accum[hist] as the temporal history
    float4* accum[2] = {nullptr, nullptr};
    float4* denoised = nullptr;
    float4* spp_sum = nullptr;  // running sum of vxrt_render_spp (allocated on first use)
    // multi-rank halo (api_halo.hip): the rows of the neighbouring ranks just outside this rank's bands, as last unpacked
    float4* halo = nullptr;            // store: two messages (kernels.h: HaloView)
    size_t halo_store_f4 = 0;          // its size in float4
    HaloView halo_view{nullptr, 0, 0, 0, 0};   // what the store holds right now (rows = 0: nothing)
    uint32_t halo_min_rows = 1;        // VXRT_OPT_HALO_ROWS: rows that travel even without a denoise window (temporal's reprojection)
    bool halo_valid = false;           // the store holds the current frame's rows (cleared by the next trace)
    hipEvent_t halo_event = nullptr;   // vxrt_stream_wait_context / vxrt_context_wait_stream
    uint16_t* d_tile_rows = nullptr;   // denoise tile rows: [interior..., edge...] (those that need no halo, those that do)
    uint32_t tile_rows_interior = 0, tile_rows_edge = 0;
    uint64_t temporal_count = 0;       // temporal stages run so far ...
    uint64_t halo_epoch = ~0ull;       // ... and its value when the halo was last imported: equal -> the halo holds the
                                       // neighbours' rows of the current temporal history
    int cur = 0;             // accum[cur] is written by the next temporal stage, accum[cur^1] is the history
    bool has_history = false;
    bool accum_is_sampled = true;  // the latest "accumulated" image is sampled_color (temporal never ran)
    int last = 0;                   // index of the most recently written accum image
    uint64_t traced = 0;            // frames traced so far
    uint64_t trace_launches = 0;    // trace launches so far (selects the trace stream)
    uint64_t timed_launches = 0;
    int batch = 1;                  // vxrt_config.frames_per_launch

    // parameters
    vxrt_uniforms uniforms{};
    vxrt_temporal temporal{};
    vxrt_denoise denoise{};
    float cam_pos[3] = {0, 0, -2}, cam_dir[3] = {0, 0, 1}, cam_fov = 1.2217305f;  // src/context.rs:618-622
    Cam cam{}, old_cam{};
    bool old_cam_valid = false;

    // stats
    unsigned long long* d_rays = nullptr;
    // wavefront tracer: two path queues (ping-pong) and three rotating sets of 64 shard counters
    struct StreamQueues {  // per trace stream; allocated only for the queue-based variants
        float4* hitq[2] = {nullptr, nullptr};  // sharded PathRec queues (variant 2: ping-pong; variant 3: [0] = primary hits)
        unsigned* counts3 = nullptr;            // three rotating sets of 64 shard counters
        unsigned launches = 0;
        RayQueue rq{};                          // variant 3
        void* rq_block = nullptr;
        // tail queue sized by need (variants 4 / 5): the counters trace_kernel wrote, copied back after every launch
        unsigned* host_counts = nullptr;        // pinned, one set: 64 counters, 16 uints apart
        hipEvent_t counts_ready = nullptr;
        bool counts_pending = false;
        unsigned counts_capacity = 0;           // the shard capacity of the launch whose counters host_counts holds
        // fused head + tail (VXRT_OPT_FUSED_TAIL; trace.hip: fused_kernel)
        void* fused_ctl = nullptr;              // the launch's cursors, zeroed on the stream before every launch
        unsigned* host_ctl = nullptr;           // pinned: the first 16 bytes of fused_ctl after a launch (word 2: a bounded wait ran out)
        bool stamps_clean = false;              // every record slot of hitq[0] carries stamp 0 (set when the queue is cleared, lost when it is re-allocated)
        unsigned fused_launches = 0;            // -> the launch's stamp, 1 .. 65535
    };
    std::vector<StreamQueues> queues;
    unsigned shard_capacity = 0;        // records per shard of the path queues
    unsigned shard_capacity_max = 0;    // ... in the worst case: every pixel of every frame of a launch hands its path over
    int tail_capacity_override = 0;     // test hook (VXRT_TAIL_CAPACITY / vxrt_set_option): > 0 pins the capacity
    uint64_t queue_overflow_paths = 0;  // paths that found their shard full and stayed in the head kernel
    uint64_t queue_bytes = 0;
    int denoise_mode = 0;               // VXRT_OPT_DENOISE_MODE: 0 exact (bit-identical to the oracle), 1 tolerant (post.hip)
    // 0 = monolithic trace_kernel (all bounces in one launch; default), 2 = wavefront launches per path segment,
    // 3 = ray queues: shade / trace launches with per-lane ray refill
    int trace_variant = 0;
    // tracer 0 (auto): scenes that do not fit the 256 MB Infinity Cache use the all-in-one kernel — compacting paths trades
    // the locality of a tile's rays for lane utilisation, which loses once SVO gathers go to HBM (config 5, 5.6 GiB:
    // 2.29 vs 3.30 ms per 4K frame)
    bool auto_tracer = false;
    int shade_blocks = 1024;
    unsigned rays_per_wave = 256;  // ray-queue tracer: fewest rays a trace wave takes (more = better lane refill, fewer waves)
    // longest-tile-first scheduling of the monolithic kernel: cost of every 16x16 tile in the last frame -> order
    struct TileSchedule {  // one per trace stream: costs of the frame it traced last, and the order made from them
        uint32_t* cost = nullptr;
        uint32_t* order = nullptr;
        uint32_t* last_cost = nullptr;  // copy for diagnostics (vxrt_debug_tile_costs)
        uint32_t* scratch = nullptr;    // per-block histograms of the sort (256 x 128)
        bool valid = false;
        int age = 0;  // frames traced since the last sort
        // VXRT_OPT_TRACE_PRIORITY: how many tiles of `order` walk (they come first: the order is plain longest-first then), read back
        // after every sort; 0 = not known (yet): the launch goes out as one grid
        unsigned* host_heavy = nullptr;     // pinned
        hipEvent_t heavy_ready = nullptr;
        bool heavy_pending = false;
        unsigned heavy = 0;
    };
    std::vector<TileSchedule> schedules;
    int last_schedule = 0;
    int use_tile_order = 1;
    int trace_blocks = 2048;
    int spread_override = -1;     // VXRT_OPT_TILE_SPREAD (tests, experiments): see launch_tile_order
    int node_order = 0;           // VXRT_OPT_NODE_ORDER: 2 / 3 = depth-first treelets of the last 2 / 3 node levels (scene_device.hip)
    int node_order_applied = 0;   // ... and whether the scene in place was reordered (vxrt_stats.node_order)
    int host_scene_build = 0;     // VXRT_OPT_HOST_SCENE_BUILD: vxrt_set_menger builds on the host even where the device builder could
    unsigned wave_slots = 5120;   // waves of trace_kernel the device holds at once: CUs x 4 SIMDs x 5 (vxrt_create)
    int frame_lanes = 1;   // trace_kernel may put 8 frames of a pixel row into a wave (TraceArgs::frame_lanes; VXRT_OPT_FRAME_LANES)
    uint32_t frame_lane_launches = 0;
    int path_blocks = 512;  // tracer 5: blocks of path_kernel (each wave takes an equal range of the queue, >= 512 paths)
    int tail_from = 1;  // tracer 4: the hit number at which live paths move to the compacted launches
    unsigned tail_split = 0;  // ... bit k: the tail compacts again and starts a new launch at path segment k
    unsigned trace_split = 0x1;  // bit k: compact live paths and start a new launch at path segment k
    uint64_t frames = 0, pixels = 0, timed_frames = 0;
    double ms[5] = {0, 0, 0, 0, 0};   // trace, temporal, denoise, halo pack, halo unpack
    uint64_t halo_exchanges = 0;
    std::vector<EventPair> pending, free_pairs;

    // vxrt_read_async (api_context.hip): a copy stream of the context's own, two slots; a slot = a device-side snapshot of the image
    // (so that later frames may overwrite the image while the snapshot travels) + "snapshot taken" / "transfer arrived" events
    struct ReadSlot {
        float4* stage = nullptr;
        size_t stage_bytes = 0;
        hipEvent_t snap = nullptr, arrived = nullptr;
        bool in_flight = false;
    };
    hipStream_t copy_stream = nullptr;
    ReadSlot read_slots[2];

    // VXRT_OPT_TRACE_PRIORITY (round 6's experiment): the trace streams at the device's highest priority, the tiles that only store
    // sky as a grid of their own on a low-priority stream per trace stream
    int trace_priority = 0;
    int xcd_affinity = 0;               // VXRT_OPT_XCD_AFFINITY: 0 off, S = side of a super-tile in tiles (api_trace.hip: xcd_affine_order)
    unsigned xcd_balance_permille = 0;  // (max - min) / max of the 8 lists' summed costs at the last such sort
    uint64_t split_launches = 0;        // trace launches that went out as two grids
    std::vector<hipStream_t> low_streams;
    std::vector<hipEvent_t> low_fork, low_join;

    // touch map (vxrt_debug_touch_map; -DVXRT_VARIANTS=1 builds): one bit per 64-byte line of the node records / the leaf words
    uint32_t* d_touch_nodes = nullptr;
    uint32_t* d_touch_leaves = nullptr;
    size_t touch_node_lines = 0, touch_leaf_lines = 0;
    // the DDA prototype's grid of the scene in place (vxrt_debug_dda_rays; -DVXRT_VARIANTS=1 builds): bricks, brick bits, super-brick bits, first leaf per brick
    void* dda_grid[4] = {nullptr, nullptr, nullptr, nullptr};
    int dda_levels = 0;
};

namespace vxrt {
// a new scene: the touch maps were sized for the old one (vxrt_debug_touch_map)
inline void drop_touch_maps(vxrt_ctx* c) {
    for (uint32_t** p : {&c->d_touch_nodes, &c->d_touch_leaves}) { if (*p) (void)hipFree(*p); *p = nullptr; }
    c->touch_node_lines = c->touch_leaf_lines = 0;
    for (void*& g : c->dda_grid) { if (g) (void)hipFree(g); g = nullptr; }   // ... and so was the DDA prototype's grid
    c->dda_levels = 0;
}
// ---- api_context.hip
size_t image_bytes(const vxrt_ctx* c);
int count_local_rows(const BandMap& b);
int local_band_count(const BandMap& b);
void free_images(vxrt_ctx* c);
int alloc_images(vxrt_ctx* c);
int sync_all(vxrt_ctx* c);
int set_band(vxrt_ctx* c, uint32_t width, uint32_t height);
EventPair take_pair(vxrt_ctx* c, int stage);
int resolve_events(vxrt_ctx* c);
bool valid_ctx(const vxrt_ctx* c);
float4* image_ptr(vxrt_ctx* c, vxrt_image which);
// ---- api_scene.hip
bool use_wide(const vxrt_ctx* c);
// the smallest box of cells of tree level min(depth, 7) that holds every voxel; recs: the first records of the tree, breadth first,
// at least those of levels 0 .. min(depth, 7) - 1.  false: no voxel at all
bool scene_box(const SvoRecord* recs, size_t count, uint32_t depth, const float root_center[3], float root_size, float box_min[3], float box_max[3]);
// ---- api_trace.hip
int resize_tail_queues(vxrt_ctx* c, unsigned want);
int apply_option(vxrt_ctx* c, uint32_t option, uint32_t value, bool at_create);   // api_context.hip
void set_cull(const vxrt_ctx* c, TraceArgs& a, const Cam* cams, uint32_t g);       // api_trace.hip
int grow_tail_queues(vxrt_ctx* c, size_t lane);
void update_bindings(vxrt_ctx* c);
void frame_constants(const vxrt_ctx* c, TraceArgs& a);
int trace_frames(vxrt_ctx* c, uint32_t g, bool timed, int* slots, Cam* cams, Cam* olds, const float (*path_pos)[3] = nullptr,
                 const float (*path_dir)[3] = nullptr, uint32_t gbuf_frames = 0xffffffffu);
// ---- api_frame.hip
int check_render(vxrt_ctx* c, uint32_t flags);
int post_stages(vxrt_ctx* c, uint32_t flags, bool timed);
// ---- api_halo.hip
void free_halo(vxrt_ctx* c);
int build_tile_rows(vxrt_ctx* c);       // after the band map changed
uint32_t halo_rows_wanted(const vxrt_ctx* c);   // rows per band edge the next exchange carries
}  // namespace vxrt
