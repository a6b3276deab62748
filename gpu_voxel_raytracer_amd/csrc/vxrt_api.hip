// vxrt_api.hip — the C ABI of libvxrt (include/vxrt.h): context, scene upload, frame sequencing.
// Stands where the reference's wgpu `Context` stands (src/context.rs); each entry point cites the
// call site it replaces in vxrt.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vxrt.h"
#include "../../include/vxrt_bluenoise.h"
#include "kernels.h"
#include "scene_host.h"
#include "vx_vec.h"

namespace vxrt {
const std::string& last_error();
int build_menger_svo(uint32_t level, uint32_t clip, const uint8_t mrgb[4], uint32_t emissive_period, std::vector<SvoRecord>* recs,
                     std::vector<int32_t>* leaves, uint32_t* depth_out);
int32_t procedural_leaf_word(uint32_t x, uint32_t y, uint32_t z, const uint8_t mrgb[4], uint32_t emissive_period);
bool menger_device_build_supported(uint32_t level, uint32_t clip);
int build_menger_svo_device(uint32_t level, uint32_t clip, const uint8_t mrgb[4], uint32_t emissive_period, hipStream_t stream,
                            SvoRecord** d_svo, size_t* svo_count, int32_t** d_leaves, size_t* leaf_count, uint32_t* depth_out, SvoRecord* root);
}
using namespace vxrt;

namespace {

constexpr size_t kNoiseCount = size_t(512) * 128 * 128;  // shaders/voxels.comp:65-67

int hip_fail(hipError_t e, const char* what) {
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return VXRT_E_DEVICE;
}
// Nothing may unwind across the C boundary: every int-returning entry point is a function-try-block ending in this.
#define VXRT_CATCH                                                                                                   \
    catch (const std::bad_alloc&) { set_error("out of host memory"); return VXRT_E_INVALID; }                        \
    catch (const std::exception& e) { set_error(std::string("internal error: ") + e.what()); return VXRT_E_INVALID; } \
    catch (...) { set_error("internal error"); return VXRT_E_INVALID; }

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) return hip_fail(e_, #expr); \
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
    int stage = 0;  // 0 trace, 1 temporal, 2 denoise
};

}  // namespace

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
        hipEvent_t trace_done = nullptr;  // recorded on the slot's trace stream
        hipEvent_t last_use = nullptr;    // recorded after the last stage that touched the slot
        bool last_use_recorded = false;
    };
    std::vector<Slot> ring;
    int inflight = 1;
    std::vector<hipStream_t> trace_streams;   // inflight entries; entry 0 is `stream` when inflight == 1
    int slot = 0;        // slot of the most recently traced frame
    int hist_slot = -1;  // slot whose normal/depth pairs with accum[hist] as the temporal history
    float4* accum[2] = {nullptr, nullptr};
    float4* denoised = nullptr;
    float4* spp_sum = nullptr;  // running sum of vxrt_render_spp (allocated on first use)
    float4* halo = nullptr;  // rows of neighbouring ranks for the denoise window
    uint32_t halo_radius = 0;
    bool halo_valid = false;
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
    };
    std::vector<TileSchedule> schedules;
    int last_schedule = 0;
    int use_tile_order = 1;
    int trace_blocks = 2048;
    int path_blocks = 512;  // tracer 5: blocks of path_kernel (each wave takes an equal range of the queue, >= 512 paths)
    int tail_from = 1;  // tracer 4: the hit number at which live paths move to the compacted launches
    unsigned tail_split = 0;  // ... bit k: the tail compacts again and starts a new launch at path segment k
    unsigned trace_split = 0x1;  // bit k: compact live paths and start a new launch at path segment k
    uint64_t frames = 0, pixels = 0, timed_frames = 0;
    double ms[3] = {0, 0, 0};
    std::vector<EventPair> pending, free_pairs;
};

namespace {

size_t image_bytes(const vxrt_ctx* c) { return size_t(c->band.local_rows) * c->band.width * sizeof(float4); }

int count_local_rows(const BandMap& b) {
    int rows = 0;
    for (int y0 = 0, band = 0; y0 < b.height; y0 += b.band_rows, band++)
        if (band % b.nranks == b.rank) rows += (y0 + b.band_rows <= b.height) ? b.band_rows : b.height - y0;
    return rows;
}

int local_band_count(const BandMap& b) {
    int bands = (b.height + b.band_rows - 1) / b.band_rows;
    return bands <= b.rank ? 0 : (bands - b.rank + b.nranks - 1) / b.nranks;
}

void free_images(vxrt_ctx* c) {
    for (vxrt_ctx::Slot& sl : c->ring) {
        for (float4** p : {&sl.sampled_color, &sl.albedo, &sl.nd}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        if (sl.trace_done) (void)hipEventDestroy(sl.trace_done);
        if (sl.last_use) (void)hipEventDestroy(sl.last_use);
    }
    c->ring.clear();
    float4** imgs[] = {&c->accum[0], &c->accum[1], &c->denoised, &c->halo, &c->spp_sum};
    for (float4** p : imgs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    for (vxrt_ctx::TileSchedule& t : c->schedules)
        for (uint32_t** p : {&t.cost, &t.order, &t.last_cost, &t.scratch}) { if (*p) (void)hipFree(*p); *p = nullptr; }
    c->schedules.clear();
    for (vxrt_ctx::StreamQueues& sq : c->queues) {
        for (float4** p : {&sq.hitq[0], &sq.hitq[1]}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        if (sq.counts3) (void)hipFree(sq.counts3);
        if (sq.rq_block) (void)hipFree(sq.rq_block);
        if (sq.host_counts) (void)hipHostFree(sq.host_counts);
        if (sq.counts_ready) (void)hipEventDestroy(sq.counts_ready);
    }
    c->queues.clear();
    c->queue_bytes = 0;
}

int alloc_images(vxrt_ctx* c) {
    free_images(c);
    size_t bytes = image_bytes(c);
    if (bytes == 0) bytes = sizeof(float4);
    if (size_t(c->band.local_rows) * size_t(c->band.width) >= (size_t(1) << kPixBits)) {
        set_error("more than 2^27 pixels per context");  // the tail queue's records carry pixel index and frame-in-launch in one word
        return VXRT_E_INVALID;
    }
    // inflight frames being traced + the frame in the post stages + the temporal history
    c->ring.resize(size_t(c->inflight) * size_t(c->batch) + 2);
    for (vxrt_ctx::Slot& sl : c->ring) {
        for (float4** p : {&sl.sampled_color, &sl.albedo, &sl.nd}) {
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), bytes));
            HIP_TRY(hipMemsetAsync(*p, 0, bytes, c->stream));
        }
        HIP_TRY(hipEventCreateWithFlags(&sl.trace_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.last_use, hipEventDisableTiming));
        sl.last_use_recorded = false;
    }
    float4** imgs[] = {&c->accum[0], &c->accum[1], &c->denoised};
    for (float4** p : imgs) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), bytes));
        HIP_TRY(hipMemsetAsync(*p, 0, bytes, c->stream));
    }
    // path queues: every 8x8-pixel wave of the primary launch appends to shard (wave index % 64)
    // Worst case: all 64 lanes of each of a shard's waves append (a launch's waves are dealt to the 64 shards round robin).
    const size_t waves = size_t((c->band.width + 15) / 16) * size_t((c->band.local_rows + 15) / 16) * 4 * size_t(c->batch);
    c->shard_capacity_max = unsigned((waves + 63) / 64 * 64);
    c->shard_capacity = c->shard_capacity_max;
    if (c->trace_variant >= 4) {
        // The compacted tail takes the paths that are alive at their second hit: 5 % of the bench frame's pixels, about half of a
        // frame filled with geometry.  An eighth of the worst case to start with (worst case: 64 B x pixels x frames per launch
        // per queue and stream = 8.5 GB for 1080p at 16 x 2); a launch that wants more keeps the excess paths in the head kernel
        // (queue_reserve) and the queues grow before the stream's next launch (grow_tail_queues).
        unsigned cap = c->shard_capacity_max / 8u;
        cap = cap < 4096u ? 4096u : cap;
        if (c->tail_capacity_override > 0) cap = unsigned(c->tail_capacity_override);
        c->shard_capacity = cap < c->shard_capacity_max ? (cap + 63u) / 64u * 64u : c->shard_capacity_max;
    }
    if (c->trace_variant != 0) {
        c->queues.resize(size_t(c->inflight));
        for (vxrt_ctx::StreamQueues& sq : c->queues) {
            const size_t hit_bytes = (size_t(c->shard_capacity) * 64 + 1) * 64;
            // the second queue: the wavefront tracer's ping-pong partner; for the compacted tail only when it compacts again
            const int nq = c->trace_variant == 3 ? 1 : ((c->trace_variant >= 4 && c->tail_split == 0u) ? 1 : 2);
            for (int i = 0; i < nq; i++) {
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sq.hitq[i]), hit_bytes));
                c->queue_bytes += hit_bytes;
            }
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sq.counts3), 3 * 64 * 64));
            HIP_TRY(hipMemsetAsync(sq.counts3, 0, 3 * 64 * 64, c->stream));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&sq.host_counts), 64 * 64, hipHostMallocDefault));
            memset(sq.host_counts, 0, 64 * 64);
            HIP_TRY(hipEventCreateWithFlags(&sq.counts_ready, hipEventDisableTiming));
            sq.counts_pending = false;
            sq.launches = 0;
            if (c->trace_variant == 3) {
                // Dense queues in kSegments segments.  A shade launch of G blocks (G a multiple of 8) hands every segment
                // G/8 blocks x 256 items per trip, so a segment receives at most its eighth of the paths rounded up to
                // a whole trip: paths/8 + 32 G.
                const size_t cap = waves * 64 / kSegments + size_t(c->shade_blocks) * 32 + 1024;
                const size_t per_path = 2 * 64 + 2 * 48 + 32;  // state x2, rays x2, results
                const size_t counts_bytes = size_t(c->cfg.max_bounces + 1) * kSegments * 64;
                HIP_TRY(hipMalloc(&sq.rq_block, kSegments * cap * per_path + counts_bytes + 256));
                char* p = static_cast<char*>(sq.rq_block);
                sq.rq.state[0] = reinterpret_cast<float4*>(p); p += kSegments * cap * 64;
                sq.rq.state[1] = reinterpret_cast<float4*>(p); p += kSegments * cap * 64;
                sq.rq.rays[0] = reinterpret_cast<float4*>(p); p += kSegments * cap * 48;
                sq.rq.rays[1] = reinterpret_cast<float4*>(p); p += kSegments * cap * 48;
                sq.rq.results = reinterpret_cast<uint4*>(p); p += kSegments * cap * 32;
                sq.rq.counts = reinterpret_cast<unsigned*>(p);
                sq.rq.seg_capacity = unsigned(cap);
            }
        }
    }
    const size_t tiles = trace_tile_count(c->band.width, c->band.local_rows);
    c->schedules.resize(size_t(c->inflight));
    for (vxrt_ctx::TileSchedule& t : c->schedules) {
        for (uint32_t** p : {&t.cost, &t.order, &t.last_cost}) {
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), (tiles + 1) * sizeof(uint32_t)));
            HIP_TRY(hipMemsetAsync(*p, 0, (tiles + 1) * sizeof(uint32_t), c->stream));
        }
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t.scratch), 256 * 128 * sizeof(uint32_t)));
        t.valid = false;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->slot = 0;
    c->hist_slot = -1;
    c->cur = 0;
    c->last = 0;
    c->has_history = false;
    c->accum_is_sampled = true;
    c->halo_valid = false;
    c->halo_radius = 0;
    return VXRT_OK;
}

int sync_all(vxrt_ctx* c) {
    for (hipStream_t t : c->trace_streams) HIP_TRY(hipStreamSynchronize(t));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VXRT_OK;
}

int set_band(vxrt_ctx* c, uint32_t width, uint32_t height) {
    const vxrt_config& cfg = c->cfg;
    BandMap b;
    b.width = int(width);
    b.height = int(height);
    b.nranks = cfg.nranks == 0 ? 1 : int(cfg.nranks);
    b.rank = int(cfg.rank);
    b.band_rows = cfg.band_rows == 0 ? 16 : int(cfg.band_rows);
    if (b.nranks == 1) b.rank = 0;
    b.local_rows = count_local_rows(b);
    c->band = b;
    return VXRT_OK;
}

Cam make_cam(const float pos[3], const CameraBasis& b) {
    Cam c;
    for (int i = 0; i < 3; i++) { c.o[i] = pos[i]; c.r[i] = b.right[i]; c.u[i] = b.up[i]; c.f[i] = b.forward_ray[i]; }
    return c;
}

// inverse of [R U F O; 0 0 0 1] (temporal.comp:75-82) by adjugate / determinant in binary64, rounded
// once to binary32: rows of A^-1 and the translation -A^-1 O.
void affine_inverse(const Cam& c, float inv[12]) {
    const double a = c.r[0], b = c.u[0], cc = c.f[0], d = c.r[1], e = c.u[1], f = c.f[1], g = c.r[2], h = c.u[2], i = c.f[2];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + cc * C;
    const double m[9] = {A, -(b * i - cc * h), b * f - cc * e, B, a * i - cc * g, -(a * f - cc * d), C, -(a * h - b * g), a * e - b * d};
    for (int r = 0; r < 3; r++) {
        const double r0 = m[3 * r] / det, r1 = m[3 * r + 1] / det, r2 = m[3 * r + 2] / det;
        inv[4 * r] = float(r0); inv[4 * r + 1] = float(r1); inv[4 * r + 2] = float(r2);
        inv[4 * r + 3] = float(-(r0 * double(c.o[0]) + r1 * double(c.o[1]) + r2 * double(c.o[2])));
    }
}

EventPair take_pair(vxrt_ctx* c, int stage) {
    EventPair p;
    if (!c->free_pairs.empty()) {
        p = c->free_pairs.back();
        c->free_pairs.pop_back();
    } else {
        (void)hipEventCreate(&p.a);
        (void)hipEventCreate(&p.b);
    }
    p.stage = stage;
    return p;
}

int resolve_events(vxrt_ctx* c) {
    for (EventPair& p : c->pending) {
        HIP_TRY(hipEventSynchronize(p.b));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b));
        c->ms[p.stage] += double(ms);
        c->free_pairs.push_back(p);
    }
    c->pending.clear();
    return VXRT_OK;
}

// reference-layout octree words -> breadth-first SVO records + leaf words (kernels.h)
int flatten_svo(const Octree& tree, std::vector<SvoRecord>* recs, std::vector<int32_t>* leaves) {
    const int32_t* nodes = tree.words.data() + 5;
    std::vector<uint32_t> order{0};
    order.reserve(tree.node_count());
    recs->clear();
    recs->reserve(tree.node_count());
    leaves->clear();
    for (size_t i = 0; i < order.size(); i++) {
        const int32_t* slot = nodes + size_t(8) * order[i];
        uint32_t child_mask = 0, leaf_mask = 0;
        for (int s = 0; s < 8; s++) {
            if (slot[s] > 0) child_mask |= 1u << s;
            else if (slot[s] < 0) leaf_mask |= 1u << s;
        }
        if (child_mask && leaf_mask) { set_error("node mixes children and leaves"); return VXRT_E_SCENE; }
        SvoRecord r;
        r.masks = child_mask | leaf_mask << 8;
        if (leaf_mask) {
            r.base = uint32_t(leaves->size());
            for (int s = 0; s < 8; s++) if (slot[s] < 0) leaves->push_back(slot[s]);
        } else {
            r.base = uint32_t(order.size());
            for (int s = 0; s < 8; s++) if (slot[s] > 0) order.push_back(uint32_t(slot[s]));
        }
        recs->push_back(r);
    }
    return VXRT_OK;
}

// 8-byte records (breadth first, children contiguous) -> wide records (kernels.h: WideRec).  Levels are paired from the bottom; the
// wide records are the nodes of the pairs' upper levels in the same breadth-first order, so a record's grandchildren are contiguous in
// (sub, slot) order, and the leaf words keep their order (the leaf parents under one node are neighbours in breadth-first order).
int widen_svo(const std::vector<SvoRecord>& recs, uint32_t depth, std::vector<WideRec>* out) {
    const uint32_t L = depth + 1;      // node levels: root 0 .. leaf parents L-1
    const uint32_t parity = L & 1u;    // odd: the root is the one sub of a virtual top
    out->clear();
    if (recs.empty()) { out->push_back(WideRec{0, 0, 0, 0}); return VXRT_OK; }
    std::vector<size_t> start(L + 1, recs.size());   // level l = recs[start[l] .. start[l+1])
    start[0] = 0;
    for (uint32_t l = 0; l + 1 < L; l++) {
        if (start[l] >= recs.size() || (recs[start[l]].masks & 0xffu) == 0u) break;   // no nodes below (an empty scene)
        start[l + 1] = recs[start[l]].base;
    }
    std::vector<size_t> woff(L + 2, 0);   // index of the first wide record whose top is level l
    size_t total = parity;
    for (uint32_t l = parity; l + 1 < L; l += 2) { woff[l] = total; total += start[l + 1] - start[l]; }
    if (total >= (size_t(1) << 32)) { set_error("too many nodes"); return VXRT_E_SCENE; }
    out->resize(total == 0 ? 1 : total);
    if (parity) {
        const SvoRecord& root = recs[0];
        const uint32_t byte = L == 1 ? (root.masks >> 8) & 0xffu : root.masks & 0xffu;
        (*out)[0] = WideRec{byte, 0u, L == 1 ? root.base : uint32_t(woff[1]), byte != 0u ? 1u : 0u};
    }
    for (uint32_t l = parity; l + 1 < L; l += 2) {
        const bool subs_are_leaf_parents = l + 2 == L;
        for (size_t i = start[l]; i < start[l + 1]; i++) {
            const SvoRecord& top = recs[i];
            const uint32_t cm = top.masks & 0xffu;
            uint64_t mask = 0;
            uint32_t k = 0;
            for (uint32_t s = 0; s < 8; s++)
                if (cm >> s & 1u) {
                    const SvoRecord& sub = recs[size_t(top.base) + k++];
                    mask |= uint64_t(subs_are_leaf_parents ? (sub.masks >> 8) & 0xffu : sub.masks & 0xffu) << (8u * s);
                }
            uint32_t base = 0;
            if (cm != 0u) {
                const SvoRecord& first = recs[top.base];
                base = subs_are_leaf_parents ? first.base : uint32_t(woff[l + 2] + (size_t(first.base) - start[l + 2]));
            }
            (*out)[woff[l] + (i - start[l])] = WideRec{uint32_t(mask), uint32_t(mask >> 32), base, cm};
        }
    }
    return VXRT_OK;
}

int upload_svo(vxrt_ctx* c, std::vector<SvoRecord>& recs, std::vector<int32_t>& leaves, uint32_t depth) {
    if (leaves.empty()) leaves.push_back(0);
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = sync_all(c)) return rc;
    // The wide records are built only when asked for (VXRT_OPT_SCENE_FORMAT 1 before the scene is set): measured on MI355X the walk
    // over them is slower than the walk over the 8-byte records on every scene tried, cache-resident or not, so the
    // default never uses them and does not pay for their memory.
    std::vector<WideRec> wide;
    if (c->scene_format == 1) { if (int rc = widen_svo(recs, depth, &wide)) return rc; }
    if (c->d_svo) (void)hipFree(c->d_svo);
    if (c->d_leaves) (void)hipFree(c->d_leaves);
    if (c->d_wide) (void)hipFree(c->d_wide);
    c->d_svo = nullptr;
    c->d_leaves = nullptr;
    c->d_wide = nullptr;
    c->wide_count = wide.size();
    c->wide_root = WideRec{0, 0, 0, 0};
    if (!wide.empty()) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_wide), wide.size() * sizeof(WideRec)));
        HIP_TRY(hipMemcpy(c->d_wide, wide.data(), wide.size() * sizeof(WideRec), hipMemcpyHostToDevice));
        c->wide_root = wide[0];
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_svo), recs.size() * sizeof(SvoRecord)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_leaves), leaves.size() * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(c->d_svo, recs.data(), recs.size() * sizeof(SvoRecord), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_leaves, leaves.data(), leaves.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    c->svo_count = recs.size();
    c->root_rec = recs.empty() ? SvoRecord{0, 0} : recs[0];
    c->leaf_count = leaves.size();
    c->root_center[0] = c->root_center[1] = c->root_center[2] = 0.0f;  // src/context.rs:782-786
    c->root_size = float(1u << depth);                                   // src/context.rs:779
    c->depth = depth;
    c->has_scene = true;
    return VXRT_OK;
}

int upload_scene(vxrt_ctx* c, const Voxel* voxels, size_t n) {
    Octree tree;
    if (int rc = build_octree(voxels, n, &tree)) return rc;
    std::vector<SvoRecord> recs;
    std::vector<int32_t> leaves;
    if (int rc = flatten_svo(tree, &recs, &leaves)) return rc;
    return upload_svo(c, recs, leaves, tree.depth);
}

// Which records trace_kernel / bounce_kernel walk: the wide ones only when VXRT_OPT_SCENE_FORMAT 1 (or VXRT_WIDE=1) asked for them
// before the scene was set.  (They halve the dependent loads of a descent, but the walk over them executes ~35 % more instructions
// per trip, and the stage is bound by instruction issue, not by those loads — measured: menger 1080p 21.0 vs 27.0 Gray/s, the
// 2048^3 scene 2.68 vs 2.22 ms per 4K frame.)
bool use_wide(const vxrt_ctx* c) { return c->d_wide != nullptr && c->scene_format == 1; }

// new capacity (records per shard) for the path queues of every stream; waits for the GPU first
int resize_tail_queues(vxrt_ctx* c, unsigned want) {
    want = want > c->shard_capacity_max ? c->shard_capacity_max : (want + 63u) / 64u * 64u;
    if (want == c->shard_capacity) return VXRT_OK;
    if (int rc = sync_all(c)) return rc;
    const size_t hit_bytes = (size_t(want) * 64 + 1) * 64;
    for (vxrt_ctx::StreamQueues& q : c->queues)
        for (float4*& p : q.hitq)
            if (p) {
                (void)hipFree(p);
                p = nullptr;
                c->queue_bytes -= (size_t(c->shard_capacity) * 64 + 1) * 64;
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p), hit_bytes));
                c->queue_bytes += hit_bytes;
            }
    c->shard_capacity = want;
    return VXRT_OK;
}

// Tail queues sized by need: look at what the stream's last launch wanted (its shard counters, copied back after the launch) and,
// if that did not fit, make the stream's queues larger before its next launch.  Paths that did not fit were followed by the head
// kernel itself, so no frame was wrong — only slower.
int grow_tail_queues(vxrt_ctx* c, size_t lane, hipStream_t ts) {
    vxrt_ctx::StreamQueues& sq = c->queues[lane];
    if (!sq.counts_pending || hipEventQuery(sq.counts_ready) != hipSuccess) return VXRT_OK;
    sq.counts_pending = false;
    unsigned peak = 0;
    for (unsigned s = 0; s < 64; s++) {
        const unsigned n = sq.host_counts[s * 16];
        peak = n > peak ? n : peak;
        if (n > c->shard_capacity) c->queue_overflow_paths += n - c->shard_capacity;
    }
    if (peak <= c->shard_capacity || c->tail_capacity_override > 0 || c->shard_capacity >= c->shard_capacity_max) return VXRT_OK;
    // every stream's queues share one capacity (PathQueue::shard_capacity travels with the launch): grow them all, at rest
    unsigned want = peak + peak / 4u;
    want = want < 2u * c->shard_capacity ? 2u * c->shard_capacity : want;
    (void)ts;
    return resize_tail_queues(c, want);
}

bool valid_ctx(const vxrt_ctx* c) {
    if (!c) { set_error("null context"); return false; }
    return true;
}

}  // namespace

extern "C" {

uint32_t vxrt_abi_version(void) { return 3; }

const char* vxrt_last_error(void) { return vxrt::last_error().c_str(); }

const char* vxrt_status_string(int status) {
    switch (status) {
        case VXRT_OK: return "ok";
        case VXRT_E_INVALID: return "invalid argument";
        case VXRT_E_DEVICE: return "HIP runtime error";
        case VXRT_E_VOX_MAGIC: return "invalid magic number";
        case VXRT_E_VOX_VERSION: return "unsupported VOX-format";
        case VXRT_E_VOX_NOMAIN: return "missing MAIN chunk";
        case VXRT_E_VOX_EOF: return "unexpected end of file";
        case VXRT_E_VOX_CHUNK: return "unexpected chunk";
        case VXRT_E_VOX_MATERIAL: return "unsupported material";
        case VXRT_E_VOX_NOMATL: return "voxel colour without material";
        case VXRT_E_VOX_NOMODEL: return "no model in file";
        case VXRT_E_IO: return "failed to read file";
        case VXRT_E_SCENE: return "voxel list cannot be represented";
        case VXRT_E_NOSCENE: return "no scene set";
        case VXRT_E_NOISE: return "failed to load blue noise";
        default: return "unknown status";
    }
}

void vxrt_default_uniforms(vxrt_uniforms* u) {
    memset(u, 0, sizeof *u);
    u->emit_strength = 4.0f;
    u->sun_strength = 4.0f;
    u->sun_size = 0.05f;
    u->sun_yaw = 1.32f;
    u->sun_pitch = 1.0f;
    u->sun_color[0] = u->sun_color[1] = u->sun_color[2] = 1.0f;
    u->sky_color[0] = 0.45f; u->sky_color[1] = 0.6f; u->sky_color[2] = 0.65f;
    u->specularity = 0.0f;
}
void vxrt_default_temporal(vxrt_temporal* t) { t->sample_blending = 0.5f; t->maximum_blending = 0.98f; t->blending_distance_cutoff = 1e-2f; }
void vxrt_default_denoise(vxrt_denoise* d) { d->radius = 0; d->sigma_distance = 2.0f; d->sigma_range = 1.5f; d->albedo_factor = 1.0f; }

int vxrt_create(const vxrt_config* cfg, vxrt_ctx** out) try {
    if (!cfg || !out) { set_error("null argument"); return VXRT_E_INVALID; }
    *out = nullptr;
    if (cfg->width == 0 || cfg->height == 0 || cfg->width > 65536 || cfg->height > 65536) { set_error("bad frame size"); return VXRT_E_INVALID; }
    if (cfg->max_bounces < 1 || cfg->max_bounces > 16) { set_error("max_bounces must be 1..16"); return VXRT_E_INVALID; }
    uint32_t nranks = cfg->nranks == 0 ? 1 : cfg->nranks;
    if (nranks > 1 && cfg->rank >= nranks) { set_error("rank >= nranks"); return VXRT_E_INVALID; }
    uint32_t band_rows = cfg->band_rows == 0 ? 16 : cfg->band_rows;
    if (band_rows % 8 != 0) { set_error("band_rows must be a multiple of 8 (of 16 for a denoise radius > 0)"); return VXRT_E_INVALID; }

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) { return hip_fail(e == hipSuccess ? hipErrorNoDevice : e, "hipGetDeviceCount"); }
    if (cfg->device < 0 || cfg->device >= ndev) { set_error("device ordinal out of range"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(cfg->device));

    vxrt_ctx* c = new vxrt_ctx();
    c->cfg = *cfg;
    c->cfg.noise = nullptr;  // borrowed for this call only
    vxrt_default_uniforms(&c->uniforms);
    vxrt_default_temporal(&c->temporal);
    vxrt_default_denoise(&c->denoise);
    int rc = VXRT_OK;
    auto fail = [&](int code) { vxrt_destroy(c); return code; };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(hip_fail(hipGetLastError(), "hipStreamCreate"));
    c->inflight = cfg->frames_in_flight == 0 ? 1 : int(cfg->frames_in_flight);
    if (const char* v = getenv("VXRT_INFLIGHT")) c->inflight = atoi(v);
    // vxrt_config.tracer: 0 auto, 1 monolithic, 2 wavefront, 3 ray queues, 4 monolithic head + compacted tail (internally
    // 0, 2, 3, 4).  Measured on MI355X with frames in flight (menger 1080p 4 bounces / monu10 4K 8 bounces, ms per frame):
    // tracer 1: 0.219 / 0.52-1.53, tracer 3: 0.275 / 0.82-1.10, tracer 4: 0.175 / 0.54-0.94 -> auto = 4 whenever a path
    // can have a second hit, with one more compaction at path segment 3 from 6 bounces on.
    if (cfg->tracer > 5) { set_error("tracer must be 0..5"); return fail(VXRT_E_INVALID); }
    c->trace_variant = cfg->tracer == 0 ? 4 : (cfg->tracer == 1 ? 0 : int(cfg->tracer));
    c->auto_tracer = cfg->tracer == 0 && getenv("VXRT_TRACE_VARIANT") == nullptr;
    c->tail_split = cfg->max_bounces >= 6 ? 0x8u : 0u;
    if (const char* v = getenv("VXRT_TRACE_VARIANT")) c->trace_variant = atoi(v);  // A/B override for benchmarks and tests
    if (c->trace_variant != 2 && c->trace_variant != 3 && c->trace_variant != 4 && c->trace_variant != 5) c->trace_variant = 0;
    if (c->trace_variant >= 4 && cfg->max_bounces < 2) c->trace_variant = 0;  // no tail to compact
    if (const char* v = getenv("VXRT_PATH_BLOCKS")) c->path_blocks = atoi(v);
    if (const char* v = getenv("VXRT_TAIL_FROM")) c->tail_from = atoi(v);
    if (c->tail_from < 0 || c->tail_from >= int(cfg->max_bounces)) c->tail_from = 1;
    if (const char* v = getenv("VXRT_TAIL_SPLIT")) c->tail_split = unsigned(strtoul(v, nullptr, 0));
    if (const char* v = getenv("VXRT_TAIL_CAPACITY")) c->tail_capacity_override = atoi(v);
    if (const char* v = getenv("VXRT_WIDE")) c->scene_format = atoi(v) == 1 ? 1 : 0;   // A/B and tests: 1 = the wide records   // test hook: force the queue-full path
    if (const char* v = getenv("VXRT_SHADE_BLOCKS")) c->shade_blocks = atoi(v);
    if (const char* v = getenv("VXRT_RAYS_PER_WAVE")) c->rays_per_wave = unsigned(atoi(v));
    c->shade_blocks = (c->shade_blocks < 8 ? 8 : (c->shade_blocks > 2048 ? 2048 : c->shade_blocks) + 7) / 8 * 8;
    if (const char* v = getenv("VXRT_TRACE_BLOCKS")) c->trace_blocks = atoi(v);
    if (c->inflight < 1 || c->inflight > 16) { set_error("frames_in_flight must be 1..16"); return fail(VXRT_E_INVALID); }
    c->batch = cfg->frames_per_launch == 0 ? 1 : int(cfg->frames_per_launch);
    if (const char* v = getenv("VXRT_BATCH")) c->batch = atoi(v);
    if (c->batch < 1 || c->batch > kMaxBatch) { set_error("frames_per_launch must be 1..32"); return fail(VXRT_E_INVALID); }
    c->trace_streams.assign(size_t(c->inflight), nullptr);
    if (c->inflight == 1) {
        c->trace_streams[0] = c->stream;
    } else {
        for (hipStream_t& t : c->trace_streams)
            if (hipStreamCreateWithFlags(&t, hipStreamNonBlocking) != hipSuccess) return fail(hip_fail(hipGetLastError(), "hipStreamCreate"));
    }
    set_band(c, cfg->width, cfg->height);
    if ((rc = alloc_images(c)) != VXRT_OK) return fail(rc);
    if (hipMalloc(reinterpret_cast<void**>(&c->d_noise), kNoiseCount * sizeof(float)) != hipSuccess) return fail(hip_fail(hipGetLastError(), "hipMalloc noise"));
    if (cfg->noise) {
        if (hipMemcpy(c->d_noise, cfg->noise, kNoiseCount * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail(hip_fail(hipGetLastError(), "noise upload"));
    } else {
        if (launch_noise_fill(c->d_noise, cfg->noise_seed, kNoiseCount, c->stream) != hipSuccess) return fail(hip_fail(hipGetLastError(), "noise fill"));
    }
    if (hipMalloc(reinterpret_cast<void**>(&c->d_rays), kRaySlots * 64) != hipSuccess) return fail(hip_fail(hipGetLastError(), "hipMalloc counter"));
    if (hipMemsetAsync(c->d_rays, 0, kRaySlots * 64, c->stream) != hipSuccess) return fail(hip_fail(hipGetLastError(), "memset counter"));
    if (const char* v = getenv("VXRT_TILE_ORDER")) c->use_tile_order = atoi(v);
    if (const char* v = getenv("VXRT_TRACE_SPLIT")) c->trace_split = unsigned(strtoul(v, nullptr, 0));
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(hip_fail(hipGetLastError(), "sync"));
    *out = c;
    return VXRT_OK;
} VXRT_CATCH

int vxrt_destroy(vxrt_ctx* c) try {
    if (!c) return VXRT_OK;
    (void)hipSetDevice(c->cfg.device);
    for (hipStream_t t : c->trace_streams) if (t) (void)hipStreamSynchronize(t);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto* v : {&c->pending, &c->free_pairs})
        for (EventPair& p : *v) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    free_images(c);
    if (c->d_svo) (void)hipFree(c->d_svo);
    if (c->d_wide) (void)hipFree(c->d_wide);
    if (c->d_leaves) (void)hipFree(c->d_leaves);
    if (c->d_noise) (void)hipFree(c->d_noise);
    if (c->d_rays) (void)hipFree(c->d_rays);
    for (hipStream_t t : c->trace_streams) if (t && t != c->stream) (void)hipStreamDestroy(t);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return VXRT_OK;
} VXRT_CATCH

int vxrt_resize(vxrt_ctx* c, uint32_t width, uint32_t height) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    if (width == 0 || height == 0 || width > 65536 || height > 65536) { set_error("bad frame size"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = sync_all(c)) return rc;
    c->cfg.width = width;
    c->cfg.height = height;
    set_band(c, width, height);
    return alloc_images(c);  // new zeroed images: the history is gone (src/context.rs:1440-1448)
} VXRT_CATCH

int vxrt_set_voxels(vxrt_ctx* c, const int16_t (*pos)[3], const uint8_t (*mrgb)[4], size_t n) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    if (n != 0 && (!pos || !mrgb)) { set_error("null voxel arrays"); return VXRT_E_INVALID; }
    std::vector<Voxel> v(n);
    for (size_t i = 0; i < n; i++) {
        v[i].x = pos[i][0]; v[i].y = pos[i][1]; v[i].z = pos[i][2];
        v[i].m = mrgb[i][0]; v[i].r = mrgb[i][1]; v[i].g = mrgb[i][2]; v[i].b = mrgb[i][3];
    }
    return upload_scene(c, v.data(), n);
} VXRT_CATCH

int vxrt_load_vox_memory(vxrt_ctx* c, const uint8_t* bytes, size_t len) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    if (!bytes) { set_error("null bytes"); return VXRT_E_INVALID; }
    VoxScene scene;
    if (int rc = decode_vox(bytes, len, &scene)) return rc;
    return upload_scene(c, scene.voxels.data(), scene.voxels.size());
} VXRT_CATCH

int vxrt_load_vox(vxrt_ctx* c, const char* path) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    if (!path) { set_error("null path"); return VXRT_E_INVALID; }
    std::ifstream f(path, std::ios::binary);
    if (!f) { set_error(std::string("failed to read file: ") + path); return VXRT_E_IO; }
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return vxrt_load_vox_memory(c, bytes.data(), bytes.size());
} VXRT_CATCH

int vxrt_set_camera(vxrt_ctx* c, const float position[3], const float direction[3], float fov) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    if (!position || !direction) { set_error("null camera vectors"); return VXRT_E_INVALID; }
    memcpy(c->cam_pos, position, sizeof c->cam_pos);
    memcpy(c->cam_dir, direction, sizeof c->cam_dir);
    c->cam_fov = fov;
    return VXRT_OK;
} VXRT_CATCH

int vxrt_set_scene_params(vxrt_ctx* c, const vxrt_uniforms* u) try {
    if (!valid_ctx(c) || !u) { set_error("null argument"); return VXRT_E_INVALID; }
    uint32_t frame = c->uniforms.frame_number;
    c->uniforms = *u;
    c->uniforms.frame_number = frame;  // owned by the library, like update_bindings (src/context.rs:2152)
    return VXRT_OK;
} VXRT_CATCH

int vxrt_set_temporal(vxrt_ctx* c, const vxrt_temporal* t) try {
    if (!valid_ctx(c) || !t) { set_error("null argument"); return VXRT_E_INVALID; }
    c->temporal = *t;
    return VXRT_OK;
} VXRT_CATCH

int vxrt_set_denoise(vxrt_ctx* c, const vxrt_denoise* d) try {
    if (!valid_ctx(c) || !d) { set_error("null argument"); return VXRT_E_INVALID; }
    if (d->radius > 8) { set_error("denoise radius must be 0..8"); return VXRT_E_INVALID; }
    c->denoise = *d;
    return VXRT_OK;
} VXRT_CATCH

int vxrt_set_option(vxrt_ctx* c, vxrt_option option, uint32_t value) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    HIP_TRY(hipSetDevice(c->cfg.device));
    switch (option) {
        case VXRT_OPT_DENOISE_MODE:
            if (value > 1) { set_error("denoise mode must be 0 (exact) or 1 (tolerant)"); return VXRT_E_INVALID; }
            c->denoise_mode = int(value);
            return VXRT_OK;
        case VXRT_OPT_SCENE_FORMAT:
            if (value > 1) { set_error("scene format must be 0 (8-byte records) or 1 (wide records)"); return VXRT_E_INVALID; }
            if (value == 1 && c->has_scene && c->d_wide == nullptr) {
                set_error("the wide records are built when a scene is set: choose the format before vxrt_set_voxels / vxrt_set_menger");
                return VXRT_E_INVALID;
            }
            c->scene_format = int(value);
            return VXRT_OK;
        case VXRT_OPT_TAIL_CAPACITY:
            if (c->trace_variant < 4) return VXRT_OK;   // the other tracers' queues are sized for the worst case
            c->tail_capacity_override = int(value > 0x7fffffffu ? 0x7fffffffu : value);
            return resize_tail_queues(c, value == 0 ? (c->shard_capacity_max / 8u < 4096u ? 4096u : c->shard_capacity_max / 8u) : value);
        default:
            set_error("unknown option");
            return VXRT_E_INVALID;
    }
} VXRT_CATCH

int vxrt_reset_history(vxrt_ctx* c) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    c->has_history = false;
    c->old_cam_valid = false;
    return VXRT_OK;
} VXRT_CATCH

int vxrt_set_frame_number(vxrt_ctx* c, uint32_t frame_number) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    c->uniforms.frame_number = frame_number;
    return VXRT_OK;
} VXRT_CATCH

namespace {

int check_render(vxrt_ctx* c, uint32_t flags) {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    if ((flags & VXRT_ALL) == 0) { set_error("no stage selected"); return VXRT_E_INVALID; }
    if (!c->has_scene) { set_error("vxrt_render before any scene was set"); return VXRT_E_NOSCENE; }
    const bool multi = c->band.nranks > 1;
    if ((flags & VXRT_DENOISE) && multi && c->denoise.radius > 0 && c->band.band_rows % 16 != 0) {
        set_error("denoise with radius > 0 on row bands needs band_rows to be a multiple of 16 (its 16x16 tiles must not straddle bands)");
        return VXRT_E_INVALID;
    }
    if ((flags & VXRT_DENOISE) && multi && c->denoise.radius > 0 && (flags & (VXRT_TRACE | VXRT_TEMPORAL))) {
        set_error("multi-rank denoise with radius > 0 needs the halo: render TRACE|TEMPORAL, exchange, then DENOISE");
        return VXRT_E_INVALID;
    }
    if ((flags & VXRT_DENOISE) && multi && c->denoise.radius > 0 && !(c->halo_valid && c->halo_radius == c->denoise.radius)) {
        set_error("multi-rank denoise: halo not imported for this frame/radius");
        return VXRT_E_INVALID;
    }
    return VXRT_OK;
}

// Context::update_bindings for one frame (src/context.rs:2136-2162): old <- current, current <- camera, frame_number + 1
void update_bindings(vxrt_ctx* c) {
    c->old_cam = c->cam;
    CameraBasis basis = camera_axis_scaled(c->cam_dir, c->cam_fov, c->cfg.width, c->cfg.height);
    c->cam = make_cam(c->cam_pos, basis);
    for (int i = 0; i < 3; i++) {
        c->uniforms.camera_origin[i] = c->cam.o[i]; c->uniforms.camera_right[i] = c->cam.r[i];
        c->uniforms.camera_up[i] = c->cam.u[i]; c->uniforms.camera_forward[i] = c->cam.f[i];
    }
    c->uniforms.still_sample += 1;
    c->uniforms.frame_number += 1;  // wrapping
}

// The fields of TraceArgs that depend on the scene and the uniforms only (not on the frame slots or the launch).
void frame_constants(const vxrt_ctx* c, TraceArgs& a) {
    const vxrt_uniforms& u = c->uniforms;
    a.svo = c->d_svo; a.leaves = c->d_leaves; a.noise = c->d_noise;
    a.root_rec = c->root_rec;
    a.wide = c->d_wide;
    a.wide_root = c->wide_root;
    a.node_levels = int(c->depth) + 1;
    memcpy(a.root_center, c->root_center, sizeof a.root_center);
    a.root_size = c->root_size;
    a.band = c->band;
    a.max_bounces = int(c->cfg.max_bounces);
    a.launch_index = 0;
    a.stack_levels = c->depth < 1 ? 1 : int(c->depth);
    // voxels.comp:296 and the other per-frame constants, evaluated once with the same operations
    f3 sun_dir = mk3(vx_cos(u.sun_yaw) * vx_cos(u.sun_pitch), -vx_sin(u.sun_pitch), vx_sin(u.sun_yaw) * vx_cos(u.sun_pitch));
    f3 sun_n = norm3(sun_dir), neg_sun_n = norm3(-sun_dir);
    f3 sun_color = u.sun_strength * mk3(u.sun_color[0], u.sun_color[1], u.sun_color[2]);
    a.sun_dir[0] = sun_dir.x; a.sun_dir[1] = sun_dir.y; a.sun_dir[2] = sun_dir.z;
    a.sun_dir_n[0] = sun_n.x; a.sun_dir_n[1] = sun_n.y; a.sun_dir_n[2] = sun_n.z;
    a.neg_sun_dir_n[0] = neg_sun_n.x; a.neg_sun_dir_n[1] = neg_sun_n.y; a.neg_sun_dir_n[2] = neg_sun_n.z;
    a.sun_color[0] = sun_color.x; a.sun_color[1] = sun_color.y; a.sun_color[2] = sun_color.z;
    a.sky_color[0] = u.sky_color[0]; a.sky_color[1] = u.sky_color[1]; a.sky_color[2] = u.sky_color[2];
    a.sun_exponent = 1.0f / (u.sun_size * u.sun_size);
    // vx_pow(x, y) = vx_exp(y * vx_log(x)) is +0 once y * log(x) < -87.3: certain for x < exp(-88 / y) * (1 - 1e-4), which leaves
    // 0.7 + 1e-4 y of margin in the exponent against vx_log's error of ~1e-6 |log x| and the product's rounding.
    a.sun_zero_below = 0.0f;
    if (a.sun_exponent > 1.0f && a.sun_exponent < 1e6f) a.sun_zero_below = float(exp(-88.0 / double(a.sun_exponent)) * (1.0 - 1e-4));
    a.sun_size = u.sun_size; a.sun_strength = u.sun_strength; a.emit_strength = u.emit_strength; a.specularity = u.specularity;
}

// The trace stage of the next g frames (parameters at rest) as ONE launch of the tracer: g ring slots, frame numbers
// frame_number+1 .. +g.  g > 1 only with the trace_kernel-based tracers (1, 4, 5).  slots[k] = ring slot of frame k.
// path: optional g camera poses (position, direction), one per frame; null = the camera stays where it is.  cams / olds
// (g entries each): the camera and the "old" camera of every frame, as temporal.comp and denoise.comp of that frame see them.
int trace_frames(vxrt_ctx* c, uint32_t g, bool timed, int* slots, Cam* cams, Cam* olds, const float (*path_pos)[3] = nullptr,
                 const float (*path_dir)[3] = nullptr, uint32_t gbuf_frames = 0xffffffffu) {
    const uint32_t first_frame_number = c->uniforms.frame_number + 1;
    for (uint32_t k = 0; k < g; k++) {
        if (path_pos) {
            memcpy(c->cam_pos, path_pos[k], sizeof c->cam_pos);
            memcpy(c->cam_dir, path_dir[k], sizeof c->cam_dir);
        }
        update_bindings(c);
        cams[k] = c->cam;
        olds[k] = c->old_cam;
    }
    // next frame slots (never the temporal history) and the trace stream of this launch
    int s = c->slot;
    for (uint32_t k = 0; k < g; k++) {
        s = (s + 1) % int(c->ring.size());
        if (c->has_history && s == c->hist_slot) s = (s + 1) % int(c->ring.size());
        slots[k] = s;
    }
    const size_t lane = size_t(c->trace_launches % uint64_t(c->inflight));
    hipStream_t ts = c->trace_streams[lane];
    vxrt_ctx::TileSchedule& sched = c->schedules[lane];
    for (uint32_t k = 0; k < g; k++) {
        vxrt_ctx::Slot& sl = c->ring[size_t(slots[k])];
        if (sl.last_use_recorded) HIP_TRY(hipStreamWaitEvent(ts, sl.last_use, 0));
    }

    TraceArgs a;
    frame_constants(c, a);
    for (uint32_t k = 0; k < g; k++) {
        const vxrt_ctx::Slot& sl = c->ring[size_t(slots[k])];
        a.out[k] = FrameOut{sl.sampled_color, sl.nd, sl.albedo};
    }
    a.out_color = a.out[0].color; a.out_nd = a.out[0].nd; a.out_albedo = a.out[0].albedo;
    a.batch = int(g);
    a.gbuf_frames = gbuf_frames;
    a.ray_counter = c->d_rays;
    a.tile_order = (c->use_tile_order && sched.valid) ? sched.order : nullptr;
    a.tile_cost = c->use_tile_order ? sched.cost : nullptr;
    a.frame_number = first_frame_number;
    a.cam = cams[0];
    for (uint32_t k = 0; k < g; k++) a.cams[k] = cams[k];
    if (c->band.local_rows > 0) {
        EventPair p;
        if (timed) { p = take_pair(c, 0); HIP_TRY(hipEventRecord(p.a, ts)); }
        a.tail = PathQueue{nullptr, nullptr, 0};
        a.tail_zero = nullptr;
        a.tail_from = 0;
        const size_t scene_bytes = c->svo_count * sizeof(SvoRecord) + c->leaf_count * sizeof(int32_t);
        // tracer 0 (auto) takes the all-in-one kernel (a) for scenes beyond the Infinity Cache (see auto_tracer) and (b) for ONE frame at
        // a time on one stream — the latency case of a render loop that calls vxrt_render per frame: the head + tail pair waits twice
        // for a longest wave (0.357 ms per 1080p bench frame), the single kernel once (0.267 ms); with frames in flight or several
        // frames per launch the pair wins (0.123 ms per frame at 16 x 2)
        const bool one_at_a_time = g == 1 && c->inflight == 1;
        const int variant = (c->auto_tracer && (scene_bytes > (size_t(256) << 20) || one_at_a_time)) ? 0 : c->trace_variant;
        if (variant == 0 || variant >= 4) {
            if (variant >= 4) {
                // count sets rotate as in launch_trace_wavefront: launch J reads set J%3, writes (J+1)%3, clears (J+2)%3
                if (int rc = grow_tail_queues(c, lane, ts)) return rc;
                vxrt_ctx::StreamQueues& sq = c->queues[lane];
                unsigned* sets[3] = {sq.counts3, sq.counts3 + 64 * 16, sq.counts3 + 2 * 64 * 16};
                const unsigned J = sq.launches;
                a.tail = PathQueue{sq.hitq[0], sets[(J + 1) % 3], c->shard_capacity};
                a.tail_zero = sets[(J + 2) % 3];
                a.tail_from = c->tail_from;
                HIP_TRY(launch_trace(a, use_wide(c) && c->trace_variant == 4, ts));
                sq.launches = J + 1;
                if (!sq.counts_pending) {   // how much room this launch wanted (the set stays untouched until launch J + 2 clears it)
                    HIP_TRY(hipMemcpyAsync(sq.host_counts, sets[(J + 1) % 3], 64 * 64, hipMemcpyDeviceToHost, ts));
                    HIP_TRY(hipEventRecord(sq.counts_ready, ts));
                    sq.counts_pending = true;
                }
                if (c->trace_variant == 5) {
                    HIP_TRY(launch_paths(a, a.tail, sets[J % 3], c->tail_from, c->path_blocks, ts));
                    sq.launches = J + 2;
                } else {
                    // without a second queue (tail_split == 0) nothing is appended to queues[1]: its capacity 0 says so
                    PathQueue queues[2] = {{sq.hitq[0], nullptr, c->shard_capacity}, {sq.hitq[1], nullptr, sq.hitq[1] ? c->shard_capacity : 0u}};
                    HIP_TRY(launch_bounces(a, use_wide(c), queues, sets, &sq.launches, c->trace_blocks, sq.hitq[1] ? c->tail_split : 0u, c->tail_from, ts));
                }
            } else {
                HIP_TRY(launch_trace(a, use_wide(c), ts));
            }
            if (timed) HIP_TRY(hipEventRecord(p.b, ts));
            // Re-sort the tiles for this stream's coming frames from the costs just measured: after its first
            // frame, then every 8th (costs keep accumulating as a running maximum in between; ~7 us per sort).
            if (c->use_tile_order && (!sched.valid || sched.age >= 8)) {
                const unsigned tiles = trace_tile_count(c->band.width, c->band.local_rows);
                HIP_TRY(launch_tile_order(sched.cost, sched.order, sched.last_cost, sched.scratch, tiles, ts));
                sched.valid = true;
                sched.age = 0;
            }
            sched.age++;
        } else {
            vxrt_ctx::StreamQueues& sq = c->queues[lane];
            PathQueue queues[2] = {{sq.hitq[0], nullptr, c->shard_capacity}, {sq.hitq[1], nullptr, c->shard_capacity}};
            unsigned* sets[3] = {sq.counts3, sq.counts3 + 64 * 16, sq.counts3 + 2 * 64 * 16};
            if (c->trace_variant == 2)
                HIP_TRY(launch_trace_wavefront(a, queues, sets, &sq.launches, c->trace_blocks, c->trace_split, ts));
            else
                HIP_TRY(launch_trace_rayqueue(a, queues[0], sets, &sq.launches, sq.rq, c->shade_blocks, c->trace_blocks, c->rays_per_wave, ts));
            if (timed) HIP_TRY(hipEventRecord(p.b, ts));
        }
        if (timed) c->pending.push_back(p);
    }
    for (uint32_t k = 0; k < g; k++) {
        vxrt_ctx::Slot& sl = c->ring[size_t(slots[k])];
        HIP_TRY(hipEventRecord(sl.trace_done, ts));
        HIP_TRY(hipEventRecord(sl.last_use, ts));  // until a later stage reads the slot, the trace is its last use
        sl.last_use_recorded = true;
    }
    c->slot = slots[g - 1];
    c->last_schedule = int(lane);
    c->trace_launches += 1;
    c->traced += g;
    c->frames += g;
    c->pixels += uint64_t(g) * uint64_t(c->band.local_rows) * c->band.width;
    if (timed) { c->timed_frames += g; c->timed_launches += 1; }
    c->accum_is_sampled = true;
    c->halo_valid = false;
    return VXRT_OK;
}

// temporal / denoise of the frame in ring slot c->slot, in the reference's order (src/context.rs:2028-2043)
int post_stages(vxrt_ctx* c, uint32_t flags, bool timed) {
    const bool multi = c->band.nranks > 1;
    vxrt_ctx::Slot& cur = c->ring[size_t(c->slot)];
    const bool post = (flags & (VXRT_TEMPORAL | VXRT_DENOISE)) != 0;
    if (post && c->traced > 0) HIP_TRY(hipStreamWaitEvent(c->stream, cur.trace_done, 0));

    bool fused_denoise = false;
    if (flags & VXRT_TEMPORAL) {
        vxrt_ctx::Slot& hist = c->ring[size_t(c->hist_slot >= 0 ? c->hist_slot : c->slot)];
        TemporalArgs a;
        a.sampled_color = cur.sampled_color; a.new_nd = cur.nd;
        a.old_color = c->accum[c->cur ^ 1]; a.old_nd = hist.nd;
        a.new_color = c->accum[c->cur];
        a.band = c->band;
        a.cam = c->cam;
        a.old_cam = c->old_cam;
        a.has_history = (c->has_history && c->old_cam_valid && c->hist_slot >= 0) ? 1 : 0;
        const bool apron = multi && a.has_history && c->halo != nullptr && c->halo_radius >= 1 && c->halo_epoch == c->temporal_count;
        a.halo = apron ? c->halo : nullptr;
        a.halo_radius = apron ? int(c->halo_radius) : 0;
        c->temporal_count += 1;
        memset(a.inv, 0, sizeof a.inv);
        if (a.has_history) affine_inverse(c->old_cam, a.inv);
        a.sample_blending = c->temporal.sample_blending;
        a.maximum_blending = c->temporal.maximum_blending;
        a.blending_distance_cutoff = c->temporal.blending_distance_cutoff;
        // radius 0: the denoise stage is a per-pixel function of this stage's output — do it in the same pass
        fused_denoise = (flags & VXRT_DENOISE) != 0 && c->denoise.radius == 0;
        a.albedo = fused_denoise ? cur.albedo : nullptr;
        a.denoised = c->denoised;
        a.albedo_factor = c->denoise.albedo_factor;
        if (c->band.local_rows > 0) {
            EventPair p;
            if (timed) { p = take_pair(c, 1); HIP_TRY(hipEventRecord(p.a, c->stream)); }
            HIP_TRY(launch_temporal(a, c->stream));
            if (timed) { HIP_TRY(hipEventRecord(p.b, c->stream)); c->pending.push_back(p); }
        }
        if (c->hist_slot >= 0 && c->hist_slot != c->slot) {  // the old history slot may be traced into again after this
            HIP_TRY(hipEventRecord(hist.last_use, c->stream));
            hist.last_use_recorded = true;
        }
        c->accum_is_sampled = false;
        c->last = c->cur;
        c->cur ^= 1;  // hand the G-buffer over: what was written becomes the history (src/context.rs:2041-2043)
        c->hist_slot = c->slot;
        c->has_history = true;
    }

    if ((flags & VXRT_DENOISE) && !fused_denoise) {
        DenoiseArgs a;
        a.colors = c->accum_is_sampled ? cur.sampled_color : c->accum[c->last];
        a.nd = cur.nd;
        a.albedo = cur.albedo;
        a.output = c->denoised;
        a.halo = (multi && c->denoise.radius > 0) ? c->halo : nullptr;
        a.band = c->band;
        a.cam = c->cam;
        a.radius = c->denoise.radius;
        a.sigma_distance_2 = 2.0f * (c->denoise.sigma_distance * c->denoise.sigma_distance);  // denoise.comp:39-40
        a.sigma_range_2 = 2.0f * (c->denoise.sigma_range * c->denoise.sigma_range);
        a.albedo_factor = c->denoise.albedo_factor;
        a.mode = c->denoise_mode;
        if (c->band.local_rows > 0) {
            EventPair p;
            if (timed) { p = take_pair(c, 2); HIP_TRY(hipEventRecord(p.a, c->stream)); }
            HIP_TRY(launch_denoise(a, c->stream));
            if (timed) { HIP_TRY(hipEventRecord(p.b, c->stream)); c->pending.push_back(p); }
        }
    }
    if (post) {
        HIP_TRY(hipEventRecord(cur.last_use, c->stream));
        cur.last_use_recorded = true;
    }
    return VXRT_OK;
}

}  // namespace

int vxrt_render(vxrt_ctx* c, uint32_t flags) try {
    if (int rc = check_render(c, flags)) return rc;
    HIP_TRY(hipSetDevice(c->cfg.device));
    const bool timed = (flags & VXRT_TIMED) != 0;
    if (flags & VXRT_TRACE) {
        int slot = 0;
        Cam cam, old;
        if (int rc = trace_frames(c, 1, timed, &slot, &cam, &old)) return rc;
    }
    if (int rc = post_stages(c, flags, timed)) return rc;
    if (flags & VXRT_TRACE) c->old_cam_valid = true;  // the next frame's "old" camera is this frame's
    return VXRT_OK;
} VXRT_CATCH

// `count` frames with the parameters at rest.  With vxrt_config.frames_per_launch = B > 1 the trace stage of up to B
// consecutive frames is one launch (see trace_frames); temporal / denoise then run per frame, in frame order.
// `count` frames; frame k through camera pose k of `path_pos` / `path_dir` (null: the camera at rest), one launch per `batch` frames
static int render_sequence(vxrt_ctx* c, uint32_t flags, uint32_t count, const float (*path_pos)[3], const float (*path_dir)[3]) {
    const bool timed = (flags & VXRT_TIMED) != 0;
    const uint32_t batch = (flags & VXRT_TRACE) && (c->trace_variant == 0 || c->trace_variant >= 4) ? uint32_t(c->batch) : 1u;
    if (batch <= 1) {
        for (uint32_t i = 0; i < count; i++) {
            if (path_pos) {
                memcpy(c->cam_pos, path_pos[i], sizeof c->cam_pos);
                memcpy(c->cam_dir, path_dir[i], sizeof c->cam_dir);
            }
            if (int rc = vxrt_render(c, flags)) return rc;
        }
        return VXRT_OK;
    }
    for (uint32_t done = 0; done < count;) {
        const uint32_t g = count - done < batch ? count - done : batch;
        int slots[kMaxBatch];
        Cam cams[kMaxBatch], olds[kMaxBatch];
        if (int rc = trace_frames(c, g, timed, slots, cams, olds, path_pos ? path_pos + done : nullptr, path_dir ? path_dir + done : nullptr)) return rc;
        for (uint32_t k = 0; k < g; k++) {   // temporal / denoise of frame k see frame k's cameras
            c->slot = slots[k];
            c->cam = cams[k];
            c->old_cam = olds[k];
            if (int rc = post_stages(c, flags, timed)) return rc;
            c->old_cam_valid = true;
        }
        done += g;
    }
    return VXRT_OK;
}

// `count` frames with the parameters at rest.  With vxrt_config.frames_per_launch = B > 1 the trace stage of up to B
// consecutive frames is one launch (see trace_frames); temporal / denoise then run per frame, in frame order.
int vxrt_render_frames(vxrt_ctx* c, uint32_t flags, uint32_t count) try {
    if (int rc = check_render(c, flags)) return rc;
    HIP_TRY(hipSetDevice(c->cfg.device));
    return render_sequence(c, flags, count, nullptr, nullptr);
} VXRT_CATCH

// `count` frames along a camera path: frame k = vxrt_set_camera(positions[k], directions[k], fov) + vxrt_render(flags), with the
// trace stage of up to frames_per_launch consecutive frames in one launch.
int vxrt_render_path(vxrt_ctx* c, uint32_t flags, uint32_t count, const float (*positions)[3], const float (*directions)[3], float fov) try {
    if (int rc = check_render(c, flags)) return rc;
    if (!positions || !directions) { set_error("null argument"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    c->cam_fov = fov;
    return render_sequence(c, flags, count, positions, directions);
} VXRT_CATCH

// One displayed frame of `spp` samples per pixel: `spp` consecutive trace frames with the parameters at rest
// (frame_number advances by spp), their colours averaged — summed in frame order, divided once — into the last frame's slot,
// then temporal / denoise once on that.  The first hit (normal/depth, albedo/node) is the same in every sample.
int vxrt_render_spp(vxrt_ctx* c, uint32_t flags, uint32_t spp) try {
    if (int rc = check_render(c, flags)) return rc;
    if (!(flags & VXRT_TRACE) || spp == 0 || spp > 4096) { set_error("vxrt_render_spp needs VXRT_TRACE and 1 <= spp <= 4096"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    const bool timed = (flags & VXRT_TIMED) != 0;
    const size_t pixels = size_t(c->band.local_rows) * c->band.width;
    if (spp > 1 && c->spp_sum == nullptr && pixels > 0) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->spp_sum), pixels * sizeof(float4)));
    const uint32_t batch = (c->trace_variant == 0 || c->trace_variant >= 4) ? uint32_t(c->batch) : 1u;
    Cam first_old{};
    for (uint32_t done = 0; done < spp;) {
        const uint32_t g = spp - done < batch ? spp - done : batch;
        int slots[kMaxBatch];
        Cam cams[kMaxBatch], olds[kMaxBatch];
        // the samples' first hits are identical: only the sample whose slot stays current writes normal/depth and albedo/node
        const uint32_t gbuf = done + g == spp ? 1u << (g - 1u) : 0u;
        if (int rc = trace_frames(c, g, timed, slots, cams, olds, nullptr, nullptr, gbuf)) return rc;
        if (done == 0) first_old = olds[0];
        if (spp > 1 && pixels > 0) {
            SppArgs a{};
            for (uint32_t k = 0; k < g; k++) {
                vxrt_ctx::Slot& sl = c->ring[size_t(slots[k])];
                HIP_TRY(hipStreamWaitEvent(c->stream, sl.trace_done, 0));
                a.frames[k] = sl.sampled_color;
            }
            a.sum = c->spp_sum;
            a.out = c->ring[size_t(slots[g - 1])].sampled_color;
            a.pixels = pixels;
            a.count = int(g); a.first = done == 0; a.last = done + g == spp; a.total = int(spp);
            HIP_TRY(launch_spp_accumulate(a, c->stream));
            for (uint32_t k = 0; k < g; k++) {  // the slots may be traced into again only after this pass has read them
                vxrt_ctx::Slot& sl = c->ring[size_t(slots[k])];
                HIP_TRY(hipEventRecord(sl.last_use, c->stream));
                sl.last_use_recorded = true;
            }
        }
        done += g;
    }
    const Cam at_rest = c->cam;
    c->old_cam = first_old;
    if (int rc = post_stages(c, flags, timed)) return rc;
    c->old_cam_valid = true;
    c->old_cam = spp > 1 ? at_rest : first_old;
    return VXRT_OK;
} VXRT_CATCH

int vxrt_sync(vxrt_ctx* c) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = sync_all(c)) return rc;
    return resolve_events(c);
} VXRT_CATCH

static float4* image_ptr(vxrt_ctx* c, vxrt_image which) {
    switch (which) {
        case VXRT_SAMPLED_COLOR: return c->ring[size_t(c->slot)].sampled_color;
        case VXRT_NORMAL_DEPTH: return c->ring[size_t(c->slot)].nd;
        case VXRT_ALBEDO_NODE: return c->ring[size_t(c->slot)].albedo;
        case VXRT_ACCUM_COLOR: return c->accum_is_sampled ? c->ring[size_t(c->slot)].sampled_color : c->accum[c->last];
        case VXRT_DENOISED: return c->denoised;
        default: return nullptr;
    }
}

int vxrt_read(vxrt_ctx* c, vxrt_image which, float* dst, size_t bytes) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    float4* src = image_ptr(c, which);
    if (!src || !dst) { set_error("bad image or null destination"); return VXRT_E_INVALID; }
    if (bytes != image_bytes(c)) { set_error("vxrt_read: bytes must equal local_rows*width*16"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = sync_all(c)) return rc;
    if (bytes) HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return VXRT_OK;
} VXRT_CATCH

int vxrt_device_image(vxrt_ctx* c, vxrt_image which, void** device_ptr, size_t* bytes) try {
    if (!valid_ctx(c) || !device_ptr) { set_error("null argument"); return VXRT_E_INVALID; }
    float4* src = image_ptr(c, which);
    if (!src) { set_error("bad image"); return VXRT_E_INVALID; }
    *device_ptr = src;
    if (bytes) *bytes = image_bytes(c);
    return VXRT_OK;
} VXRT_CATCH

int vxrt_local_rows(const vxrt_ctx* c, uint32_t* count, uint32_t* rows) try {
    if (!valid_ctx(c) || !count) { set_error("null argument"); return VXRT_E_INVALID; }
    const BandMap& b = c->band;
    uint32_t n = 0;
    for (int y = 0; y < b.height; y++)
        if ((y / b.band_rows) % b.nranks == b.rank) {
            if (rows) rows[n] = uint32_t(y);
            n++;
        }
    *count = n;
    return VXRT_OK;
} VXRT_CATCH

int vxrt_get_stats(vxrt_ctx* c, vxrt_stats* out) try {
    if (!valid_ctx(c) || !out) { set_error("null argument"); return VXRT_E_INVALID; }
    if (int rc = vxrt_sync(c)) return rc;
    std::vector<unsigned long long> slots(size_t(kRaySlots) * 8);
    HIP_TRY(hipMemcpy(slots.data(), c->d_rays, slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long rays = 0;
    for (size_t i = 0; i < slots.size(); i += 8) rays += slots[i];
    memset(out, 0, sizeof *out);
    out->frames = c->frames;
    out->rays = rays;
    out->pixels = c->pixels;
    out->trace_ms = c->ms[0];
    out->temporal_ms = c->ms[1];
    out->denoise_ms = c->ms[2];
    out->timed_frames = c->timed_frames;
    out->timed_launches = c->timed_launches;
    out->scene_bytes = c->svo_count * sizeof(SvoRecord) + c->leaf_count * sizeof(int32_t);
    out->noise_bytes = kNoiseCount * sizeof(float);
    out->local_rows = uint32_t(c->band.local_rows);
    out->octree_depth = c->depth;
    out->octree_nodes = c->svo_count;
    for (size_t lane = 0; lane < c->queues.size(); lane++)   // fold in what the last launches wanted (the GPU is idle here)
        if (c->trace_variant >= 4) { if (int rc = grow_tail_queues(c, lane, nullptr)) return rc; }
    out->wide_nodes = c->wide_count;
    out->scene_format = use_wide(c) ? 1u : 0u;
    out->queue_bytes = c->queue_bytes;
    out->queue_overflow_paths = c->queue_overflow_paths;
    return VXRT_OK;
} VXRT_CATCH

int vxrt_reset_stats(vxrt_ctx* c) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    if (int rc = vxrt_sync(c)) return rc;
    // on the context's own stream and waited for: a null-stream hipMemset is neither ordered against the
    // non-blocking trace streams nor guaranteed to have finished when it returns
    HIP_TRY(hipMemsetAsync(c->d_rays, 0, kRaySlots * 64, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->frames = c->pixels = c->timed_frames = c->timed_launches = 0;
    c->queue_overflow_paths = 0;
    c->ms[0] = c->ms[1] = c->ms[2] = 0.0;
    return VXRT_OK;
} VXRT_CATCH

// Diagnostics: shader-clock duration of every 16x16 tile in the last traced frame (monolithic kernel).
int vxrt_debug_tile_costs(vxrt_ctx* c, uint32_t* out, size_t n) try {
    if (!valid_ctx(c) || !out) { set_error("null argument"); return VXRT_E_INVALID; }
    // reported per 16x16 pixels whatever the kernel's own tile is: the maximum over the kernel tiles inside
    const size_t out_x = size_t((c->band.width + 15) / 16), out_y = size_t((c->band.local_rows + 15) / 16);
    if (n != out_x * out_y || c->schedules.empty()) { set_error("tile count mismatch"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = sync_all(c)) return rc;
    int tw = 16, th = 16;
    trace_tile_dims(&tw, &th);
    const size_t kx = size_t((c->band.width + tw - 1) / tw), ky = size_t((c->band.local_rows + th - 1) / th);
    std::vector<uint32_t> raw(kx * ky);
    HIP_TRY(hipMemcpy(raw.data(), c->schedules[size_t(c->last_schedule)].last_cost, raw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) out[i] = 0;
    for (size_t y = 0; y < ky; y++)
        for (size_t x = 0; x < kx; x++) {
            uint32_t& o = out[(y * size_t(th) / 16) * out_x + x * size_t(tw) / 16];
            o = raw[y * kx + x] > o ? raw[y * kx + x] : o;
        }
    return VXRT_OK;
} VXRT_CATCH

// ---- denoise halo ---------------------------------------------------------------------------------
// Message to a neighbour: for each of ITS local bands j (up to max_bands), r rows x 3 images x width
// float4.  A band's "above" rows come from the previous rank, its "below" rows from the next rank.
static int max_bands(const BandMap& b) {
    int bands = (b.height + b.band_rows - 1) / b.band_rows;
    return (bands + b.nranks - 1) / b.nranks;
}

int vxrt_halo_bytes(vxrt_ctx* c, size_t* bytes) try {
    if (!valid_ctx(c) || !bytes) { set_error("null argument"); return VXRT_E_INVALID; }
    *bytes = size_t(max_bands(c->band)) * c->denoise.radius * 3 * c->band.width * sizeof(float4);
    return VXRT_OK;
} VXRT_CATCH

int vxrt_halo_export(vxrt_ctx* c, void* dev_to_prev, void* dev_to_next) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    const BandMap& b = c->band;
    const int r = int(c->denoise.radius);
    if (b.nranks < 2 || r == 0) return VXRT_OK;
    if (!dev_to_prev || !dev_to_next) { set_error("null halo buffer"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    const vxrt_ctx::Slot& cur = c->ring[size_t(c->slot)];
    if (c->traced > 0) HIP_TRY(hipStreamWaitEvent(c->stream, cur.trace_done, 0));
    const float4* imgs[3] = {c->accum_is_sampled ? cur.sampled_color : c->accum[c->last], cur.nd, cur.albedo};
    const size_t row_bytes = size_t(b.width) * sizeof(float4);
    const int nlb = local_band_count(b);
    for (int lb = 0; lb < nlb; lb++) {
        const int gb = lb * b.nranks + b.rank;
        const int y0 = gb * b.band_rows;
        const int rows = (y0 + b.band_rows <= b.height) ? b.band_rows : b.height - y0;
        // my top rows are the "below" rows of band gb-1 (previous rank, its local band (gb-1)/nranks)
        if (gb >= 1) {
            const int j = (gb - 1) / b.nranks;
            for (int k = 0; k < r && k < rows; k++)
                for (int im = 0; im < 3; im++) {
                    float4* dst = static_cast<float4*>(dev_to_prev) + (size_t(j * r + k) * 3 + im) * b.width;
                    HIP_TRY(hipMemcpyAsync(dst, imgs[im] + size_t(lb * b.band_rows + k) * b.width, row_bytes, hipMemcpyDeviceToDevice, c->stream));
                }
        }
        // my bottom rows are the "above" rows of band gb+1 (next rank, its local band (gb+1)/nranks)
        if (rows == b.band_rows && (gb + 1) * b.band_rows < b.height) {
            const int j = (gb + 1) / b.nranks;
            for (int k = 0; k < r; k++)
                for (int im = 0; im < 3; im++) {
                    float4* dst = static_cast<float4*>(dev_to_next) + (size_t(j * r + k) * 3 + im) * b.width;
                    HIP_TRY(hipMemcpyAsync(dst, imgs[im] + size_t(lb * b.band_rows + (b.band_rows - r) + k) * b.width, row_bytes, hipMemcpyDeviceToDevice, c->stream));
                }
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VXRT_OK;
} VXRT_CATCH

int vxrt_halo_import(vxrt_ctx* c, const void* dev_from_prev, const void* dev_from_next) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    const BandMap& b = c->band;
    const int r = int(c->denoise.radius);
    if (b.nranks < 2 || r == 0) return VXRT_OK;
    if (!dev_from_prev || !dev_from_next) { set_error("null halo buffer"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    const int nlb = local_band_count(b);
    const size_t per_side = size_t(r) * 3 * b.width;  // float4 per (band, side)
    if (c->halo == nullptr || c->halo_radius != uint32_t(r)) {
        if (c->halo) (void)hipFree(c->halo);
        c->halo = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->halo), size_t(nlb > 0 ? nlb : 1) * 2 * per_side * sizeof(float4)));
    }
    for (int lb = 0; lb < nlb; lb++) {
        // side 0 ("above") was sent by the previous rank as its to_next message slot lb, side 1 by the next rank
        HIP_TRY(hipMemcpyAsync(c->halo + size_t(lb * 2 + 0) * per_side, static_cast<const float4*>(dev_from_prev) + size_t(lb) * per_side,
                               per_side * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->halo + size_t(lb * 2 + 1) * per_side, static_cast<const float4*>(dev_from_next) + size_t(lb) * per_side,
                               per_side * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    }
    // the caller's buffers are borrowed for this call only (they are torch tensors that may be freed or re-used at once)
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->halo_radius = uint32_t(r);
    c->halo_valid = true;
    c->halo_epoch = c->temporal_count;
    return VXRT_OK;
} VXRT_CATCH

// ---- host-only helpers -------------------------------------------------------------------------------
int vxrt_vox_to_voxels(const uint8_t* bytes, size_t len, int16_t (*pos)[3], uint8_t (*mrgb)[4], size_t cap, size_t* n,
                       uint32_t size_xyz[3]) try {
    if (!bytes || !n) { set_error("null argument"); return VXRT_E_INVALID; }
    VoxScene scene;
    if (int rc = decode_vox(bytes, len, &scene)) return rc;
    *n = scene.voxels.size();
    if (size_xyz) memcpy(size_xyz, scene.size, sizeof scene.size);
    for (size_t i = 0; i < scene.voxels.size() && i < cap; i++) {
        const Voxel& v = scene.voxels[i];
        if (pos) { pos[i][0] = v.x; pos[i][1] = v.y; pos[i][2] = v.z; }
        if (mrgb) { mrgb[i][0] = v.m; mrgb[i][1] = v.r; mrgb[i][2] = v.g; mrgb[i][3] = v.b; }
    }
    return VXRT_OK;
} VXRT_CATCH

int vxrt_build_octree(const int16_t (*pos)[3], const uint8_t (*mrgb)[4], size_t n, int32_t* words, size_t cap, size_t* n_words,
                      uint32_t* depth) try {
    if (!n_words || (n != 0 && (!pos || !mrgb))) { set_error("null argument"); return VXRT_E_INVALID; }
    std::vector<Voxel> v(n);
    for (size_t i = 0; i < n; i++) {
        v[i].x = pos[i][0]; v[i].y = pos[i][1]; v[i].z = pos[i][2];
        v[i].m = mrgb[i][0]; v[i].r = mrgb[i][1]; v[i].g = mrgb[i][2]; v[i].b = mrgb[i][3];
    }
    Octree tree;
    if (int rc = build_octree(v.data(), n, &tree)) return rc;
    *n_words = tree.words.size();
    if (depth) *depth = tree.depth;
    if (words && cap >= tree.words.size()) memcpy(words, tree.words.data(), tree.words.size() * sizeof(int32_t));
    return VXRT_OK;
} VXRT_CATCH

// The scene as the kernels read it, for a voxel list (host only): 8-byte records, wide records, leaf words.  Arrays may be null
// (sizes only); nothing is written past the caps.
int vxrt_build_records(const int16_t (*pos)[3], const uint8_t (*mrgb)[4], size_t n, uint32_t* svo, size_t svo_cap, size_t* n_svo,
                       uint32_t* wide, size_t wide_cap, size_t* n_wide, int32_t* leaves, size_t leaf_cap, size_t* n_leaves, uint32_t* depth) try {
    if (!n_svo || !n_wide || !n_leaves || (n != 0 && (!pos || !mrgb))) { set_error("null argument"); return VXRT_E_INVALID; }
    std::vector<Voxel> v(n);
    for (size_t i = 0; i < n; i++) {
        v[i].x = pos[i][0]; v[i].y = pos[i][1]; v[i].z = pos[i][2];
        v[i].m = mrgb[i][0]; v[i].r = mrgb[i][1]; v[i].g = mrgb[i][2]; v[i].b = mrgb[i][3];
    }
    Octree tree;
    if (int rc = build_octree(v.data(), n, &tree)) return rc;
    std::vector<SvoRecord> recs;
    std::vector<int32_t> lw;
    std::vector<WideRec> wr;
    if (int rc = flatten_svo(tree, &recs, &lw)) return rc;
    if (int rc = widen_svo(recs, tree.depth, &wr)) return rc;
    *n_svo = recs.size(); *n_wide = wr.size(); *n_leaves = lw.size();
    if (depth) *depth = tree.depth;
    if (svo && svo_cap >= recs.size()) memcpy(svo, recs.data(), recs.size() * sizeof(SvoRecord));
    if (wide && wide_cap >= wr.size()) memcpy(wide, wr.data(), wr.size() * sizeof(WideRec));
    if (leaves && leaf_cap >= lw.size()) memcpy(leaves, lw.data(), lw.size() * sizeof(int32_t));
    return VXRT_OK;
} VXRT_CATCH

int vxrt_camera_axis_scaled(const float position[3], const float direction[3], float fov, uint32_t width, uint32_t height,
                            float right[3], float up[3], float forward_ray[3]) try {
    (void)position;
    if (!direction || !right || !up || !forward_ray) { set_error("null argument"); return VXRT_E_INVALID; }
    CameraBasis b = camera_axis_scaled(direction, fov, width, height);
    memcpy(right, b.right, sizeof b.right);
    memcpy(up, b.up, sizeof b.up);
    memcpy(forward_ray, b.forward_ray, sizeof b.forward_ray);
    return VXRT_OK;
} VXRT_CATCH

int vxrt_noise_table(uint32_t seed, float* out, size_t n) try {
    if (!out) { set_error("null argument"); return VXRT_E_INVALID; }
    for (size_t i = 0; i < n; i++) out[i] = noise_value(seed, uint32_t(i));
    return VXRT_OK;
} VXRT_CATCH

int vxrt_menger_voxels_ex(uint32_t level, uint32_t clip, const uint8_t mrgb[4], uint32_t emissive_period, int16_t (*pos)[3],
                          uint8_t (*out_mrgb)[4], size_t cap, size_t* n) try {
    if (!n || !mrgb || level > 9) { set_error("bad argument"); return VXRT_E_INVALID; }
    uint32_t side = 1;
    for (uint32_t l = 0; l < level; l++) side *= 3;
    if (clip != 0 && clip < side) side = clip;
    if (side > 32767) { set_error("menger side exceeds i16"); return VXRT_E_INVALID; }
    size_t count = 0;
    for (uint32_t x = 0; x < side; x++)
        for (uint32_t y = 0; y < side; y++)
            for (uint32_t z = 0; z < side; z++)
                if (menger_solid(level, x, y, z)) {
                    if (count < cap) {
                        if (pos) { pos[count][0] = int16_t(x); pos[count][1] = int16_t(y); pos[count][2] = int16_t(z); }
                        if (out_mrgb) {
                            const int32_t w = procedural_leaf_word(x, y, z, mrgb, emissive_period);
                            out_mrgb[count][0] = uint8_t((uint32_t(w) >> 24) & 0x7fu);
                            out_mrgb[count][1] = mrgb[1]; out_mrgb[count][2] = mrgb[2]; out_mrgb[count][3] = mrgb[3];
                        }
                    }
                    count++;
                }
    *n = count;
    return VXRT_OK;
} VXRT_CATCH

int vxrt_menger_voxels(uint32_t level, const uint8_t mrgb[4], int16_t (*pos)[3], uint8_t (*out_mrgb)[4], size_t cap, size_t* n) try {
    return vxrt_menger_voxels_ex(level, 0, mrgb, 0, pos, out_mrgb, cap, n);
} VXRT_CATCH

int vxrt_set_menger(vxrt_ctx* c, uint32_t level, uint32_t clip, const uint8_t mrgb[4], uint32_t emissive_period) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    if (!mrgb) { set_error("null colour"); return VXRT_E_INVALID; }
    uint32_t side = 1;
    for (uint32_t l = 0; l < level && l < 10; l++) side *= 3;
    if (clip == 0 || clip > side) clip = side;
    uint32_t depth = 0;
    // On the device when the scene fits its dense sweep (side <= 2048: 1.15 GB of scratch) and only the 8-byte records are wanted:
    // the 2048^3 scene in a fraction of a second instead of ~9 s of host threads + a 5.6 GB upload.  VXRT_HOST_BUILD=1: the host builder.
    if (menger_device_build_supported(level, clip) && c->scene_format != 1 && getenv("VXRT_HOST_BUILD") == nullptr) {
        HIP_TRY(hipSetDevice(c->cfg.device));
        if (int rc = sync_all(c)) return rc;
        SvoRecord* svo = nullptr;
        int32_t* lw = nullptr;
        size_t nsvo = 0, nlw = 0;
        SvoRecord root{0, 0};
        if (int rc = build_menger_svo_device(level, clip, mrgb, emissive_period, c->stream, &svo, &nsvo, &lw, &nlw, &depth, &root)) return rc;
        if (c->d_svo) (void)hipFree(c->d_svo);
        if (c->d_leaves) (void)hipFree(c->d_leaves);
        if (c->d_wide) (void)hipFree(c->d_wide);
        c->d_svo = svo; c->d_leaves = lw; c->d_wide = nullptr;
        c->svo_count = nsvo; c->leaf_count = nlw; c->wide_count = 0;
        c->root_rec = root;
        c->wide_root = WideRec{0, 0, 0, 0};
        c->root_center[0] = c->root_center[1] = c->root_center[2] = 0.0f;
        c->root_size = float(1u << depth);
        c->depth = depth;
        c->has_scene = true;
        return VXRT_OK;
    }
    std::vector<SvoRecord> recs;
    std::vector<int32_t> leaves;
    if (int rc = build_menger_svo(level, clip, mrgb, emissive_period, &recs, &leaves, &depth)) return rc;
    return upload_svo(c, recs, leaves, depth);
} VXRT_CATCH

// Test hook: the scene as the device holds it (8-byte records: 2 words each; leaf words).  Null arrays: sizes only.
int vxrt_debug_read_scene(vxrt_ctx* c, uint32_t* svo, size_t svo_cap, size_t* n_svo, int32_t* leaves, size_t leaf_cap, size_t* n_leaves) try {
    if (!valid_ctx(c) || !n_svo || !n_leaves) { set_error("null argument"); return VXRT_E_INVALID; }
    if (!c->has_scene) { set_error("no scene set"); return VXRT_E_NOSCENE; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = sync_all(c)) return rc;
    *n_svo = c->svo_count;
    *n_leaves = c->leaf_count;
    if (svo && svo_cap >= c->svo_count) HIP_TRY(hipMemcpy(svo, c->d_svo, c->svo_count * sizeof(SvoRecord), hipMemcpyDeviceToHost));
    if (leaves && leaf_cap >= c->leaf_count) HIP_TRY(hipMemcpy(leaves, c->d_leaves, c->leaf_count * sizeof(int32_t), hipMemcpyDeviceToHost));
    return VXRT_OK;
} VXRT_CATCH

// ---- blue noise (include/vxrt_bluenoise.h, csrc/noise.hip, csrc/noise_zip.cpp) ----------------------------------
int vxrt_blue_noise(int32_t device, uint32_t seed, uint32_t size, uint32_t first_layer, uint32_t layers, float* out) try {
    if (!out || layers == 0) { set_error("null argument"); return VXRT_E_INVALID; }
    if (size < 16 || size > VXBN_MAX_SIZE || (size & (size - 1)) != 0) { set_error("blue-noise size must be a power of two in 16..128"); return VXRT_E_INVALID; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return hip_fail(e == hipSuccess ? hipErrorNoDevice : e, "hipGetDeviceCount");
    if (device < 0 || device >= ndev) { set_error("device ordinal out of range"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(device));
    const size_t bytes = size_t(layers) * size * size * sizeof(float);
    ScratchBuffer b;
    HIP_TRY(b.alloc(bytes));
    HIP_TRY(launch_blue_noise(b.as<float>(), seed, first_layer, layers, int(size), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, b.as<float>(), bytes, hipMemcpyDeviceToHost));
    return VXRT_OK;
} VXRT_CATCH

int vxrt_noise_zip_read(const char* path, float* out, size_t cap_floats, uint32_t* size, uint32_t* layers) try {
    if (!path || !size || !layers) { set_error("null argument"); return VXRT_E_INVALID; }
    std::vector<float> px;
    if (int rc = noise_zip_read(path, &px, size, layers)) return rc;
    if (out) {
        if (cap_floats < px.size()) { set_error("buffer too small for the archive's images"); return VXRT_E_INVALID; }
        memcpy(out, px.data(), px.size() * sizeof(float));
    }
    return VXRT_OK;
} VXRT_CATCH

int vxrt_noise_zip_write(const char* path, const float* table, uint32_t size, uint32_t layers) try {
    if (!path || !table) { set_error("null argument"); return VXRT_E_INVALID; }
    return noise_zip_write(path, table, size, layers);
} VXRT_CATCH

int vxrt_set_noise(vxrt_ctx* c, const float* table) try {
    if (!c || !table) { set_error("null argument"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = vxrt_sync(c)) return rc;
    HIP_TRY(hipMemcpy(c->d_noise, table, kNoiseCount * sizeof(float), hipMemcpyHostToDevice));
    return VXRT_OK;
} VXRT_CATCH

// ---- wider scene input (csrc/vox_scene.cpp) ------------------------------------------------------------------------
int vxrt_vox_scene_to_voxels(const uint8_t* bytes, size_t len, uint32_t flags, int16_t (*pos)[3], uint8_t (*mrgb)[4], size_t cap,
                             size_t* n, int32_t bounds_min[3], int32_t bounds_max[3]) try {
    if (!bytes || !n) { set_error("null argument"); return VXRT_E_INVALID; }
    if (flags & ~uint32_t(VXRT_VOX_ALL_MODELS | VXRT_VOX_LENIENT_MATERIALS | VXRT_VOX_REBASE)) { set_error("unknown flag"); return VXRT_E_INVALID; }
    VoxScene scene;
    int32_t lo[3], hi[3];
    if (int rc = decode_vox_scene(bytes, len, flags, &scene, lo, hi)) return rc;
    *n = scene.voxels.size();
    for (int a = 0; a < 3; a++) {
        if (bounds_min) bounds_min[a] = lo[a];
        if (bounds_max) bounds_max[a] = hi[a];
    }
    for (size_t i = 0; i < scene.voxels.size() && i < cap; i++) {
        const Voxel& v = scene.voxels[i];
        if (pos) { pos[i][0] = v.x; pos[i][1] = v.y; pos[i][2] = v.z; }
        if (mrgb) { mrgb[i][0] = v.m; mrgb[i][1] = v.r; mrgb[i][2] = v.g; mrgb[i][3] = v.b; }
    }
    return VXRT_OK;
} VXRT_CATCH

int vxrt_default_scene_voxels(uint32_t seed, int16_t (*pos)[3], uint8_t (*mrgb)[4], size_t cap, size_t* n) try {
    if (!n) { set_error("null argument"); return VXRT_E_INVALID; }
    std::vector<Voxel> voxels;
    default_scene(seed, &voxels);
    *n = voxels.size();
    for (size_t i = 0; i < voxels.size() && i < cap; i++) {
        const Voxel& v = voxels[i];
        if (pos) { pos[i][0] = v.x; pos[i][1] = v.y; pos[i][2] = v.z; }
        if (mrgb) { mrgb[i][0] = v.m; mrgb[i][1] = v.r; mrgb[i][2] = v.g; mrgb[i][3] = v.b; }
    }
    return VXRT_OK;
} VXRT_CATCH

// Test hook: cast_bounded_ray (voxels.comp:134-247) as the kernels implement it, for caller-given rays of the current scene.
int vxrt_debug_cast_rays(vxrt_ctx* c, const float* origins, const float* dirs, size_t n, uint8_t* hit, float* time, int32_t* node, float* normal) try {
    if (!valid_ctx(c) || !origins || !dirs || !hit || !time || !node || !normal) { set_error("null argument"); return VXRT_E_INVALID; }
    if (!c->has_scene) { set_error("no scene set"); return VXRT_E_NOSCENE; }
    if (n == 0) return VXRT_OK;
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = vxrt_sync(c)) return rc;
    ScratchBuffer b_o, b_d, b_out;
    HIP_TRY(b_o.alloc(n * 12));
    HIP_TRY(b_d.alloc(n * 12));
    HIP_TRY(b_out.alloc(n * 32));
    float *d_o = b_o.as<float>(), *d_d = b_d.as<float>(), *d_out = b_out.as<float>();
    HIP_TRY(hipMemcpy(d_o, origins, n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_d, dirs, n * 12, hipMemcpyHostToDevice));
    TraceArgs a{};
    a.svo = c->d_svo; a.leaves = c->d_leaves;
    a.root_rec = c->root_rec;
    a.wide = c->d_wide;
    a.wide_root = c->wide_root;
    a.node_levels = int(c->depth) + 1;
    memcpy(a.root_center, c->root_center, sizeof a.root_center);
    a.root_size = c->root_size;
    a.stack_levels = c->depth < 1 ? 1 : int(c->depth);
    HIP_TRY(launch_cast_probe(a, use_wide(c), d_o, d_d, d_out, unsigned(n), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<float> out(n * 8);
    HIP_TRY(hipMemcpy(out.data(), d_out, n * 32, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {
        hit[i] = out[8 * i] != 0.0f;
        time[i] = out[8 * i + 1];
        memcpy(&node[i], &out[8 * i + 2], 4);
        normal[3 * i] = out[8 * i + 3]; normal[3 * i + 1] = out[8 * i + 4]; normal[3 * i + 2] = out[8 * i + 5];
    }
    return VXRT_OK;
} VXRT_CATCH

// Test hook: the path of ONE pixel of the next frame (frame_number + 1, the camera as set), cast by cast, as the kernels compute it
// (cast_ray and shade_hit of trace_common.h, in voxels.comp's order).  log: 12 floats per cast = origin, direction, hit flag, time,
// bits(leaf word), normal; at most 32 casts.  Nothing is rendered and no context state changes, apart from the camera basis.
int vxrt_debug_path_log(vxrt_ctx* c, int32_t x, int32_t y, float* log, int32_t* casts) try {
    if (!valid_ctx(c) || !log || !casts) { set_error("null argument"); return VXRT_E_INVALID; }
    if (!c->has_scene) { set_error("no scene set"); return VXRT_E_NOSCENE; }
    if (x < 0 || y < 0 || x >= int(c->cfg.width) || y >= int(c->cfg.height)) { set_error("pixel outside the frame"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = vxrt_sync(c)) return rc;
    const Cam keep_cam = c->cam, keep_old = c->old_cam;
    const vxrt_uniforms keep_u = c->uniforms;
    update_bindings(c);
    TraceArgs a{};
    frame_constants(c, a);
    a.frame_number = c->uniforms.frame_number;
    a.cam = c->cam;
    a.batch = 1;
    c->cam = keep_cam; c->old_cam = keep_old; c->uniforms = keep_u;
    ScratchBuffer b_log;
    HIP_TRY(b_log.alloc((12 * 32 + 1) * sizeof(float)));
    float* d_log = b_log.as<float>();
    HIP_TRY(launch_path_log(a, use_wide(c), x, y, d_log, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<float> out(12 * 32 + 1);
    HIP_TRY(hipMemcpy(out.data(), d_log, out.size() * sizeof(float), hipMemcpyDeviceToHost));
    *casts = int32_t(out[12 * 32]);
    memcpy(log, out.data(), size_t(*casts) * 12 * sizeof(float));
    return VXRT_OK;
} VXRT_CATCH

// device-vs-host bit equality probe of include/vxrt_detmath.h (test hook; host arrays in and out)
int vxrt_detmath_probe(int32_t device, int32_t fn, const float* x, const float* y, float* out, size_t n) try {
    if (!x || !y || !out) { set_error("null argument"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(device));
    ScratchBuffer bx, by, bout;
    HIP_TRY(bx.alloc(n * 4));
    HIP_TRY(by.alloc(n * 4));
    HIP_TRY(bout.alloc(n * 4));
    float *dx = bx.as<float>(), *dy = by.as<float>(), *dout = bout.as<float>();
    HIP_TRY(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice));
    HIP_TRY(launch_detmath_probe(fn, dx, dy, dout, n, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
    return VXRT_OK;
} VXRT_CATCH

}  // extern "C"
