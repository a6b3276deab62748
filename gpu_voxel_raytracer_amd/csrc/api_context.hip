// api_context.hip — the context of libvxrt (include/vxrt.h): creation, image and queue memory, per-frame parameters, options,
// outputs and statistics.  Stands where Context::new / create_bindings / resize / update_bindings stand in the reference
// (src/context.rs:595-660, 936-1016, 1430-1461, 2136-2162); each entry point cites its call site in vxrt.h.
#include <chrono>

#include "ctx.h"

namespace vxrt {

size_t image_bytes(const vxrt_ctx* c) { return size_t(c->band.local_rows) * c->band.width * sizeof(float4); }

int count_local_rows(const BandMap& b) {
    int rows = 0;
    for (int gb = b.rank; gb < band_count(b); gb += b.nranks) {
        const int y0 = band_first_row(b, gb), n = band_nominal_rows(b, gb);
        rows += (y0 + n <= b.height) ? n : b.height - y0;
    }
    return rows;
}

int local_band_count(const BandMap& b) {
    const int bands = band_count(b);
    return bands <= b.rank ? 0 : (bands - b.rank + b.nranks - 1) / b.nranks;
}

void free_images(vxrt_ctx* c) {
    for (vxrt_ctx::Slot& sl : c->ring) {
        sl.sampled_color = sl.albedo = sl.nd = nullptr;   // pieces of ring_arena
        if (sl.own) (void)hipEventDestroy(sl.own);
    }
    if (c->ring_arena) (void)hipFree(c->ring_arena);
    c->ring_arena = nullptr;
    c->ring.clear();
    free_halo(c);
    float4** imgs[] = {&c->accum[0], &c->accum[1], &c->denoised, &c->spp_sum};
    for (float4** p : imgs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    for (vxrt_ctx::TileSchedule& t : c->schedules)
    {
        for (uint32_t** p : {&t.cost, &t.order, &t.last_cost, &t.scratch}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        if (t.host_heavy) (void)hipHostFree(t.host_heavy);
        if (t.heavy_ready) (void)hipEventDestroy(t.heavy_ready);
        t.host_heavy = nullptr; t.heavy_ready = nullptr;
    }
    c->schedules.clear();
    for (vxrt_ctx::StreamQueues& sq : c->queues) {
        for (float4** p : {&sq.hitq[0], &sq.hitq[1]}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        if (sq.counts3) (void)hipFree(sq.counts3);
        if (sq.rq_block) (void)hipFree(sq.rq_block);
        if (sq.host_counts) (void)hipHostFree(sq.host_counts);
        if (sq.host_ctl) (void)hipHostFree(sq.host_ctl);
        if (sq.fused_ctl) (void)hipFree(sq.fused_ctl);
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
    // ONE allocation for the whole ring (each image on a 2 MiB boundary): a hundred separate 33 MB allocations land wherever the
    // driver finds room, and the trace stage's rate then differs by up to 4 % from process to process (101.3 .. 105.6 ms per 960
    // bench frames, each process steady to 0.1 %)
    c->ring.resize(size_t(c->inflight) * size_t(c->batch) + 2);
    const size_t pitch = (bytes + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ring_arena), pitch * 3 * c->ring.size()));
    HIP_TRY(hipMemsetAsync(c->ring_arena, 0, pitch * 3 * c->ring.size(), c->stream));
    size_t piece = 0;
    for (vxrt_ctx::Slot& sl : c->ring) {
        for (float4** p : {&sl.sampled_color, &sl.albedo, &sl.nd}) *p = reinterpret_cast<float4*>(c->ring_arena + pitch * piece++);
        HIP_TRY(hipEventCreateWithFlags(&sl.own, hipEventDisableTiming));
        sl.trace_done = sl.last_use = nullptr;
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
                const size_t per_path = 2 * 64 + 2 * 64 + 2 * 32;  // state x2, rays x2, results x2
                const size_t counts_bytes = size_t(c->cfg.max_bounces + 1) * (kSegments + 1) * 64;   // + the ray pool's cursor per stage
                HIP_TRY(hipMalloc(&sq.rq_block, kSegments * cap * per_path + counts_bytes + 256));
                char* p = static_cast<char*>(sq.rq_block);
                sq.rq.state[0] = reinterpret_cast<float4*>(p); p += kSegments * cap * 64;
                sq.rq.state[1] = reinterpret_cast<float4*>(p); p += kSegments * cap * 64;
                sq.rq.rays[0] = reinterpret_cast<float4*>(p); p += kSegments * cap * 64;
                sq.rq.rays[1] = reinterpret_cast<float4*>(p); p += kSegments * cap * 64;
                sq.rq.results[0] = reinterpret_cast<uint4*>(p); p += kSegments * cap * 32;
                sq.rq.results[1] = reinterpret_cast<uint4*>(p); p += kSegments * cap * 32;
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
        t.heavy = 0; t.heavy_pending = false;
        if (c->trace_priority) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&t.host_heavy), 64, hipHostMallocDefault));
            t.host_heavy[0] = 0u;
            HIP_TRY(hipEventCreateWithFlags(&t.heavy_ready, hipEventDisableTiming));
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->slot = 0;
    c->hist_slot = -1;
    c->cur = 0;
    c->last = 0;
    c->has_history = false;
    c->accum_is_sampled = true;
    return build_tile_rows(c);
}

// Wait for a stream.  hipStreamSynchronize parks the thread and is woken by an interrupt: 15-25 us after the stream has drained on this
// stack, which is 5 % of a rank's 20-frame block on 8 GPUs (0.4 ms).  So the first 2 ms are spent polling hipStreamQuery (the
// stream's last completion signal, ~1 us a look); a wait that lasts longer falls back to the blocking call.
static int wait_stream(hipStream_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return VXRT_OK;
        if (e != hipErrorNotReady) return hip_fail(e, "hipStreamQuery");
        if ((spins & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    HIP_TRY(hipStreamSynchronize(s));
    return VXRT_OK;
}

int sync_all(vxrt_ctx* c) {
    for (hipStream_t t : c->trace_streams) { if (int rc = wait_stream(t)) return rc; }
    if (int rc = wait_stream(c->stream)) return rc;
    unsigned fused_bits = 0;
    for (vxrt_ctx::StreamQueues& sq : c->queues)      // fused_kernel's report of the last launch (copied back behind it)
        if (sq.host_ctl != nullptr && sq.host_ctl[2] != 0u) { c->fused_errors++; fused_bits |= sq.host_ctl[2]; sq.host_ctl[2] = 0u; }
    if (c->fused_errors != 0) {
        c->fused_errors = 0;
        set_error(("fused head + tail: a bounded wait of fused_kernel ran out (frames of the last launches are incomplete); bits " + std::to_string(fused_bits) +
                   ": 1 idle, 2 a record's stamp, 4 -, 8 a part-filled chunk").c_str());
        return VXRT_E_DEVICE;
    }
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
    // whole rounds of nranks bands at band_rows rows; the LAST round takes what is left as well, in taller bands (kernels.h: BandMap)
    const int tile = b.band_rows % 16 == 0 ? 16 : (b.band_rows % 8 == 0 ? 8 : b.band_rows);
    const int round_rows = b.nranks * b.band_rows;
    int rounds = b.height / round_rows;
    if (b.height % round_rows != 0 && rounds > 0) rounds -= 1;               // fold the remainder into the last whole round
    b.full_bands = rounds * b.nranks;
    b.tail_y0 = b.full_bands * b.band_rows;
    const int rest = b.height - b.tail_y0;                                  // 0, or < 2 * nranks * band_rows
    b.tail_rows = rest == 0 ? b.band_rows : ((rest + b.nranks - 1) / b.nranks + tile - 1) / tile * tile;
    b.local_rows = count_local_rows(b);
    c->band = b;
    return VXRT_OK;
}

EventPair take_pair(vxrt_ctx* c, int stage) {
    EventPair p;
    if (c->pending.size() >= 512) {   // a loop that never calls vxrt_sync: fold in the pairs that have finished (no waiting)
        size_t keep = 0;
        for (EventPair& q : c->pending) {
            float ms = 0.0f;
            if (hipEventQuery(q.b) == hipSuccess && hipEventElapsedTime(&ms, q.a, q.b) == hipSuccess) {
                c->ms[q.stage] += double(ms);
                c->free_pairs.push_back(q);
            } else {
                c->pending[keep++] = q;
            }
        }
        c->pending.resize(keep);
    }
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

bool valid_ctx(const vxrt_ctx* c) {
    if (!c) { set_error("null context"); return false; }
    return true;
}

float4* image_ptr(vxrt_ctx* c, vxrt_image which) {
    switch (which) {
        case VXRT_SAMPLED_COLOR: return c->ring[size_t(c->slot)].sampled_color;
        case VXRT_NORMAL_DEPTH: return c->ring[size_t(c->slot)].nd;
        case VXRT_ALBEDO_NODE: return c->ring[size_t(c->slot)].albedo;
        case VXRT_ACCUM_COLOR: return c->accum_is_sampled ? c->ring[size_t(c->slot)].sampled_color : c->accum[c->last];
        case VXRT_DENOISED: return c->denoised;
        default: return nullptr;
    }
}


}  // namespace vxrt

namespace vxrt {
// One option, at vxrt_create_tuned (at_create: nothing is allocated yet) or later through vxrt_set_option.  The options that size
// what vxrt_create allocates exist at creation only.
int apply_option(vxrt_ctx* c, uint32_t option, uint32_t value, bool at_create) {
    auto create_only = [&]() { if (!at_create) set_error("this option exists at vxrt_create_tuned only (it sizes what the context allocates)"); return at_create; };
    auto needs_variants = [&](const char* what) {
#if VXRT_VARIANTS
        (void)what; return true;
#else
        set_error(what); return false;
#endif
    };
    switch (int(option)) {
        case VXRT_OPT_DENOISE_MODE:
            if (value > 3) { set_error("denoise mode must be 0 (exact) or 1 (tolerant), + 2 for the generic kernel"); return VXRT_E_INVALID; }
            c->denoise_mode = int(value);
            return VXRT_OK;
        case VXRT_OPT_SCENE_FORMAT:
            if (value > 1) { set_error("scene format must be 0 (8-byte records) or 1 (wide records)"); return VXRT_E_INVALID; }
            if (value == 1 && !needs_variants("the wide scene records are not in this build of libvxrt (compile with -DVXRT_VARIANTS=1)")) return VXRT_E_INVALID;
            if (value == 1 && c->has_scene && c->d_wide == nullptr) {
                set_error("the wide records are built when a scene is set: choose the format before vxrt_set_voxels / vxrt_set_menger");
                return VXRT_E_INVALID;
            }
            c->scene_format = int(value);
            return VXRT_OK;
        case VXRT_OPT_TAIL_CAPACITY:
            if (at_create) { c->tail_capacity_override = int(value > 0x7fffffffu ? 0x7fffffffu : value); return VXRT_OK; }
            if (c->trace_variant < 4) return VXRT_OK;   // the other tracers' queues are sized for the worst case
            c->tail_capacity_override = int(value > 0x7fffffffu ? 0x7fffffffu : value);
            return resize_tail_queues(c, value == 0 ? (c->shard_capacity_max / 8u < 4096u ? 4096u : c->shard_capacity_max / 8u) : value);
        case VXRT_OPT_SKY_CULL:
            if (value > 1) { set_error("sky cull must be 0 or 1"); return VXRT_E_INVALID; }
            c->sky_cull = int(value);
            return VXRT_OK;
        case VXRT_OPT_FRAME_LANES:
            if (value > 1) { set_error("frame lanes must be 0 or 1"); return VXRT_E_INVALID; }
            c->frame_lanes = int(value);
            return VXRT_OK;
        case VXRT_OPT_HALO_ROWS:
            if (value > 4096) { set_error("halo rows must be 0..4096"); return VXRT_E_INVALID; }
            c->halo_min_rows = value;
            return VXRT_OK;
        case VXRT_OPT_TILE_ORDER:
            if (value > 1) { set_error("tile order must be 0 (raster) or 1 (longest first)"); return VXRT_E_INVALID; }
            c->use_tile_order = int(value);
            return VXRT_OK;
        case VXRT_OPT_TILE_SPREAD:      // 0..256: that many 256ths of the launch; VXRT_TILE_SPREAD_AUTO: the sort decides
            if (value > 256 && value != VXRT_TILE_SPREAD_AUTO) { set_error("tile spread must be 0..256 or VXRT_TILE_SPREAD_AUTO"); return VXRT_E_INVALID; }
            c->spread_override = value == VXRT_TILE_SPREAD_AUTO ? -1 : int(value);
            return VXRT_OK;
        case VXRT_OPT_TRACE_BLOCKS:
            if (value < 1 || value > 65535) { set_error("trace blocks must be 1..65535"); return VXRT_E_INVALID; }
            c->trace_blocks = int(value);
            return VXRT_OK;
        case VXRT_OPT_TAIL_FROM:        // sizes nothing, but a launch in flight must not see it change: creation only
            if (!create_only()) return VXRT_E_INVALID;
            c->tail_from = int(value > 64 ? 64 : value);
            return VXRT_OK;
        case VXRT_OPT_TAIL_SPLIT:
            if (!create_only()) return VXRT_E_INVALID;
            c->tail_split = value;
            return VXRT_OK;
        case VXRT_OPT_HOST_SCENE_BUILD:
            if (value > 1) { set_error("host scene build must be 0 or 1"); return VXRT_E_INVALID; }
            c->host_scene_build = int(value);
            return VXRT_OK;
        case VXRT_OPT_NODE_ORDER:       // read when a scene is set
            if (value != 0 && value != 2 && value != 3) { set_error("node order must be 0 (breadth-first), 2 or 3 (treelets of the last 2 / 3 node levels)"); return VXRT_E_INVALID; }
            c->node_order = int(value);
            return VXRT_OK;
        case VXRT_OPT_TRACER_OVERRIDE:  // the internal tracer number, past vxrt_config.tracer's auto rule (A/B runs of the variants)
            if (!create_only()) return VXRT_E_INVALID;
            if ((value == 2 || value == 3 || value == 5) && !needs_variants("tracers 2, 3 and 5 are not in this build of libvxrt (compile with -DVXRT_VARIANTS=1: scripts/test_variants.sh)")) return VXRT_E_INVALID;
            c->trace_variant = (value == 2 || value == 3 || value == 4 || value == 5) ? int(value) : 0;
            if (c->trace_variant >= 4 && c->cfg.max_bounces < 2) c->trace_variant = 0;
            c->auto_tracer = false;
            return VXRT_OK;
        case VXRT_OPT_TRACE_SPLIT:
            if (!create_only()) return VXRT_E_INVALID;
            c->trace_split = value;
            return VXRT_OK;
        case VXRT_OPT_PATH_BLOCKS:
            if (!create_only()) return VXRT_E_INVALID;
            c->path_blocks = int(value < 1 ? 1 : (value > 65535 ? 65535 : value));
            return VXRT_OK;
        case VXRT_OPT_SHADE_BLOCKS:
            if (!create_only()) return VXRT_E_INVALID;
            c->shade_blocks = int(value > 65535 ? 65535 : value);
            return VXRT_OK;
        case VXRT_OPT_RAYS_PER_WAVE:
            if (!create_only()) return VXRT_E_INVALID;
            c->rays_per_wave = value;
            return VXRT_OK;
        case VXRT_OPT_FUSED_TAIL:
            if (value > 1) { set_error("fused tail must be 0 or 1"); return VXRT_E_INVALID; }
            if (value == 1 && !needs_variants("the fused head + tail kernel is not in this build of libvxrt (compile with -DVXRT_VARIANTS=1: it measured slower)")) return VXRT_E_INVALID;
            c->fused_tail = int(value);
            return VXRT_OK;
        case VXRT_OPT_LONG_TILES:
            if (value > 500) { set_error("long tiles: 0 (off) .. 500 per mille of the tiles"); return VXRT_E_INVALID; }
            if (value != 0 && !needs_variants("the split launch of the longest tiles is not in this build of libvxrt (compile with -DVXRT_VARIANTS=1: it measured slower)")) return VXRT_E_INVALID;
            c->long_tiles_permille = value;
            return VXRT_OK;
        case VXRT_OPT_TRACE_PRIORITY:   // the streams are made at creation
            if (!create_only()) return VXRT_E_INVALID;
            if (value > 1) { set_error("trace priority must be 0 or 1"); return VXRT_E_INVALID; }
            if (value != 0 && !needs_variants("the priority split of a trace launch is not in this build of libvxrt (compile with -DVXRT_VARIANTS=1: it measured slower)")) return VXRT_E_INVALID;
            c->trace_priority = int(value);
            return VXRT_OK;
        case VXRT_OPT_XCD_AFFINITY:
            if (value > 64) { set_error("xcd affinity: 0 (off) or the side of a super-tile in tiles, 1..64"); return VXRT_E_INVALID; }
            if (value != 0 && !needs_variants("the XCD-affine launch order is not in this build of libvxrt (compile with -DVXRT_VARIANTS=1: an experiment, - 2 % at best)")) return VXRT_E_INVALID;
            c->xcd_affinity = int(value);
            for (vxrt_ctx::TileSchedule& t : c->schedules) t.age = 8;     // the next launch re-sorts
            return VXRT_OK;
        case VXRT_OPT_HEAD_STAGGER:
            if (value > 1) { set_error("head stagger must be 0 or 1"); return VXRT_E_INVALID; }
            if (value != 0 && !needs_variants("the head stagger is not in this build of libvxrt (compile with -DVXRT_VARIANTS=1: it measured slower)")) return VXRT_E_INVALID;
            c->head_stagger = int(value);
            return VXRT_OK;
        default:
            set_error("unknown option");
            return VXRT_E_INVALID;
    }
}
}  // namespace vxrt

extern "C" {

uint32_t vxrt_abi_version(void) { return 6; }   // 6: vxrt_read_async / vxrt_read_wait / vxrt_host_alloc / vxrt_host_free, vxrt_stats.split_launches; the header in three
uint32_t vxrt_build_features(void) { return VXRT_VARIANTS ? uint32_t(VXRT_FEATURE_VARIANTS) : 0u; }

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

int vxrt_create(const vxrt_config* cfg, vxrt_ctx** out) { return vxrt_create_tuned(cfg, nullptr, 0, out); }

int vxrt_create_tuned(const vxrt_config* cfg, const vxrt_tuning* tuning, size_t ntuning, vxrt_ctx** out) try {
    if (!cfg || !out || (ntuning != 0 && !tuning)) { set_error("null argument"); return VXRT_E_INVALID; }
    *out = nullptr;
    if (cfg->width == 0 || cfg->height == 0 || cfg->width > 65536 || cfg->height > 65536) { set_error("bad frame size"); return VXRT_E_INVALID; }
    if (cfg->max_bounces < 1 || cfg->max_bounces > 16) { set_error("max_bounces must be 1..16"); return VXRT_E_INVALID; }
    uint32_t nranks = cfg->nranks == 0 ? 1 : cfg->nranks;
    if (nranks > 1 && cfg->rank >= nranks) { set_error("rank >= nranks"); return VXRT_E_INVALID; }
    uint32_t band_rows = cfg->band_rows == 0 ? 16 : cfg->band_rows;
    // 16 for a denoise window (its 16 x 16 tiles must not straddle bands: check_render); 8 = the tracer's tile height, what a launch of
    // single frames wants (a wave is an 8 x 8 pixel tile); 2 or 4 rows are for launches of frame groups, whose waves hold 2 rows x 4
    // frames or 1 row x 8 frames (VXRT_OPT_FRAME_LANES): the finer the interleave, the more alike the ranks' shares
    if (!(band_rows == 2 || band_rows == 4 || band_rows % 8 == 0)) { set_error("band_rows must be 2, 4 or a multiple of 8 (of 16 for a denoise radius > 0)"); return VXRT_E_INVALID; }

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
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) c->wave_slots = unsigned(cus) * 4u * 5u;
    }
    if (hipEventCreateWithFlags(&c->halo_event, hipEventDisableTiming) != hipSuccess) return fail(hip_fail(hipGetLastError(), "hipEventCreate"));
    c->inflight = cfg->frames_in_flight == 0 ? 1 : int(cfg->frames_in_flight);
    // vxrt_config.tracer: 0 auto, 1 monolithic, 2 wavefront, 3 ray queues, 4 monolithic head + compacted tail (internally
    // 0, 2, 3, 4).  Measured on MI355X with frames in flight (menger 1080p 4 bounces / monu10 4K 8 bounces, ms per frame):
    // tracer 1: 0.219 / 0.52-1.53, tracer 3: 0.275 / 0.82-1.10, tracer 4: 0.175 / 0.54-0.94 -> auto = 4 whenever a path
    // can have a second hit, with further compactions from 6 bounces on (below).
    if (cfg->tracer > 5) { set_error("tracer must be 0..5"); return fail(VXRT_E_INVALID); }
    c->trace_variant = cfg->tracer == 0 ? 4 : (cfg->tracer == 1 ? 0 : int(cfg->tracer));
    c->auto_tracer = cfg->tracer == 0;
    // From 6 bounces on the tail compacts its live paths again, in launches of their own, at path segments 2, 3 and 5 (round 4; one
    // compaction at segment 3 until then).  Measured at 3840x2160, 4 spp, 8 bounces (scripts/exp_tail_split.py, ms per displayed frame,
    // castle close up / monu10 from outside): none 4.21 / 1.21, 0x08 3.53 / 1.13-1.15, 0x14 3.37 / 1.11-1.12, 0x2c 3.31-3.33 / 1.11,
    // 0x54 3.34-3.36 / 1.11-1.13, every segment (0xfc) 3.30 / 1.15.  At 4 bounces every extra compaction loses.
    c->tail_split = cfg->max_bounces >= 6 ? 0x2cu : 0u;
#if !VXRT_VARIANTS
    if (c->trace_variant == 2 || c->trace_variant == 3 || c->trace_variant == 5) {
        set_error("tracers 2, 3 and 5 are not in this build of libvxrt (compile with -DVXRT_VARIANTS=1: scripts/test_variants.sh)");
        return fail(VXRT_E_INVALID);
    }
#endif
    if (c->trace_variant >= 4 && cfg->max_bounces < 2) c->trace_variant = 0;  // no tail to compact
    // The library reads NO environment variable: what an experiment or a test wants different from the defaults arrives here, as
    // (option, value) pairs (vxrt_create_tuned), or later through vxrt_set_option.
    for (size_t i = 0; i < ntuning; i++)
        if ((rc = apply_option(c, tuning[i].option, tuning[i].value, true)) != VXRT_OK) return fail(rc);
    if (c->tail_from < 0 || c->tail_from >= int(cfg->max_bounces)) c->tail_from = 1;
    c->shade_blocks = (c->shade_blocks < 8 ? 8 : (c->shade_blocks > 2048 ? 2048 : c->shade_blocks) + 7) / 8 * 8;
    if (c->inflight < 1 || c->inflight > 16) { set_error("frames_in_flight must be 1..16"); return fail(VXRT_E_INVALID); }
    c->batch = cfg->frames_per_launch == 0 ? 1 : int(cfg->frames_per_launch);
    if (c->batch < 1 || c->batch > kMaxBatch) { set_error("frames_per_launch must be 1..32"); return fail(VXRT_E_INVALID); }
    // VXRT_OPT_TRACE_PRIORITY: the streams that carry trace launches at the device's highest priority (numerically lowest), one
    // low-priority stream per trace stream for the grid of tiles that only store sky (trace_frames)
    int prio_least = 0, prio_greatest = 0;
    if (c->trace_priority) (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    auto make_stream = [&](hipStream_t* st, bool trace) {
        return (c->trace_priority && trace) ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_greatest) : hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    };
    if (make_stream(&c->stream, c->inflight == 1) != hipSuccess) return fail(hip_fail(hipGetLastError(), "hipStreamCreate"));
    c->trace_streams.assign(size_t(c->inflight), nullptr);
    if (c->inflight == 1) {
        c->trace_streams[0] = c->stream;
    } else {
        for (hipStream_t& t : c->trace_streams)
            if (make_stream(&t, true) != hipSuccess) return fail(hip_fail(hipGetLastError(), "hipStreamCreate"));
    }
    if (c->trace_priority) {
        c->low_streams.assign(size_t(c->inflight), nullptr);
        c->low_fork.assign(size_t(c->inflight), nullptr);
        c->low_join.assign(size_t(c->inflight), nullptr);
        for (size_t i = 0; i < size_t(c->inflight); i++) {
            if (hipStreamCreateWithPriority(&c->low_streams[i], hipStreamNonBlocking, prio_least) != hipSuccess) return fail(hip_fail(hipGetLastError(), "hipStreamCreateWithPriority"));
            if (hipEventCreateWithFlags(&c->low_fork[i], hipEventDisableTiming) != hipSuccess) return fail(hip_fail(hipGetLastError(), "hipEventCreate"));
            if (hipEventCreateWithFlags(&c->low_join[i], hipEventDisableTiming) != hipSuccess) return fail(hip_fail(hipGetLastError(), "hipEventCreate"));
        }
    }
    c->launch_events.assign(size_t(c->inflight) * 2, nullptr);
    c->launch_event_turn.assign(size_t(c->inflight), 0u);
    c->head_events.assign(size_t(c->inflight), nullptr);
    for (hipEvent_t& e : c->head_events)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(hip_fail(hipGetLastError(), "hipEventCreate"));
    for (hipEvent_t& e : c->launch_events)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(hip_fail(hipGetLastError(), "hipEventCreate"));
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
    if (c->halo_event) (void)hipEventDestroy(c->halo_event);
    for (hipEvent_t e : c->launch_events) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->head_events) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->aux_fork) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->aux_join) if (e) (void)hipEventDestroy(e);
    for (hipStream_t t : c->aux_streams) if (t) (void)hipStreamDestroy(t);
    for (hipStream_t t : c->low_streams) if (t) { (void)hipStreamSynchronize(t); (void)hipStreamDestroy(t); }
    for (hipEvent_t e : c->low_fork) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->low_join) if (e) (void)hipEventDestroy(e);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (vxrt_ctx::ReadSlot& rs : c->read_slots) {
        if (rs.stage) (void)hipFree(rs.stage);
        if (rs.snap) (void)hipEventDestroy(rs.snap);
        if (rs.arrived) (void)hipEventDestroy(rs.arrived);
    }
    drop_touch_maps(c);
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
    return apply_option(c, option, value, false);
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

int vxrt_sync(vxrt_ctx* c) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = sync_all(c)) return rc;
    return resolve_events(c);
} VXRT_CATCH

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

// The non-blocking read-back (vxrt.h).  Order of events for slot k:
//   context stream:  [wait arrived_k of the slot's previous use] [wait the trace launch, for a trace image] D2D image -> stage_k, record snap_k
//   copy stream:     wait snap_k, D2H stage_k -> dst, record arrived_k
// The image itself is free for the next frame as soon as the context stream has passed the D2D copy (33 MB at 1080p: ~0.03 ms of HBM
// time), the stage is not reused before its transfer has arrived, and the host only ever waits in vxrt_read_wait.
int vxrt_read_async(vxrt_ctx* c, vxrt_image which, float* dst, size_t bytes, uint32_t slot) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    float4* src = image_ptr(c, which);
    if (!src || !dst) { set_error("bad image or null destination"); return VXRT_E_INVALID; }
    if (slot > 1) { set_error("vxrt_read_async: slot must be 0 or 1"); return VXRT_E_INVALID; }
    if (bytes != image_bytes(c)) { set_error("vxrt_read_async: bytes must equal local_rows*width*16"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    vxrt_ctx::ReadSlot& rs = c->read_slots[slot];
    if (c->copy_stream == nullptr) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (rs.snap == nullptr) {
        HIP_TRY(hipEventCreateWithFlags(&rs.snap, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&rs.arrived, hipEventDisableTiming));
    }
    if (rs.stage_bytes < bytes) {    // first use, or after vxrt_resize: a new stage (the old one's transfer must have arrived)
        if (rs.in_flight) HIP_TRY(hipEventSynchronize(rs.arrived));
        rs.in_flight = false;
        if (rs.stage) (void)hipFree(rs.stage);
        rs.stage = nullptr; rs.stage_bytes = 0;
        if (bytes) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&rs.stage), bytes));
        rs.stage_bytes = bytes;
    }
    if (bytes == 0) { rs.in_flight = false; return VXRT_OK; }
    if (rs.in_flight) HIP_TRY(hipStreamWaitEvent(c->stream, rs.arrived, 0));
    // a trace output lives in a ring slot written by a trace stream: the context stream waits for that launch (the post stages do the same)
    const bool trace_image = which == VXRT_SAMPLED_COLOR || which == VXRT_NORMAL_DEPTH || which == VXRT_ALBEDO_NODE || (which == VXRT_ACCUM_COLOR && c->accum_is_sampled);
    vxrt_ctx::Slot& cur = c->ring[size_t(c->slot)];
    if (trace_image && cur.trace_done != nullptr) HIP_TRY(hipStreamWaitEvent(c->stream, cur.trace_done, 0));
    HIP_TRY(hipMemcpyAsync(rs.stage, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipEventRecord(rs.snap, c->stream));
    if (trace_image) {   // the ring slot may be traced into again only after the snapshot has read it
        HIP_TRY(hipEventRecord(cur.own, c->stream));
        cur.last_use = cur.own;
        cur.last_use_recorded = true;
    }
    HIP_TRY(hipStreamWaitEvent(c->copy_stream, rs.snap, 0));
    HIP_TRY(hipMemcpyAsync(dst, rs.stage, bytes, hipMemcpyDeviceToHost, c->copy_stream));
    HIP_TRY(hipEventRecord(rs.arrived, c->copy_stream));
    rs.in_flight = true;
    return VXRT_OK;
} VXRT_CATCH

int vxrt_read_wait(vxrt_ctx* c, uint32_t slot) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    if (slot > 1) { set_error("vxrt_read_wait: slot must be 0 or 1"); return VXRT_E_INVALID; }
    vxrt_ctx::ReadSlot& rs = c->read_slots[slot];
    if (!rs.in_flight) return VXRT_OK;          // nothing was asked for (or it has been waited for already)
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipEventSynchronize(rs.arrived));
    rs.in_flight = false;
    return VXRT_OK;
} VXRT_CATCH

int vxrt_host_alloc(size_t bytes, void** out) try {
    if (!out) { set_error("null argument"); return VXRT_E_INVALID; }
    *out = nullptr;
    if (bytes == 0) return VXRT_OK;
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return VXRT_OK;
} VXRT_CATCH

int vxrt_host_free(void* p) try {
    if (p) HIP_TRY(hipHostFree(p));
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
        if (band_of_row(b, y) % b.nranks == b.rank) {
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
    out->halo_pack_ms = c->ms[3];
    out->halo_unpack_ms = c->ms[4];
    out->halo_exchanges = c->halo_exchanges;
    out->cull_box_valid = c->box_valid ? 1u : 0u;
    out->frame_lane_launches = c->frame_lane_launches;
    out->split_launches = uint32_t(c->split_launches);
    memcpy(out->cull_box_min, c->box_min, sizeof c->box_min);
    memcpy(out->cull_box_max, c->box_max, sizeof c->box_max);
    out->timed_frames = c->timed_frames;
    out->timed_launches = c->timed_launches;
    out->scene_bytes = c->svo_count * sizeof(SvoRecord) + c->leaf_count * sizeof(int32_t);
    out->noise_bytes = kNoiseCount * sizeof(float);
    out->local_rows = uint32_t(c->band.local_rows);
    out->octree_depth = c->depth;
    out->node_order = uint32_t(c->node_order_applied);
    out->octree_nodes = c->svo_count;
    for (size_t lane = 0; lane < c->queues.size(); lane++)   // fold in what the last launches wanted (the GPU is idle here)
        if (c->trace_variant >= 4) { if (int rc = grow_tail_queues(c, lane)) return rc; }
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
    c->ms[0] = c->ms[1] = c->ms[2] = c->ms[3] = c->ms[4] = 0.0;
    c->halo_exchanges = 0;
    c->frame_lane_launches = 0;
    c->split_launches = 0;
    return VXRT_OK;
} VXRT_CATCH


}  // extern "C"
