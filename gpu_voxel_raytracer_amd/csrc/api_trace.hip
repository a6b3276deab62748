// api_trace.hip — scheduling of the trace stage of libvxrt: which ring slots, stream, tile order and tail queue a launch of
// trace_kernel (+ bounce_kernel) gets, for 1..32 consecutive frames per launch and several launches in flight.
#include <algorithm>
#include <cmath>

#include "ctx.h"
#include "vx_vec.h"

namespace vxrt {

// new capacity (records per shard) for the path queues of every stream; waits for the GPU first.  All new buffers are allocated before
// any old one is released: if an allocation fails the context keeps its queues, its capacity and its accounting as they were.
int resize_tail_queues(vxrt_ctx* c, unsigned want) {
    want = want > c->shard_capacity_max ? c->shard_capacity_max : (want + 63u) / 64u * 64u;
    if (want == c->shard_capacity) return VXRT_OK;
    if (int rc = sync_all(c)) return rc;
    const size_t new_bytes = (size_t(want) * 64 + 1) * 64, old_bytes = (size_t(c->shard_capacity) * 64 + 1) * 64;
    std::vector<float4*> fresh;
    for (vxrt_ctx::StreamQueues& q : c->queues)
        for (float4* p : q.hitq)
            if (p) {
                float4* n = nullptr;
                const hipError_t e = hipMalloc(reinterpret_cast<void**>(&n), new_bytes);
                if (e != hipSuccess) {
                    for (float4* f : fresh) (void)hipFree(f);
                    return hip_fail(e, "hipMalloc (tail queues)");
                }
                fresh.push_back(n);
            }
    size_t k = 0;
    for (vxrt_ctx::StreamQueues& q : c->queues)
        for (float4*& p : q.hitq)
            if (p) {
                (void)hipFree(p);
                p = fresh[k++];
                c->queue_bytes += new_bytes - old_bytes;
            }
    c->shard_capacity = want;
    for (vxrt_ctx::StreamQueues& q : c->queues) q.stamps_clean = false;   // fresh memory: fused_kernel's stamps must be cleared before its next launch
    return VXRT_OK;
}

// Tail queues sized by need: look at what the stream's last launch wanted (its shard counters, copied back after the launch, with the
// capacity that launch ran with) and, if that did not fit, make the queues larger before the stream's next launch.  Paths that did not
// fit were followed by the head kernel itself, so no frame was wrong — only slower.
int grow_tail_queues(vxrt_ctx* c, size_t lane) {
    vxrt_ctx::StreamQueues& sq = c->queues[lane];
    if (!sq.counts_pending || hipEventQuery(sq.counts_ready) != hipSuccess) return VXRT_OK;
    sq.counts_pending = false;
    if (sq.host_ctl != nullptr && sq.host_ctl[2] != 0u) { c->fused_errors++; sq.host_ctl[2] = 0u; }
    unsigned peak = 0;
    for (unsigned s = 0; s < 64; s++) {
        const unsigned n = sq.host_counts[s * 16];
        peak = n > peak ? n : peak;
        if (n > sq.counts_capacity) c->queue_overflow_paths += n - sq.counts_capacity;   // against the capacity THAT launch had
    }
    if (peak <= c->shard_capacity || c->tail_capacity_override > 0 || c->shard_capacity >= c->shard_capacity_max) return VXRT_OK;
    // every stream's queues share one capacity (PathQueue::shard_capacity travels with the launch): grow them all, at rest
    unsigned want = peak + peak / 4u;
    want = want < 2u * c->shard_capacity ? 2u * c->shard_capacity : want;
    return resize_tail_queues(c, want);
}

Cam make_cam(const float pos[3], const CameraBasis& b) {
    Cam c;
    for (int i = 0; i < 3; i++) { c.o[i] = pos[i]; c.r[i] = b.right[i]; c.u[i] = b.up[i]; c.f[i] = b.forward_ray[i]; }
    return c;
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
    a.block_first = 0;
    a.cull = 0;
    a.stack_levels = c->depth < 1 ? 1 : int(c->depth);
#if VXRT_VARIANTS
    a.touch_nodes = c->d_touch_nodes;
    a.touch_leaves = c->d_touch_leaves;
#endif
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

// Sky cull: the scene's box grown by a margin m that dwarfs every rounding error of the walk and of the test itself.  The walk
// visits a cell only if the ray passes within ~2^-21 (|origin| + root_size) of it (its plane times are fl(fl(p - o) * inv): two
// roundings of quantities no larger than that); m = 0.01 + 2^-16 (max |origin| + 2 root_size) is at least 32 times as much.
void set_cull(const vxrt_ctx* c, TraceArgs& a, const Cam* cams, uint32_t g) {
    a.cull = 0;
    if (c->sky_cull && c->box_valid) {
        float far = 0.0f;
        bool sane = true;
        for (uint32_t k = 0; k < g; k++)
            for (int i = 0; i < 3; i++) {
                const float v = fabsf(cams[k].o[i]);
                sane = sane && v < 1e6f && std::isfinite(cams[k].r[i]) && std::isfinite(cams[k].u[i]) && std::isfinite(cams[k].f[i]);   // NaN fails v < 1e6
                far = v > far ? v : far;
            }
        if (sane) {
            const float m = 0.01f + ldexpf(far + 2.0f * c->root_size, -16);
            for (int i = 0; i < 3; i++) { a.cull_min[i] = c->box_min[i] - m; a.cull_max[i] = c->box_max[i] + m; }
            a.cull = 1;
        }
    }
}

// VXRT_OPT_XCD_AFFINITY (round 6's experiment for scenes that live in HBM, BASELINE config 5): a launch order in which the tiles that
// walk reach the XCDs by SCREEN REGION instead of round robin.  Blocks are dealt to the 8 XCDs in turn (block b and b + 8 share one:
// observed, not promised — a wrong guess costs speed only), every XCD has an L2 of its own, and neighbouring tiles read neighbouring
// parts of the tree: dealt round robin, each XCD fetches its own copy of every line (config 5 from outside: 9.8 x the unique bytes
// of a frame come in from beyond the L2s, 31 x inside a tunnel — bench.py: extra.config5_*.roofline.refetch).  So: the screen in
// super-tiles of S x S tiles; each super-tile's walking tiles belong to ONE of 8 lists, the super-tiles dealt longest first to the list
// with the least summed cost so far (the measured tile costs: the lists end up within a few per cent of each other, which matters
// because the dispatcher's round robin waits for the slowest XCD); every list longest first; order[8 i + x] = list x's i-th tile, a
// list that has run out filled up with tiles of sky (no scene traffic: any XCD may have them), or with the shortest chains of the
// longest list when there is no sky left; what remains of the sky comes last.  Made on the HOST from the costs copied back (a stall
// of its own every 8th launch: an experiment's price), one frame per launch only.
static int xcd_affine_order(vxrt_ctx* c, vxrt_ctx::TileSchedule& sched, hipStream_t ts) {
    const unsigned tiles = trace_tile_count(c->band.width, c->band.local_rows);
    int tw = 8, th = 8;
    trace_tile_dims(&tw, &th);
    const unsigned tx = unsigned(c->band.width + tw - 1) / unsigned(tw), ty = tiles / (tx ? tx : 1u);
    const unsigned S = unsigned(c->xcd_affinity);
    std::vector<uint32_t> cost(tiles), order;
    order.reserve(tiles);
    HIP_TRY(hipStreamSynchronize(ts));
    HIP_TRY(hipMemcpy(cost.data(), sched.cost, size_t(tiles) * 4, hipMemcpyDeviceToHost));
    const unsigned sx = (tx + S - 1) / S, sy = (ty + S - 1) / S;
    std::vector<unsigned long long> scost(size_t(sx) * sy, 0ull);
    for (unsigned t = 0; t < tiles; t++)
        if (cost[t] >= 4u) scost[size_t(t / tx / S) * sx + (t % tx) / S] += cost[t];
    std::vector<unsigned> supers(scost.size());
    for (unsigned i = 0; i < supers.size(); i++) supers[i] = i;
    std::stable_sort(supers.begin(), supers.end(), [&](unsigned a, unsigned b) { return scost[a] > scost[b]; });
    std::vector<int> owner(scost.size(), 0);
    unsigned long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (unsigned s : supers) {
        int best = 0;
        for (int x = 1; x < 8; x++) if (load[x] < load[best]) best = x;
        owner[s] = best;
        load[best] += scost[s];
    }
    std::vector<uint32_t> lists[8], sky;
    for (unsigned t = 0; t < tiles; t++) {
        if (cost[t] >= 4u) lists[owner[size_t(t / tx / S) * sx + (t % tx) / S]].push_back(t);
        else sky.push_back(t);
    }
    for (auto& l : lists) std::stable_sort(l.begin(), l.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
    size_t head[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tail[8], sky_next = 0;
    for (int x = 0; x < 8; x++) tail[x] = lists[x].size();
    auto left = [&](int x) { return tail[x] - head[x]; };
    for (;;) {
        bool any = false;
        for (int x = 0; x < 8; x++) any = any || left(x) != 0;
        if (!any) break;
        for (int x = 0; x < 8; x++) {
            if (left(x) != 0) { order.push_back(lists[x][head[x]++]); continue; }
            if (sky_next < sky.size()) { order.push_back(sky[sky_next++]); continue; }
            int longest = -1;                               // no sky left: the cheapest tile of the list with the most left
            for (int y = 0; y < 8; y++) if (left(y) > 1 && (longest < 0 || left(y) > left(longest))) longest = y;
            if (longest >= 0) order.push_back(lists[longest][--tail[longest]]);
            // else: nothing to pad with — this slot is skipped (the launch's last blocks; the lists that still hold a tile follow in this round)
        }
    }
    while (sky_next < sky.size()) order.push_back(sky[sky_next++]);
    if (order.size() != tiles) { set_error("xcd_affine_order: internal error (order incomplete)"); return VXRT_E_INVALID; }
    HIP_TRY(hipMemcpy(sched.order, order.data(), size_t(tiles) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(sched.last_cost, cost.data(), size_t(tiles) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(sched.cost, 0, size_t(tiles) * 4, ts));
    unsigned long long lo = load[0], hi = load[0];
    for (int x = 1; x < 8; x++) { lo = load[x] < lo ? load[x] : lo; hi = load[x] > hi ? load[x] : hi; }
    c->xcd_balance_permille = hi ? unsigned((hi - lo) * 1000ull / hi) : 0u;
    return VXRT_OK;
}

// One launch of trace_kernel over all tiles — or, with VXRT_OPT_TRACE_PRIORITY and a tile order whose walking tiles are known, as TWO
// grids: the tiles that walk (the first `heavy` of the plain longest-first order) on the high-priority trace stream, the tiles that
// only store sky on the lane's low-priority stream, forked and joined by events, so that the dispatcher takes blocks of the long
// chains first whenever both grids have blocks waiting.  Same blocks, same arithmetic.
static int launch_trace_split(vxrt_ctx* c, const TraceArgs& a, bool wide, bool hbm, hipStream_t ts, size_t lane, unsigned g, unsigned first = 0) {
    vxrt_ctx::TileSchedule& sched = c->schedules[lane];
    if (c->trace_priority && sched.heavy_pending && hipEventQuery(sched.heavy_ready) == hipSuccess) {
        sched.heavy = sched.host_heavy[0];
        sched.heavy_pending = false;
    }
    const unsigned tiles = trace_tile_count(c->band.width, c->band.local_rows);
    const unsigned heavy = (c->trace_priority && a.tile_order != nullptr && !sched.heavy_pending && first == 0) ? sched.heavy : 0u;
    if (heavy == 0u || heavy >= tiles) {
        HIP_TRY(launch_trace(a, wide, hbm, ts, first, 0u));
        return VXRT_OK;
    }
    HIP_TRY(hipEventRecord(c->low_fork[lane], ts));                         // after everything this launch waits for
    HIP_TRY(hipStreamWaitEvent(c->low_streams[lane], c->low_fork[lane], 0));
    HIP_TRY(launch_trace(a, wide, hbm, ts, 0u, heavy * g));                  // high priority: the chains
    HIP_TRY(launch_trace(a, wide, hbm, c->low_streams[lane], heavy * g, 0u));   // low priority: the stores
    HIP_TRY(hipEventRecord(c->low_join[lane], c->low_streams[lane]));
    c->split_launches++;
    return VXRT_OK;
}

// The trace stage of the next g frames (parameters at rest) as ONE launch of the tracer: g ring slots, frame numbers
// frame_number+1 .. +g.  g > 1 only with the trace_kernel-based tracers (1, 4, 5).  slots[k] = ring slot of frame k.
// path: optional g camera poses (position, direction), one per frame; null = the camera stays where it is.  cams / olds
// (g entries each): the camera and the "old" camera of every frame, as temporal.comp and denoise.comp of that frame see them.
int trace_frames(vxrt_ctx* c, uint32_t g, bool timed, int* slots, Cam* cams, Cam* olds, const float (*path_pos)[3], const float (*path_dir)[3],
                 uint32_t gbuf_frames) {
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
    hipEvent_t waited = nullptr;
    for (uint32_t k = 0; k < g; k++) {   // the slots of an earlier launch share its event: wait for each event once
        vxrt_ctx::Slot& sl = c->ring[size_t(slots[k])];
        if (sl.last_use_recorded && sl.last_use != waited) {
            HIP_TRY(hipStreamWaitEvent(ts, sl.last_use, 0));
            waited = sl.last_use;
        }
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
    // a wave of trace_kernel holds 8 (4) frames of one (two) rows of 8 pixels when the launch has whole groups of 8 (4) frames of ONE camera
    a.frame_lanes = (c->frame_lanes && !use_wide(c)) ? (g % 8u == 0u ? 8 : (g % 4u == 0u ? 4 : 0)) : 0;
    if (a.frame_lanes && path_pos != nullptr) {
        // a camera path: the frames of a wave have cameras of their own, and what they share shrinks with the image motion across the
        // group — estimated in pixels from the poses (rotation, and translation seen from the scene's centre).  Measured on an orbit
        // of the bench scene: + 4 % at 0.16-0.4 degrees (2-5 pixels) per 8 frames, + 2 % at 0.8 (11 pixels), - 1 % at 1.6 (21), - 3 % at 4 degrees (55)
        const float f_px = 0.5f * float(c->cfg.height) / tanf(0.5f * c->cam_fov);
        float worst = 0.0f;
        const uint32_t F = uint32_t(a.frame_lanes);
        for (uint32_t k = 0; k + F <= g; k += F) {
            const float* p0 = path_pos[k]; const float* p1 = path_pos[k + F - 1];
            const float* d0 = path_dir[k]; const float* d1 = path_dir[k + F - 1];
            const float n0 = sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]), n1 = sqrtf(d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2]);
            float cosang = (d0[0] * d1[0] + d0[1] * d1[1] + d0[2] * d1[2]) / (n0 * n1);
            cosang = cosang > 1.0f ? 1.0f : (cosang < -1.0f ? -1.0f : cosang);
            const float ang = acosf(cosang);
            const float dp = sqrtf((p1[0] - p0[0]) * (p1[0] - p0[0]) + (p1[1] - p0[1]) * (p1[1] - p0[1]) + (p1[2] - p0[2]) * (p1[2] - p0[2]));
            float dist = sqrtf((p0[0] - c->root_center[0]) * (p0[0] - c->root_center[0]) + (p0[1] - c->root_center[1]) * (p0[1] - c->root_center[1]) +
                               (p0[2] - c->root_center[2]) * (p0[2] - c->root_center[2]));
            dist = dist < 0.05f * c->root_size ? 0.05f * c->root_size : dist;
            const float px = f_px * (ang > dp / dist ? ang : dp / dist);
            worst = !(px <= worst) ? px : worst;   // NaN poses: no frame lanes
        }
        if (!(worst <= 16.0f)) a.frame_lanes = 0;
    }
    set_cull(c, a, cams, g);
    a.ray_counter = c->d_rays;
    a.tile_order = (c->use_tile_order && sched.valid) ? sched.order : nullptr;
    a.tile_cost = c->use_tile_order ? sched.cost : nullptr;
    a.frame_number = first_frame_number;
    a.cam = cams[0];
    for (uint32_t k = 0; k < g; k++) a.cams[k] = cams[k];
    if (c->band.local_rows > 0) {
        // the tail queues grow (host wait + reallocation) BEFORE the launch's timed region starts
        if (c->trace_variant >= 4 && !c->queues.empty()) { if (int rc = grow_tail_queues(c, lane)) return rc; }
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
                vxrt_ctx::StreamQueues& sq = c->queues[lane];
                unsigned* sets[3] = {sq.counts3, sq.counts3 + 64 * 16, sq.counts3 + 2 * 64 * 16};
                const unsigned J = sq.launches;
                a.tail = PathQueue{sq.hitq[0], sets[(J + 1) % 3], c->shard_capacity};
                a.tail_zero = sets[(J + 2) % 3];
                a.tail_from = c->tail_from;
                // head stagger (VXRT_OPT_HEAD_STAGGER): this launch's head starts when the previous launch's head (another stream) has
                // finished, so that a head runs beside the previous launch's tail instead of beside its head
#if VXRT_VARIANTS
                if (c->head_stagger && c->last_head_lane >= 0 && size_t(c->last_head_lane) != lane)
                    HIP_TRY(hipStreamWaitEvent(ts, c->head_events[size_t(c->last_head_lane)], 0));
#endif
                // Fused head + tail (VXRT_OPT_FUSED_TAIL): one grid of persistent waves takes the launch's blocks and then its queued
                // paths, chunk by chunk as they become complete (trace.hip: fused_kernel).  For tails of ONE launch (the 4-bounce
                // benchmark; a tail that compacts again keeps its launches), the 8-byte records, scenes in cache.
                bool fused_done = false;
#if VXRT_VARIANTS
                bool one_tail_launch = true;
                for (int k = c->tail_from + 1; k < int(c->cfg.max_bounces); k++)
                    if (sq.hitq[1] && ((c->tail_split >> k) & 1u)) one_tail_launch = false;
                if (c->fused_tail && c->trace_variant == 4 && one_tail_launch && !use_wide(c) && scene_bytes <= (size_t(256) << 20) && c->tail_from < int(c->cfg.max_bounces)) {
                    if (sq.fused_ctl == nullptr) {
                        HIP_TRY(hipMalloc(&sq.fused_ctl, fused_ctl_bytes()));
                        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&sq.host_ctl), 64, hipHostMallocDefault));    // word 2: the kernel's error word
                        memset(sq.host_ctl, 0, 64);
                    }
                    if (!sq.stamps_clean) {   // once per allocation of the queue: every slot's stamp must read 0
                        HIP_TRY(hipMemsetAsync(sq.hitq[0], 0, (size_t(c->shard_capacity) * 64 + 1) * 64, ts));
                        sq.stamps_clean = true;
                    }
                    HIP_TRY(hipMemsetAsync(sq.fused_ctl, 0, fused_ctl_bytes(), ts));
                    const uint32_t stamp = (sq.fused_launches++ % 65535u) + 1u;
                    const unsigned all_blocks = trace_tile_count(c->band.width, c->band.local_rows) * g;
                    const unsigned fused_slots = c->wave_slots / 5u * 4u;      // fused_kernel is compiled for 4 waves per SIMD (trace.hip: VXRT_FUSED_WAVES)
                    HIP_TRY(launch_fused(a, sq.fused_ctl, fused_slots < all_blocks ? fused_slots : all_blocks, a.tile_order ? sched.scratch : nullptr, stamp, ts));
                    if (a.frame_lanes) c->frame_lane_launches++;
                    sq.launches = J + 1;          // one kernel: it wrote set (J + 1) % 3 and cleared (J + 2) % 3, which the next launch writes
                    if (!sq.counts_pending) {
                        HIP_TRY(hipMemcpyAsync(sq.host_counts, sets[(J + 1) % 3], 64 * 64, hipMemcpyDeviceToHost, ts));
                        HIP_TRY(hipMemcpyAsync(sq.host_ctl + 2, static_cast<char*>(sq.fused_ctl) + fused_ctl_error_offset(), 4, hipMemcpyDeviceToHost, ts));
                        HIP_TRY(hipEventRecord(sq.counts_ready, ts));
                        sq.counts_pending = true;
                        sq.counts_capacity = c->shard_capacity;
                    }
                    fused_done = true;
                }
#endif
                if (!fused_done) {
                // (-DVXRT_VARIANTS=1 only: measured slower — the rest of the tiles are as chain-bound as the longest.)  The longest tiles apart (VXRT_OPT_LONG_TILES, per mille of the tiles): the first tiles of the launch order — the
                // longest chains of the last frames — run as an all-in-one grid of their own on a second stream (no hand-over: a
                // path's whole chain in ONE wave, begun at the launch's start), beside the head + tail pair of all other tiles.  A
                // launch that is little more than its chains (a rank's share of a short block on 8 GPUs) is two chains long as head +
                // tail and one as the all-in-one kernel, which in turn needs 1.3 x the instructions: this takes the one chain where
                // it matters and the cheaper instructions everywhere else.  Same pixels, same arithmetic, either way.
                unsigned long_blocks = 0;
#if VXRT_VARIANTS
                if (c->long_tiles_permille > 0 && a.tile_order != nullptr && c->trace_variant == 4 && !use_wide(c) && scene_bytes <= (size_t(256) << 20)) {
                    const unsigned tiles = trace_tile_count(c->band.width, c->band.local_rows);
                    unsigned n = unsigned((unsigned long long)tiles * c->long_tiles_permille / 1000u);
                    n = n < 1u ? 1u : (n >= tiles ? tiles - 1u : n);
                    long_blocks = n * g;
                    if (c->aux_streams.size() <= lane) c->aux_streams.resize(lane + 1, nullptr);
                    if (c->aux_fork.size() <= lane) { c->aux_fork.resize(lane + 1, nullptr); c->aux_join.resize(lane + 1, nullptr); }
                    if (c->aux_streams[lane] == nullptr) {
                        HIP_TRY(hipStreamCreateWithFlags(&c->aux_streams[lane], hipStreamNonBlocking));
                        HIP_TRY(hipEventCreateWithFlags(&c->aux_fork[lane], hipEventDisableTiming));
                        HIP_TRY(hipEventCreateWithFlags(&c->aux_join[lane], hipEventDisableTiming));
                    }
                    TraceArgs al = a;
                    al.tail = PathQueue{nullptr, nullptr, 0};     // nothing is handed over: trace_kernel follows these paths to their end
                    al.tail_zero = nullptr;
                    HIP_TRY(hipEventRecord(c->aux_fork[lane], ts));                       // after everything this launch waits for
                    HIP_TRY(hipStreamWaitEvent(c->aux_streams[lane], c->aux_fork[lane], 0));
                    HIP_TRY(launch_trace(al, false, false, c->aux_streams[lane], 0u, long_blocks));
                    HIP_TRY(hipEventRecord(c->aux_join[lane], c->aux_streams[lane]));
                }
#endif
                if (int rc = launch_trace_split(c, a, use_wide(c) && c->trace_variant == 4, scene_bytes > (size_t(256) << 20), ts, lane, g, long_blocks)) return rc;
#if VXRT_VARIANTS
                if (c->head_stagger) {
                    HIP_TRY(hipEventRecord(c->head_events[lane], ts));
                    c->last_head_lane = int(lane);
                }
#endif
                if (a.frame_lanes && !(use_wide(c) && c->trace_variant == 4) && scene_bytes <= (size_t(256) << 20)) c->frame_lane_launches++;
                sq.launches = J + 1;
                // how much room this launch wanted: its counter set, copied back for grow_tail_queues.  The set stays untouched until
                // launch J + 2 clears it, so when the tail is ONE launch (J + 1) the copy goes behind it — a copy between the head and
                // the tail costs the stream ~10 us of copy-engine hand-over on a block's critical path (round 5: the time line of a
                // rank of 8 showed trace_kernel -> 4 us copy -> 5.6 us gap -> bounce_kernel); a tail of several launches clears it sooner
                bool single_tail_launch = c->trace_variant == 4;
                for (int k = c->tail_from + 1; k < int(c->cfg.max_bounces); k++)
                    if (sq.hitq[1] && ((c->tail_split >> k) & 1u)) single_tail_launch = false;
                auto copy_counts = [&]() -> int {
                    if (sq.counts_pending) return VXRT_OK;
                    HIP_TRY(hipMemcpyAsync(sq.host_counts, sets[(J + 1) % 3], 64 * 64, hipMemcpyDeviceToHost, ts));
                    HIP_TRY(hipEventRecord(sq.counts_ready, ts));
                    sq.counts_pending = true;
                    sq.counts_capacity = c->shard_capacity;
                    return VXRT_OK;
                };
                if (!single_tail_launch) { if (int rc = copy_counts()) return rc; }
#if VXRT_VARIANTS
                if (c->trace_variant == 5) {
                    HIP_TRY(launch_paths(a, a.tail, sets[J % 3], c->tail_from, c->path_blocks, ts));
                    sq.launches = J + 2;
                } else
#endif
                {
                    // without a second queue (tail_split == 0) nothing is appended to queues[1]: its capacity 0 says so
                    PathQueue queues[2] = {{sq.hitq[0], nullptr, c->shard_capacity}, {sq.hitq[1], nullptr, sq.hitq[1] ? c->shard_capacity : 0u}};
                    HIP_TRY(launch_bounces(a, use_wide(c), queues, sets, &sq.launches, c->trace_blocks, sq.hitq[1] ? c->tail_split : 0u, c->tail_from, ts));
                }
                if (single_tail_launch) { if (int rc = copy_counts()) return rc; }
                if (long_blocks != 0u) HIP_TRY(hipStreamWaitEvent(ts, c->aux_join[lane], 0));   // the launch is over when both grids are
                }   // !fused_done
            } else {
                if (int rc = launch_trace_split(c, a, use_wide(c), scene_bytes > (size_t(256) << 20), ts, lane, g)) return rc;
                if (a.frame_lanes && !use_wide(c) && scene_bytes <= (size_t(256) << 20)) c->frame_lane_launches++;
            }
            if (c->trace_priority) HIP_TRY(hipStreamWaitEvent(ts, c->low_join[lane], 0));   // the launch is over when both grids are (a no-op before the first split)
            if (timed) HIP_TRY(hipEventRecord(p.b, ts));
            // Re-sort the tiles for this stream's coming frames from the costs just measured: after its first
            // frame, then every 8th (costs keep accumulating as a running maximum in between; ~7 us per sort).
            if (c->use_tile_order && (!sched.valid || sched.age >= 8)) {
                const unsigned tiles = trace_tile_count(c->band.width, c->band.local_rows);
                if (c->xcd_affinity > 0 && g == 1 && scene_bytes > (size_t(256) << 20)) {
                    if (int rc = xcd_affine_order(c, sched, ts)) return rc;
                } else {
                // (the priority split needs the walking tiles FIRST in the order: no spreading; and their number on the host)
                HIP_TRY(launch_tile_order(sched.cost, sched.order, sched.last_cost, sched.scratch, tiles, g * unsigned(c->inflight), c->wave_slots,
                                          c->trace_priority ? 0 : c->spread_override, ts));
                if (c->trace_priority) {
                    if (sched.heavy_pending) HIP_TRY(hipEventSynchronize(sched.heavy_ready));
                    HIP_TRY(hipMemcpyAsync(sched.host_heavy, sched.scratch + 128 * 64, sizeof(unsigned), hipMemcpyDeviceToHost, ts));
                    HIP_TRY(hipEventRecord(sched.heavy_ready, ts));
                    sched.heavy_pending = true;
                }
                }
                sched.valid = true;
                sched.age = 0;
            }
            sched.age++;
        }
#if VXRT_VARIANTS
        else {
            vxrt_ctx::StreamQueues& sq = c->queues[lane];
            PathQueue queues[2] = {{sq.hitq[0], nullptr, c->shard_capacity}, {sq.hitq[1], nullptr, c->shard_capacity}};
            unsigned* sets[3] = {sq.counts3, sq.counts3 + 64 * 16, sq.counts3 + 2 * 64 * 16};
            if (c->trace_variant == 2)
                HIP_TRY(launch_trace_wavefront(a, queues, sets, &sq.launches, c->trace_blocks, c->trace_split, ts));
            else
                HIP_TRY(launch_trace_rayqueue(a, queues[0], sets, &sq.launches, sq.rq, c->shade_blocks, c->trace_blocks, c->rays_per_wave, ts));
            if (timed) HIP_TRY(hipEventRecord(p.b, ts));
        }
#endif
        if (timed) c->pending.push_back(p);
    }
    hipEvent_t done = c->launch_events[lane * 2 + (c->launch_event_turn[lane]++ & 1u)];
    HIP_TRY(hipEventRecord(done, ts));   // one event for the whole launch
    for (uint32_t k = 0; k < g; k++) {
        vxrt_ctx::Slot& sl = c->ring[size_t(slots[k])];
        sl.trace_done = done;
        sl.last_use = done;              // until a later stage reads the slot, the trace is its last use
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


}  // namespace vxrt
