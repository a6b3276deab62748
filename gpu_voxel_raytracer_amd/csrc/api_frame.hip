// api_frame.hip — frame sequencing of libvxrt: vxrt_render and its batched forms run the stages in the order of Context::render
// (src/context.rs:2004-2075: voxels -> temporal -> denoise, then the G-buffer hand-over, here a ping-pong).
#include "ctx.h"

namespace vxrt {

int check_render(vxrt_ctx* c, uint32_t flags) {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    const uint32_t stages = VXRT_ALL | VXRT_DENOISE_INTERIOR | VXRT_DENOISE_EDGE;
    if ((flags & stages) == 0) { set_error("no stage selected"); return VXRT_E_INVALID; }
    if ((flags & VXRT_DENOISE) && (flags & (VXRT_DENOISE_INTERIOR | VXRT_DENOISE_EDGE))) {
        set_error("VXRT_DENOISE is the whole stage: it does not combine with VXRT_DENOISE_INTERIOR / VXRT_DENOISE_EDGE");
        return VXRT_E_INVALID;
    }
    if (!c->has_scene) { set_error("vxrt_render before any scene was set"); return VXRT_E_NOSCENE; }
    const bool window = c->band.nranks > 1 && c->denoise.radius > 0;   // a denoise window that reaches into other ranks' rows
    const bool any_denoise = (flags & (VXRT_DENOISE | VXRT_DENOISE_INTERIOR | VXRT_DENOISE_EDGE)) != 0;
    const bool reads_halo = (flags & (VXRT_DENOISE | VXRT_DENOISE_EDGE)) != 0;
    if (any_denoise && window && c->band.band_rows % 16 != 0) {
        set_error("denoise with radius > 0 on row bands needs band_rows to be a multiple of 16 (its 16x16 tiles must not straddle bands)");
        return VXRT_E_INVALID;
    }
    if (reads_halo && window && (flags & (VXRT_TRACE | VXRT_TEMPORAL))) {
        set_error("multi-rank denoise with radius > 0 needs the halo: render TRACE|TEMPORAL, exchange, then DENOISE (or DENOISE_INTERIOR "
                  "while the messages travel and DENOISE_EDGE after vxrt_halo_unpack)");
        return VXRT_E_INVALID;
    }
    if (reads_halo && window && !(c->halo_valid && c->halo_view.rows >= int(c->denoise.radius))) {
        set_error("multi-rank denoise: no halo unpacked for this frame (or with fewer rows than the radius)");
        return VXRT_E_INVALID;
    }
    return VXRT_OK;
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


// temporal / denoise of the frame in ring slot c->slot, in the reference's order (src/context.rs:2028-2043)
int post_stages(vxrt_ctx* c, uint32_t flags, bool timed) {
    const bool multi = c->band.nranks > 1;
    vxrt_ctx::Slot& cur = c->ring[size_t(c->slot)];
    const bool post = (flags & (VXRT_TEMPORAL | VXRT_DENOISE | VXRT_DENOISE_INTERIOR | VXRT_DENOISE_EDGE)) != 0;
    if (post && cur.trace_done != nullptr) HIP_TRY(hipStreamWaitEvent(c->stream, cur.trace_done, 0));

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
        // the halo unpacked after the previous temporal stage holds the neighbours' rows of exactly this history
        const bool apron = multi && a.has_history && c->halo_view.base != nullptr && c->halo_view.rows >= 1 && c->halo_epoch == c->temporal_count;
        a.halo = apron ? c->halo_view : HaloView{nullptr, 0, 0, 0, 0};
        c->temporal_count += 1;
        memset(a.inv, 0, sizeof a.inv);
        if (a.has_history) affine_inverse(c->old_cam, a.inv);
        a.sample_blending = c->temporal.sample_blending;
        a.maximum_blending = c->temporal.maximum_blending;
        a.blending_distance_cutoff = c->temporal.blending_distance_cutoff;
        // radius 0: the denoise stage is a per-pixel function of this stage's output — do it in the same pass
        fused_denoise = (flags & (VXRT_DENOISE | VXRT_DENOISE_INTERIOR)) != 0 && c->denoise.radius == 0;
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
            HIP_TRY(hipEventRecord(hist.own, c->stream));
            hist.last_use = hist.own;
            hist.last_use_recorded = true;
        }
        c->accum_is_sampled = false;
        c->last = c->cur;
        c->cur ^= 1;  // hand the G-buffer over: what was written becomes the history (src/context.rs:2041-2043)
        c->hist_slot = c->slot;
        c->has_history = true;
    }

    // The denoise stage: whole (VXRT_DENOISE), or in two launches around a halo exchange — the 16x16 tiles whose window stays inside
    // this rank's rows (VXRT_DENOISE_INTERIOR: needs nothing from the neighbours) and the tiles that read halo rows
    // (VXRT_DENOISE_EDGE).  Radius 0 has no window: INTERIOR is then the whole stage and EDGE has nothing to do.
    const uint32_t parts = flags & (VXRT_DENOISE | VXRT_DENOISE_INTERIOR | VXRT_DENOISE_EDGE);
    if (parts != 0 && !fused_denoise && !(c->denoise.radius == 0 && parts == VXRT_DENOISE_EDGE)) {
        DenoiseArgs a;
        a.colors = c->accum_is_sampled ? cur.sampled_color : c->accum[c->last];
        a.nd = cur.nd;
        a.albedo = cur.albedo;
        a.output = c->denoised;
        a.halo = (multi && c->denoise.radius > 0 && (parts & (VXRT_DENOISE | VXRT_DENOISE_EDGE))) ? c->halo_view : HaloView{nullptr, 0, 0, 0, 0};
        a.band = c->band;
        a.cam = c->cam;
        a.radius = c->denoise.radius;
        a.sigma_distance_2 = 2.0f * (c->denoise.sigma_distance * c->denoise.sigma_distance);  // denoise.comp:39-40
        a.sigma_range_2 = 2.0f * (c->denoise.sigma_range * c->denoise.sigma_range);
        a.albedo_factor = c->denoise.albedo_factor;
        a.mode = c->denoise_mode;
        a.tile_rows = nullptr;
        a.tile_row_count = 0;
        const bool both = (parts & VXRT_DENOISE) || parts == (VXRT_DENOISE_INTERIOR | VXRT_DENOISE_EDGE);
        bool launch = c->band.local_rows > 0;
        if (!both && c->denoise.radius > 0 && launch) {
            if (c->d_tile_rows == nullptr) { set_error("no denoise tile lists for this band layout"); return VXRT_E_INVALID; }
            const bool interior = parts == VXRT_DENOISE_INTERIOR;
            a.tile_rows = c->d_tile_rows + (interior ? 0u : c->tile_rows_interior);
            a.tile_row_count = interior ? c->tile_rows_interior : c->tile_rows_edge;
            launch = launch && a.tile_row_count > 0;
        }
        if (launch) {
            EventPair p;
            if (timed) { p = take_pair(c, 2); HIP_TRY(hipEventRecord(p.a, c->stream)); }
            HIP_TRY(launch_denoise(a, c->stream));
            if (timed) { HIP_TRY(hipEventRecord(p.b, c->stream)); c->pending.push_back(p); }
        }
    }
    if (post) {
        HIP_TRY(hipEventRecord(cur.own, c->stream));
        cur.last_use = cur.own;
        cur.last_use_recorded = true;
    }
    return VXRT_OK;
}

}  // namespace vxrt

extern "C" {

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
                if (sl.trace_done != nullptr && (k == 0 || sl.trace_done != c->ring[size_t(slots[k - 1])].trace_done))
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
                if (k == 0) HIP_TRY(hipEventRecord(sl.own, c->stream));   // one event for the pass; the batch's slots share it
                sl.last_use = c->ring[size_t(slots[0])].own;
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


}  // extern "C"
