// api_halo.hip — the multi-rank halo of libvxrt (include/vxrt.h "halo").  A rank owns interleaved bands of rows; its
// denoise window (shaders/denoise.comp:51-57) and its temporal reprojection (shaders/temporal.comp:85-113) read up to `rows` rows of
// the bands above and below each of its own, which live on rank - 1 and rank + 1.  Per frame, after the temporal stage:
//
//     vxrt_halo_pack   -> two messages leave (RCCL send/recv, one per neighbour, each on its own xGMI link)
//     vxrt_render(VXRT_DENOISE_INTERIOR)      the tiles whose window stays inside this rank's rows: runs while the messages travel
//     vxrt_halo_unpack <- two messages arrived
//     vxrt_render(VXRT_DENOISE_EDGE)          the tiles that read halo rows
//
// Everything is enqueued on the context's stream; vxrt_stream_wait_context / vxrt_context_wait_stream order it against the caller's
// communication stream with events, so nothing on this path waits on the host.  The message layout is kernels.h: HaloView (36 B/px).
#include "ctx.h"

namespace vxrt {

namespace {
int max_bands(const BandMap& b) {
    return (band_count(b) + b.nranks - 1) / b.nranks;
}

// the layout of an exchange with `rows` rows per band edge
HaloView view_for(const vxrt_ctx* c, uint32_t rows) {
    HaloView v{nullptr, int(rows), max_bands(c->band), 0, 0};
    v.plane = size_t(v.slots) * rows * size_t(c->band.width);
    const size_t f4 = 2 * v.plane + (v.plane + 3) / 4;           // A, B: a float4 per pixel; C: a float per pixel
    v.message = (f4 + 15) / 16 * 16;                              // whole 256-byte lines
    return v;
}
}  // namespace

// The most rows per band edge the layout can carry: a neighbour's rows beyond its band belong to a third rank, so the lowest band
// that has a band below it.  With the remainder folded into the last round (kernels.h: BandMap) that is band_rows; only a frame
// lower than one round of bands has lower ones.  (The frame's very last band may be clipped lower still: nothing lies below it.)
uint32_t halo_rows_max(const vxrt_ctx* c) {
    const BandMap& b = c->band;
    if (b.nranks < 2) return 0;
    const int lowest = b.tail_y0 < b.height ? (b.full_bands > 0 && b.band_rows < b.tail_rows ? b.band_rows : b.tail_rows) : b.band_rows;
    return uint32_t(lowest);
}

uint32_t halo_rows_wanted(const vxrt_ctx* c) {
    if (c->band.nranks < 2) return 0;
    const uint32_t rows = c->denoise.radius > c->halo_min_rows ? c->denoise.radius : c->halo_min_rows;
    const uint32_t most = halo_rows_max(c);
    return rows > most ? most : rows;
}

void free_halo(vxrt_ctx* c) {
    if (c->halo) (void)hipFree(c->halo);
    if (c->d_tile_rows) (void)hipFree(c->d_tile_rows);
    c->halo = nullptr;
    c->d_tile_rows = nullptr;
    c->halo_store_f4 = 0;
    c->halo_view = HaloView{nullptr, 0, 0, 0, 0};
    c->halo_valid = false;
    c->tile_rows_interior = c->tile_rows_edge = 0;
}

// Which rows of 16x16 denoise tiles read rows of another rank: the window of a tile reaches at most 8 < 16 rows beyond it, so these
// are a band's first tile row when a band lies above it and its last one when a band lies below — whatever the radius (>= 1).
int build_tile_rows(vxrt_ctx* c) {
    const BandMap& b = c->band;
    if (c->d_tile_rows) (void)hipFree(c->d_tile_rows);
    c->d_tile_rows = nullptr;
    c->tile_rows_interior = c->tile_rows_edge = 0;
    const int tile_rows = (b.local_rows + 15) / 16;
    if (tile_rows == 0 || tile_rows > 65535 || b.band_rows % 16 != 0) return VXRT_OK;   // 8-row bands: no denoise window (check_render)
    std::vector<uint16_t> interior, edge;
    for (int t = 0; t < tile_rows; t++) {
        const int lrow0 = t * 16, lband = local_band_of(b, lrow0), gb = lband * b.nranks + b.rank;
        const int band_y0 = band_first_row(b, gb);
        const int band_end = band_y0 + band_nominal_rows(b, gb) < b.height ? band_y0 + band_nominal_rows(b, gb) : b.height;
        const int y0 = band_y0 + (lrow0 - local_band_first_row(b, lband)), y1 = y0 + 16 < band_end ? y0 + 16 : band_end;   // the tile's rows [y0, y1)
        const bool above = b.nranks > 1 && y0 == band_y0 && band_y0 > 0;
        const bool below = b.nranks > 1 && y1 == band_end && band_end < b.height;
        (above || below ? edge : interior).push_back(uint16_t(t));
    }
    c->tile_rows_interior = uint32_t(interior.size());
    c->tile_rows_edge = uint32_t(edge.size());
    interior.insert(interior.end(), edge.begin(), edge.end());
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_tile_rows), interior.size() * sizeof(uint16_t)));
    HIP_TRY(hipMemcpy(c->d_tile_rows, interior.data(), interior.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return VXRT_OK;
}

}  // namespace vxrt

extern "C" {

int vxrt_halo_info_get(vxrt_ctx* c, vxrt_halo_info* out) try {
    if (!valid_ctx(c) || !out) { set_error("null argument"); return VXRT_E_INVALID; }
    memset(out, 0, sizeof *out);
    const uint32_t rows = halo_rows_wanted(c);
    if (rows == 0) return VXRT_OK;
    const HaloView v = view_for(c, rows);
    out->rows = rows;
    out->max_rows = halo_rows_max(c);
    out->slots = uint32_t(v.slots);
    out->bytes_per_pixel = 36;
    out->message_bytes = v.message * sizeof(float4);
    out->interior_tile_rows = c->tile_rows_interior;
    out->edge_tile_rows = c->tile_rows_edge;
    return VXRT_OK;
} VXRT_CATCH

int vxrt_halo_bytes(vxrt_ctx* c, size_t* bytes) try {
    if (!valid_ctx(c) || !bytes) { set_error("null argument"); return VXRT_E_INVALID; }
    const uint32_t rows = halo_rows_wanted(c);
    *bytes = rows == 0 ? 0 : view_for(c, rows).message * sizeof(float4);
    return VXRT_OK;
} VXRT_CATCH

// ONE pack launch on the context's stream (after the stages already enqueued there); returns at once.
int vxrt_halo_pack(vxrt_ctx* c, void* dev_to_prev, void* dev_to_next) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    const uint32_t rows = halo_rows_wanted(c);
    if (rows == 0) return VXRT_OK;
    if (!dev_to_prev || !dev_to_next) { set_error("null halo buffer"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    vxrt_ctx::Slot& cur = c->ring[size_t(c->slot)];
    if (cur.trace_done != nullptr) HIP_TRY(hipStreamWaitEvent(c->stream, cur.trace_done, 0));
    const HaloView v = view_for(c, rows);
    HaloPackArgs a;
    a.color = c->accum_is_sampled ? cur.sampled_color : c->accum[c->last];
    a.nd = cur.nd;
    a.albedo = cur.albedo;
    a.to_prev = static_cast<float4*>(dev_to_prev);
    a.to_next = static_cast<float4*>(dev_to_next);
    a.band = c->band;
    a.rows = int(rows);
    a.slots = v.slots;
    a.local_bands = local_band_count(c->band);
    a.plane = v.plane;
    EventPair p = take_pair(c, 3);
    HIP_TRY(hipEventRecord(p.a, c->stream));
    HIP_TRY(launch_halo_pack(a, c->stream));
    HIP_TRY(hipEventRecord(p.b, c->stream));
    c->pending.push_back(p);
    HIP_TRY(hipEventRecord(cur.own, c->stream));   // the slot may be traced into again only after its rows were packed
    cur.last_use = cur.own;
    cur.last_use_recorded = true;
    return VXRT_OK;
} VXRT_CATCH

// ONE unpack launch on the context's stream: both received messages -> the context's halo store.  Returns at once; the two
// buffers must stay as they are until the launch has run (order the next receive after it: vxrt_stream_wait_context).
int vxrt_halo_unpack(vxrt_ctx* c, const void* dev_from_prev, const void* dev_from_next) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    const uint32_t rows = halo_rows_wanted(c);
    if (rows == 0) return VXRT_OK;
    if (!dev_from_prev || !dev_from_next) { set_error("null halo buffer"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    HaloView v = view_for(c, rows);
    if (c->halo == nullptr || c->halo_store_f4 < 2 * v.message) {
        HIP_TRY(hipStreamSynchronize(c->stream));   // a stage that reads the old store may still be running (only when the layout grows)
        if (c->halo) (void)hipFree(c->halo);
        c->halo = nullptr;
        c->halo_store_f4 = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->halo), 2 * v.message * sizeof(float4)));
        c->halo_store_f4 = 2 * v.message;
    }
    EventPair p = take_pair(c, 4);
    HIP_TRY(hipEventRecord(p.a, c->stream));
    HIP_TRY(launch_halo_unpack(c->halo, static_cast<const float4*>(dev_from_prev), static_cast<const float4*>(dev_from_next), v.message, c->stream));
    HIP_TRY(hipEventRecord(p.b, c->stream));
    c->pending.push_back(p);
    c->halo_exchanges += 1;
    v.base = c->halo;
    c->halo_view = v;
    c->halo_valid = true;
    c->halo_epoch = c->temporal_count;
    return VXRT_OK;
} VXRT_CATCH

// The synchronous forms: on return the rows are in the caller's buffers / the caller's buffers may be re-used.
int vxrt_halo_export(vxrt_ctx* c, void* dev_to_prev, void* dev_to_next) try {
    if (int rc = vxrt_halo_pack(c, dev_to_prev, dev_to_next)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VXRT_OK;
} VXRT_CATCH

int vxrt_halo_import(vxrt_ctx* c, const void* dev_from_prev, const void* dev_from_next) try {
    if (int rc = vxrt_halo_unpack(c, dev_from_prev, dev_from_next)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VXRT_OK;
} VXRT_CATCH

// `stream` (a hipStream_t of the same device; null = the legacy default stream) waits for everything enqueued on the context so far.
int vxrt_stream_wait_context(vxrt_ctx* c, void* stream) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    HIP_TRY(hipSetDevice(c->cfg.device));
    for (hipStream_t t : c->trace_streams)
        if (t != c->stream) {
            HIP_TRY(hipEventRecord(c->halo_event, t));
            HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->halo_event, 0));
        }
    HIP_TRY(hipEventRecord(c->halo_event, c->stream));
    HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->halo_event, 0));
    return VXRT_OK;
} VXRT_CATCH

// The context's main stream (post stages, halo pack / unpack) waits for everything enqueued on `stream` so far.
int vxrt_context_wait_stream(vxrt_ctx* c, void* stream) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipEventRecord(c->halo_event, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->halo_event, 0));
    return VXRT_OK;
} VXRT_CATCH

}  // extern "C"
