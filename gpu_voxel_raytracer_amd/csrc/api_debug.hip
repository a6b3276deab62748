// api_debug.hip — test hooks and diagnostics of libvxrt (include/vxrt.h "Test hook" entries) and the blue-noise generator's entry point.
#include "../../include/vxrt_bluenoise.h"
#include "ctx.h"

extern "C" {

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

// Diagnostics: the launch order the last sort made (kernel tiles, 8x8 pixels), the costs it was made from, how many tiles walked and
// how far they were spread.  order / cost: n = ceil(width/8) * ceil(local_rows/8) entries each (either may be null).
int vxrt_debug_tile_order(vxrt_ctx* c, uint32_t* order, uint32_t* cost, size_t n, uint32_t* walking_tiles, uint32_t* spread_256) try {
    if (!valid_ctx(c)) { set_error("null argument"); return VXRT_E_INVALID; }
    const size_t tiles = trace_tile_count(c->band.width, c->band.local_rows);
    if (n != tiles || c->schedules.empty()) { set_error("tile count mismatch"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = sync_all(c)) return rc;
    const vxrt_ctx::TileSchedule& t = c->schedules[size_t(c->last_schedule)];
    if (!t.valid) { set_error("no tile order yet (it is made after a stream's first launch)"); return VXRT_E_INVALID; }
    if (order) HIP_TRY(hipMemcpy(order, t.order, tiles * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (cost) HIP_TRY(hipMemcpy(cost, t.last_cost, tiles * sizeof(uint32_t), hipMemcpyDeviceToHost));
    uint32_t two[2] = {0, 0};
    HIP_TRY(hipMemcpy(two, t.scratch + 128 * 64, sizeof two, hipMemcpyDeviceToHost));   // tile_scan_kernel: [bins * blocks] walking tiles, [+ 1] spread
    if (walking_tiles) *walking_tiles = two[0];
    if (spread_256) *spread_256 = two[1];
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

// Diagnostics: of this rank's pixels, how many primary rays of the NEXT frame (the camera as set) the sky cull decides without walking
// the octree (0 with VXRT_OPT_SKY_CULL off).  Those rays are counted in vxrt_stats.rays — a ray is one cast_bounded_ray invocation
// of the shader, and the cull is an implementation of it for rays that provably miss — so rays - frames x this = the rays that
// walked.  Nothing is rendered and no context state changes.
int vxrt_debug_culled_pixels(vxrt_ctx* c, uint64_t* count) try {
    if (!valid_ctx(c) || !count) { set_error("null argument"); return VXRT_E_INVALID; }
    if (!c->has_scene) { set_error("no scene set"); return VXRT_E_NOSCENE; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = vxrt_sync(c)) return rc;
    const Cam keep_cam = c->cam, keep_old = c->old_cam;
    const vxrt_uniforms keep_u = c->uniforms;
    update_bindings(c);
    TraceArgs a{};
    frame_constants(c, a);
    a.cam = c->cam;
    a.batch = 1;
    set_cull(c, a, &a.cam, 1);
    c->cam = keep_cam; c->old_cam = keep_old; c->uniforms = keep_u;
    ScratchBuffer b;
    HIP_TRY(b.alloc(sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(b.as<unsigned long long>(), 0, sizeof(unsigned long long), c->stream));
    HIP_TRY(launch_count_culled(a, b.as<unsigned long long>(), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long n = 0;
    HIP_TRY(hipMemcpy(&n, b.as<unsigned long long>(), sizeof n, hipMemcpyDeviceToHost));
    *count = n;
    return VXRT_OK;
} VXRT_CATCH

#if VXRT_VARIANTS
// PROTOTYPE (csrc/trace_dda.hip; -DVXRT_VARIANTS=1 builds only): the same rays through the exact walk (cast_probe_kernel over the
// context's scene format) and through the three-level DDA over a bit grid that is built on the device from the scene in place (first
// call; kept until the scene changes), both timed with HIP events (the second of two launches each).  flags: bit 0 = the certificate,
// bit 1 = the super-brick bits staged in LDS, bit 2 = skip the walk (out_walk untouched, ms[0] = 0), bit 3 = skip the DDA (no grid is built: for timing the walk of a
// wide-record context on the same rays).  max_steps: the DDA gives a ray
// up (flags it) beyond that many steps.  out_*: 8 floats per ray = hit, time, bits(leaf word), normal xyz, flagged, steps.
// ms[0] = walk, ms[1] = DDA, ms[2] = the grid's build (0 when it was there), ms[3] = its bytes.  origins / dirs / out_*: HOST arrays.
int vxrt_debug_dda_rays(vxrt_ctx* c, const float* origins, const float* dirs, size_t n, uint32_t flags, float margin_scale, uint32_t max_steps,
                        float* out_walk, float* out_dda, double ms[4]) try {
    if (!valid_ctx(c) || !origins || !dirs || !out_dda || !ms || (!out_walk && !(flags & 4u))) { set_error("null argument"); return VXRT_E_INVALID; }
    if (!c->has_scene) { set_error("no scene set"); return VXRT_E_NOSCENE; }
    const int levels = int(c->depth) + 1;
    if (levels < 3 || levels > 12) { set_error("the DDA grid wants a tree of depth 2..11 (4 .. 4096 cells per axis)"); return VXRT_E_INVALID; }
    if (n == 0) return VXRT_OK;
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = vxrt_sync(c)) return rc;
    TraceArgs a{};
    a.svo = c->d_svo; a.leaves = c->d_leaves;
    a.root_rec = c->root_rec;
    a.wide = c->d_wide;
    a.wide_root = c->wide_root;
    a.node_levels = int(c->depth) + 1;
    memcpy(a.root_center, c->root_center, sizeof a.root_center);
    a.root_size = c->root_size;
    a.stack_levels = c->depth < 1 ? 1 : int(c->depth);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    float t = 0.0f;
    ms[0] = ms[1] = ms[2] = ms[3] = 0.0;
    size_t sz[4] = {0, 0, 0, 0};
    dda_grid_sizes(levels, &sz[0], &sz[1], &sz[2], &sz[3]);
    const bool dda = !(flags & 8u);
    if (dda && (c->dda_grid[0] == nullptr || c->dda_levels != levels)) {
        drop_touch_maps(c);
        for (int k = 0; k < 4; k++) {
            if (sz[k] == 0) continue;
            HIP_TRY(hipMalloc(&c->dda_grid[k], sz[k]));
            HIP_TRY(hipMemsetAsync(c->dda_grid[k], 0, sz[k], c->stream));
        }
        HIP_TRY(hipEventRecord(e0, c->stream));
        HIP_TRY(launch_dda_build(a, levels, c->dda_grid[0], c->dda_grid[1], c->dda_grid[2], c->dda_grid[3], c->stream));
        HIP_TRY(hipEventRecord(e1, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipEventElapsedTime(&t, e0, e1));
        ms[2] = double(t);
        c->dda_levels = levels;
    }
    ms[3] = double(sz[0] + sz[1] + sz[2] + sz[3]);
    ScratchBuffer b_o, b_d, b_w, b_x;
    HIP_TRY(b_o.alloc(n * 12)); HIP_TRY(b_d.alloc(n * 12)); HIP_TRY(b_w.alloc(n * 32)); HIP_TRY(b_x.alloc(n * 32));
    HIP_TRY(hipMemcpy(b_o.as<float>(), origins, n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b_d.as<float>(), dirs, n * 12, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; rep++) {
        if (!(flags & 4u)) {
            HIP_TRY(hipEventRecord(e0, c->stream));
            HIP_TRY(launch_cast_probe(a, use_wide(c), b_o.as<float>(), b_d.as<float>(), b_w.as<float>(), unsigned(n), c->stream));
            HIP_TRY(hipEventRecord(e1, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipEventElapsedTime(&t, e0, e1));
            ms[0] = double(t);
        }
        if (!dda) continue;
        HIP_TRY(hipEventRecord(e0, c->stream));
        HIP_TRY(launch_dda_probe(a, c->dda_grid[0], c->dda_grid[1], c->dda_grid[2], c->dda_grid[3], levels, b_o.as<float>(), b_d.as<float>(), b_x.as<float>(),
                                 unsigned(n), int(flags & 1u), margin_scale, max_steps, int((flags >> 1) & 1u), c->stream));
        HIP_TRY(hipEventRecord(e1, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipEventElapsedTime(&t, e0, e1));
        ms[1] = double(t);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (!(flags & 4u)) HIP_TRY(hipMemcpy(out_walk, b_w.as<float>(), n * 32, hipMemcpyDeviceToHost));
    if (dda) HIP_TRY(hipMemcpy(out_dda, b_x.as<float>(), n * 32, hipMemcpyDeviceToHost));
    return VXRT_OK;
} VXRT_CATCH
#endif

}  // extern "C"
// ---- touch map (vxrt_debug.h): which 64-byte lines of the scene does a frame read? ---------------------------------------------
namespace {
// set bits of a bitmap, at one bit per line (out[0]) and with neighbouring pairs of lines folded (128-byte lines, out[1])
__global__ __launch_bounds__(256) void touch_count_kernel(const uint32_t* map, size_t words, unsigned long long* out) {
    unsigned long long n64 = 0, n128 = 0;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < words; i += size_t(gridDim.x) * 256) {
        const uint32_t w = map[i];
        n64 += unsigned(__popc(w));
        n128 += unsigned(__popc((w | (w >> 1)) & 0x55555555u));
    }
    for (int off = 32; off > 0; off >>= 1) { n64 += __shfl_down(n64, off, 64); n128 += __shfl_down(n128, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(out, n64); atomicAdd(out + 1, n128); }
}
}  // namespace
extern "C" {

int vxrt_debug_touch_map(vxrt_ctx* c, uint32_t enable) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
#if !VXRT_VARIANTS
    (void)enable;
    set_error("the touch map is not in this build of libvxrt (compile with -DVXRT_VARIANTS=1)");
    return VXRT_E_INVALID;
#else
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = vxrt_sync(c)) return rc;
    for (uint32_t** p : {&c->d_touch_nodes, &c->d_touch_leaves}) { if (*p) (void)hipFree(*p); *p = nullptr; }
    c->touch_node_lines = c->touch_leaf_lines = 0;
    if (!enable) return VXRT_OK;
    if (!c->has_scene) { set_error("no scene set"); return VXRT_E_NOSCENE; }
    const size_t node_bytes = use_wide(c) ? c->wide_count * sizeof(WideRec) : c->svo_count * sizeof(SvoRecord);
    c->touch_node_lines = (node_bytes + 63) / 64;
    c->touch_leaf_lines = (c->leaf_count * sizeof(int32_t) + 63) / 64;
    const size_t wn = (c->touch_node_lines + 31) / 32 + 1, wl = (c->touch_leaf_lines + 31) / 32 + 1;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_touch_nodes), wn * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_touch_leaves), wl * 4));
    HIP_TRY(hipMemsetAsync(c->d_touch_nodes, 0, wn * 4, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_touch_leaves, 0, wl * 4, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VXRT_OK;
#endif
} VXRT_CATCH

int vxrt_debug_touch_count(vxrt_ctx* c, uint64_t out[6], uint32_t reset) try {
    if (!valid_ctx(c) || !out) { set_error("null argument"); return VXRT_E_INVALID; }
    if (c->d_touch_nodes == nullptr) { set_error("no touch map (vxrt_debug_touch_map(ctx, 1) first; -DVXRT_VARIANTS=1 builds only)"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = vxrt_sync(c)) return rc;
    ScratchBuffer b;
    HIP_TRY(b.alloc(4 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(b.p, 0, 4 * sizeof(unsigned long long), c->stream));
    const size_t wn = (c->touch_node_lines + 31) / 32 + 1, wl = (c->touch_leaf_lines + 31) / 32 + 1;
    hipLaunchKernelGGL(touch_count_kernel, dim3(1024), dim3(256), 0, c->stream, c->d_touch_nodes, wn, b.as<unsigned long long>());
    hipLaunchKernelGGL(touch_count_kernel, dim3(1024), dim3(256), 0, c->stream, c->d_touch_leaves, wl, b.as<unsigned long long>() + 2);
    HIP_TRY(hipGetLastError());
    if (reset) {
        HIP_TRY(hipMemsetAsync(c->d_touch_nodes, 0, wn * 4, c->stream));
        HIP_TRY(hipMemsetAsync(c->d_touch_leaves, 0, wl * 4, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long h[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(h, b.p, sizeof h, hipMemcpyDeviceToHost));
    out[0] = h[0]; out[1] = h[2]; out[2] = h[1]; out[3] = h[3];
    out[4] = c->touch_node_lines; out[5] = c->touch_leaf_lines;
    return VXRT_OK;
} VXRT_CATCH

// Diagnostics of the last fused launch of trace stream 0 (VXRT_OPT_FUSED_TAIL; trace.hip: FusedCtl::prof; every 64th wave reports):
// out[0] ticks of the 100 MHz clock from the first wave's start to the moment every head block was finished, out[1] ... to the last wave's end, out[2] / out[3] chunks taken
// before / after that moment, out[4] idle sleeps, out[5] polls for a record's stamp, out[6] head claims, out[7] 0.
int vxrt_debug_fused_profile(vxrt_ctx* c, uint64_t out[8]) try {
    if (!valid_ctx(c) || !out) { set_error("null argument"); return VXRT_E_INVALID; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = vxrt_sync(c)) return rc;
    memset(out, 0, 8 * sizeof(uint64_t));
    if (c->queues.empty() || c->queues[0].fused_ctl == nullptr) return VXRT_OK;
    unsigned long long p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#if VXRT_VARIANTS
    HIP_TRY(hipMemcpy(p, static_cast<char*>(c->queues[0].fused_ctl) + fused_ctl_profile_offset(), sizeof p, hipMemcpyDeviceToHost));
#endif
    const unsigned long long t0 = ~p[0];
    out[0] = p[1] > t0 ? p[1] - t0 : 0; out[1] = p[2] > t0 ? p[2] - t0 : 0;
    out[2] = p[3]; out[3] = p[4]; out[4] = p[5]; out[5] = p[6]; out[6] = p[7];
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
