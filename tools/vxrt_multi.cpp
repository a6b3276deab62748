// vxrt_multi — a multi-GPU host for the C ABI without torch or Python: one host thread per rank, one vxrt context per rank, the halo
// over RCCL (ncclSend / ncclRecv).  It is the frame of INTEGRATION.md §5 as a program — what a Rust host that replaces
// Context::render (src/context.rs:2004-2075) on a node would mirror:
//
//     vxrt_render(TRACE | TEMPORAL) -> vxrt_halo_pack -> vxrt_stream_wait_context(comm)
//         -> ncclGroupStart; ncclSend x2; ncclRecv x2; ncclGroupEnd                       (on the communication stream)
//     -> vxrt_render(DENOISE_INTERIOR) -> vxrt_context_wait_stream(comm) -> vxrt_halo_unpack -> vxrt_render(DENOISE_EDGE)
//
//   vxrt_multi <scene.vox | menger:<level>[:clip[:period]] | default[:seed]> <width> <height> <frames> <bounces> <radius> <out.ppm>
//              [--ranks N] [--transport rccl|copy] [--band ROWS] [--spp S] [--check] [--halo-rows R] [--pan DY [--near D]]
//
// --transport rccl (default): rank r drives device r; needs N devices (RCCL refuses two ranks on one device; N = 1 works and sends
//     nothing).  ncclCommInitAll makes the communicators in this one process.
// --transport copy: the four transfers of a frame are hipMemcpyPeerAsync on the communication streams, ordered by events and two
//     thread barriers per frame; rank r drives device r mod (device count), so any N runs on one GPU.  Everything but the four
//     nccl calls is the same code: it is how this host is tested where there is one GPU.
// --pan DY: a moving camera — frame f looks along direction + f * DY * (0, 1, 0) (a vertical pan, the motion that crosses band
//     edges); every frame's exchange is sized for the NEXT frame's reprojection with vxrt_halo_rows_for_motion (points no nearer
//     than --near, default 0.25), so the temporal history survives the band edges exactly as on one GPU.
// --self-loop (with --transport rccl --ranks 1): ncclSend / ncclRecv on ONE GPU — the context is rank 0 of 2 and both neighbours are
//     mapped onto the rank itself, so the messages it sends come back to it (not what a frame needs: the image is not checked).  The
//     transport is: after every frame `to_prev` must have arrived as `from_next` and `to_next` as `from_prev`, bit for bit — the
//     in-order matching of two sends and two receives to one peer that a 2-rank job relies on.
// --check: the same frames in ONE context on device 0, compared with the stitched frame bit for bit.
// Prints one JSON line: ranks, devices, ms per frame, halo bytes per rank and frame, differing values.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <thread>

#include "../include/vxrt.hpp"

namespace {

struct Barrier {   // all ranks' threads meet; an error in one of them releases the others
    explicit Barrier(int n) : n_(n) {}
    bool wait() {
        std::unique_lock<std::mutex> l(m_);
        if (failed_) return false;
        const unsigned gen = gen_;
        if (++count_ == n_) { count_ = 0; gen_++; cv_.notify_all(); return !failed_; }
        cv_.wait(l, [&] { return gen_ != gen || failed_; });
        return !failed_;
    }
    void fail() { std::lock_guard<std::mutex> l(m_); failed_ = true; cv_.notify_all(); }
    std::mutex m_; std::condition_variable cv_; int n_, count_ = 0; unsigned gen_ = 0; bool failed_ = false;
};

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void nccl_check(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}

unsigned char srgb8(float x) {  // what a Bgra8UnormSrgb swap chain stores (src/context.rs:696-706)
    if (!(x == x)) x = 0.0f;
    x = std::min(1.0f, std::max(0.0f, x));
    const float y = x <= 0.0031308f ? 12.92f * x : 1.055f * std::pow(x, 1.0f / 2.4f) - 0.055f;
    return static_cast<unsigned char>(y * 255.0f + 0.5f);
}

struct Scene {
    std::string name;
    vxrt::VoxelList voxels;        // voxel scenes
    unsigned level = 0, clip = 0, period = 0;   // menger:<level>...
    bool procedural = false, start_camera = false;
    float extent[3] = {0, 0, 0};
};

Scene load_scene(const std::string& name) {
    Scene s;
    s.name = name;
    if (name.rfind("default", 0) == 0) {
        unsigned seed = 1;
        std::sscanf(name.c_str() + 7, ":%u", &seed);
        s.voxels = vxrt::create_voxels(seed);
        s.start_camera = true;
    } else if (name.rfind("menger:", 0) == 0) {
        std::sscanf(name.c_str() + 7, "%u:%u:%u", &s.level, &s.clip, &s.period);
        s.procedural = true;
        unsigned side = 1;
        for (unsigned l = 0; l < s.level; l++) side *= 3;
        if (s.clip && s.clip < side) side = s.clip;
        s.extent[0] = s.extent[1] = s.extent[2] = side * 0.5f;
    } else {
        std::ifstream f(name, std::ios::binary);
        if (!f) throw std::runtime_error("cannot open " + name);
        std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        s.voxels = vxrt::voxels_from_vox(bytes);
        s.extent[0] = s.voxels.size[0] * 0.5f; s.extent[1] = s.voxels.size[2] * 0.5f; s.extent[2] = s.voxels.size[1] * 0.5f;
    }
    return s;
}

void place(vxrt::Context& ctx, const Scene& s, unsigned radius) {
    if (s.procedural) ctx.set_menger(s.level, s.clip, {0, 0x7b, 0xa2, 0x3f}, s.period);
    else ctx.recreate_octree(s.voxels);
    ctx.denoise_uniforms.radius = radius;
    // close to the model, looking into it (geometry fills the frame: every band edge has something to denoise):
    // position = c + e * (-0.45, 0.3, -0.6), looking at c
    const float e = std::max(s.extent[0], std::max(s.extent[1], s.extent[2]));
    const float k[3] = {-0.45f, 0.3f, -0.6f};
    for (int i = 0; i < 3 && !s.start_camera; i++) {
        const float c = s.extent[i] * 0.5f;
        ctx.camera.position[i] = c + e * k[i];
        ctx.camera.direction[i] = c - ctx.camera.position[i];
    }
}

// frame f of a vertical pan: the placed camera's direction + f * pan * (0, 1, 0)
vxrt::Camera camera_of_frame(const vxrt::Camera& base, float pan, int f) {
    vxrt::Camera c = base;
    c.direction[1] += pan * float(f);
    return c;
}

struct Options {
    std::string scene, out;
    uint32_t width = 0, height = 0, bounces = 3, radius = 0, band = 0, spp = 1, halo_rows = 0;
    int frames = 1, ranks = 0;
    bool rccl = true, check = false, self_loop = false;
    float pan = 0.0f, near_distance = 0.25f;
};

struct Rank {   // what the ranks' threads share with each other (the copy transport reads a neighbour's buffers)
    int device = 0;
    void* bufs[4] = {nullptr, nullptr, nullptr, nullptr};   // to_prev, to_next, from_prev, from_next
    hipStream_t comm = nullptr;
    hipEvent_t packed = nullptr, copied = nullptr;
    std::vector<uint32_t> rows;
    std::vector<float> image;
    size_t message_bytes = 0;
    uint32_t halo_rows = 0;
    int self_loop_frames = 0;
    double seconds = 0.0;
    uint64_t rays = 0;
    std::string error;
};

}  // namespace

int main(int argc, char** argv) {
    if (argc < 8) {
        std::fprintf(stderr, "usage: %s <scene.vox|menger:L[:clip[:period]]|default[:seed]> <width> <height> <frames> <bounces> <radius> <out.ppm> "
                             "[--ranks N] [--transport rccl|copy] [--band ROWS] [--spp S] [--halo-rows R] [--check]\n", argv[0]);
        return 2;
    }
    Options o;
    o.scene = argv[1]; o.width = std::atoi(argv[2]); o.height = std::atoi(argv[3]); o.frames = std::atoi(argv[4]);
    o.bounces = std::atoi(argv[5]); o.radius = std::atoi(argv[6]); o.out = argv[7];
    for (int i = 8; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--ranks") o.ranks = std::atoi(next());
        else if (a == "--transport") { const std::string t = next(); if (t != "rccl" && t != "copy") { std::fprintf(stderr, "transport: rccl or copy\n"); return 2; } o.rccl = t == "rccl"; }
        else if (a == "--band") o.band = std::atoi(next());
        else if (a == "--spp") o.spp = std::atoi(next());
        else if (a == "--halo-rows") o.halo_rows = std::atoi(next());
        else if (a == "--check") o.check = true;
        else if (a == "--self-loop") o.self_loop = true;
        else if (a == "--pan") o.pan = float(std::atof(next()));
        else if (a == "--near") o.near_distance = float(std::atof(next()));
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    try {
        int ndev = 0;
        hip_check(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
        if (ndev < 1) throw std::runtime_error("no HIP device");
        const int n = o.ranks > 0 ? o.ranks : ndev;
        if (o.self_loop && !(o.rccl && n == 1 && !o.check)) throw std::runtime_error("--self-loop goes with --transport rccl --ranks 1 (and without --check)");
        const int ctx_ranks = o.self_loop ? 2 : n;       // what the contexts believe
        if (o.rccl && n > ndev) throw std::runtime_error("the rccl transport needs one device per rank (RCCL refuses two ranks on one device): " +
                                                         std::to_string(n) + " ranks, " + std::to_string(ndev) + " devices; use --transport copy");
        if (o.band == 0) {   // >= 8 radius in whole 16-row tiles (the halo at most a quarter of a rank's rows), 16 without a window
            o.band = o.radius == 0 ? 16u : std::max(48u, (8u * o.radius + 15u) / 16u * 16u);
        }
        const Scene scene = load_scene(o.scene);
        std::vector<Rank> ranks(static_cast<size_t>(n));
        std::vector<int> devices(static_cast<size_t>(n));
        for (int r = 0; r < n; r++) ranks[size_t(r)].device = devices[size_t(r)] = o.rccl ? r : r % ndev;
        std::vector<ncclComm_t> comms(static_cast<size_t>(n), nullptr);
        if (o.rccl) nccl_check(ncclCommInitAll(comms.data(), n, devices.data()), "ncclCommInitAll");
        if (!o.rccl)   // a rank's copies read its neighbours' memory
            for (int a = 0; a < ndev; a++)
                for (int b = 0; b < ndev; b++)
                    if (a != b) { (void)hipSetDevice(a); (void)hipDeviceEnablePeerAccess(b, 0); }

        Barrier barrier(n);
        std::mutex abort_mutex;
        bool aborted = false;
        auto run = [&](int r) {
            Rank& me = ranks[size_t(r)];
            try {
                hip_check(hipSetDevice(me.device), "hipSetDevice");
                vxrt::Context ctx(o.width, o.height, o.bounces, me.device, /*frames_in_flight=*/1, uint32_t(r), uint32_t(ctx_ranks),
                                  /*frames_per_launch=*/o.spp > 1 ? std::min(o.spp, 32u) : 1u, o.band);
                place(ctx, scene, o.radius);
                if (o.halo_rows) ctx.set_option(VXRT_OPT_HALO_ROWS, o.halo_rows);
                const vxrt::Camera base = ctx.camera;
                if (o.pan != 0.0f && n > 1) {                     // the messages are sized for the largest motion of the path (one size for the run)
                    uint32_t most = 1;
                    const uint32_t cap = std::max<uint32_t>(ctx.halo_info().max_rows, 1u);    // what this band layout can carry (<= the band height)
                    for (int f = 0; f + 1 < o.frames; f++) {
                        const vxrt::Camera a = camera_of_frame(base, o.pan, f), b = camera_of_frame(base, o.pan, f + 1);
                        uint32_t rows = 0;
                        vxrt::check(vxrt_halo_rows_for_motion(a.position.data(), a.direction.data(), b.position.data(), b.direction.data(), a.fov, o.width,
                                                              o.height, o.near_distance, cap, &rows), "vxrt_halo_rows_for_motion");
                        most = std::max(most, rows);
                    }
                    ctx.set_option(VXRT_OPT_HALO_ROWS, most);
                    me.halo_rows = most;
                }
                const vxrt_halo_info info = ctx.halo_info();
                me.message_bytes = ctx_ranks > 1 ? size_t(info.message_bytes) : 0;
                me.halo_rows = info.rows;
                hip_check(hipStreamCreateWithFlags(&me.comm, hipStreamNonBlocking), "hipStreamCreate");
                hip_check(hipEventCreateWithFlags(&me.packed, hipEventDisableTiming), "hipEventCreate");
                hip_check(hipEventCreateWithFlags(&me.copied, hipEventDisableTiming), "hipEventCreate");
                for (void*& b : me.bufs) {
                    hip_check(hipMalloc(&b, std::max<size_t>(me.message_bytes, 256)), "hipMalloc");
                    hip_check(hipMemset(b, 0, std::max<size_t>(me.message_bytes, 256)), "hipMemset");
                }
                const int prev = (r + n - 1) % n, next = (r + 1) % n;
                if (!barrier.wait()) return;                      // every rank's buffers and events exist
                std::chrono::steady_clock::time_point t0;
                for (int f = 0; f < o.frames; f++) {
                    if (f == 1) { ctx.sync(); if (!barrier.wait()) return; t0 = std::chrono::steady_clock::now(); }
                    ctx.camera = camera_of_frame(base, o.pan, f);
                    // without a window (radius 0) the denoise stage is a per-pixel pass the library fuses into the temporal kernel; the
                    // halo then carries the history rows for the next frame's reprojection only
                    const uint32_t first = (ctx_ranks == 1 || o.radius == 0) ? uint32_t(VXRT_ALL) : uint32_t(VXRT_TRACE | VXRT_TEMPORAL);
                    if (o.spp > 1) ctx.render_spp(first, o.spp);
                    else ctx.render(first);
                    if (ctx_ranks == 1) continue;
                    if (!o.rccl) vxrt::check(vxrt_context_wait_stream(ctx.handle(), me.comm), "vxrt_context_wait_stream");   // the neighbours have read last frame's messages
                    ctx.halo_pack(me.bufs[0], me.bufs[1], me.comm);      // one kernel; the communication stream waits for it (event)
                    if (o.rccl) {
                        const size_t bytes = me.message_bytes;
                        nccl_check(ncclGroupStart(), "ncclGroupStart");
                        nccl_check(ncclSend(me.bufs[0], bytes, ncclChar, prev, comms[size_t(r)], me.comm), "ncclSend");
                        nccl_check(ncclSend(me.bufs[1], bytes, ncclChar, next, comms[size_t(r)], me.comm), "ncclSend");
                        nccl_check(ncclRecv(me.bufs[3], bytes, ncclChar, next, comms[size_t(r)], me.comm), "ncclRecv");   // what `next` addressed to ITS prev
                        nccl_check(ncclRecv(me.bufs[2], bytes, ncclChar, prev, comms[size_t(r)], me.comm), "ncclRecv");   // what `prev` addressed to ITS next
                        nccl_check(ncclGroupEnd(), "ncclGroupEnd");
                    } else {
                        hip_check(hipEventRecord(me.packed, me.comm), "hipEventRecord");
                        if (!barrier.wait()) return;               // the neighbours' `packed` events of this frame are recorded
                        Rank& p = ranks[size_t(prev)];
                        Rank& q = ranks[size_t(next)];
                        hip_check(hipStreamWaitEvent(me.comm, p.packed, 0), "hipStreamWaitEvent");
                        hip_check(hipStreamWaitEvent(me.comm, q.packed, 0), "hipStreamWaitEvent");
                        hip_check(hipMemcpyPeerAsync(me.bufs[2], me.device, p.bufs[1], p.device, me.message_bytes, me.comm), "hipMemcpyPeerAsync");
                        hip_check(hipMemcpyPeerAsync(me.bufs[3], me.device, q.bufs[0], q.device, me.message_bytes, me.comm), "hipMemcpyPeerAsync");
                        hip_check(hipEventRecord(me.copied, me.comm), "hipEventRecord");
                        if (!barrier.wait()) return;               // ... and everybody's `copied`
                        hip_check(hipStreamWaitEvent(me.comm, p.copied, 0), "hipStreamWaitEvent");   // my next pack (which waits for my
                        hip_check(hipStreamWaitEvent(me.comm, q.copied, 0), "hipStreamWaitEvent");   // comm stream) comes after their reads
                    }
                    if (o.radius > 0) ctx.render_stage(VXRT_DENOISE_INTERIOR);     // runs while the messages travel
                    ctx.halo_unpack(me.bufs[2], me.bufs[3], me.comm);              // the context's stream waits for the receives (event)
                    if (o.radius > 0) ctx.render_stage(VXRT_DENOISE_EDGE);
                    if (o.self_loop) {      // the transport test: what left must have come back crossed, bit for bit
                        ctx.sync();
                        hip_check(hipStreamSynchronize(me.comm), "hipStreamSynchronize");
                        std::vector<uint8_t> host[4];
                        for (int b = 0; b < 4; b++) {
                            host[b].resize(me.message_bytes);
                            hip_check(hipMemcpy(host[b].data(), me.bufs[b], me.message_bytes, hipMemcpyDeviceToHost), "hipMemcpy");
                        }
                        const bool crossed = host[0] == host[3] && host[1] == host[2];
                        const bool distinct = host[0] != host[1] && std::any_of(host[0].begin(), host[0].end(), [](uint8_t v) { return v != 0; });
                        if (!crossed || !distinct) throw std::runtime_error("self-loop: the messages did not come back crossed (frame " + std::to_string(f) + ")");
                        me.self_loop_frames++;
                    }
                }
                ctx.sync();
                hip_check(hipStreamSynchronize(me.comm), "hipStreamSynchronize");
                if (!barrier.wait()) return;
                me.seconds = o.frames > 1 ? std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() : 0.0;
                me.image = ctx.read(VXRT_DENOISED);
                uint32_t count = 0;
                vxrt::check(vxrt_local_rows(ctx.handle(), &count, nullptr), "vxrt_local_rows");
                me.rows.resize(count);
                vxrt::check(vxrt_local_rows(ctx.handle(), &count, me.rows.data()), "vxrt_local_rows");
                me.rays = ctx.stats().rays;
                if (!barrier.wait()) return;                       // nobody frees what a neighbour may still read
                for (void* b : me.bufs) (void)hipFree(b);
                (void)hipEventDestroy(me.packed); (void)hipEventDestroy(me.copied); (void)hipStreamDestroy(me.comm);
            } catch (const std::exception& ex) {
                me.error = ex.what();
                barrier.fail();
                // Over RCCL the peers do not meet at the barrier every frame: they may be inside ncclGroupEnd, or their streams may be
                // waiting on sends and receives this rank will never post.  Abort every communicator so that they come out (with an
                // error of their own) instead of sitting there until somebody's timeout; the process then exits 1.
                if (o.rccl) {
                    std::lock_guard<std::mutex> l(abort_mutex);
                    if (!aborted) {
                        aborted = true;
                        for (ncclComm_t cm : comms) if (cm) (void)ncclCommAbort(cm);      // (an aborted communicator is not destroyed again below)
                    }
                }
            }
        };
        std::vector<std::thread> threads;
        for (int r = 0; r < n; r++) threads.emplace_back(run, r);
        for (std::thread& t : threads) t.join();
        if (o.rccl && !aborted) for (ncclComm_t c : comms) if (c) (void)ncclCommDestroy(c);
        for (int r = 0; r < n; r++)
            if (!ranks[size_t(r)].error.empty()) throw std::runtime_error("rank " + std::to_string(r) + ": " + ranks[size_t(r)].error);

        // stitch the ranks' rows
        std::vector<float> full(size_t(o.width) * o.height * 4, 0.0f);
        uint64_t rays = 0;
        double seconds = 0.0;
        for (const Rank& k : ranks) {
            for (size_t i = 0; i < k.rows.size(); i++)
                std::memcpy(&full[size_t(k.rows[i]) * o.width * 4], &k.image[i * o.width * 4], size_t(o.width) * 16);
            rays += k.rays;
            seconds = std::max(seconds, k.seconds);
        }
        long long differing = -1;
        bool rays_equal = true;
        if (o.check) {
            hip_check(hipSetDevice(0), "hipSetDevice");
            vxrt::Context one(o.width, o.height, o.bounces, 0, 1, 0, 1, o.spp > 1 ? std::min(o.spp, 32u) : 1u);
            place(one, scene, o.radius);
            const vxrt::Camera base = one.camera;
            for (int f = 0; f < o.frames; f++) {
                one.camera = camera_of_frame(base, o.pan, f);
                if (o.spp > 1) one.render_spp(VXRT_ALL, o.spp);
                else one.render(VXRT_ALL);
            }
            const std::vector<float> want = one.read(VXRT_DENOISED);
            differing = 0;
            for (size_t i = 0; i < want.size(); i++) {
                const float a = full[i], b = want[i];
                if (!(a == b || (a != a && b != b))) differing++;
            }
            rays_equal = one.stats().rays == rays;
        }
        std::ofstream ppm(o.out, std::ios::binary);
        ppm << "P6\n" << o.width << " " << o.height << "\n255\n";
        for (size_t p = 0; p < size_t(o.width) * o.height; p++) {
            const unsigned char rgb[3] = {srgb8(full[4 * p]), srgb8(full[4 * p + 1]), srgb8(full[4 * p + 2])};
            ppm.write(reinterpret_cast<const char*>(rgb), 3);
        }
        std::string devs;
        for (int r = 0; r < n; r++) devs += (r ? ", " : "") + std::to_string(devices[size_t(r)]);
        int version = 0;
        (void)ncclGetVersion(&version);
        std::printf("{\"tool\": \"vxrt_multi\", \"scene\": \"%s\", \"width\": %u, \"height\": %u, \"frames\": %d, \"spp\": %u, \"radius\": %u, \"ranks\": %d, "
                    "\"transport\": \"%s\", \"rccl_version\": %d, \"devices\": [%s], \"band_rows\": %u, \"halo_rows\": %u, \"halo_bytes_per_rank_per_frame\": %zu, "
                    "\"ms_per_frame\": %.4f, \"rays\": %llu, \"checked\": %s, \"differing_values\": %lld, \"rays_equal\": %s, \"self_loop_frames_crossed\": %d}\n",
                    o.scene.c_str(), o.width, o.height, o.frames, o.spp, o.radius, n, o.rccl ? "rccl" : "copy", version, devs.c_str(), o.band, ranks[0].halo_rows,
                    2 * ranks[0].message_bytes, o.frames > 1 ? seconds / (o.frames - 1) * 1e3 : 0.0, (unsigned long long)rays,
                    o.check ? "true" : "false", differing, rays_equal ? "true" : "false", ranks[0].self_loop_frames);
        return (o.check && (differing != 0 || !rays_equal)) ? 3 : 0;
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "vxrt_multi: %s\n", ex.what());
        return 1;
    }
}
