// vxrt.hpp — C++17 host-side mirror of the reference's render-loop types over the C ABI (vxrt.h).
//
// The reference's host is Rust (src/main.rs, src/context.rs, src/camera.rs); no Rust toolchain exists in the build
// image, so the compiled-language host above the C ABI is this header: the same names and argument meaning —
//   Camera{position, direction, fov}                         src/camera.rs:5-9
//   Uniforms / TemporalUniforms / DenoiseUniforms            src/context.rs:425-525, 304-325
//   Context::new / resize / recreate_octree / render         src/context.rs:595-660, 1430-1461, 799-810, 2004-2075
// Errors are reported as vxrt::Error exceptions carrying the vxrt_status and the library's message, where the
// reference returns anyhow::Error (or panics).  Header-only; link with -lvxrt.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "vxrt.h"
#include "vxrt_host.h"

namespace vxrt {

class Error : public std::runtime_error {
  public:
    Error(int status, const std::string& where)
        : std::runtime_error(where + ": " + vxrt_status_string(status) + " (" + vxrt_last_error() + ")"), status_(status) {}
    int status() const { return status_; }

  private:
    int status_;
};

inline void check(int status, const char* where) {
    if (status != VXRT_OK) throw Error(status, where);
}

using Vec3 = std::array<float, 3>;

// src/camera.rs:5-9; the default is the reference's start camera (src/context.rs:618-622), fov = 70 degrees.
struct Camera {
    Vec3 position{0.0f, 0.0f, -2.0f};
    Vec3 direction{0.0f, 0.0f, 1.0f};
    float fov = 70.0f * (3.14159274f / 180.0f);

    // Camera::axis_scaled (src/camera.rs:19-28): right, up, forward_ray
    std::array<Vec3, 3> axis_scaled(uint32_t width, uint32_t height) const {
        std::array<Vec3, 3> out{};
        check(vxrt_camera_axis_scaled(position.data(), direction.data(), fov, width, height, out[0].data(), out[1].data(), out[2].data()),
              "vxrt_camera_axis_scaled");
        return out;
    }
};

struct Uniforms : vxrt_uniforms {
    Uniforms() { vxrt_default_uniforms(this); }   // Uniforms::default(), src/context.rs:471-498
};
struct TemporalUniforms : vxrt_temporal {
    TemporalUniforms() { vxrt_default_temporal(this); }
};
struct DenoiseUniforms : vxrt_denoise {
    DenoiseUniforms() { vxrt_default_denoise(this); }
};

// A voxel as the reference's adapters produce it: ([i16; 3], [material, r, g, b]) (src/context.rs:777, 913-933).
struct VoxelList {
    std::vector<std::array<int16_t, 3>> pos;
    std::vector<std::array<uint8_t, 4>> mrgb;
    std::array<uint32_t, 3> size{};
};

// vox::parse + Context::voxels_from_vox (src/vox.rs:11-70, src/context.rs:913-933)
inline VoxelList voxels_from_vox(const std::vector<uint8_t>& bytes) {
    VoxelList v;
    size_t n = 0;
    check(vxrt_vox_to_voxels(bytes.data(), bytes.size(), nullptr, nullptr, 0, &n, v.size.data()), "vxrt_vox_to_voxels");
    v.pos.resize(n);
    v.mrgb.resize(n);
    check(vxrt_vox_to_voxels(bytes.data(), bytes.size(), reinterpret_cast<int16_t(*)[3]>(v.pos.data()),
                             reinterpret_cast<uint8_t(*)[4]>(v.mrgb.data()), n, &n, v.size.data()),
          "vxrt_vox_to_voxels");
    return v;
}

// Whole MagicaVoxel scenes: every shape instance of the scene graph (flags: VXRT_VOX_*, see vxrt.h).
inline VoxelList voxels_from_vox_scene(const std::vector<uint8_t>& bytes, uint32_t flags = VXRT_VOX_ALL_MODELS) {
    VoxelList v;
    size_t n = 0;
    check(vxrt_vox_scene_to_voxels(bytes.data(), bytes.size(), flags, nullptr, nullptr, 0, &n, nullptr, nullptr), "vxrt_vox_scene_to_voxels");
    v.pos.resize(n);
    v.mrgb.resize(n);
    check(vxrt_vox_scene_to_voxels(bytes.data(), bytes.size(), flags, reinterpret_cast<int16_t(*)[3]>(v.pos.data()),
                                   reinterpret_cast<uint8_t(*)[4]>(v.mrgb.data()), n, &n, nullptr, nullptr),
          "vxrt_vox_scene_to_voxels");
    return v;
}

// Context::create_voxels (src/context.rs:838-910): the scene the reference starts with, seeded.
inline VoxelList create_voxels(uint32_t seed = 1) {
    VoxelList v;
    size_t n = 0;
    check(vxrt_default_scene_voxels(seed, nullptr, nullptr, 0, &n), "vxrt_default_scene_voxels");
    v.pos.resize(n);
    v.mrgb.resize(n);
    check(vxrt_default_scene_voxels(seed, reinterpret_cast<int16_t(*)[3]>(v.pos.data()), reinterpret_cast<uint8_t(*)[4]>(v.mrgb.data()), n, &n),
          "vxrt_default_scene_voxels");
    return v;
}

// Context::load_blue_noise (src/context.rs:1042-1085): (image size, all images' pixels appended).
inline std::pair<uint32_t, std::vector<float>> load_blue_noise(const std::string& path) {
    uint32_t size = 0, layers = 0;
    check(vxrt_noise_zip_read(path.c_str(), nullptr, 0, &size, &layers), "vxrt_noise_zip_read");
    std::vector<float> px(size_t(layers) * size * size);
    check(vxrt_noise_zip_read(path.c_str(), px.data(), px.size(), &size, &layers), "vxrt_noise_zip_read");
    return {size, std::move(px)};
}

class Context {
  public:
    Camera camera;
    Uniforms uniforms;
    TemporalUniforms temporal_uniforms;
    DenoiseUniforms denoise_uniforms;

    Context(uint32_t width, uint32_t height, uint32_t max_bounces = 3, int device = 0, uint32_t frames_in_flight = 1,
            uint32_t rank = 0, uint32_t nranks = 1, uint32_t frames_per_launch = 1, uint32_t band_rows = 16)
        : width_(width), height_(height) {
        vxrt_config cfg{};
        cfg.width = width; cfg.height = height; cfg.device = device; cfg.max_bounces = max_bounces;
        cfg.noise_seed = 0x5EED0001u; cfg.noise = nullptr; cfg.rank = rank; cfg.nranks = nranks; cfg.band_rows = band_rows;
        cfg.frames_in_flight = frames_in_flight; cfg.tracer = 0; cfg.frames_per_launch = frames_per_launch;
        check(vxrt_create(&cfg, &ctx_), "vxrt_create");
    }
    ~Context() { vxrt_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;

    // Context::recreate_octree (src/context.rs:799-810)
    void recreate_octree(const VoxelList& voxels) {
        check(vxrt_set_voxels(ctx_, reinterpret_cast<const int16_t(*)[3]>(voxels.pos.data()),
                              reinterpret_cast<const uint8_t(*)[4]>(voxels.mrgb.data()), voxels.pos.size()),
              "vxrt_set_voxels");
    }
    void load_vox(const std::string& path) { check(vxrt_load_vox(ctx_, path.c_str()), "vxrt_load_vox"); }
    // Context::create_blue_noise_buffer (src/context.rs:1016-1040): the archive must hold 128x128 images, 512 of them
    // (BLUE_NOISE_SIZE, and the table length shaders/voxels.comp:65-71 indexes).
    void load_blue_noise(const std::string& path) {
        auto [size, px] = vxrt::load_blue_noise(path);
        if (size != 128 || px.size() != size_t(512) * 128 * 128) throw Error(VXRT_E_NOISE, "blue noise images must be 512 x 128 x 128");
        check(vxrt_set_noise(ctx_, px.data()), "vxrt_set_noise");
    }
    // Makes the table the reference's repository does not ship, on the GPU (include/vxrt_bluenoise.h).
    void generate_blue_noise(uint32_t seed = 0x5EED0001u, int device = 0) {
        std::vector<float> table(size_t(512) * 128 * 128);
        check(vxrt_blue_noise(device, seed, 128, 0, 512, table.data()), "vxrt_blue_noise");
        check(vxrt_set_noise(ctx_, table.data()), "vxrt_set_noise");
    }
    void set_menger(uint32_t level, uint32_t clip, std::array<uint8_t, 4> mrgb, uint32_t emissive_period) {
        check(vxrt_set_menger(ctx_, level, clip, mrgb.data(), emissive_period), "vxrt_set_menger");
    }
    // Context::resize (src/context.rs:1430-1461)
    void resize(uint32_t width, uint32_t height) {
        check(vxrt_resize(ctx_, width, height), "vxrt_resize");
        width_ = width; height_ = height;
    }
    // Context::update_bindings + render (src/context.rs:2136-2162, 2004-2075)
    void render(uint32_t flags = VXRT_ALL) {
        check(vxrt_set_camera(ctx_, camera.position.data(), camera.direction.data(), camera.fov), "vxrt_set_camera");
        check(vxrt_set_scene_params(ctx_, &uniforms), "vxrt_set_scene_params");
        check(vxrt_set_temporal(ctx_, &temporal_uniforms), "vxrt_set_temporal");
        check(vxrt_set_denoise(ctx_, &denoise_uniforms), "vxrt_set_denoise");
        check(vxrt_render(ctx_, flags), "vxrt_render");
    }
    // `count` frames with the camera and parameters at rest (one trace launch per frames_per_launch frames)
    void render_frames(uint32_t flags, uint32_t count) {
        check(vxrt_set_camera(ctx_, camera.position.data(), camera.direction.data(), camera.fov), "vxrt_set_camera");
        check(vxrt_set_scene_params(ctx_, &uniforms), "vxrt_set_scene_params");
        check(vxrt_set_temporal(ctx_, &temporal_uniforms), "vxrt_set_temporal");
        check(vxrt_set_denoise(ctx_, &denoise_uniforms), "vxrt_set_denoise");
        check(vxrt_render_frames(ctx_, flags, count), "vxrt_render_frames");
    }
    // frames along a camera path (vxrt_render_path); `camera` ends at the last pose
    void render_path(uint32_t flags, const std::vector<Vec3>& positions, const std::vector<Vec3>& directions) {
        if (positions.size() != directions.size()) throw Error(VXRT_E_INVALID, "render_path: one direction per position");
        check(vxrt_set_scene_params(ctx_, &uniforms), "vxrt_set_scene_params");
        check(vxrt_set_temporal(ctx_, &temporal_uniforms), "vxrt_set_temporal");
        check(vxrt_set_denoise(ctx_, &denoise_uniforms), "vxrt_set_denoise");
        check(vxrt_render_path(ctx_, flags, uint32_t(positions.size()), reinterpret_cast<const float(*)[3]>(positions.data()),
                               reinterpret_cast<const float(*)[3]>(directions.data()), camera.fov), "vxrt_render_path");
        if (!positions.empty()) { camera.position = positions.back(); camera.direction = directions.back(); }
    }
    // one displayed frame of `spp` samples per pixel (vxrt_render_spp)
    void render_spp(uint32_t flags, uint32_t spp) {
        check(vxrt_set_camera(ctx_, camera.position.data(), camera.direction.data(), camera.fov), "vxrt_set_camera");
        check(vxrt_set_scene_params(ctx_, &uniforms), "vxrt_set_scene_params");
        check(vxrt_set_temporal(ctx_, &temporal_uniforms), "vxrt_set_temporal");
        check(vxrt_set_denoise(ctx_, &denoise_uniforms), "vxrt_set_denoise");
        check(vxrt_render_spp(ctx_, flags, spp), "vxrt_render_spp");
    }
    // vxrt_render without re-pushing the parameter blocks: the stages of a frame around a halo exchange
    void render_stage(uint32_t flags) { check(vxrt_render(ctx_, flags), "vxrt_render"); }
    // Multi-GPU halo (vxrt.h "halo"): one frame of a rank = render(VXRT_TRACE | VXRT_TEMPORAL); halo_pack; [send / receive on comm_stream];
    // render_stage(VXRT_DENOISE_INTERIOR); halo_unpack; render_stage(VXRT_DENOISE_EDGE) — ordered by events, the host never waits.
    vxrt_halo_info halo_info() {
        check(vxrt_set_denoise(ctx_, &denoise_uniforms), "vxrt_set_denoise");   // the halo's row count follows the radius
        vxrt_halo_info info{};
        check(vxrt_halo_info_get(ctx_, &info), "vxrt_halo_info_get");
        return info;
    }
    void halo_pack(void* dev_to_prev, void* dev_to_next, void* comm_stream) {
        check(vxrt_halo_pack(ctx_, dev_to_prev, dev_to_next), "vxrt_halo_pack");
        check(vxrt_stream_wait_context(ctx_, comm_stream), "vxrt_stream_wait_context");
    }
    void halo_unpack(const void* dev_from_prev, const void* dev_from_next, void* comm_stream) {
        check(vxrt_context_wait_stream(ctx_, comm_stream), "vxrt_context_wait_stream");
        check(vxrt_halo_unpack(ctx_, dev_from_prev, dev_from_next), "vxrt_halo_unpack");
    }
    // run-time options (vxrt.h: VXRT_OPT_DENOISE_MODE, VXRT_OPT_TAIL_CAPACITY, VXRT_OPT_SCENE_FORMAT, VXRT_OPT_HALO_ROWS, VXRT_OPT_SKY_CULL)
    void set_option(vxrt_option option, uint32_t value) { check(vxrt_set_option(ctx_, option, value), "vxrt_set_option"); }
    void sync() { check(vxrt_sync(ctx_), "vxrt_sync"); }
    std::vector<float> read(vxrt_image which) {
        uint32_t rows = 0;
        check(vxrt_local_rows(ctx_, &rows, nullptr), "vxrt_local_rows");
        std::vector<float> img(size_t(rows) * width_ * 4);
        check(vxrt_read(ctx_, which, img.data(), img.size() * sizeof(float)), "vxrt_read");
        return img;
    }
    // floats of one image of this context (its local rows x width x 4)
    size_t image_floats() const {
        uint32_t rows = 0;
        check(vxrt_local_rows(ctx_, &rows, nullptr), "vxrt_local_rows");
        return size_t(rows) * width_ * 4;
    }
    // The non-blocking read-back (vxrt_read_async / vxrt_read_wait): what a host that shows or stores EVERY frame calls where the
    // reference presents (src/context.rs:2046-2070).  Frame k: render(...); read_async(VXRT_DENOISED, buf[k & 1], k & 1); then
    // read_wait((k + 1) & 1) and frame k - 1 is in buf[(k + 1) & 1] — it travelled while frame k rendered.
    void read_async(vxrt_image which, class PinnedImage& dst, uint32_t slot);
    void read_wait(uint32_t slot) { check(vxrt_read_wait(ctx_, slot), "vxrt_read_wait"); }
    vxrt_stats stats() {
        vxrt_stats s{};
        check(vxrt_get_stats(ctx_, &s), "vxrt_get_stats");
        return s;
    }
    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }
    vxrt_ctx* handle() { return ctx_; }

  private:
    vxrt_ctx* ctx_ = nullptr;
    uint32_t width_, height_;
};

// Pinned host memory for Context::read_async (vxrt_host_alloc / vxrt_host_free): a host needs no HIP binding of its own for it.
class PinnedImage {
  public:
    explicit PinnedImage(size_t floats) : floats_(floats) {
        void* p = nullptr;
        check(vxrt_host_alloc(floats * sizeof(float), &p), "vxrt_host_alloc");
        data_ = static_cast<float*>(p);
    }
    ~PinnedImage() { if (data_) (void)vxrt_host_free(data_); }
    PinnedImage(const PinnedImage&) = delete;
    PinnedImage& operator=(const PinnedImage&) = delete;
    float* data() { return data_; }
    const float* data() const { return data_; }
    size_t size() const { return floats_; }
    size_t bytes() const { return floats_ * sizeof(float); }

  private:
    float* data_ = nullptr;
    size_t floats_;
};

inline void Context::read_async(vxrt_image which, PinnedImage& dst, uint32_t slot) {
    check(vxrt_read_async(ctx_, which, dst.data(), dst.bytes(), slot), "vxrt_read_async");
}

}  // namespace vxrt
