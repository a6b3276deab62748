// scene_host.h — host-side scene preparation of libvxrt: .vox decoding, the voxel-list adapter,
// the reference-layout sparse octree, the camera basis, the noise table and procedural scenes.
// Pure C++17, no GPU; compiled into libvxrt.so by hipcc as ordinary host code.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/vxrt.h"
#include "../../include/vxrt_debug.h"
#include "../../include/vxrt_host.h"

namespace vxrt {

void set_error(const std::string& msg);  // thread-local text behind vxrt_last_error()

struct Voxel {
    int16_t x, y, z;
    uint8_t m, r, g, b;
};

struct VoxScene {
    uint32_t size[3] = {0, 0, 0};  // SIZE chunk of model 0 (x, y, z in file axes)
    std::vector<Voxel> voxels;     // already in the renderer's axes (x, z_file, y_file)
};

// MagicaVoxel v150 -> voxel list with the semantics of vox::parse + Context::voxels_from_vox
// (src/vox.rs:11-101, src/context.rs:913-933).
int decode_vox(const uint8_t* bytes, size_t len, VoxScene* out);

// Whole MagicaVoxel scenes (vox_scene.cpp): flags = VXRT_VOX_* of vxrt.h; 0 = decode_vox.  bounds: inclusive voxel
// bounding box in the renderer's axes.
int decode_vox_scene(const uint8_t* bytes, size_t len, uint32_t flags, VoxScene* out, int32_t bounds_lo[3], int32_t bounds_hi[3]);
// Context::create_voxels (src/context.rs:838-910) with a seeded generator.
void default_scene(uint32_t seed, std::vector<Voxel>* out);

// The reference's blue-noise archive format (noise_zip.cpp; src/context.rs:1042-1116).
int noise_zip_read(const char* path, std::vector<float>* pixels, uint32_t* size, uint32_t* layers);
int noise_zip_write(const char* path, const float* table, uint32_t size, uint32_t layers);

// Sparse octree in the layout shaders/voxels.comp:58-63 reads: 5-word header
// [cx, cy, cz, root_size, child_size] (f32 bits) followed by 8 int32 slots per node
// (0 empty, >0 child node, <0 leaf word).  Built with the insertion order and overwrite rule of
// Context::create_octree_nodes (src/context.rs:710-773).
struct Octree {
    std::vector<int32_t> words;
    uint32_t depth = 0;  // root_size = 2^depth
    size_t node_count() const { return words.size() < 5 ? 0 : (words.size() - 5) / 8; }
};
int build_octree(const Voxel* voxels, size_t n, Octree* out);

struct CameraBasis {
    float right[3], up[3], forward_ray[3];
};
// Camera::axis_scaled (src/camera.rs:19-28).
CameraBasis camera_axis_scaled(const float dir[3], float fov, uint32_t width, uint32_t height);

// the grammar of Rust's str::parse::<f32>() (`_flux` in MATL chunks, src/vox.rs:93-96)
bool is_rust_f32_literal(const std::string& text);

float noise_value(uint32_t seed, uint32_t index);  // documented in vxrt.h (vxrt_noise_table)

// Level-L Menger sponge: cell (x,y,z) in [0,3^L)^3 is solid unless, at some base-3 digit position,
// at least two of its three digits equal 1.
bool menger_solid(uint32_t level, uint32_t x, uint32_t y, uint32_t z);

}  // namespace vxrt
