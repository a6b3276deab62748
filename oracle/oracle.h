// ORACLE — test infrastructure only.  Nothing in the product path may include, link or call this.
//
// CPU restatement of the hot path of nolanderc/gpu-voxel-raytracer (scene prep + the three compute
// shaders + the dead src/cpu.rs ray caster).  Parity status: the three SHADERS are pinned by the
// reference's own compiled modules, executed by ospirv.cpp (tests/test_oracle_spirv_exec.py), up to the
// operations SPIR-V leaves to a driver (oshaders.cpp: U1-U8); the HOST-SIDE restatements (parser, octree
// builder, camera, src/cpu.rs) stay UNPINNED by any reference output — it ships no tests or golden vectors
// and cannot be built here — and are held by review, the Appendix-C node-count table,
// sha256 known answers and an independent dense-grid DDA (odda.cpp).
#pragma once
#include <cstddef>
#include <cstdint>

#include "ovec.h"

enum {
    ORC_E_MAGIC = -1,     // "invalid magic number"            src/vox.rs:12-14
    ORC_E_VERSION = -2,   // "unsupported VOX-format"          src/vox.rs:17-19
    ORC_E_NOMAIN = -3,    // "missing MAIN chunk"              src/vox.rs:21
    ORC_E_EOF = -4,       // "unexpected end of file"          src/vox.rs:262-283
    ORC_E_CHUNK = -5,     // "expected chunk X, found chunk Y" src/vox.rs:230-241
    ORC_E_MATERIAL = -6,  // unsupported material / bad _flux  src/vox.rs:82-96
    ORC_E_NOMATL = -7,    // voxels_from_vox .unwrap() panic   src/context.rs:919
    ORC_E_NOMODEL = -8,   // models[0] out of bounds           src/context.rs:916
    ORC_E_SPLITLEAF = -9, // todo!("split leaf ...")           src/context.rs:746
};

// Uniforms (src/context.rs:425-469), std140 offsets as in shaders/voxels.comp:28-49 — 148 bytes.
struct OrcUniforms {
    float camera_origin[4];
    float camera_right[4];
    float camera_up[4];
    float camera_forward[4];
    float light[4];
    float global_time;
    uint32_t still_sample;
    uint32_t frame_number;
    float emit_strength;
    float sun_strength;
    float sun_size;
    float sun_yaw;
    float sun_pitch;
    float sun_color[4];
    float sky_color[4];
    float specularity;
};
static_assert(sizeof(OrcUniforms) == 148, "Uniforms must be 148 bytes");

// TemporalUniforms (src/context.rs:502-515), DenoiseUniforms (src/context.rs:304-314).
struct OrcTemporal { float sample_blending, maximum_blending, blending_distance_cutoff; };
struct OrcDenoise { uint32_t radius; float sigma_distance, sigma_range, albedo_factor; };

namespace orc {
struct Hit { float time; int32_t node; V3 normal; int iterations; };
// cast_bounded_ray (shaders/voxels.comp:134-247) on a reference-layout octree buffer.
bool cast_bounded_ray(const int32_t* octree, V3 origin, V3 dir, float max_distance, Hit* hit);

// The octree buffer as the walk reads it: the header fields and nodes[] — stored (nodes) or, for a procedural scene too large
// to store (BASELINE config 5: 261 M nodes = 8.4 GB in the reference's layout), materialised on first touch (lazy, oprocedural.cpp).
struct LazyTree;
struct LazyPool;
struct Scene { V3 root_center; float root_size; const int32_t* nodes; LazyTree* lazy; };
Scene scene_of(const int32_t* octree);
bool cast_bounded_ray(const Scene& scene, V3 origin, V3 dir, float max_distance, Hit* hit);
int32_t lazy_fetch(LazyTree* tree, int32_t node, uint32_t octant);   // nodes[8 * node + octant]
LazyPool* lazy_pool(uint32_t level, uint32_t clip, const uint8_t mrgb[4], uint32_t emissive_period);   // one per scene, kept for the process
LazyTree* lazy_acquire(LazyPool* pool);      // a tree for the calling thread's exclusive use ...
void lazy_release(LazyPool* pool, LazyTree* tree);   // ... handed back with what it has materialised
Scene lazy_scene(LazyTree* tree);

// Two of the driver-defined choices as functions, shared by oshaders.cpp and the SPIR-V interpreter (ospirv.cpp):
// U5: affine inverse of [R U F O; 0 0 0 1] — rows of A^-1 and t = -A^-1 O, as 12 floats (temporal.comp:75-82)
void affine_inverse(const float* R, const float* U, const float* F, const float* O, float inv[12]);
// U4: an rgba32f image behind the Linear / ClampToEdge sampler of src/context.rs:980-989
struct Tex {
    const float* data; int w, h;
    void fetch(int x, int y, float* o) const;
    void sample(float u, float v, float* o) const;
};
}  // namespace orc
