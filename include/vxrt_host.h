/*
 * vxrt_host.h — host-side helpers of libvxrt.so that need no context: what src/vox.rs (the .vox decoder), Context::voxels_from_vox and
 * create_octree (src/context.rs:710-834, 913-933) and Camera::axis_scaled (src/camera.rs:19-28) do in the reference, callable on
 * their own for tools and tests; the generator and the archive format of the blue-noise table the reference loads
 * (src/context.rs:1016-1116); the voxel-list forms of the procedural scenes.  A host that only renders needs none of these:
 * vxrt_load_vox / vxrt_set_voxels / vxrt_set_camera of vxrt.h call the same code inside the library.
 * All of them but vxrt_blue_noise run without a GPU.  Conventions as in vxrt.h: 0 or a negative vxrt_status, counts returned
 * through *n, nothing written past cap.
 */
#ifndef VXRT_HOST_H
#define VXRT_HOST_H

#include "vxrt.h"

#ifdef __cplusplus
extern "C" {
#endif

int vxrt_vox_to_voxels(const uint8_t* bytes, size_t len, int16_t (*pos)[3], uint8_t (*mrgb)[4], size_t cap,
                       size_t* n, uint32_t size_xyz[3]);
int vxrt_build_octree(const int16_t (*pos)[3], const uint8_t (*mrgb)[4], size_t n, int32_t* words, size_t cap,
                      size_t* n_words, uint32_t* depth);
int vxrt_camera_axis_scaled(const float position[3], const float direction[3], float fov, uint32_t width,
                            uint32_t height, float right[3], float up[3], float forward_ray[3]);
/* Stand-in for the blue-noise table the reference does not ship (resources/blue-noise-128.zip,
 * .MISSING_LARGE_BLOBS): value i of seed s is  z = i*0x9E3779B9 + s;  z ^= z>>16; z *= 0x85EBCA6B;
 * z ^= z>>13; z *= 0xC2B2AE35; z ^= z>>16;  (z >> 8) * 2^-24. */
int vxrt_noise_table(uint32_t seed, float* out, size_t n);
/* ---- blue noise.  The reference indexes 512 layers of 128x128 blue noise loaded from
 *      resources/blue-noise-128.zip (src/context.rs:1016-1116), a file its repository does not ship.
 *      vxrt_blue_noise makes such a table on the GPU (void-and-cluster, include/vxrt_bluenoise.h): `layers`
 *      layers first_layer.. of size x size floats into the HOST buffer out (size: power of two, 16..128).
 *      The zip functions read / write the reference's archive format (entries in archive order, each
 *      BE u32 width | BE u32 height | BE f32 pixels; square, all the same size), so a generated table can be
 *      dropped into the reference as its missing resource.  vxrt_set_noise replaces a context's table
 *      (512*128*128 floats, host memory). ---------------------------------------------------------------- */
int vxrt_blue_noise(int32_t device, uint32_t seed, uint32_t size, uint32_t first_layer, uint32_t layers, float* out);
int vxrt_noise_zip_read(const char* path, float* out, size_t cap_floats, uint32_t* size, uint32_t* layers);
int vxrt_noise_zip_write(const char* path, const float* table, uint32_t size, uint32_t layers);

/* ---- wider scene input.  vxrt_vox_scene_to_voxels: like vxrt_vox_to_voxels, plus, per flags:
 *      every shape instance of the file's nTRN/nGRP/nSHP scene graph with its translation and axis rotation
 *      (the reference renders models[0] at its raw coordinates and skips those chunks, src/context.rs:916,
 *      src/vox.rs:61); material types other than _diffuse/_emit taken as diffuse instead of rejected
 *      (src/vox.rs:82-89); the result shifted so that its minimum corner is the origin.  bounds: inclusive voxel
 *      bounding box in renderer axes (optional).  vxrt_default_scene_voxels: Context::create_voxels
 *      (src/context.rs:838-910), the start-up scene, with a seeded generator for its random colours. ------- */
enum { VXRT_VOX_ALL_MODELS = 1, VXRT_VOX_LENIENT_MATERIALS = 2, VXRT_VOX_REBASE = 4 };
int vxrt_vox_scene_to_voxels(const uint8_t* bytes, size_t len, uint32_t flags, int16_t (*pos)[3], uint8_t (*mrgb)[4],
                             size_t cap, size_t* n, int32_t bounds_min[3], int32_t bounds_max[3]);
int vxrt_default_scene_voxels(uint32_t seed, int16_t (*pos)[3], uint8_t (*mrgb)[4], size_t cap, size_t* n);

/* Procedural level-`level` Menger sponge, side 3^level voxels at the origin (config 5 generator;
 * level 4 reproduces the voxel set of vox/menger.vox). */
int vxrt_menger_voxels(uint32_t level, const uint8_t mrgb[4], int16_t (*pos)[3], uint8_t (*out_mrgb)[4], size_t cap,
                       size_t* n);
/* The same sponge clipped to [0, clip)^3 (0 = no clip) with every voxel whose hash(x,y,z) % emissive_period == 0
 * marked emissive (0 = none): the voxel-list form of the scene vxrt_set_menger builds procedurally. */
int vxrt_menger_voxels_ex(uint32_t level, uint32_t clip, const uint8_t mrgb[4], uint32_t emissive_period, int16_t (*pos)[3],
                          uint8_t (*out_mrgb)[4], size_t cap, size_t* n);

#ifdef __cplusplus
}
#endif
#endif /* VXRT_HOST_H */
