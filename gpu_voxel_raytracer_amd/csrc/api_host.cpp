// api_host.cpp — the host-only entry points of libvxrt (no GPU needed): .vox decoding, the reference-layout octree, the device record
// formats, the camera basis, the stand-in noise table, the blue-noise archive format, procedural voxel lists
// (src/vox.rs, src/context.rs:710-834, 913-933, 1042-1116, src/camera.rs).
#include <cmath>

#include "ctx.h"

namespace vxrt {
int32_t procedural_leaf_word(uint32_t x, uint32_t y, uint32_t z, const uint8_t mrgb[4], uint32_t emissive_period);
int flatten_svo(const Octree& tree, std::vector<SvoRecord>* recs, std::vector<int32_t>* leaves);
int widen_svo(const std::vector<SvoRecord>& recs, uint32_t depth, std::vector<WideRec>* out);
}

extern "C" {

// ---- host-only helpers -------------------------------------------------------------------------------
int vxrt_vox_to_voxels(const uint8_t* bytes, size_t len, int16_t (*pos)[3], uint8_t (*mrgb)[4], size_t cap, size_t* n,
                       uint32_t size_xyz[3]) try {
    if (!bytes || !n) { set_error("null argument"); return VXRT_E_INVALID; }
    VoxScene scene;
    if (int rc = decode_vox(bytes, len, &scene)) return rc;
    *n = scene.voxels.size();
    if (size_xyz) memcpy(size_xyz, scene.size, sizeof scene.size);
    for (size_t i = 0; i < scene.voxels.size() && i < cap; i++) {
        const Voxel& v = scene.voxels[i];
        if (pos) { pos[i][0] = v.x; pos[i][1] = v.y; pos[i][2] = v.z; }
        if (mrgb) { mrgb[i][0] = v.m; mrgb[i][1] = v.r; mrgb[i][2] = v.g; mrgb[i][3] = v.b; }
    }
    return VXRT_OK;
} VXRT_CATCH

int vxrt_build_octree(const int16_t (*pos)[3], const uint8_t (*mrgb)[4], size_t n, int32_t* words, size_t cap, size_t* n_words,
                      uint32_t* depth) try {
    if (!n_words || (n != 0 && (!pos || !mrgb))) { set_error("null argument"); return VXRT_E_INVALID; }
    std::vector<Voxel> v(n);
    for (size_t i = 0; i < n; i++) {
        v[i].x = pos[i][0]; v[i].y = pos[i][1]; v[i].z = pos[i][2];
        v[i].m = mrgb[i][0]; v[i].r = mrgb[i][1]; v[i].g = mrgb[i][2]; v[i].b = mrgb[i][3];
    }
    Octree tree;
    if (int rc = build_octree(v.data(), n, &tree)) return rc;
    *n_words = tree.words.size();
    if (depth) *depth = tree.depth;
    if (words && cap >= tree.words.size()) memcpy(words, tree.words.data(), tree.words.size() * sizeof(int32_t));
    return VXRT_OK;
} VXRT_CATCH

// The scene as the kernels read it, for a voxel list (host only): 8-byte records, wide records, leaf words.  Arrays may be null
// (sizes only); nothing is written past the caps.
int vxrt_build_records(const int16_t (*pos)[3], const uint8_t (*mrgb)[4], size_t n, uint32_t* svo, size_t svo_cap, size_t* n_svo,
                       uint32_t* wide, size_t wide_cap, size_t* n_wide, int32_t* leaves, size_t leaf_cap, size_t* n_leaves, uint32_t* depth) try {
    if (!n_svo || !n_wide || !n_leaves || (n != 0 && (!pos || !mrgb))) { set_error("null argument"); return VXRT_E_INVALID; }
    std::vector<Voxel> v(n);
    for (size_t i = 0; i < n; i++) {
        v[i].x = pos[i][0]; v[i].y = pos[i][1]; v[i].z = pos[i][2];
        v[i].m = mrgb[i][0]; v[i].r = mrgb[i][1]; v[i].g = mrgb[i][2]; v[i].b = mrgb[i][3];
    }
    Octree tree;
    if (int rc = build_octree(v.data(), n, &tree)) return rc;
    std::vector<SvoRecord> recs;
    std::vector<int32_t> lw;
    std::vector<WideRec> wr;
    if (int rc = flatten_svo(tree, &recs, &lw)) return rc;
    if (int rc = widen_svo(recs, tree.depth, &wr)) return rc;
    *n_svo = recs.size(); *n_wide = wr.size(); *n_leaves = lw.size();
    if (depth) *depth = tree.depth;
    if (svo && svo_cap >= recs.size()) memcpy(svo, recs.data(), recs.size() * sizeof(SvoRecord));
    if (wide && wide_cap >= wr.size()) memcpy(wide, wr.data(), wr.size() * sizeof(WideRec));
    if (leaves && leaf_cap >= lw.size()) memcpy(leaves, lw.data(), lw.size() * sizeof(int32_t));
    return VXRT_OK;
} VXRT_CATCH

int vxrt_camera_axis_scaled(const float position[3], const float direction[3], float fov, uint32_t width, uint32_t height,
                            float right[3], float up[3], float forward_ray[3]) try {
    (void)position;
    if (!direction || !right || !up || !forward_ray) { set_error("null argument"); return VXRT_E_INVALID; }
    CameraBasis b = camera_axis_scaled(direction, fov, width, height);
    memcpy(right, b.right, sizeof b.right);
    memcpy(up, b.up, sizeof b.up);
    memcpy(forward_ray, b.forward_ray, sizeof b.forward_ray);
    return VXRT_OK;
} VXRT_CATCH

// Multi-GPU: how many rows of the neighbouring ranks' history the exchange after a frame rendered from camera A must carry
// (VXRT_OPT_HALO_ROWS) so that temporal.comp's reprojection (shaders/temporal.comp:75-113) of the NEXT frame, rendered from camera B,
// stays inside what a rank can see: the largest vertical image motion, in rows, of a point at distance >= `near` along any pixel's
// ray of B, + 2 (the bilinear footprint's second row and rounding), capped at band_rows.  Along a ray the reprojected row is a
// linear-fractional function of 1 / distance, so both ends of [near, inf) are evaluated, on a 33 x 33 grid of pixels that holds the
// frame's borders.  Every rank computes the same number from the same cameras (the message sizes of an exchange must agree).
// The same arithmetic as distributed.halo_rows_for_motion (tests/test_host_logic.py compares them).
int vxrt_halo_rows_for_motion(const float pos_a[3], const float dir_a[3], const float pos_b[3], const float dir_b[3], float fov, uint32_t width,
                              uint32_t height, float near_distance, uint32_t band_rows, uint32_t* rows) try {
    if (!pos_a || !dir_a || !pos_b || !dir_b || !rows || width == 0 || height == 0) { set_error("null argument"); return VXRT_E_INVALID; }
    *rows = band_rows;
    const CameraBasis a = camera_axis_scaled(dir_a, fov, width, height), b = camera_axis_scaled(dir_b, fov, width, height);
    // inverse of M = [right_a | up_a | forward_a] (columns), by cofactors in binary64
    const double m[3][3] = {{a.right[0], a.up[0], a.forward_ray[0]}, {a.right[1], a.up[1], a.forward_ray[1]}, {a.right[2], a.up[2], a.forward_ray[2]}};
    const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                       m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return VXRT_OK;       // no bound: every row travels
    double inv[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            inv[j][i] = (m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1]) / det;
        }
    double worst = 0.0;
    for (int gy = 0; gy <= 32; gy++)
        for (int gx = 0; gx <= 32; gx++) {
            const double x = double(width - 1) * gx / 32.0, y = double(height - 1) * gy / 32.0;
            double d[3], len = 0.0;
            for (int k = 0; k < 3; k++) { d[k] = x * b.right[k] - y * b.up[k] + b.forward_ray[k]; len += d[k] * d[k]; }
            len = std::sqrt(len);
            for (const double dist : {double(near_distance), 1e9}) {
                double p[3], s[3];
                for (int k = 0; k < 3; k++) p[k] = double(pos_b[k]) + dist * d[k] / len - double(pos_a[k]);
                for (int r = 0; r < 3; r++) s[r] = inv[r][0] * p[0] + inv[r][1] * p[1] + inv[r][2] * p[2];
                if (!(s[2] > 1e-12)) return VXRT_OK;                              // a point behind the old camera: no bound
                const double moved = std::fabs(-(s[1] / s[2]) - y);
                worst = moved > worst ? moved : worst;
            }
        }
    if (!std::isfinite(worst)) return VXRT_OK;
    const double want = std::ceil(worst - 1e-6) + 2.0;
    *rows = want < double(band_rows) ? uint32_t(want) : band_rows;
    return VXRT_OK;
} VXRT_CATCH

int vxrt_noise_table(uint32_t seed, float* out, size_t n) try {
    if (!out) { set_error("null argument"); return VXRT_E_INVALID; }
    for (size_t i = 0; i < n; i++) out[i] = noise_value(seed, uint32_t(i));
    return VXRT_OK;
} VXRT_CATCH

int vxrt_menger_voxels_ex(uint32_t level, uint32_t clip, const uint8_t mrgb[4], uint32_t emissive_period, int16_t (*pos)[3],
                          uint8_t (*out_mrgb)[4], size_t cap, size_t* n) try {
    if (!n || !mrgb || level > 9) { set_error("bad argument"); return VXRT_E_INVALID; }
    uint32_t side = 1;
    for (uint32_t l = 0; l < level; l++) side *= 3;
    if (clip != 0 && clip < side) side = clip;
    if (side > 32767) { set_error("menger side exceeds i16"); return VXRT_E_INVALID; }
    size_t count = 0;
    for (uint32_t x = 0; x < side; x++)
        for (uint32_t y = 0; y < side; y++)
            for (uint32_t z = 0; z < side; z++)
                if (menger_solid(level, x, y, z)) {
                    if (count < cap) {
                        if (pos) { pos[count][0] = int16_t(x); pos[count][1] = int16_t(y); pos[count][2] = int16_t(z); }
                        if (out_mrgb) {
                            const int32_t w = procedural_leaf_word(x, y, z, mrgb, emissive_period);
                            out_mrgb[count][0] = uint8_t((uint32_t(w) >> 24) & 0x7fu);
                            out_mrgb[count][1] = mrgb[1]; out_mrgb[count][2] = mrgb[2]; out_mrgb[count][3] = mrgb[3];
                        }
                    }
                    count++;
                }
    *n = count;
    return VXRT_OK;
} VXRT_CATCH

int vxrt_menger_voxels(uint32_t level, const uint8_t mrgb[4], int16_t (*pos)[3], uint8_t (*out_mrgb)[4], size_t cap, size_t* n) try {
    return vxrt_menger_voxels_ex(level, 0, mrgb, 0, pos, out_mrgb, cap, n);
} VXRT_CATCH

// ---- wider scene input (csrc/vox_scene.cpp) ------------------------------------------------------------------------
int vxrt_vox_scene_to_voxels(const uint8_t* bytes, size_t len, uint32_t flags, int16_t (*pos)[3], uint8_t (*mrgb)[4], size_t cap,
                             size_t* n, int32_t bounds_min[3], int32_t bounds_max[3]) try {
    if (!bytes || !n) { set_error("null argument"); return VXRT_E_INVALID; }
    if (flags & ~uint32_t(VXRT_VOX_ALL_MODELS | VXRT_VOX_LENIENT_MATERIALS | VXRT_VOX_REBASE)) { set_error("unknown flag"); return VXRT_E_INVALID; }
    VoxScene scene;
    int32_t lo[3], hi[3];
    if (int rc = decode_vox_scene(bytes, len, flags, &scene, lo, hi)) return rc;
    *n = scene.voxels.size();
    for (int a = 0; a < 3; a++) {
        if (bounds_min) bounds_min[a] = lo[a];
        if (bounds_max) bounds_max[a] = hi[a];
    }
    for (size_t i = 0; i < scene.voxels.size() && i < cap; i++) {
        const Voxel& v = scene.voxels[i];
        if (pos) { pos[i][0] = v.x; pos[i][1] = v.y; pos[i][2] = v.z; }
        if (mrgb) { mrgb[i][0] = v.m; mrgb[i][1] = v.r; mrgb[i][2] = v.g; mrgb[i][3] = v.b; }
    }
    return VXRT_OK;
} VXRT_CATCH

int vxrt_default_scene_voxels(uint32_t seed, int16_t (*pos)[3], uint8_t (*mrgb)[4], size_t cap, size_t* n) try {
    if (!n) { set_error("null argument"); return VXRT_E_INVALID; }
    std::vector<Voxel> voxels;
    default_scene(seed, &voxels);
    *n = voxels.size();
    for (size_t i = 0; i < voxels.size() && i < cap; i++) {
        const Voxel& v = voxels[i];
        if (pos) { pos[i][0] = v.x; pos[i][1] = v.y; pos[i][2] = v.z; }
        if (mrgb) { mrgb[i][0] = v.m; mrgb[i][1] = v.r; mrgb[i][2] = v.g; mrgb[i][3] = v.b; }
    }
    return VXRT_OK;
} VXRT_CATCH

}  // extern "C"
