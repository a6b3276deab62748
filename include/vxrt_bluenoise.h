/*
 * vxrt_bluenoise.h — definition of the blue-noise table libvxrt generates.
 *
 * The reference indexes 512 layers of 128x128 blue noise (shaders/voxels.comp:65-71, 268-275) loaded from
 * resources/blue-noise-128.zip (src/context.rs:1016-1116), a file its repository does not ship.  There is
 * therefore no reference data or generator to match; what is pinned here is the algorithm this build uses to make
 * a table of the same shape and kind: void-and-cluster (Ulichney 1993), one independent layer per (seed, layer).
 * Both implementations — the HIP kernel (gpu_voxel_raytracer_amd/csrc/noise.hip) and the CPU checker
 * (oracle/onoise.cpp) — follow this text and give bit-identical tables.
 *
 * A layer is an N x N torus (N a power of two, 16..128: the window must not overlap itself), cell c = x + N*y.
 *   energy      E[c] = sum over the (2R+1)^2 window, rows dy = -R..R outer, dx = -R..R inner, of K[dy][dx] for
 *               every window cell that belongs to the minority set, accumulated in that order in binary32
 *               ("gather").  Afterwards inserting / removing a point p adds / subtracts K[dy][dx] at every
 *               window cell of p ("splat"), one binary32 operation per cell.
 *   K[dy][dx]   = vx_exp_unfused(-(dx*dx + dy*dy) / (2 * 1.9 * 1.9)) from include/vxrt_detmath.h, R = 7.
 *   tightest cluster = the minority cell of maximal E; largest void = the majority cell of minimal E; ties go to
 *               the lowest cell index.
 *   1. initial pattern: n0 = N*N/10 ones at cells vxbn_hash(seed, layer, i) % (N*N), i = 0, 1, ... (cells already
 *      set are skipped); gather E over the ones.
 *   2. relax: remove the tightest cluster c1, insert into the largest void c0, until c0 == c1 (or 4*n0 rounds).
 *   3. ranks n0-1 .. 0: remove the tightest cluster of a copy of the relaxed pattern, one by one.
 *   4. ranks n0 .. N*N/2-1: from the relaxed pattern (E gathered afresh) insert into the largest void.
 *   5. ranks N*N/2 .. N*N-1: the zeros are now the minority: gather E over the zeros; repeatedly turn the
 *      tightest cluster of zeros into a one.
 *   value[c] = (rank[c] + 0.5) / (N*N)  — exact in binary32, in (0, 1), every value used once per layer.
 */
#ifndef VXRT_BLUENOISE_H
#define VXRT_BLUENOISE_H

#include <stdint.h>

#include "vxrt_detmath.h"

#define VXBN_RADIUS 7
#define VXBN_TAPS (2 * VXBN_RADIUS + 1)
#define VXBN_MAX_SIZE 128

/* counter-based hash of (seed, layer, i): three rounds of the murmur3 finaliser */
VX_HD uint32_t vxbn_hash(uint32_t seed, uint32_t layer, uint32_t i) {
    uint32_t z = seed ^ (layer * 0x9E3779B9u) ^ (i * 0x85EBCA6Bu + 0x165667B1u);
    for (int r = 0; r < 3; r++) {
        z ^= z >> 16; z *= 0x85EBCA6Bu;
        z ^= z >> 13; z *= 0xC2B2AE35u;
        z ^= z >> 16; z += layer + 0x27D4EB2Fu * (uint32_t)(r + 1);
    }
    return z;
}

/* K[dy + R][dx + R] */
VX_HD float vxbn_kernel(int dx, int dy) {
    return vx_exp_unfused(-(float)(dx * dx + dy * dy) / (2.0f * 1.9f * 1.9f));
}

#endif /* VXRT_BLUENOISE_H */
