// trace_dda.hip — PROTOTYPE (-DVXRT_VARIANTS=1 only; round 5): cast_bounded_ray's RESULT by a two-level DDA over a dense bit grid, with
// a certificate that says when the result is the octree walk's, bit for bit.
//
// Why it can be exact.  Every time the walk of voxels.comp:134-247 computes is the crossing time of a grid plane, fl(fl(p - o) * inv)
// — node centres and cube faces are dyadic and exact, so the value depends on the plane only, not on the node whose slab test or
// mid-plane test made it (trace_common.h: "all cube planes are dyadic").  Which plane is crossed next, whether a node is left before its
// mid plane is reached, which sibling follows: all of that is decided by COMPARING such times.  A DDA that steps from plane to plane
// by the same computed times makes the same comparisons on the same numbers and visits the same cells.  The one thing the walk decides
// differently is the octant it lands in when it DESCENDS into a child node: current_octant (voxels.comp:119-125) compares the POSITION
// o + d * time with the node's centre, and a position has roundings of its own.  Position and plane times agree unless the point
// lies within a few ulps of a grid plane; so: whenever the DDA enters a cell of a non-empty brick it computes that position, takes the
// other two axes' cells from it (as the walk does), and checks that it keeps a margin from every unit plane.  A ray that ever fails
// the check — or has a zero / non-finite direction component, or takes more steps than the walk's 2048-trip cap could allow — is
// FLAGGED: its result is not used, the exact walk decides it.  The hit time is the time of the plane through which the hit cell was
// entered (or max(0, root entry) for a ray that starts in a solid cell), the normal comes from hit_normal's own formula, the leaf
// word from the cell: what finish_ray produces.
//
// This file measures two things through vxrt_debug_dda_rays: that unflagged rays equal cast_ray bit for bit (tests), how many rays
// are flagged, and what a lock-step wave of DDA rays costs beside a wave of octree walks (the kernels are timed with HIP events).
#define VXRT_STACK_STRIDE 64
#include "trace_common.h"

namespace vxrt {
namespace {

// Round 6: the grid is built ON THE DEVICE from the scene in place (dda_build_kernel below: one thread per brick descends the 8-byte
// records along the brick's coordinates), so that the prototype also runs on BASELINE config 5's 2048^3 scene (512^3 bricks, 8.6 GB of
// address space of which the sponge's octant, 1 GiB, is populated).  Three levels now: one bit per 64^3-cell SUPER-BRICK (32 KB for
// config 5: optionally staged in LDS, north_star's "LDS-staged"), one bit per 8^3 brick, 64 bytes per brick.  No dense leaf array any
// more: the leaves of a brick's subtree are one run of the tree's breadth-first leaf array, in exactly the order of the brick's 512
// mask bits, so a hit's leaf word is leaves[first_leaf[brick] + popcount(mask bits below the cell's)].
struct DdaGrid {
    const unsigned long long* bricks;   // per 8^3 brick 8 words: word = the brick's 4^3 octant (x>>2, y>>2, z>>2), byte = the 2^3 node inside it, bit = the cell
    const uint32_t* brick_bits;         // one bit per brick: holds a voxel
    const uint32_t* super_bits;         // one bit per super-brick of 8^3 bricks (64^3 cells); null: levels < 6
    const uint32_t* first_leaf;         // per brick: index of its first leaf word in `leaves`
    const int32_t* leaves;              // the scene's leaf words (breadth-first order)
    int levels;                         // cells (leaf octants) per axis = 1 << levels = 2 << depth
};

// One thread per brick: descend the SVO from the root along the brick's coordinates (levels - 3 records), then gather the leaf masks
// of the up to 64 leaf parents below the brick's node into its 8 words.  slot = 4 x + 2 y + z (src/context.rs:726-729) at every level,
// which is also the bit order of cell_bit_index.
__global__ __launch_bounds__(256) void dda_build_kernel(const SvoRecord* svo, SvoRecord root, int levels, unsigned long long* bricks, uint32_t* brick_bits,
                                                         uint32_t* super_bits, uint32_t* first_leaf) {
    const unsigned nb = 1u << (levels - 3);
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= size_t(nb) * nb * nb) return;
    const unsigned bz = unsigned(i % nb), by = unsigned((i / nb) % nb), bx = unsigned(i / (size_t(nb) * nb));
    SvoRecord rec = root;
    for (int l = 0; l < levels - 3; l++) {            // node level l: its slot from bit (levels - 4 - l) of the brick coordinates
        const int sh = levels - 4 - l;
        const unsigned slot = ((bx >> sh) & 1u) << 2 | ((by >> sh) & 1u) << 1 | ((bz >> sh) & 1u);
        const unsigned bit = 1u << slot;
        if ((rec.masks & bit) == 0u) return;          // empty (the arrays were cleared)
        rec = svo[rec.base + __popc(rec.masks & (bit - 1u))];
    }
    // rec: the node of 8^3 cells; its children: 4^3 cells; theirs: the leaf parents (2^3 cells, masks >> 8 = leaf mask, base = first leaf)
    uint32_t first = 0xffffffffu;
    bool any = false;
    for (unsigned n4 = 0; n4 < 8; n4++) {
        unsigned long long word = 0ull;
        if (rec.masks & (1u << n4)) {
            const SvoRecord r4 = svo[rec.base + __popc(rec.masks & ((1u << n4) - 1u))];
            for (unsigned n2 = 0; n2 < 8; n2++)
                if (r4.masks & (1u << n2)) {
                    const SvoRecord r2 = svo[r4.base + __popc(r4.masks & ((1u << n2) - 1u))];
                    word |= (unsigned long long)((r2.masks >> 8) & 0xffu) << (8u * n2);
                    if (first == 0xffffffffu) first = r2.base;
                }
        }
        bricks[i * 8u + n4] = word;
        any |= word != 0ull;
    }
    if (any) {
        first_leaf[i] = first;
        atomicOr(brick_bits + (i >> 5), 1u << (i & 31u));
        if (super_bits) {
            const unsigned ns = nb >> 3;
            const size_t si = (size_t(bx >> 3) * ns + (by >> 3)) * ns + (bz >> 3);
            atomicOr(super_bits + (si >> 5), 1u << (si & 31u));
        }
    }
}

__device__ __forceinline__ unsigned cell_bit_index(int x, int y, int z) {   // within a brick: 0..511
    const unsigned n4 = unsigned((x >> 2) & 1) << 2 | unsigned((y >> 2) & 1) << 1 | unsigned((z >> 2) & 1);
    const unsigned n2 = unsigned((x >> 1) & 1) << 2 | unsigned((y >> 1) & 1) << 1 | unsigned((z >> 1) & 1);
    const unsigned c = unsigned(x & 1) << 2 | unsigned(y & 1) << 1 | unsigned(z & 1);
    return n4 << 6 | n2 << 3 | c;
}

// the cell (per axis) that holds coordinate p: cell j spans (lo_j, lo_j + cell] — the strict > of current_octant, decided with exact
// compares against the (dyadic) planes; clamped to the grid
__device__ __forceinline__ int cell_of(float p, float rmin, float cell, float inv_cell, int top) {
    int c = int(__builtin_floorf((p - rmin) * inv_cell));
    c = c < 0 ? 0 : (c > top ? top : c);
    const float lo = rmin + float(c) * cell;
    c += !(lo < p) ? -1 : (p > lo + cell ? 1 : 0);
    return c < 0 ? 0 : (c > top ? top : c);
}

// distance of p from the nearest unit plane, and the margin the certificate asks for at a point reached after time t along d
__device__ __forceinline__ bool near_plane(float p, float dt, float rmin, float cell, float inv_cell, float margin_scale) {
    const float u = (p - rmin) * inv_cell;
    const float f = u - __builtin_floorf(u);
    const float dist = (f < 0.5f ? f : 1.0f - f) * cell;
    // roundings: the position fl(o + fl(d t)) (half an ulp of each), the plane times it is compared with (two roundings each, seen
    // through d), the time itself (two roundings, seen through d): <= 2^-24 (5 |d t| + |p|), times a safety factor
    const float eps = margin_scale * 5.9604645e-8f * (5.0f * vx_abs(dt) + vx_abs(p) + vx_abs(rmin));
    return !(dist > eps);
}

// out: hit flag, time, bits(leaf word), normal xyz, flags (1 = flagged: the exact walk must decide), steps
template <int kCertify, int kLdsTop>
__global__ __launch_bounds__(256) void dda_probe_kernel(const TraceArgs a, const DdaGrid g, const float* origins, const float* dirs, float* out, unsigned n,
                                                         float margin_scale, unsigned max_steps) {
    // the top level in LDS (kLdsTop: the super-brick bits are <= 32 KB, i.e. levels <= 12): every block copies them once
    __shared__ uint32_t top_lds[kLdsTop ? 8192 : 1];
    if (kLdsTop) {
        const unsigned ns = 1u << (g.levels - 6);
        const unsigned words = (ns * ns * ns + 31u) / 32u;
        for (unsigned k = threadIdx.x; k < words; k += 256u) top_lds[k] = g.super_bits[k];
        __syncthreads();
    }
    const uint32_t* sup = kLdsTop ? top_lds : g.super_bits;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const SceneView sc = make_scene(a);
    const f3 o = ld3(origins + 3 * i), d = ld3(dirs + 3 * i);
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    float* res = out + 8 * size_t(i);
    res[0] = 0.0f; res[1] = 0.0f; res[2] = 0.0f; res[3] = res[4] = res[5] = 0.0f; res[6] = 0.0f; res[7] = 0.0f;
    if (!ray_is_regular(inv)) { res[6] = 1.0f; return; }
    f3 en, ex;
    plane_times(o, inv, sc.root_center, 0.5f * sc.root_size, en, ex);
    const float entry = vx_max3(en.x, en.y, en.z), exit = vx_min3(ex.x, ex.y, ex.z);
    if (!(exit >= 0.0f && entry < exit)) return;                                       // misses the root cube (voxels.comp:154-156)
    float time = vx_max(0.0f, entry);
    const int top = (1 << g.levels) - 1;
    const float cell = __builtin_ldexpf(sc.root_size, -g.levels), inv_cell = 1.0f / cell;   // a leaf octant: half the finest node (sc.cell)
    const float rm[3] = {sc.root_min.x, sc.root_min.y, sc.root_min.z};
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, ii[3] = {inv.x, inv.y, inv.z};
    int j[3];
    bool flagged = false;
    {
        const f3 p = o + d * time;
        const float pp[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            j[k] = cell_of(pp[k], rm[k], cell, inv_cell, top);
            // (at time 0 the position IS the origin and the walk's compares are exact: nothing to certify)
            if (kCertify && time != 0.0f) flagged |= near_plane(pp[k], dd[k] * time, rm[k], cell, inv_cell, margin_scale);
        }
    }
    unsigned steps = 0;
    int entered_axis = -1;
    for (;;) {
        // what is here?
        const int bx = j[0] >> 3, by = j[1] >> 3, bz = j[2] >> 3;
        const unsigned nb = 1u << (g.levels - 3);
        const unsigned blin = (unsigned(bx) * nb + unsigned(by)) * nb + unsigned(bz);
        int stride = 8;
        bool brick_full;
        if (sup != nullptr) {                                                              // the super-brick first: 64 cells in one step when it is empty
            const unsigned ns = nb >> 3;
            const unsigned slin = (unsigned(bx >> 3) * ns + unsigned(by >> 3)) * ns + unsigned(bz >> 3);
            const bool super_full = (sup[slin >> 5] >> (slin & 31u)) & 1u;
            brick_full = super_full && ((g.brick_bits[blin >> 5] >> (blin & 31u)) & 1u);
            if (!super_full) stride = 64;
        } else {
            brick_full = (g.brick_bits[blin >> 5] >> (blin & 31u)) & 1u;
        }
        if (brick_full) {
            const unsigned bit = cell_bit_index(j[0], j[1], j[2]);
            const unsigned long long word = g.bricks[size_t(blin) * 8u + (bit >> 6)];
            if ((word >> (bit & 63u)) & 1ull) {                                         // a voxel: the hit
                unsigned rank = unsigned(__popcll(word & ((1ull << (bit & 63u)) - 1ull)));  // its leaf word: the brick's leaves are one run, in mask-bit order
                for (unsigned wq = 0; wq < (bit >> 6); wq++) rank += unsigned(__popcll(g.bricks[size_t(blin) * 8u + wq]));
                const size_t lin = size_t(g.first_leaf[blin]) + rank;
                const f3 oc = mk3(rm[0] + (float(j[0]) + 0.5f) * cell, rm[1] + (float(j[1]) + 0.5f) * cell, rm[2] + (float(j[2]) + 0.5f) * cell);
                const f3 nrm = hit_normal(o, d, time, oc);
                res[0] = 1.0f; res[1] = time; res[2] = __int_as_float(g.leaves[lin]); res[3] = nrm.x; res[4] = nrm.y; res[5] = nrm.z;
                break;
            }
            const unsigned byte = unsigned(word >> (bit & 56u)) & 0xffu;
            stride = word == 0ull ? 4 : (byte == 0u ? 2 : 1);                           // an empty 4^3 / 2^3 node is crossed in one step, as the walk crosses it
        }
        // the next plane of the current stride per axis, its crossing time by the walk's formula
        float tmax[3];
        int plane[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int base = j[k] & ~(stride - 1);
            plane[k] = dd[k] > 0.0f ? base + stride : base;
            tmax[k] = ((rm[k] + float(plane[k]) * cell) - oo[k]) * ii[k];
        }
        // the first crossing: x before y before z on equal times (voxels.comp:196-200); an equal time is a flagged case anyway
        int ax = tmax[0] <= tmax[1] ? (tmax[0] <= tmax[2] ? 0 : 2) : (tmax[1] <= tmax[2] ? 1 : 2);
        time = tmax[ax];
        const int nj = dd[ax] > 0.0f ? plane[ax] : plane[ax] - 1;
        if (nj < 0 || nj > top) break;                                                   // left the root cube: a miss (voxels.comp:226)
        j[ax] = nj;
        entered_axis = ax;
        // the other axes from the position, as a descend takes them (current_octant), and the certificate
        const f3 p = o + d * time;
        const float pp[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (k == ax) continue;
            if (stride > 1) j[k] = cell_of(pp[k], rm[k], cell, inv_cell, top);
            if (kCertify) flagged |= near_plane(pp[k], dd[k] * time, rm[k], cell, inv_cell, margin_scale);
        }
        if (++steps > max_steps) { flagged = true; break; }                                // no result: the exact walk decides (its own cap is 2048 trips)
    }
    (void)entered_axis;
    res[6] = flagged ? 1.0f : 0.0f;
    res[7] = float(steps);
}

}  // namespace

// grid arrays for a tree of `levels` cell levels: bricks 64 B per brick, one bit per brick, one bit per super-brick, 4 B per brick
void dda_grid_sizes(int levels, size_t* brick_bytes, size_t* brick_bit_bytes, size_t* super_bit_bytes, size_t* first_leaf_bytes) {
    const size_t nb = size_t(1) << (levels - 3), bricks = nb * nb * nb;
    *brick_bytes = bricks * 64;
    *brick_bit_bytes = (bricks + 31) / 32 * 4;
    const size_t ns = levels >= 6 ? size_t(1) << (levels - 6) : 0, supers = ns * ns * ns;
    *super_bit_bytes = supers ? (supers + 31) / 32 * 4 : 0;
    *first_leaf_bytes = bricks * 4;
}

// fills the (zeroed) grid arrays from the 8-byte records in a.svo
hipError_t launch_dda_build(const TraceArgs& a, int levels, void* bricks, void* brick_bits, void* super_bits, void* first_leaf, hipStream_t s) {
    const size_t nb = size_t(1) << (levels - 3), n = nb * nb * nb;
    hipLaunchKernelGGL(dda_build_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, a.svo, a.root_rec, levels, static_cast<unsigned long long*>(bricks),
                       static_cast<uint32_t*>(brick_bits), static_cast<uint32_t*>(super_bits), static_cast<uint32_t*>(first_leaf));
    return hipGetLastError();
}

hipError_t launch_dda_probe(const TraceArgs& a, const void* bricks, const void* brick_bits, const void* super_bits, const void* first_leaf, int levels,
                            const float* origins, const float* dirs, float* out, unsigned n, int certify, float margin_scale, unsigned max_steps, int lds_top,
                            hipStream_t s) {
    DdaGrid g{static_cast<const unsigned long long*>(bricks), static_cast<const uint32_t*>(brick_bits), static_cast<const uint32_t*>(super_bits),
              static_cast<const uint32_t*>(first_leaf), a.leaves, levels};
    const dim3 grid((n + 255u) / 256u), block(256);
    const bool lds = lds_top && super_bits != nullptr && levels <= 12;
    if (certify) {
        if (lds) hipLaunchKernelGGL((dda_probe_kernel<1, 1>), grid, block, 0, s, a, g, origins, dirs, out, n, margin_scale, max_steps);
        else hipLaunchKernelGGL((dda_probe_kernel<1, 0>), grid, block, 0, s, a, g, origins, dirs, out, n, margin_scale, max_steps);
    } else {
        if (lds) hipLaunchKernelGGL((dda_probe_kernel<0, 1>), grid, block, 0, s, a, g, origins, dirs, out, n, margin_scale, max_steps);
        else hipLaunchKernelGGL((dda_probe_kernel<0, 0>), grid, block, 0, s, a, g, origins, dirs, out, n, margin_scale, max_steps);
    }
    return hipGetLastError();
}

}  // namespace vxrt
