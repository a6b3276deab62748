// trace_common.h — device-side building blocks shared by the tracer's kernels (trace.hip, trace_wavefront.hip):
// the octree walk, the shading helpers, the blue-noise RNG, ray counting and the 64-byte path record.
// Everything is in an anonymous namespace: each translation unit gets its own copies, all force-inlined.
#pragma once
#include "kernels.h"
#include "vx_vec.h"

namespace vxrt {
namespace {


constexpr float kAlmostInfinity = 1073741824.0f;  // float(1 << 30)  voxels.comp:8
constexpr int32_t kLeafBit = int32_t(0x80000000u);
constexpr int32_t kEmitBit = 1 << 30;
constexpr int kBlock = 256;
#ifndef VXRT_STACK_STRIDE
#define VXRT_STACK_STRIDE 256   // threads per block of the including kernel file = columns of the LDS stack
#endif
constexpr int kStackStride = VXRT_STACK_STRIDE;
constexpr uint32_t kNoiseLayer = 128u * 128u;
constexpr uint32_t kNoiseTotal = kNoiseLayer * 512u;

struct RayHit {
    float time;
    int32_t node;
    f3 normal;
};

// ray_cube_intersection, voxels.comp:73-90
__device__ __forceinline__ bool slab(f3 o, f3 inv, f3 sg, f3 c, float half, float& entry, float& exit) {
    f3 hs = half * sg;
    f3 en = ((c - hs) - o) * inv;
    f3 ex = ((c + hs) - o) * inv;
    entry = vx_max(vx_max(en.x, en.y), en.z);
    exit = vx_min(vx_min(ex.x, ex.y), ex.z);
    return exit >= 0.0f && entry < exit;
}

// current_octant, voxels.comp:119-125 (strict >: ties go to the low side)
__device__ __forceinline__ uint32_t octant_of(f3 p, f3 c) {
    return ((p.x - c.x) > 0.0f ? 4u : 0u) + ((p.y - c.y) > 0.0f ? 2u : 0u) + ((p.z - c.z) > 0.0f ? 1u : 0u);
}

struct SceneView {
    const SvoRecord* svo;
    const int32_t* leaves;
    SvoRecord root_rec;
    f3 root_center;
    f3 root_min;
    float root_size;
    float cell, inv_cell;  // edge of the finest cell a node's octant can be (root_size * 2^-levels), and its reciprocal
    int levels;            // node levels 0 .. levels-1
    uint32_t pop_mask;     // ~1u when the root holds ONE occupied slot (walkf_step: a pop back to the root is then a certain miss), else ~0u
#if VXRT_VARIANTS
    uint32_t* touch_nodes;   // TraceArgs::touch_nodes / touch_leaves (null: off)
    uint32_t* touch_leaves;
#endif
};

// Touch map (variants build only; TraceArgs::touch_nodes): mark the 64-byte line `line` of a scene array as read.  The test before the
// atomic keeps the map's own traffic down (a stale read only costs a redundant atomic).  Compiles to nothing in the product.
#if VXRT_VARIANTS
__device__ __forceinline__ void touch_line(uint32_t* map, uint32_t line) {
    if (map == nullptr) return;
    const uint32_t bit = 1u << (line & 31u);
    if ((__builtin_nontemporal_load(map + (line >> 5)) & bit) == 0u) atomicOr(map + (line >> 5), bit);
}
#define VX_TOUCH(map, line) touch_line(map, line)
#else
#define VX_TOUCH(map, line) ((void)0)
#endif

// cast_bounded_ray, voxels.comp:134-247.  `stack` points at this thread's column of the LDS stack
// (entry l at stack[l * kStackStride]).  On the iteration cap the shader returns true without writing the
// normal; it is defined as 0 here (oracle U1).
//
// Shape of the loop (what differs from the shader's text, none of it changes a result):
//  * descend (voxels.comp:205-221) and pop (:225-243) share one code path for everything they have in
//    common — new integer path coordinates, node size and centre, the slab test — so a wave whose lanes
//    are split between the two executes that code once, not twice;
//  * a saved frame is {masks | next_octant << 16, base}: the sibling to resume with travels with the
//    node record in LDS, and only frames that can still advance are stored (the shader's node == -1
//    "complete" frames are never read back: a pop goes straight to the highest level whose bit is set
//    in has_next_mask);
//  * leaving the loop (leaf, miss, iteration cap) only sets a status; the leaf word load and the
//    normal computation happen once after the loop for all lanes of the wave together, instead of
//    inside the loop each time a single lane hits.
struct Walk {  // the loop-carried state of cast_bounded_ray
    f3 o, d, inv, sg, center;
    float time, exit, size;
    uint32_t ix, iy, iz;      // integer path coordinates of the current node, `lvl` bits each
    uint32_t lvl;
    uint32_t has_next_mask;   // bit l: level l can still advance to a sibling (frame.node != -1)
    uint32_t octant, dir_mask;
    SvoRecord rec;
    int iterations;
};
enum : int { kWalkOn = 0, kWalkLeaf = 1, kWalkMiss = 2, kWalkCap = 3 };

// voxels.comp:138-160.  false: the ray misses the root cube.
__device__ __forceinline__ bool walk_begin(Walk& w, const SceneView& sc, f3 o, f3 d) {
    w.o = o;
    w.d = d;
    w.dir_mask = (d.x < 0.0f ? 4u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 1u : 0u);
    w.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    w.sg = mk3(vx_sign(w.inv.x), vx_sign(w.inv.y), vx_sign(w.inv.z));
    float entry;
    if (!slab(o, w.inv, w.sg, sc.root_center, 0.5f * sc.root_size, entry, w.exit)) return false;
    w.time = vx_max(0.0f, entry);
    w.size = sc.root_size;
    w.center = sc.root_center;
    w.ix = w.iy = w.iz = w.lvl = w.has_next_mask = 0;
    w.rec = sc.root_rec;
    w.octant = octant_of(o + d * w.time, w.center);
    w.iterations = 0;
    return true;
}

// One trip of the while(true) loop, voxels.comp:163-246.
__device__ __forceinline__ int walk_step(Walk& w, const SceneView& sc, float max_distance, uint2* stack) {
    if (++w.iterations >= 2048) return kWalkCap;             // voxels.comp:166-169
    if (w.time > max_distance) return kWalkMiss;             // voxels.comp:171-173
    const uint32_t bit = 1u << w.octant;
    if (w.rec.masks & (bit << 8)) return kWalkLeaf;          // value < 0

    // next sibling through the node's mid planes                     voxels.comp:191-203
    const f3 t_mid = (w.center - w.o) * w.inv;
    const uint32_t directional = w.octant ^ w.dir_mask;
    const float mx = (directional & 4u) ? kAlmostInfinity : t_mid.x;
    const float my = (directional & 2u) ? kAlmostInfinity : t_mid.y;
    const float mz = (directional & 1u) ? kAlmostInfinity : t_mid.z;
    const float next_time = vx_min(vx_min(mx, my), mz);
    const uint32_t transition = (mx == next_time) ? 4u : ((my == next_time) ? 2u : ((mz == next_time) ? 1u : 0u));
    const uint32_t next_octant = w.octant ^ transition;
    const bool has_next = next_time <= w.exit && transition != 0u && (directional & transition) == 0u;
    const bool is_child = (w.rec.masks & bit) != 0u;         // value > 0

    if (is_child || !has_next) {
        uint2 raw;
        if (is_child) {  // descend: remember where to resume, fetch the child record   voxels.comp:205-214
            if (has_next) {
                stack[w.lvl * kStackStride] = make_uint2(w.rec.masks | next_octant << 16, w.rec.base);
                w.has_next_mask |= 1u << w.lvl;
            }
            raw = *reinterpret_cast<const uint2*>(sc.svo + (w.rec.base + __popc(w.rec.masks & (bit - 1u))));
            VX_TOUCH(sc.touch_nodes, (w.rec.base + __popc(w.rec.masks & (bit - 1u))) >> 3);
            w.ix = (w.ix << 1) | ((w.octant >> 2) & 1u);
            w.iy = (w.iy << 1) | ((w.octant >> 1) & 1u);
            w.iz = (w.iz << 1) | (w.octant & 1u);
            w.lvl++;
        } else {  // pop to the nearest level that can still advance                     voxels.comp:225-234
            if (w.has_next_mask == 0u) return kWalkMiss;
            const uint32_t l = 31u - uint32_t(__clz(int(w.has_next_mask)));
            w.has_next_mask &= ~(1u << l);
            const uint32_t up = w.lvl - l;
            w.ix >>= up; w.iy >>= up; w.iz >>= up;
            w.lvl = l;
            raw = stack[l * kStackStride];
            // consume the LDS read here: left alone, the compiler merges it with the descend branch's global
            // load into one flat_load (either address space), which is slower and waits on both counters
            asm volatile("" : "+v"(raw.x), "+v"(raw.y));
        }
        w.size = __builtin_ldexpf(sc.root_size, -int(w.lvl));
        w.center = sc.root_min + mk3(float(w.ix) + 0.5f, float(w.iy) + 0.5f, float(w.iz) + 0.5f) * w.size;
        float node_entry, node_exit;
        slab(w.o, w.inv, w.sg, w.center, 0.5f * w.size, node_entry, node_exit);
        if (is_child) {  // voxels.comp:216-221
            w.octant = octant_of(w.o + w.d * w.time, w.center);
            w.time = vx_max(w.time, node_entry);
        } else {         // voxels.comp:236-242
            w.time = w.exit;
            w.octant = (raw.x >> 16) & 7u;
        }
        w.exit = node_exit;
        w.rec.masks = raw.x & 0xffffu;
        w.rec.base = raw.y;
    } else {  // empty slot, step to the sibling                                         voxels.comp:222-224
        w.octant = next_octant;
        w.time = next_time;
    }
    return kWalkOn;
}

// ---- the walk for regular rays ---------------------------------------------------------------------------------
// A ray is "regular" when every component of 1/dir is finite and non-zero (all but ~1e-7 of the rays: a direction
// component that is exactly 0 needs an exactly-0.5 noise sample).  For such rays no NaN can arise in the walk — every
// time is fl(fl(plane - origin) * inv) with finite inv — so GLSL's min/max (vx_min / vx_max, compare + select) equal the
// hardware's v_min3 / v_max3, the transition of voxels.comp:198-201 is never 0, and dir_mask agrees with sign(inv).
//
// On top of that, a descend needs no arithmetic for its slab test: all cube planes are dyadic, so the time at which the
// ray crosses a plane, fl(fl(p - o) * inv), depends on the plane's coordinate p only — not on the node whose test
// computed it.  A child's near / far plane along an axis is the parent's near plane and mid plane (child on the near
// side) or its mid plane and far plane (child on the far side), and "far side" is the step's own `directional` bit.
// So with the per-axis crossing times of the current node kept in registers (en: near planes, ex: far planes), the
// child's are a select between them and the t_mid the step has just computed:
//     ray_cube_intersection(child) (voxels.comp:73-90, :218)  ==  max3(en'), min3(ex')      — bit for bit.
// A pop recomputes en / ex of the node it returns to by the shader's formula (the frames in LDS stay 8 bytes).
struct WalkF {
    f3 o, d, inv, center;
    f3 en, ex;                // crossing times of the current node's near / far planes
    float time, exit;
    uint32_t ix, iy, iz, lvl;
    uint32_t has_next_mask, octant, dir_mask;
    SvoRecord rec;
    int iterations;
};

__device__ __forceinline__ float vx_min3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ float vx_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float vx_copysign(float mag, float sgn) { return __builtin_copysignf(mag, sgn); }

__device__ __forceinline__ bool ray_is_regular(f3 inv) {
    const float big = __builtin_inff();
    return vx_abs(inv.x) > 0.0f && vx_abs(inv.x) < big && vx_abs(inv.y) > 0.0f && vx_abs(inv.y) < big && vx_abs(inv.z) > 0.0f &&
           vx_abs(inv.z) < big;
}

// crossing times of the near / far planes of the cube (c, half): the operands of ray_cube_intersection's max / min
__device__ __forceinline__ void plane_times(f3 o, f3 inv, f3 c, float half, f3& en, f3& ex) {
    const f3 hs = mk3(vx_copysign(half, inv.x), vx_copysign(half, inv.y), vx_copysign(half, inv.z));  // half * sign(inv)
    en = ((c - hs) - o) * inv;
    ex = ((c + hs) - o) * inv;
}

// voxels.comp:138-160 for a regular ray (inv already computed).  false: the ray misses the root cube.
__device__ __forceinline__ bool walkf_begin(WalkF& w, const SceneView& sc, f3 o, f3 d, f3 inv) {
    w.o = o;
    w.d = d;
    w.inv = inv;
    w.dir_mask = (d.x < 0.0f ? 4u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 1u : 0u);
    plane_times(o, inv, sc.root_center, 0.5f * sc.root_size, w.en, w.ex);
    const float entry = vx_max3(w.en.x, w.en.y, w.en.z);
    w.exit = vx_min3(w.ex.x, w.ex.y, w.ex.z);
    if (!(w.exit >= 0.0f && entry < w.exit)) return false;
    w.time = vx_max(0.0f, entry);
    w.center = sc.root_center;
    w.ix = w.iy = w.iz = w.lvl = w.has_next_mask = 0;
    w.rec = sc.root_rec;
    w.octant = octant_of(o + d * w.time, w.center);
    w.iterations = 0;
    return true;
}

#ifndef VXRT_LOCATE
#define VXRT_LOCATE 0   // measured: 0.157 vs 0.126 ms per bench frame with it (see the note below and DESIGN.md section 8)
#endif

// The initial descent of a ray that STARTS INSIDE the root cube (walkf_begin left time = 0: every secondary ray), in a loop of
// its own.  While time is 0 the loop of voxels.comp:163-246 only descends: current_octant(origin + dir * 0, centre) picks the
// child that holds the origin, level after level, until that octant is empty or a leaf — the path is a function of the origin
// alone, and a secondary ray spends 6-7 of its ~10 trips on it, ~115 instructions each in walkf_step.  Here:
//   1. the origin's cell at the finest level as integers j: cell j spans (lo_j, lo_j + cell] per axis — the strict > of
//      current_octant — decided with exact compares against the (dyadic, exactly representable) planes;
//   2. per level: the octant is one bit of each j; has_next and the sibling to resume with (voxels.comp:191-204, time = 0)
//      go into the frame exactly as a walked trip would leave them; the child's far-plane times by selection; no leaf test
//      (the octant is a child), no position, no time update (every near plane is behind the origin: max(0, <= 0) = 0);
//   3. the walk state at the level D where the octant is no child: D trips counted, near-plane times from plane_times (what
//      the selections would have produced: a plane's crossing time depends on the plane only).
// Results are bit-identical to walking those trips (tests/test_gpu_trace.py, test_gpu_degenerate.py, test_gpu_scenes.py with
// -DVXRT_LOCATE=1).  NOT the default: the loop is ~62 instructions per level against ~95 for a walked descend, but a wave in lock
// step pays max(D) of these and then max(R) walk trips instead of max(D + R) — the descent is 7 of a wave's ~36 trips, not 6 of a
// ray's 10 — and the second copy of the code costs 16-20 spilled registers in trace_kernel and bounce_kernel: 25 % slower.
__device__ __forceinline__ void walkf_locate(WalkF& w, const SceneView& sc, uint2* stack) {
    const int top = (1 << sc.levels) - 1;
    int j[3];
    const float oc[3] = {w.o.x, w.o.y, w.o.z}, rm[3] = {sc.root_min.x, sc.root_min.y, sc.root_min.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int c = int(__builtin_floorf((oc[k] - rm[k]) * sc.inv_cell));   // within 1 of the truth (one rounding of the difference)
        c = c < 0 ? 0 : (c > top ? top : c);
        const float lo = rm[k] + float(c) * sc.cell;                   // exact
        c += !(lo < oc[k]) ? -1 : (oc[k] > lo + sc.cell ? 1 : 0);
        j[k] = c < 0 ? 0 : (c > top ? top : c);                        // on the root's faces: low side / high side, as the compares go
    }
    uint32_t l = 0, oct, mask = 0;
    SvoRecord rec = w.rec;
    f3 center = w.center, ex = w.ex;
    float exit = w.exit, quarter = 0.25f * sc.root_size;
    for (;;) {
        const int sh = sc.levels - 1 - int(l);
        const uint32_t bx = uint32_t(j[0] >> sh) & 1u, by = uint32_t(j[1] >> sh) & 1u, bz = uint32_t(j[2] >> sh) & 1u;
        oct = bx << 2 | by << 1 | bz;
        const uint32_t bit = 1u << oct;
        if ((rec.masks & bit) == 0u || sh == 0) break;                 // empty or a leaf: walkf_step takes over here
        const f3 tm = (center - w.o) * w.inv;                          // voxels.comp:191
        const uint32_t directional = oct ^ w.dir_mask;
        const bool far_x = (directional & 4u) != 0u, far_y = (directional & 2u) != 0u, far_z = (directional & 1u) != 0u;
        const float mx = far_x ? kAlmostInfinity : tm.x;
        const float my = far_y ? kAlmostInfinity : tm.y;
        const float mz = far_z ? kAlmostInfinity : tm.z;
        const float next_time = vx_min3(mx, my, mz);
        const uint32_t transition = (mx == next_time) ? 4u : ((my == next_time) ? 2u : 1u);
        const bool has_next = next_time <= exit && (directional & transition) == 0u;
        stack[l * kStackStride] = make_uint2(rec.masks | (oct ^ transition) << 16, rec.base);   // read back only if has_next
        mask |= (has_next ? 1u : 0u) << l;
        const uint2 raw = *reinterpret_cast<const uint2*>(sc.svo + (rec.base + __popc(rec.masks & (bit - 1u))));
        VX_TOUCH(sc.touch_nodes, (rec.base + __popc(rec.masks & (bit - 1u))) >> 3);
        ex = mk3(far_x ? ex.x : tm.x, far_y ? ex.y : tm.y, far_z ? ex.z : tm.z);
        exit = vx_min3(ex.x, ex.y, ex.z);
        center = center + mk3(bx ? quarter : -quarter, by ? quarter : -quarter, bz ? quarter : -quarter);   // exact
        quarter *= 0.5f;
        rec.masks = raw.x & 0xffffu;
        rec.base = raw.y;
        l++;
    }
    if (l == 0u) return;                                               // nothing to skip
    const int up = sc.levels - int(l);
    w.ix = uint32_t(j[0] >> up); w.iy = uint32_t(j[1] >> up); w.iz = uint32_t(j[2] >> up);
    w.lvl = l;
    w.has_next_mask = mask;
    w.iterations = int(l);
    w.rec = rec;
    w.octant = oct;
    w.center = center;
    w.ex = ex;
    w.exit = exit;
    const f3 hs = mk3(vx_copysign(2.0f * quarter, w.inv.x), vx_copysign(2.0f * quarter, w.inv.y), vx_copysign(2.0f * quarter, w.inv.z));
    w.en = ((center - hs) - w.o) * w.inv;
}

// One trip of the while(true) loop (voxels.comp:163-246) for a regular ray, max_distance = 2^30.
__device__ __forceinline__ int walkf_step(WalkF& w, const SceneView& sc, uint2* stack) {
    if (++w.iterations >= 2048) return kWalkCap;             // voxels.comp:166-169
    const uint32_t bit = 1u << w.octant;
    {   // voxels.comp:171-177 as ONE branch (every divergent region costs exec-mask bookkeeping: measured + 1.7 %); the
        // distance check comes first in the shader, so it wins when both hold
        const bool too_far = w.time > kAlmostInfinity, leaf = (w.rec.masks & (bit << 8)) != 0u;
        if (too_far | leaf) return too_far ? kWalkMiss : kWalkLeaf;
    }

#ifndef VXRT_EARLY_FETCH
#define VXRT_EARLY_FETCH 0   // experiment: ask for the child's record before the sibling search, so that the search runs under the load
#endif
#if VXRT_EARLY_FETCH
    uint2 early = make_uint2(0u, 0u);
    if ((w.rec.masks & bit) != 0u) {
        early = *reinterpret_cast<const uint2*>(sc.svo + (w.rec.base + __popc(w.rec.masks & (bit - 1u))));
        VX_TOUCH(sc.touch_nodes, (w.rec.base + __popc(w.rec.masks & (bit - 1u))) >> 3);
        asm volatile("" ::: "memory");   // the load is issued here (nothing that touches memory moves across), waited for where it is used
    }
#endif
    const f3 tm = (w.center - w.o) * w.inv;                  // voxels.comp:191
    const uint32_t directional = w.octant ^ w.dir_mask;
    const bool far_x = (directional & 4u) != 0u, far_y = (directional & 2u) != 0u, far_z = (directional & 1u) != 0u;
    const float mx = far_x ? kAlmostInfinity : tm.x;
    const float my = far_y ? kAlmostInfinity : tm.y;
    const float mz = far_z ? kAlmostInfinity : tm.z;
    const float next_time = vx_min3(mx, my, mz);
    const uint32_t transition = (mx == next_time) ? 4u : ((my == next_time) ? 2u : 1u);
    const bool has_next = next_time <= w.exit && (directional & transition) == 0u;
    const bool is_child = (w.rec.masks & bit) != 0u;         // value > 0
#if VXRT_EARLY_FETCH
    (void)is_child;
#endif

    if (is_child || !has_next) {
        uint2 raw;
        if (is_child) {  // voxels.comp:205-214
            if (has_next) {
                stack[w.lvl * kStackStride] = make_uint2(w.rec.masks | (w.octant ^ transition) << 16, w.rec.base);
                w.has_next_mask |= 1u << w.lvl;
            }
#if VXRT_EARLY_FETCH
            raw = early;
#else
            raw = *reinterpret_cast<const uint2*>(sc.svo + (w.rec.base + __popc(w.rec.masks & (bit - 1u))));
            VX_TOUCH(sc.touch_nodes, (w.rec.base + __popc(w.rec.masks & (bit - 1u))) >> 3);
#endif
            w.ix = (w.ix << 1) | ((w.octant >> 2) & 1u);
            w.iy = (w.iy << 1) | ((w.octant >> 1) & 1u);
            w.iz = (w.iz << 1) | (w.octant & 1u);
            w.lvl++;
        } else {         // voxels.comp:225-234
            // A pop that would go back to the ROOT of a tree whose root holds one occupied slot (every scene whose coordinates are
            // >= 0: the root is centred on 0, src/context.rs:782-786) can only walk the root's other, empty octants — a sibling is
            // never revisited — and end in `top == 0 -> false` (voxels.comp:226): a certain miss, 2-4 trips early, unless those trips
            // would have run into the 2048-trip cap first (then the walk goes on and finds out).  Bit-exact (the parity suite ran with it),
            // and measured SLOWER on one box, three alternations (round 6, scripts/r06/06_root_exit.sh): 0.1085-0.1095 against 0.1063-0.1084 ms
            // per bench frame, 0.2698-0.2707 against 0.2647-0.2665 in the close view — a lane that leaves early does not shorten its
            // wave, and the two extra instructions sit in every pop.  Off.
#ifndef VXRT_ROOT_EXIT
#define VXRT_ROOT_EXIT 0
#endif
#if VXRT_ROOT_EXIT
            if ((w.has_next_mask & sc.pop_mask) == 0u && (w.has_next_mask == 0u || w.iterations < 2040)) return kWalkMiss;
#else
            if (w.has_next_mask == 0u) return kWalkMiss;
#endif
            const uint32_t l = 31u - uint32_t(__clz(int(w.has_next_mask)));
            w.has_next_mask &= ~(1u << l);
            const uint32_t up = w.lvl - l;
            w.ix >>= up; w.iy >>= up; w.iz >>= up;
            w.lvl = l;
            raw = stack[l * kStackStride];
            asm volatile("" : "+v"(raw.x), "+v"(raw.y));  // keep it an LDS read (see walk_step)
        }
        const float size = __builtin_ldexpf(sc.root_size, -int(w.lvl));
        w.center = sc.root_min + mk3(float(w.ix) + 0.5f, float(w.iy) + 0.5f, float(w.iz) + 0.5f) * size;
        if (is_child) {  // voxels.comp:216-221, the slab test by selection
            w.en = mk3(far_x ? tm.x : w.en.x, far_y ? tm.y : w.en.y, far_z ? tm.z : w.en.z);
            w.ex = mk3(far_x ? w.ex.x : tm.x, far_y ? w.ex.y : tm.y, far_z ? w.ex.z : tm.z);
            w.octant = octant_of(w.o + w.d * w.time, w.center);
            w.time = vx_max(w.time, vx_max3(w.en.x, w.en.y, w.en.z));
        } else {         // voxels.comp:236-242
            plane_times(w.o, w.inv, w.center, 0.5f * size, w.en, w.ex);
            w.time = w.exit;
            w.octant = (raw.x >> 16) & 7u;
        }
        w.exit = vx_min3(w.ex.x, w.ex.y, w.ex.z);
        w.rec.masks = raw.x & 0xffffu;
        w.rec.base = raw.y;
    } else {  // voxels.comp:222-224
        w.octant ^= transition;
        w.time = next_time;
    }
    return kWalkOn;
}

// index of the leaf word the walk stopped at (status kWalkLeaf)
__device__ __forceinline__ uint32_t walk_leaf_index(const Walk& w) {
    const uint32_t bit = 1u << w.octant;
    return w.rec.base + __popc((w.rec.masks >> 8) & (bit - 1u));
}

// voxels.comp:177-189: normal of a hit at `time` on the unit voxel whose centre is `oc`
__device__ __forceinline__ f3 hit_normal(f3 o, f3 d, float time, f3 oc) {
    f3 p = o + time * d;
    f3 dist = mk3(vx_abs(p.x - oc.x), vx_abs(p.y - oc.y), vx_abs(p.z - oc.z));
    float m = vx_max(vx_max(dist.x, dist.y), dist.z);
    f3 mask = mk3(dist.x == m ? 1.0f : 0.0f, dist.y == m ? 1.0f : 0.0f, dist.z == m ? 1.0f : 0.0f);
    return mask * mk3(-vx_sign(d.x), -vx_sign(d.y), -vx_sign(d.z));
}

// hit resolution shared by both walks: leaf word and normal (voxels.comp:177-189)
__device__ __forceinline__ bool finish_ray(const SceneView& sc, int status, f3 o, f3 d, float time, f3 center, uint32_t lvl,
                                           uint32_t octant, uint32_t leaf_index, RayHit& hit) {
    hit.time = time;
    hit.normal = splat3(0.0f);
    if (status == kWalkMiss) return false;
    if (status == kWalkCap) {
        hit.node = kLeafBit;
        return true;
    }
    hit.node = sc.leaves[leaf_index];
    VX_TOUCH(sc.touch_leaves, leaf_index >> 4);
    const float size = __builtin_ldexpf(sc.root_size, -int(lvl));
    f3 delta = mk3(float((octant >> 2) & 1u), float((octant >> 1) & 1u), float(octant & 1u));
    f3 oc = center + (0.5f * size) * (delta - splat3(0.5f));
    hit.normal = hit_normal(o, d, time, oc);
    return true;
}

__device__ __forceinline__ bool cast_ray(const SceneView& sc, f3 o, f3 d, float max_distance, uint2* stack, RayHit& hit) {
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int status = kWalkMiss;
    f3 center = splat3(0.0f);
    float time = 0.0f;
    uint32_t lvl = 0, octant = 0, leaf = 0;
    bool entered;
    if (ray_is_regular(inv) && max_distance == kAlmostInfinity) {
        WalkF w;
        entered = walkf_begin(w, sc, o, d, inv);
        if (entered) {
#if VXRT_LOCATE
            if (w.time == 0.0f) walkf_locate(w, sc, stack);
#endif
            do { status = walkf_step(w, sc, stack); } while (status == kWalkOn);
            const uint32_t bit = 1u << w.octant;
            center = w.center; time = w.time; lvl = w.lvl; octant = w.octant;
            leaf = w.rec.base + __popc((w.rec.masks >> 8) & (bit - 1u));
        }
    } else {  // a direction component is 0 (or NaN): the shader's text, NaN and all
        Walk w;
        entered = walk_begin(w, sc, o, d);
        if (entered) {
            do { status = walk_step(w, sc, max_distance, stack); } while (status == kWalkOn);
            center = w.center; time = w.time; lvl = w.lvl; octant = w.octant;
            leaf = walk_leaf_index(w);
        }
    }
    if (!entered) return false;
    return finish_ray(sc, status, o, d, time, center, lvl, octant, leaf, hit);
}

__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
#if VXRT_VARIANTS
#include "walk_wide.h"
#endif

__device__ __forceinline__ f3 node_rgb(int32_t node) {
    return mk3(float((node >> 16) & 0xff), float((node >> 8) & 0xff), float(node & 0xff));
}
// c / 255.0f for c = 0..255 (IEEE division, folded by the compiler): node_color's three divisions are a table lookup.
#define VX_D1(c) float(c) / 255.0f
#define VX_D4(c) VX_D1(c), VX_D1(c + 1), VX_D1(c + 2), VX_D1(c + 3)
#define VX_D16(c) VX_D4(c), VX_D4(c + 4), VX_D4(c + 8), VX_D4(c + 12)
#define VX_D64(c) VX_D16(c), VX_D16(c + 16), VX_D16(c + 32), VX_D16(c + 48)
__device__ const float kOver255[256] = {VX_D64(0), VX_D64(64), VX_D64(128), VX_D64(192)};
// node_color, voxels.comp:253-258
__device__ __forceinline__ f3 node_color(int32_t node) {
    return mk3(kOver255[(node >> 16) & 0xff], kOver255[(node >> 8) & 0xff], kOver255[node & 0xff]);
}
// node_emmitance, voxels.comp:260-266: ((e * emit_strength) * rgb) / 255 with e = 0 or 1.  For a voxel that does not emit
// the numerator is 0 * emit_strength * rgb = 0 (for any finite emit_strength) and the three divisions are skipped.
__device__ __forceinline__ f3 node_emittance(int32_t node, float emit_strength) {
    const bool emits = (node & kEmitBit) != 0;
    if (!emits && 0.0f * emit_strength == 0.0f) return splat3(0.0f);
    float e = emits ? 1.0f : 0.0f;
    return ((e * emit_strength) * node_rgb(node)) / 255.0f;
}

struct Rng {  // rand(), voxels.comp:268-275
    uint32_t index;
    const float* noise;
    __device__ __forceinline__ float next() {
        index = (index + kNoiseLayer) % kNoiseTotal;
        return noise[index];
    }
};

// random_hemisphere, voxels.comp:277-287
__device__ __forceinline__ f3 random_hemisphere(f3 n, Rng& rng) {
    float phi = (2.0f * 3.14159265358979f) * rng.next();
    f3 r;
    r.x = 2.0f * rng.next() - 1.0f;
    float plane_radius = vx_sqrt(1.0f - r.x * r.x);
    r.y = plane_radius * vx_cos(phi);
    r.z = plane_radius * vx_sin(phi);
    return r - n * vx_min0(2.0f * dot3(n, r));  // min(0.0, .): see vx_min0
}

__device__ __forceinline__ f3 xyz4(float4 v) { return mk3(v.x, v.y, v.z); }

// Rays cast by this wave -> one atomic on one of kRaySlots counters, each on a 64-byte line of its own.
// (A single counter word saturates at ~88 atomics/us chip-wide: with one atomic per wave that alone
// put a 0.4 ms floor under a 1080p frame.)
__device__ __forceinline__ void count_rays(unsigned long long* slots, uint32_t rays, int lane) {
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
    if (lane == 0 && rays != 0) {
        const unsigned slot = (blockIdx.x + blockIdx.y * gridDim.x) * 4u + (threadIdx.x >> 6);
        atomicAdd(slots + size_t(slot % kRaySlots) * 8u, (unsigned long long)rays);
    }
}

// ---- path records and queues (wavefront variants) ------------------------------------------------------
constexpr unsigned kShards = 64;
constexpr unsigned kCountStride = 16;  // uints: one 64-byte line per shard counter

struct PathRec {  // 4 x float4
    f3 hit_pos; int32_t node;
    f3 dir; uint32_t normal_ambient;   // normal: 2 bits per axis (0:+0, 1:+1, 2:-1, 3:-0); ambient_rays << 8
    f3 sample; uint32_t rng_index;
    f3 blend; uint32_t pix;
};

__device__ __forceinline__ uint32_t pack_axis(float v) { return v == 0.0f ? ((vx_f2u(v) >> 31) ? 3u : 0u) : (v > 0.0f ? 1u : 2u); }
__device__ __forceinline__ float unpack_axis(uint32_t c) { return c == 0u ? 0.0f : (c == 1u ? 1.0f : (c == 2u ? -1.0f : -0.0f)); }

__device__ __forceinline__ void store_rec(float4* q, const PathRec& r) {
    q[0] = make_float4(r.hit_pos.x, r.hit_pos.y, r.hit_pos.z, __int_as_float(r.node));
    q[1] = make_float4(r.dir.x, r.dir.y, r.dir.z, __uint_as_float(r.normal_ambient));
    q[2] = make_float4(r.sample.x, r.sample.y, r.sample.z, __uint_as_float(r.rng_index));
    q[3] = make_float4(r.blend.x, r.blend.y, r.blend.z, __uint_as_float(r.pix));
}
__device__ __forceinline__ PathRec load_rec(const float4* q) {
    const float4 a = q[0], b = q[1], c = q[2], d = q[3];
    PathRec r;
    r.hit_pos = mk3(a.x, a.y, a.z); r.node = __float_as_int(a.w);
    r.dir = mk3(b.x, b.y, b.z); r.normal_ambient = __float_as_uint(b.w);
    r.sample = mk3(c.x, c.y, c.z); r.rng_index = __float_as_uint(c.w);
    r.blend = mk3(d.x, d.y, d.z); r.pix = __float_as_uint(d.w);
    return r;
}

// Append the records of the lanes with `keep` to shard `shard` of the queue (called by all 64 lanes).
__device__ __forceinline__ void queue_append(const PathQueue& q, unsigned shard, bool keep, const PathRec& rec, int lane) {
    const unsigned long long m = __ballot(keep);
    if (m == 0ull) return;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(q.counts + shard * kCountStride, unsigned(__popcll(m)));
    base = __builtin_amdgcn_readfirstlane(base);
    if (keep) {
        const unsigned rank = __builtin_amdgcn_mbcnt_hi(unsigned(m >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(m), 0u));
        store_rec(q.recs + (size_t(shard) * q.shard_capacity + base + rank) * 4u, rec);
    }
}

// The same, for a queue that is sized by need instead of for the worst case (the compacted tail: api_trace.hip sizes it from what
// earlier launches queued): called from INSIDE the path loop by exactly the lanes that want to hand their path over.  Returns the
// lane's record slot in the shard, or kNoSlot when the shard is full — the lane then keeps following its path itself, which gives
// the same result (the hand-over only moves work).  The shard's counter keeps counting past the capacity, so the host sees how
// much room was wanted; consumers clamp it (queue_count).
constexpr uint32_t kNoSlot = 0xffffffffu;
__device__ __forceinline__ uint32_t queue_reserve(const PathQueue& q, unsigned shard) {
    const unsigned long long m = __ballot(1);   // the lanes that are here
    const unsigned rank = __builtin_amdgcn_mbcnt_hi(unsigned(m >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(m), 0u));
    unsigned base = 0;
    if (rank == 0u) base = atomicAdd(q.counts + shard * kCountStride, unsigned(__popcll(m)));
    base = __builtin_amdgcn_readfirstlane(base);
    const unsigned slot = base + rank;
    return slot < q.shard_capacity ? slot : kNoSlot;
}
__device__ __forceinline__ void queue_store(const PathQueue& q, unsigned shard, uint32_t slot, const PathRec& rec) {
    store_rec(q.recs + (size_t(shard) * q.shard_capacity + slot) * 4u, rec);
}
// fused_kernel's form (trace.hip): the consumer may be polling this slot already.  Seven 8-byte stores at agent scope — write-through, so
// that a wave on another XCD reads them from memory — then, when those have been written (s_waitcnt), the eighth: dir.z and normal_ambient
// with the launch's stamp in its top 16 bits, which is what the consumer waits for.
__device__ __forceinline__ void queue_store_fused(const PathQueue& q, unsigned shard, uint32_t slot, const PathRec& r, uint32_t stamp) {
    unsigned long long* w = reinterpret_cast<unsigned long long*>(q.recs + (size_t(shard) * q.shard_capacity + slot) * 4u);
    auto pack = [](uint32_t lo, uint32_t hi) { return (unsigned long long)lo | (unsigned long long)hi << 32; };
    const unsigned long long v[8] = {pack(vx_f2u(r.hit_pos.x), vx_f2u(r.hit_pos.y)), pack(vx_f2u(r.hit_pos.z), uint32_t(r.node)),
                                     pack(vx_f2u(r.dir.x), vx_f2u(r.dir.y)), pack(vx_f2u(r.dir.z), (r.normal_ambient & 0xffffu) | stamp << 16),
                                     pack(vx_f2u(r.sample.x), vx_f2u(r.sample.y)), pack(vx_f2u(r.sample.z), r.rng_index),
                                     pack(vx_f2u(r.blend.x), vx_f2u(r.blend.y)), pack(vx_f2u(r.blend.z), r.pix)};
    for (int i = 0; i < 8; i++)
        if (i != 3) __hip_atomic_store(w + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);      // vmcnt(0): the seven are written through before the stamp can be seen
    __hip_atomic_store(w + 3, v[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... and the consumer's side: w3 = the polled word (dir.z | normal_ambient with the stamp), the rest read after it at agent scope
__device__ __forceinline__ PathRec load_rec_fused(const float4* q, unsigned long long w3) {
    const unsigned long long* w = reinterpret_cast<const unsigned long long*>(q);
    unsigned long long v[8];
    for (int i = 0; i < 8; i++) v[i] = i == 3 ? w3 : __hip_atomic_load(w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    auto lo = [](unsigned long long x) { return uint32_t(x); };
    auto hi = [](unsigned long long x) { return uint32_t(x >> 32); };
    PathRec r;
    r.hit_pos = mk3(vx_u2f(lo(v[0])), vx_u2f(hi(v[0])), vx_u2f(lo(v[1]))); r.node = int32_t(hi(v[1]));
    r.dir = mk3(vx_u2f(lo(v[2])), vx_u2f(hi(v[2])), vx_u2f(lo(v[3]))); r.normal_ambient = hi(v[3]) & 0xffffu;
    r.sample = mk3(vx_u2f(lo(v[4])), vx_u2f(hi(v[4])), vx_u2f(lo(v[5]))); r.rng_index = hi(v[5]);
    r.blend = mk3(vx_u2f(lo(v[6])), vx_u2f(hi(v[6])), vx_u2f(lo(v[7]))); r.pix = hi(v[7]);
    return r;
}
// records a consumer finds in shard `shard`
__device__ __forceinline__ unsigned queue_count(const PathQueue& q, unsigned shard) {
    const unsigned n = q.counts[shard * kCountStride];
    return n < q.shard_capacity ? n : q.shard_capacity;
}

constexpr unsigned kFlagSun = 1u, kFlagBounce = 2u;
constexpr unsigned kFlagSunTraced = 4u, kFlagBounceTraced = 8u;   // ray queues: the ray's result is already stored (an irregular ray)
struct Shaded {  // what shading a hit produces
    f3 sample, blend, pend_sun, pend_emit, origin, sun_dir, bounce_dir;
    uint32_t ambient_rays, flags;
};

// voxels.comp:314-371 for a hit at path segment `bounce` — everything between two cast_ray calls.
__device__ __forceinline__ Shaded shade_hit(const TraceArgs& a, int bounce, f3 hit_pos, f3 dir, f3 n, int32_t node, f3 sample, f3 blend,
                                            uint32_t ambient_rays, Rng& rng, f3 sun_dir, f3 sun_color) {
    Shaded r;
    const f3 color = bounce == 0 ? splat3(1.0f) : node_color(node);   // voxels.comp:317
    const f3 emit = node_emittance(node, a.emit_strength);
    r.origin = hit_pos + 1e-5f * n;                                    // voxels.comp:333,353,370
    r.pend_sun = r.pend_emit = r.sun_dir = splat3(0.0f);
    r.flags = bounce + 1 < a.max_bounces ? kFlagBounce : 0u;
    if (rng.next() < a.specularity) {  // specular                     voxels.comp:326-334
        r.bounce_dir = norm3(reflect3(dir, n));
        sample = sample + emit * blend;
        blend = blend * ((2.0f * color) * dot3(r.bounce_dir, n));
    } else if (a.sun_strength > 0.0f) {  // diffuse + sun sample         voxels.comp:339-371
        float r0 = rng.next(), r1 = rng.next(), r2 = rng.next();
        f3 up_dir = norm3(cross3(mk3(r0, r1, r2), sun_dir));
        f3 right_dir = norm3(cross3(sun_dir, up_dir));
        float dx = 2.0f * rng.next() - 1.0f;
        float dy = 2.0f * rng.next() - 1.0f;
        f3 light_dir = ld3(a.sun_dir_n) + (dx * right_dir + dy * up_dir) * a.sun_size;
        r.sun_dir = norm3(-light_dir);
        ambient_rays++;
        r.pend_sun = ((sun_color * color) * blend) * vx_max(0.0f, dot3(n, r.sun_dir));
        r.bounce_dir = random_hemisphere(n, rng);
        r.pend_emit = emit * blend;
        blend = blend * (color * dot3(n, r.bounce_dir));
        r.flags |= kFlagSun;
    } else {  // diffuse, sun switched off
        r.bounce_dir = random_hemisphere(n, rng);
        sample = sample + emit * blend;
        blend = blend * (color * dot3(n, r.bounce_dir));
    }
    r.sample = sample;
    r.blend = blend;
    r.ambient_rays = ambient_rays;
    return r;
}

// pow(max(0, dot(dir, -sun)), 1 / sun_size^2) of voxels.comp:378-381 — exponent 400 by default, so the sun's disc is a few degrees
// wide and the power is EXACTLY zero for every direction further from the sun: vx_pow = vx_exp(y * vx_log(x)), and vx_exp
// returns +0 for every argument below -87.3.  TraceArgs::sun_zero_below is a bound (made on the host with a safety margin three
// orders of magnitude above vx_log's error) below which that is certain, so most sky pixels skip the ~90 instructions of log and
// exp; the result is the same bit pattern either way (the bound: tests/test_detmath_cpu.py::test_sun_power_is_exactly_zero_below_the_kernels_bound;
// on the device, other sun sizes with the sun in view: tests/test_gpu_trace.py::test_sun_power_shortcut_with_other_sun_sizes; the
// build without the shortcut, -DVXRT_SUN_SHORTCUT=0, goes through the parity tests in scripts/test_variants.sh).
__device__ __forceinline__ float sun_power_of(const TraceArgs& a, f3 d) {
    const float x = vx_max(0.0f, dot3(d, ld3(a.neg_sun_dir_n)));
#ifndef VXRT_SUN_SHORTCUT
#define VXRT_SUN_SHORTCUT 1
#endif
#if VXRT_SUN_SHORTCUT
    float p = 0.0f;
    if (!(x < a.sun_zero_below)) p = vx_pow(x, a.sun_exponent);
    return p;
#else
    return vx_pow(x, a.sun_exponent);
#endif
}

// The G-buffer is written once and never read back by the tracer: stream it (global_store ... nt) so that the SVO records
// and noise layers keep their cache lines.  With plain stores rocprofv3's WRITE_SIZE read 1.45 x the bytes stored (dirty
// lines written back more than once); with nt stores 1.06 x, and the frame is 2 % faster.
typedef float vx_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_out(float4* p, float4 v) {
    vx_v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<vx_v4f*>(p));
}

__device__ __forceinline__ SceneView make_scene(const TraceArgs& a) {
    SceneView sc;
    sc.svo = a.svo;
    sc.leaves = a.leaves;
    sc.root_rec = a.root_rec;
    sc.root_center = ld3(a.root_center);
    sc.root_size = a.root_size;
    sc.root_min = sc.root_center - splat3(0.5f * a.root_size);
    sc.levels = a.stack_levels;
    sc.cell = __builtin_ldexpf(a.root_size, -a.stack_levels);
    sc.inv_cell = 1.0f / sc.cell;
    const uint32_t occupied = (a.root_rec.masks | a.root_rec.masks >> 8) & 0xffu;
    sc.pop_mask = (occupied & (occupied - 1u)) == 0u ? ~1u : ~0u;
#if VXRT_VARIANTS
    sc.touch_nodes = a.touch_nodes;
    sc.touch_leaves = a.touch_leaves;
#endif
    return sc;
}

// cast_ray over either scene format, with the thread's column of the block's LDS frames (`lds`: the kernel's dynamic shared memory;
// 8 bytes per tree level and thread for the 8-byte records, 16 bytes per PAIR of levels for the wide ones).
template <bool kWide> struct Caster;
template <> struct Caster<false> {
    SceneView sc;
    uint2* stack;
    __device__ __forceinline__ Caster(const TraceArgs& a, uint4* lds, int tid) : sc(make_scene(a)), stack(reinterpret_cast<uint2*>(lds) + tid) {}
    __device__ __forceinline__ bool cast(f3 o, f3 d, RayHit& hit) const { return cast_ray(sc, o, d, kAlmostInfinity, stack, hit); }
};
#if VXRT_VARIANTS
template <> struct Caster<true> {
    SceneW sc;
    uint4* stack;
    __device__ __forceinline__ Caster(const TraceArgs& a, uint4* lds, int tid) : sc(make_scene_w(a)), stack(lds + tid) {}
    __device__ __forceinline__ bool cast(f3 o, f3 d, RayHit& hit) const { return cast_ray_w(sc, o, d, kAlmostInfinity, stack, hit); }
};
#endif
// dynamic shared memory a block of `threads` threads needs for its frames
inline size_t caster_lds_bytes(const TraceArgs& a, bool wide, int threads) {
    if (!wide) return size_t(a.stack_levels) * size_t(threads) * sizeof(uint2);
    const int slots = (a.node_levels + (a.node_levels & 1)) / 2;
    return size_t(slots < 1 ? 1 : slots) * size_t(threads) * sizeof(uint4);
}

__device__ __forceinline__ void zero_counts(unsigned* counts, int tid) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid < int(kShards)) {
        counts[tid * kCountStride] = 0u;
    }
}

}  // namespace
}  // namespace vxrt
