// walk_wide.h — cast_bounded_ray (shaders/voxels.comp:134-247) over WIDE records (kernels.h: WideRec, two tree levels per 16-byte
// record).  Included by trace_common.h; everything lives in its anonymous namespace.
//
// The walk visits exactly the (node, octant, time) sequence of the 8-byte walk above (walk_step / walkf_step) — the float code
// is the same text — only where a node's masks come from differs:
//   * the current node is either the TOP of the wide record in registers or one of its SUBS (`sub`); which, is the parity of its
//     level (SceneW::parity): levels are paired from the bottom, so leaf parents are always subs;
//   * descend top -> sub: the sub's masks are a byte of the record — no memory access;
//   * descend sub -> child: one 16-byte load of the grandchild record (base + popcount of the 64-bit mask below the slot);
//   * a pop to the top of the wide record the lane is in needs nothing but registers; a pop further up reads the 16-byte frame of
//     that wide level from LDS ([wide level][thread], written when the lane left the record downwards with a sibling pending
//     on either of its two levels).
// Results are bit-identical to the 8-byte walk (the whole GPU parity suite runs with both formats: VXRT_WIDE=0 / 1).
#pragma once

struct SceneW {
    const WideRec* wide;
    const int32_t* leaves;
    WideRec root;
    f3 root_center;
    f3 root_min;
    float root_size;
    int levels;        // node levels L: root = 0, leaf parents = L - 1
    uint32_t parity;   // L & 1: level l is a TOP level iff ((l + parity) & 1) == 0 (parity 1: the root is the one sub of a virtual top)
    uint32_t* touch_nodes;   // TraceArgs::touch_nodes / touch_leaves (null: off)
    uint32_t* touch_leaves;
};

__device__ __forceinline__ SceneW make_scene_w(const TraceArgs& a) {
    SceneW sc;
    sc.wide = a.wide;
    sc.leaves = a.leaves;
    sc.root = a.wide_root;
    sc.root_center = ld3(a.root_center);
    sc.root_size = a.root_size;
    sc.root_min = sc.root_center - splat3(0.5f * a.root_size);
    sc.levels = a.node_levels;
    sc.parity = uint32_t(a.node_levels) & 1u;
    sc.touch_nodes = a.touch_nodes;
    sc.touch_leaves = a.touch_leaves;
    return sc;
}

// byte s of the record's 64-bit mask: the occupancy of sub s
__device__ __forceinline__ uint32_t wide_sub_byte(uint32_t mlo, uint32_t mhi, uint32_t s) {
    return (((s & 4u) ? mhi : mlo) >> ((s & 3u) * 8u)) & 0xffu;
}
// set bits of the 64-bit mask below bit `pos` (0..63): the rank of grandchild / leaf `pos` among the record's
__device__ __forceinline__ uint32_t wide_rank(uint32_t mlo, uint32_t mhi, uint32_t pos) {
    const uint32_t below = (1u << (pos & 31u)) - 1u;
    return (pos & 32u) ? uint32_t(__popc(mlo)) + uint32_t(__popc(mhi & below)) : uint32_t(__popc(mlo & below));
}

// Loop-carried state.  `info` packs what identifies the current node inside its wide record:
//   bits 0-7 child mask and 8-15 leaf mask of the CURRENT node (what walk_step calls rec.masks), 16-18 sub (when the node is a sub),
//   19-21 the sibling the record's top resumes with (valid while the top's bit is set in has_next_mask), 24-31 the top's mask.
template <bool kRegular>
struct WalkW {
    f3 o, d, inv, center;
    f3 en, ex;                // regular rays: crossing times of the current node's near / far planes (see WalkF)
    f3 sg;                    // other rays: sign(inv), and the node size below
    float size;
    float time, exit;
    uint32_t ix, iy, iz, lvl;
    uint32_t has_next_mask, octant, dir_mask;
    uint32_t mlo, mhi, base, info;
    int iterations;
};

template <bool kRegular>
__device__ __forceinline__ void walkw_enter_root(WalkW<kRegular>& w, const SceneW& sc) {
    w.mlo = sc.root.mlo; w.mhi = sc.root.mhi; w.base = sc.root.base;
    const uint32_t top = sc.root.top & 0xffu;
    if (sc.parity) {   // the root is sub 0 of a virtual top
        const uint32_t byte = sc.root.mlo & 0xffu;
        w.info = top << 24 | (sc.levels == 1 ? byte << 8 : byte);
    } else {
        w.info = top << 24 | top;
    }
    w.ix = w.iy = w.iz = w.lvl = w.has_next_mask = 0;
    w.iterations = 0;
}

// voxels.comp:138-160.  false: the ray misses the root cube.
__device__ __forceinline__ bool walkw_begin(WalkW<true>& w, const SceneW& sc, f3 o, f3 d, f3 inv) {
    w.o = o; w.d = d; w.inv = inv;
    w.dir_mask = (d.x < 0.0f ? 4u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 1u : 0u);
    plane_times(o, inv, sc.root_center, 0.5f * sc.root_size, w.en, w.ex);
    const float entry = vx_max3(w.en.x, w.en.y, w.en.z);
    w.exit = vx_min3(w.ex.x, w.ex.y, w.ex.z);
    if (!(w.exit >= 0.0f && entry < w.exit)) return false;
    w.time = vx_max(0.0f, entry);
    w.center = sc.root_center;
    walkw_enter_root(w, sc);
    w.octant = octant_of(o + d * w.time, w.center);
    return true;
}
__device__ __forceinline__ bool walkw_begin(WalkW<false>& w, const SceneW& sc, f3 o, f3 d) {
    w.o = o; w.d = d;
    w.dir_mask = (d.x < 0.0f ? 4u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 1u : 0u);
    w.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    w.sg = mk3(vx_sign(w.inv.x), vx_sign(w.inv.y), vx_sign(w.inv.z));
    float entry;
    if (!slab(o, w.inv, w.sg, sc.root_center, 0.5f * sc.root_size, entry, w.exit)) return false;
    w.time = vx_max(0.0f, entry);
    w.size = sc.root_size;
    w.center = sc.root_center;
    walkw_enter_root(w, sc);
    w.octant = octant_of(o + d * w.time, w.center);
    return true;
}

// One trip of the while(true) loop, voxels.comp:163-246.  `stack`: this thread's column of the LDS frames, entry j at
// stack[j * kStackStride] (16 bytes per wide level).
template <bool kRegular>
__device__ __forceinline__ int walkw_step(WalkW<kRegular>& w, const SceneW& sc, float max_distance, uint4* stack) {
    if (++w.iterations >= 2048) return kWalkCap;             // voxels.comp:166-169
    const uint32_t bit = 1u << w.octant;
    {   // voxels.comp:171-177; the distance check comes first in the shader, so it wins when both hold
        const bool too_far = w.time > max_distance, leaf = (w.info & (bit << 8)) != 0u;
        if (too_far | leaf) return too_far ? kWalkMiss : kWalkLeaf;
    }

    // next sibling through the node's mid planes                     voxels.comp:191-203
    const f3 tm = (w.center - w.o) * w.inv;
    const uint32_t directional = w.octant ^ w.dir_mask;
    const bool far_x = (directional & 4u) != 0u, far_y = (directional & 2u) != 0u, far_z = (directional & 1u) != 0u;
    const float mx = far_x ? kAlmostInfinity : tm.x;
    const float my = far_y ? kAlmostInfinity : tm.y;
    const float mz = far_z ? kAlmostInfinity : tm.z;
    float next_time;
    uint32_t transition;
    bool has_next;
    if (kRegular) {   // no NaN can arise: GLSL's min equals v_min3, the transition is never 0 (see walkf_step)
        next_time = vx_min3(mx, my, mz);
        transition = (mx == next_time) ? 4u : ((my == next_time) ? 2u : 1u);
        has_next = next_time <= w.exit && (directional & transition) == 0u;
    } else {
        next_time = vx_min(vx_min(mx, my), mz);
        transition = (mx == next_time) ? 4u : ((my == next_time) ? 2u : ((mz == next_time) ? 1u : 0u));
        has_next = next_time <= w.exit && transition != 0u && (directional & transition) == 0u;
    }
    const uint32_t next_octant = w.octant ^ transition;
    const bool is_child = (w.info & bit) != 0u;              // value > 0

    if (is_child || !has_next) {
        uint32_t popped_octant = 0u;
        if (is_child) {  // voxels.comp:205-214
            const bool from_top = ((w.lvl + sc.parity) & 1u) == 0u;
            if (has_next) w.has_next_mask |= 1u << w.lvl;
            if (from_top) {   // the child is a sub of the record in registers
                const uint32_t byte = wide_sub_byte(w.mlo, w.mhi, w.octant);
                const bool leaf_parent = w.lvl + 2u == uint32_t(sc.levels);
                w.info = (w.info & 0xff000000u) | next_octant << 19 | w.octant << 16 | (leaf_parent ? byte << 8 : byte);
            } else {          // the child is a wide record of its own: leave this one, keeping what a pop back into it will need
                const uint32_t above = (w.has_next_mask >> ((w.lvl + 31u) & 31u)) & (w.lvl != 0u ? 1u : 0u);   // the top's sibling is pending
                if (has_next || above != 0u)
                    stack[((w.lvl + sc.parity) >> 1) * kStackStride] =
                        make_uint4(w.mlo, w.mhi, w.base, ((w.info >> 19) & 7u) | ((w.info >> 16) & 7u) << 3 | next_octant << 6 | (w.info >> 24) << 16);
                const uint32_t slot = ((w.info >> 16) & 7u) * 8u + w.octant;
                const uint4 raw = *reinterpret_cast<const uint4*>(sc.wide + (w.base + wide_rank(w.mlo, w.mhi, slot)));
                VX_TOUCH(sc.touch_nodes, (w.base + wide_rank(w.mlo, w.mhi, slot)) >> 2);
                w.mlo = raw.x; w.mhi = raw.y; w.base = raw.z;
                w.info = (raw.w & 0xffu) << 24 | (raw.w & 0xffu);
            }
            w.ix = (w.ix << 1) | ((w.octant >> 2) & 1u);
            w.iy = (w.iy << 1) | ((w.octant >> 1) & 1u);
            w.iz = (w.iz << 1) | (w.octant & 1u);
            w.lvl++;
        } else {         // voxels.comp:225-234: pop to the nearest level that can still advance
            if (w.has_next_mask == 0u) return kWalkMiss;
            const uint32_t l = 31u - uint32_t(__clz(int(w.has_next_mask)));
            w.has_next_mask &= ~(1u << l);
            const uint32_t up = w.lvl - l;
            const bool in_sub = ((w.lvl + sc.parity) & 1u) != 0u;
            w.ix >>= up; w.iy >>= up; w.iz >>= up;
            w.lvl = l;
            if (in_sub && up == 1u) {   // to the top of the record the lane is in
                popped_octant = (w.info >> 19) & 7u;
                w.info = (w.info & 0xff000000u) | (w.info >> 24);
            } else {
                uint4 raw = stack[((l + sc.parity) >> 1) * kStackStride];
                asm volatile("" : "+v"(raw.x), "+v"(raw.y), "+v"(raw.z), "+v"(raw.w));   // keep it an LDS read (see walk_step)
                w.mlo = raw.x; w.mhi = raw.y; w.base = raw.z;
                const uint32_t top = (raw.w >> 16) & 0xffu;
                if (((l + sc.parity) & 1u) == 0u) {   // a top level
                    popped_octant = raw.w & 7u;
                    w.info = top << 24 | top;
                } else {                              // a sub level
                    const uint32_t s = (raw.w >> 3) & 7u;
                    const uint32_t byte = wide_sub_byte(raw.x, raw.y, s);
                    popped_octant = (raw.w >> 6) & 7u;
                    w.info = top << 24 | (raw.w & 7u) << 19 | s << 16 | (l + 1u == uint32_t(sc.levels) ? byte << 8 : byte);
                }
            }
        }
        const float size = __builtin_ldexpf(sc.root_size, -int(w.lvl));
        w.center = sc.root_min + mk3(float(w.ix) + 0.5f, float(w.iy) + 0.5f, float(w.iz) + 0.5f) * size;
        if (kRegular) {
            if (is_child) {  // voxels.comp:216-221, the slab test by selection (see walkf_step)
                w.en = mk3(far_x ? tm.x : w.en.x, far_y ? tm.y : w.en.y, far_z ? tm.z : w.en.z);
                w.ex = mk3(far_x ? w.ex.x : tm.x, far_y ? w.ex.y : tm.y, far_z ? w.ex.z : tm.z);
                w.octant = octant_of(w.o + w.d * w.time, w.center);
                w.time = vx_max(w.time, vx_max3(w.en.x, w.en.y, w.en.z));
            } else {         // voxels.comp:236-242
                plane_times(w.o, w.inv, w.center, 0.5f * size, w.en, w.ex);
                w.time = w.exit;
                w.octant = popped_octant;
            }
            w.exit = vx_min3(w.ex.x, w.ex.y, w.ex.z);
        } else {
            w.size = size;
            float node_entry, node_exit;
            slab(w.o, w.inv, w.sg, w.center, 0.5f * size, node_entry, node_exit);
            if (is_child) {  // voxels.comp:216-221
                w.octant = octant_of(w.o + w.d * w.time, w.center);
                w.time = vx_max(w.time, node_entry);
            } else {         // voxels.comp:236-242
                w.time = w.exit;
                w.octant = popped_octant;
            }
            w.exit = node_exit;
        }
    } else {  // empty slot, step to the sibling                                         voxels.comp:222-224
        w.octant = next_octant;
        w.time = next_time;
    }
    return kWalkOn;
}

// index of the leaf word the walk stopped at (status kWalkLeaf: the current node is a sub, a leaf parent)
template <bool kRegular>
__device__ __forceinline__ uint32_t walkw_leaf_index(const WalkW<kRegular>& w) {
    return w.base + wide_rank(w.mlo, w.mhi, ((w.info >> 16) & 7u) * 8u + w.octant);
}

// hit resolution: finish_ray with the scene's leaf words (SceneW has no 8-byte records)
__device__ __forceinline__ bool finish_ray_w(const SceneW& sc, int status, f3 o, f3 d, float time, f3 center, uint32_t lvl, uint32_t octant,
                                             uint32_t leaf_index, RayHit& hit) {
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

__device__ __forceinline__ bool cast_ray_w(const SceneW& sc, f3 o, f3 d, float max_distance, uint4* stack, RayHit& hit) {
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int status = kWalkMiss;
    f3 center = splat3(0.0f);
    float time = 0.0f;
    uint32_t lvl = 0, octant = 0, leaf = 0;
    bool entered;
    if (ray_is_regular(inv) && max_distance == kAlmostInfinity) {
        WalkW<true> w;
        entered = walkw_begin(w, sc, o, d, inv);
        if (entered) {
            do { status = walkw_step<true>(w, sc, kAlmostInfinity, stack); } while (status == kWalkOn);
            center = w.center; time = w.time; lvl = w.lvl; octant = w.octant;
            leaf = walkw_leaf_index(w);
        }
    } else {  // a direction component is 0 (or NaN): the shader's text, NaN and all
        WalkW<false> w;
        entered = walkw_begin(w, sc, o, d);
        if (entered) {
            do { status = walkw_step<false>(w, sc, max_distance, stack); } while (status == kWalkOn);
            center = w.center; time = w.time; lvl = w.lvl; octant = w.octant;
            leaf = walkw_leaf_index(w);
        }
    }
    if (!entered) return false;
    return finish_ray_w(sc, status, o, d, time, center, lvl, octant, leaf, hit);
}
