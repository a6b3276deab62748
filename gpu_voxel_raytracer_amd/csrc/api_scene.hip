// api_scene.hip — scene upload of libvxrt: voxel list -> reference-layout octree (scene_host.cpp) -> breadth-first device records
// (kernels.h: SvoRecord, WideRec) -> HBM.  Replaces Context::recreate_octree (src/context.rs:799-810); the procedural scene of
// BASELINE config 5 is built on the device (scene_device.hip).
#include <cmath>
#include <fstream>
#include <iterator>

#include "ctx.h"

namespace vxrt {
int build_menger_svo(uint32_t level, uint32_t clip, const uint8_t mrgb[4], uint32_t emissive_period, std::vector<SvoRecord>* recs,
                     std::vector<int32_t>* leaves, uint32_t* depth_out);
bool menger_device_build_supported(uint32_t level, uint32_t clip);
int reorder_bottom_treelets(SvoRecord** d_svo, size_t n, const std::vector<size_t>& level_first, uint32_t depth, int levels, hipStream_t stream);
int build_menger_svo_device(uint32_t level, uint32_t clip, const uint8_t mrgb[4], uint32_t emissive_period, hipStream_t stream,
                            SvoRecord** d_svo, size_t* svo_count, int32_t** d_leaves, size_t* leaf_count, uint32_t* depth_out, SvoRecord* root);
}

namespace vxrt {

// reference-layout octree words -> breadth-first SVO records + leaf words (kernels.h)
int flatten_svo(const Octree& tree, std::vector<SvoRecord>* recs, std::vector<int32_t>* leaves) {
    const int32_t* nodes = tree.words.data() + 5;
    std::vector<uint32_t> order{0};
    order.reserve(tree.node_count());
    recs->clear();
    recs->reserve(tree.node_count());
    leaves->clear();
    for (size_t i = 0; i < order.size(); i++) {
        const int32_t* slot = nodes + size_t(8) * order[i];
        uint32_t child_mask = 0, leaf_mask = 0;
        for (int s = 0; s < 8; s++) {
            if (slot[s] > 0) child_mask |= 1u << s;
            else if (slot[s] < 0) leaf_mask |= 1u << s;
        }
        if (child_mask && leaf_mask) { set_error("node mixes children and leaves"); return VXRT_E_SCENE; }
        SvoRecord r;
        r.masks = child_mask | leaf_mask << 8;
        if (leaf_mask) {
            r.base = uint32_t(leaves->size());
            for (int s = 0; s < 8; s++) if (slot[s] < 0) leaves->push_back(slot[s]);
        } else {
            r.base = uint32_t(order.size());
            for (int s = 0; s < 8; s++) if (slot[s] > 0) order.push_back(uint32_t(slot[s]));
        }
        recs->push_back(r);
    }
    return VXRT_OK;
}

// 8-byte records (breadth first, children contiguous) -> wide records (kernels.h: WideRec).  Levels are paired from the bottom; the
// wide records are the nodes of the pairs' upper levels in the same breadth-first order, so a record's grandchildren are contiguous in
// (sub, slot) order, and the leaf words keep their order (the leaf parents under one node are neighbours in breadth-first order).
int widen_svo(const std::vector<SvoRecord>& recs, uint32_t depth, std::vector<WideRec>* out) {
    const uint32_t L = depth + 1;      // node levels: root 0 .. leaf parents L-1
    const uint32_t parity = L & 1u;    // odd: the root is the one sub of a virtual top
    out->clear();
    if (recs.empty()) { out->push_back(WideRec{0, 0, 0, 0}); return VXRT_OK; }
    std::vector<size_t> start(L + 1, recs.size());   // level l = recs[start[l] .. start[l+1])
    start[0] = 0;
    for (uint32_t l = 0; l + 1 < L; l++) {
        if (start[l] >= recs.size() || (recs[start[l]].masks & 0xffu) == 0u) break;   // no nodes below (an empty scene)
        start[l + 1] = recs[start[l]].base;
    }
    std::vector<size_t> woff(L + 2, 0);   // index of the first wide record whose top is level l
    size_t total = parity;
    for (uint32_t l = parity; l + 1 < L; l += 2) { woff[l] = total; total += start[l + 1] - start[l]; }
    if (total >= (size_t(1) << 32)) { set_error("too many nodes"); return VXRT_E_SCENE; }
    out->resize(total == 0 ? 1 : total);
    if (parity) {
        const SvoRecord& root = recs[0];
        const uint32_t byte = L == 1 ? (root.masks >> 8) & 0xffu : root.masks & 0xffu;
        (*out)[0] = WideRec{byte, 0u, L == 1 ? root.base : uint32_t(woff[1]), byte != 0u ? 1u : 0u};
    }
    for (uint32_t l = parity; l + 1 < L; l += 2) {
        const bool subs_are_leaf_parents = l + 2 == L;
        for (size_t i = start[l]; i < start[l + 1]; i++) {
            const SvoRecord& top = recs[i];
            const uint32_t cm = top.masks & 0xffu;
            uint64_t mask = 0;
            uint32_t k = 0;
            for (uint32_t s = 0; s < 8; s++)
                if (cm >> s & 1u) {
                    const SvoRecord& sub = recs[size_t(top.base) + k++];
                    mask |= uint64_t(subs_are_leaf_parents ? (sub.masks >> 8) & 0xffu : sub.masks & 0xffu) << (8u * s);
                }
            uint32_t base = 0;
            if (cm != 0u) {
                const SvoRecord& first = recs[top.base];
                base = subs_are_leaf_parents ? first.base : uint32_t(woff[l + 2] + (size_t(first.base) - start[l + 2]));
            }
            (*out)[woff[l] + (i - start[l])] = WideRec{uint32_t(mask), uint32_t(mask >> 32), base, cm};
        }
    }
    return VXRT_OK;
}

// The sky cull's box (kernels.h: TraceArgs::cull): the smallest box of level-L cells, L = min(depth, 7), that holds every occupied
// cell of that level — node levels run 0 (the root, edge root_size) .. depth (the leaf parents, edge 1).  The tree's records are
// breadth first with a node's children contiguous, so levels 0 .. L - 1 are a prefix of `recs`.
bool scene_box(const SvoRecord* recs, size_t count, uint32_t depth, const float root_center[3], float root_size, float box_min[3], float box_max[3]) {
    if (count == 0 || (recs[0].masks & 0xffffu) == 0u) return false;
    const uint32_t L = depth < 7u ? depth : 7u;
    struct Cell { uint32_t index, x, y, z; };
    std::vector<Cell> level{{0u, 0u, 0u, 0u}}, next;
    for (uint32_t l = 0; l < L; l++) {
        next.clear();
        for (const Cell& n : level) {
            if (n.index >= count) return false;               // the prefix handed in is too short: no box, no cull
            const uint32_t cm = recs[n.index].masks & 0xffu;
            uint32_t k = 0;
            for (uint32_t s = 0; s < 8; s++)
                if (cm >> s & 1u) next.push_back(Cell{recs[n.index].base + k++, n.x * 2 + ((s >> 2) & 1u), n.y * 2 + ((s >> 1) & 1u), n.z * 2 + (s & 1u)});
        }
        level.swap(next);
        if (level.empty()) return false;
    }
    uint32_t lo[3] = {~0u, ~0u, ~0u}, hi[3] = {0, 0, 0};
    for (const Cell& n : level) {
        const uint32_t p[3] = {n.x, n.y, n.z};
        for (int a = 0; a < 3; a++) { lo[a] = p[a] < lo[a] ? p[a] : lo[a]; hi[a] = p[a] > hi[a] ? p[a] : hi[a]; }
    }
    const float cell = ldexpf(root_size, -int(L));             // exact: powers of two
    for (int a = 0; a < 3; a++) {
        const float root_min = root_center[a] - 0.5f * root_size;
        box_min[a] = root_min + float(lo[a]) * cell;
        box_max[a] = root_min + float(hi[a] + 1u) * cell;
    }
    return true;
}

// VXRT_OPT_NODE_ORDER 2 / 3: the last two / three node levels of the (breadth-first) records just built become depth-first treelets
// (scene_device.hip: reorder_bottom_treelets); trees of depth >= 4.  (The sky cull's box was made from the breadth-first records
// before this runs.)  The level starts are found by following the first node of every level
// (in a breadth-first array the first child of a level's first node opens the next level; every inner node has a child).
int apply_node_order(vxrt_ctx* c) {
    c->node_order_applied = 0;
    if ((c->node_order != 2 && c->node_order != 3) || c->depth < 4u || c->d_svo == nullptr || c->svo_count < 2) return VXRT_OK;
    std::vector<size_t> first(size_t(c->depth) + 1, 0);
    SvoRecord r = c->root_rec;
    for (uint32_t l = 1; l <= c->depth; l++) {
        if ((r.masks & 0xffu) == 0u || r.base >= c->svo_count) return VXRT_OK;      // no such level (an empty scene): nothing to reorder
        first[l] = r.base;
        HIP_TRY(hipMemcpy(&r, c->d_svo + r.base, sizeof r, hipMemcpyDeviceToHost));
    }
    if (int rc = reorder_bottom_treelets(&c->d_svo, c->svo_count, first, c->depth, c->node_order, c->stream)) return rc;
    c->node_order_applied = c->node_order;
    return VXRT_OK;
}

int upload_svo(vxrt_ctx* c, std::vector<SvoRecord>& recs, std::vector<int32_t>& leaves, uint32_t depth) {
    if (leaves.empty()) leaves.push_back(0);
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = sync_all(c)) return rc;
    // The wide records are built only when asked for (VXRT_OPT_SCENE_FORMAT 1 before the scene is set): measured on MI355X the walk
    // over them is slower than the walk over the 8-byte records on every scene tried, cache-resident or not, so the
    // default never uses them and does not pay for their memory.
    std::vector<WideRec> wide;
    if (c->scene_format == 1) { if (int rc = widen_svo(recs, depth, &wide)) return rc; }
    if (c->d_svo) (void)hipFree(c->d_svo);
    if (c->d_leaves) (void)hipFree(c->d_leaves);
    if (c->d_wide) (void)hipFree(c->d_wide);
    drop_touch_maps(c);
    c->d_svo = nullptr;
    c->d_leaves = nullptr;
    c->d_wide = nullptr;
    c->wide_count = wide.size();
    c->wide_root = WideRec{0, 0, 0, 0};
    if (!wide.empty()) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_wide), wide.size() * sizeof(WideRec)));
        HIP_TRY(hipMemcpy(c->d_wide, wide.data(), wide.size() * sizeof(WideRec), hipMemcpyHostToDevice));
        c->wide_root = wide[0];
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_svo), recs.size() * sizeof(SvoRecord)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_leaves), leaves.size() * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(c->d_svo, recs.data(), recs.size() * sizeof(SvoRecord), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_leaves, leaves.data(), leaves.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    c->svo_count = recs.size();
    c->root_rec = recs.empty() ? SvoRecord{0, 0} : recs[0];
    c->leaf_count = leaves.size();
    c->root_center[0] = c->root_center[1] = c->root_center[2] = 0.0f;  // src/context.rs:782-786
    c->root_size = float(1u << depth);                                   // src/context.rs:779
    c->depth = depth;
    c->has_scene = true;
    c->box_valid = scene_box(recs.data(), recs.size(), depth, c->root_center, c->root_size, c->box_min, c->box_max);
    return c->scene_format == 1 ? VXRT_OK : apply_node_order(c);
}

int upload_scene(vxrt_ctx* c, const Voxel* voxels, size_t n) {
    Octree tree;
    if (int rc = build_octree(voxels, n, &tree)) return rc;
    std::vector<SvoRecord> recs;
    std::vector<int32_t> leaves;
    if (int rc = flatten_svo(tree, &recs, &leaves)) return rc;
    return upload_svo(c, recs, leaves, tree.depth);
}

// Which records trace_kernel / bounce_kernel walk: the wide ones only when VXRT_OPT_SCENE_FORMAT 1 (or VXRT_WIDE=1) asked for them
// before the scene was set.  (They halve the dependent loads of a descent, but the walk over them executes ~35 % more instructions
// per trip, and the stage is bound by instruction issue, not by those loads — measured: menger 1080p 21.0 vs 27.0 Gray/s, the
// 2048^3 scene 2.68 vs 2.22 ms per 4K frame.)
bool use_wide(const vxrt_ctx* c) { return c->d_wide != nullptr && c->scene_format == 1; }

}  // namespace vxrt

extern "C" {

int vxrt_set_voxels(vxrt_ctx* c, const int16_t (*pos)[3], const uint8_t (*mrgb)[4], size_t n) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    if (n != 0 && (!pos || !mrgb)) { set_error("null voxel arrays"); return VXRT_E_INVALID; }
    std::vector<Voxel> v(n);
    for (size_t i = 0; i < n; i++) {
        v[i].x = pos[i][0]; v[i].y = pos[i][1]; v[i].z = pos[i][2];
        v[i].m = mrgb[i][0]; v[i].r = mrgb[i][1]; v[i].g = mrgb[i][2]; v[i].b = mrgb[i][3];
    }
    return upload_scene(c, v.data(), n);
} VXRT_CATCH

int vxrt_load_vox_memory(vxrt_ctx* c, const uint8_t* bytes, size_t len) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    if (!bytes) { set_error("null bytes"); return VXRT_E_INVALID; }
    VoxScene scene;
    if (int rc = decode_vox(bytes, len, &scene)) return rc;
    return upload_scene(c, scene.voxels.data(), scene.voxels.size());
} VXRT_CATCH

int vxrt_load_vox(vxrt_ctx* c, const char* path) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    if (!path) { set_error("null path"); return VXRT_E_INVALID; }
    std::ifstream f(path, std::ios::binary);
    if (!f) { set_error(std::string("failed to read file: ") + path); return VXRT_E_IO; }
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return vxrt_load_vox_memory(c, bytes.data(), bytes.size());
} VXRT_CATCH

int vxrt_set_menger(vxrt_ctx* c, uint32_t level, uint32_t clip, const uint8_t mrgb[4], uint32_t emissive_period) try {
    if (!valid_ctx(c)) return VXRT_E_INVALID;
    if (!mrgb) { set_error("null colour"); return VXRT_E_INVALID; }
    uint32_t side = 1;
    for (uint32_t l = 0; l < level && l < 10; l++) side *= 3;
    if (clip == 0 || clip > side) clip = side;
    uint32_t depth = 0;
    // On the device when the scene fits its dense sweep (side <= 2048: 1.15 GB of scratch) and only the 8-byte records are wanted:
    // the 2048^3 scene in a fraction of a second instead of ~9 s of host threads + a 5.6 GB upload.  VXRT_HOST_BUILD=1: the host builder.
    if (menger_device_build_supported(level, clip) && c->scene_format != 1 && c->host_scene_build == 0) {
        HIP_TRY(hipSetDevice(c->cfg.device));
        if (int rc = sync_all(c)) return rc;
        SvoRecord* svo = nullptr;
        int32_t* lw = nullptr;
        size_t nsvo = 0, nlw = 0;
        SvoRecord root{0, 0};
        if (int rc = build_menger_svo_device(level, clip, mrgb, emissive_period, c->stream, &svo, &nsvo, &lw, &nlw, &depth, &root)) return rc;
        if (c->d_svo) (void)hipFree(c->d_svo);
        if (c->d_leaves) (void)hipFree(c->d_leaves);
        if (c->d_wide) (void)hipFree(c->d_wide);
        drop_touch_maps(c);
        c->d_svo = svo; c->d_leaves = lw; c->d_wide = nullptr;
        c->svo_count = nsvo; c->leaf_count = nlw; c->wide_count = 0;
        c->root_rec = root;
        c->wide_root = WideRec{0, 0, 0, 0};
        c->root_center[0] = c->root_center[1] = c->root_center[2] = 0.0f;
        c->root_size = float(1u << depth);
        c->depth = depth;
        c->has_scene = true;
        // the sky cull's box from the top of the tree: levels 0 .. 6 are a prefix of the records (find where level 7 starts, copy that much)
        c->box_valid = false;
        size_t prefix = 1;
        SvoRecord first = root;
        for (uint32_t l = 0; l < (depth < 7u ? depth : 7u) && (first.masks & 0xffu) != 0u; l++) {
            prefix = first.base;
            if (prefix >= nsvo) break;
            HIP_TRY(hipMemcpy(&first, svo + prefix, sizeof first, hipMemcpyDeviceToHost));
        }
        if (prefix <= nsvo && prefix <= (size_t(1) << 22)) {
            std::vector<SvoRecord> top(prefix);
            HIP_TRY(hipMemcpy(top.data(), svo, prefix * sizeof(SvoRecord), hipMemcpyDeviceToHost));
            c->box_valid = scene_box(top.data(), top.size(), depth, c->root_center, c->root_size, c->box_min, c->box_max);
        }
        return apply_node_order(c);
    }
    std::vector<SvoRecord> recs;
    std::vector<int32_t> leaves;
    if (int rc = build_menger_svo(level, clip, mrgb, emissive_period, &recs, &leaves, &depth)) return rc;
    return upload_svo(c, recs, leaves, depth);
} VXRT_CATCH

// Test hook: the scene as the device holds it (8-byte records: 2 words each; leaf words).  Null arrays: sizes only.
int vxrt_debug_read_scene(vxrt_ctx* c, uint32_t* svo, size_t svo_cap, size_t* n_svo, int32_t* leaves, size_t leaf_cap, size_t* n_leaves) try {
    if (!valid_ctx(c) || !n_svo || !n_leaves) { set_error("null argument"); return VXRT_E_INVALID; }
    if (!c->has_scene) { set_error("no scene set"); return VXRT_E_NOSCENE; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = sync_all(c)) return rc;
    *n_svo = c->svo_count;
    *n_leaves = c->leaf_count;
    if (svo && svo_cap >= c->svo_count) HIP_TRY(hipMemcpy(svo, c->d_svo, c->svo_count * sizeof(SvoRecord), hipMemcpyDeviceToHost));
    if (leaves && leaf_cap >= c->leaf_count) HIP_TRY(hipMemcpy(leaves, c->d_leaves, c->leaf_count * sizeof(int32_t), hipMemcpyDeviceToHost));
    return VXRT_OK;
} VXRT_CATCH

}  // extern "C"
