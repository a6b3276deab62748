// ORACLE — test infrastructure only.  Nothing in the product path may include, link or call this.
//
// oprocedural.cpp: BASELINE.json config 5 (a level-7 Menger sponge, 3^7 = 2187, clipped to [0, 2048)^3,
// ~1.05e9 solid voxels, sparse emissive seeds from a hash) for the oracle, at FULL size, without ever storing the scene:
//
//  1. the voxel predicate — solid(x, y, z) and the voxel's leaf word — written from the sponge's definition (a voxel is
//     removed iff, at some base-3 digit position, two or more of its three coordinates have the digit 1);
//  2. LazyTree: the octree buffer create_octree (src/context.rs:710-796) would build for that voxel set, materialised slot by
//     slot the first time the walk of voxels.comp:175 reads `nodes[8 * node + octant]`.  Topology and leaf words are a
//     function of the voxel set alone; node NUMBERS differ from the reference's insertion order, and never reach an output.
//     cast_bounded_ray (oshaders.cpp) runs on it unchanged, so times, normals, leaf words and every shaded value are the
//     ones the restated shader produces for the full 2048^3 scene;
//  3. orc_dda_menger: an INDEPENDENT first-hit finder over the predicate (binary64 Amanatides & Woo over unit cells; no
//     octree, no dense grid, no code shared with the walk) for the result contract of SURVEY.md §0 D1.
//
// Follows no reference file for the scene itself (the reference has no procedural Menger scene; config 5 is BASELINE.json's);
// the leaf word layout is src/context.rs:732-735, the depth rule src/context.rs:813-834, the node geometry :749-753.
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "oracle.h"

namespace orc {

// ---- the voxel predicate -------------------------------------------------------------------------------------------
struct Sponge {
    uint32_t level, clip, side;   // side = min(3^level, clip): voxels live in [0, side)^3
    uint8_t mrgb[4];
    uint32_t emissive_period;
    std::vector<uint16_t> ones;   // ones[c] bit k: base-3 digit k of coordinate c is 1

    Sponge(uint32_t level_, uint32_t clip_, const uint8_t* mrgb_, uint32_t period) : level(level_), clip(clip_), emissive_period(period) {
        uint32_t p = 1;
        for (uint32_t l = 0; l < level; l++) p *= 3;
        side = (clip != 0 && clip < p) ? clip : p;
        memcpy(mrgb, mrgb_, 4);
        ones.resize(side);
        for (uint32_t c = 0; c < side; c++) {
            uint32_t m = 0, v = c;
            for (uint32_t k = 0; k < level; k++) { if (v % 3 == 1) m |= 1u << k; v /= 3; }
            ones[c] = uint16_t(m);
        }
    }
    bool solid(int64_t x, int64_t y, int64_t z) const {
        if (x < 0 || y < 0 || z < 0 || x >= side || y >= side || z >= side) return false;
        const uint32_t a = ones[size_t(x)], b = ones[size_t(y)], c = ones[size_t(z)];
        return ((a & b) | (a & c) | (b & c)) == 0;
    }
    // any solid voxel with lo <= coordinate < hi per axis?  Descends the sponge's own 3x3x3 subdivision: a sub-cube of
    // side 3^k at a kept position is a smaller sponge, which always has solid voxels — so a kept sub-cube that lies
    // wholly inside the box answers yes, and only sub-cubes cut by the box's faces are opened.
    bool any_in_box(const int64_t lo[3], const int64_t hi[3]) const {
        int64_t l[3], h[3];
        for (int a = 0; a < 3; a++) {
            l[a] = lo[a] < 0 ? 0 : lo[a];
            h[a] = hi[a] > int64_t(side) ? int64_t(side) : hi[a];
            if (l[a] >= h[a]) return false;
        }
        uint32_t full = 1;
        for (uint32_t k = 0; k < level; k++) full *= 3;
        return any_rec(full, 0, 0, 0, l, h);
    }
    bool any_rec(int64_t size, int64_t cx, int64_t cy, int64_t cz, const int64_t l[3], const int64_t h[3]) const {
        if (cx >= h[0] || cy >= h[1] || cz >= h[2] || cx + size <= l[0] || cy + size <= l[1] || cz + size <= l[2]) return false;
        if (size == 1) return true;
        if (cx >= l[0] && cy >= l[1] && cz >= l[2] && cx + size <= h[0] && cy + size <= h[1] && cz + size <= h[2]) return true;
        const int64_t t = size / 3;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) {
                    if ((i == 1) + (j == 1) + (k == 1) >= 2) continue;   // removed by the sponge's rule
                    if (any_rec(t, cx + i * t, cy + j * t, cz + k * t, l, h)) return true;
                }
        return false;
    }
    // Specification of the scene's leaf words (the build's own: include/vxrt.h vxrt_set_menger): material bit 0x40 where
    // hash(x, y, z) % emissive_period == 0; word = 1<<31 | (m & 0x7f)<<24 | r<<16 | g<<8 | b   (src/context.rs:732-735)
    int32_t leaf_word(uint32_t x, uint32_t y, uint32_t z) const {
        uint32_t hsh = x * 0x8DA6B343u ^ y * 0xD8163841u ^ z * 0xCB1AB31Fu;
        hsh ^= hsh >> 15; hsh *= 0x2C1B3C6Du;
        hsh ^= hsh >> 12; hsh *= 0x297A2D39u;
        hsh ^= hsh >> 15;
        uint32_t m = mrgb[0] & 0x7fu;
        if (emissive_period != 0 && hsh % emissive_period == 0) m |= 0x40u;
        return int32_t(0x80000000u | m << 24 | uint32_t(mrgb[1]) << 16 | uint32_t(mrgb[2]) << 8 | mrgb[3]);
    }
    // Context::voxel_depth (src/context.rs:813-834) of coordinates 0 .. side-1
    int depth() const {
        int tz = 0;
        for (uint32_t p = 1; p < side; p <<= 1) tz++;   // (max + 1).next_power_of_two().trailing_zeros(); min = 0 contributes 0
        return tz;
    }
};

// ---- the octree buffer, materialised on demand -------------------------------------------------------------------------
// Node geometry of create_octree_nodes (src/context.rs:749-753): a node centred c with extent e covers [c-e, c+e); slot bit
// set <=> coordinate >= c; child centre c -+ e/2, extent e/2; at extent 1 the slots are unit voxels c-1 / c.
struct LazyTree {
    const Sponge* sp;
    struct Node { int32_t slot[8]; int32_t cx, cy, cz, extent; uint8_t known; };
    std::vector<Node> nodes;

    explicit LazyTree(const Sponge* s) : sp(s) {
        Node root{};
        root.extent = 1 << sp->depth();   // src/context.rs:779
        nodes.push_back(root);
    }
    int32_t fetch(int32_t node, uint32_t octant) {
        if (nodes[size_t(node)].known >> octant & 1u) return nodes[size_t(node)].slot[octant];
        const Node n = nodes[size_t(node)];
        const int dx = int(octant >> 2) & 1, dy = int(octant >> 1) & 1, dz = int(octant) & 1;
        int32_t value;
        if (n.extent == 1) {
            const int64_t x = n.cx - 1 + dx, y = n.cy - 1 + dy, z = n.cz - 1 + dz;
            value = sp->solid(x, y, z) ? sp->leaf_word(uint32_t(x), uint32_t(y), uint32_t(z)) : 0;
        } else {
            const int64_t lo[3] = {dx ? n.cx : n.cx - n.extent, dy ? n.cy : n.cy - n.extent, dz ? n.cz : n.cz - n.extent};
            const int64_t hi[3] = {lo[0] + n.extent, lo[1] + n.extent, lo[2] + n.extent};
            if (sp->any_in_box(lo, hi)) {
                Node c{};
                c.extent = n.extent / 2;
                c.cx = n.cx - n.extent / 2 + dx * n.extent;   // src/context.rs:749-753
                c.cy = n.cy - n.extent / 2 + dy * n.extent;
                c.cz = n.cz - n.extent / 2 + dz * n.extent;
                value = int32_t(nodes.size());
                nodes.push_back(c);
            } else {
                value = 0;
            }
        }
        nodes[size_t(node)].slot[octant] = value;
        nodes[size_t(node)].known |= uint8_t(1u << octant);
        return value;
    }
};

struct LazyPool {
    Sponge sp;
    std::mutex m;
    std::vector<LazyTree*> idle;
    LazyPool(uint32_t level, uint32_t clip, const uint8_t* mrgb, uint32_t period) : sp(level, clip, mrgb, period) {}
};

int32_t lazy_fetch(LazyTree* tree, int32_t node, uint32_t octant) { return tree->fetch(node, octant); }

LazyPool* lazy_pool(uint32_t level, uint32_t clip, const uint8_t mrgb[4], uint32_t emissive_period) {
    static std::mutex m;
    static std::map<std::vector<uint32_t>, LazyPool*> pools;
    std::lock_guard<std::mutex> g(m);
    const std::vector<uint32_t> key{level, clip, mrgb[0], mrgb[1], mrgb[2], mrgb[3], emissive_period};
    LazyPool*& p = pools[key];
    if (!p) p = new LazyPool(level, clip, mrgb, emissive_period);
    return p;
}
LazyTree* lazy_acquire(LazyPool* pool) {
    {
        std::lock_guard<std::mutex> g(pool->m);
        if (!pool->idle.empty()) { LazyTree* t = pool->idle.back(); pool->idle.pop_back(); return t; }
    }
    return new LazyTree(&pool->sp);
}
void lazy_release(LazyPool* pool, LazyTree* tree) {
    std::lock_guard<std::mutex> g(pool->m);
    pool->idle.push_back(tree);
}
Scene lazy_scene(LazyTree* tree) {
    // header of create_octree (src/context.rs:782-791): centre 0, root_size = 2^depth
    return Scene{v3s(0.0f), float(1 << tree->sp->depth()), nullptr, tree};
}

}  // namespace orc

using namespace orc;

extern "C" {

int orc_menger_depth(uint32_t level, uint32_t clip) {
    const uint8_t c[4] = {0, 0, 0, 0};
    return Sponge(level, clip, c, 0).depth();
}

// the voxel predicate for caller-given cells: solid flag and leaf word (0 where empty)
void orc_menger_cells(uint32_t level, uint32_t clip, const uint8_t* mrgb, uint32_t emissive_period, const int32_t* cells, size_t n,
                      uint8_t* solid, int32_t* word) {
    const Sponge sp(level, clip, mrgb, emissive_period);
    for (size_t i = 0; i < n; i++) {
        const bool s = sp.solid(cells[3 * i], cells[3 * i + 1], cells[3 * i + 2]);
        solid[i] = s;
        word[i] = s ? sp.leaf_word(uint32_t(cells[3 * i]), uint32_t(cells[3 * i + 1]), uint32_t(cells[3 * i + 2])) : 0;
    }
}

// how many nodes the lazily materialised trees of a scene hold (diagnostic: the walk touches a sliver of the 261 M)
long long orc_menger_lazy_nodes(uint32_t level, uint32_t clip, const uint8_t* mrgb, uint32_t emissive_period) {
    LazyPool* p = lazy_pool(level, clip, mrgb, emissive_period);
    std::lock_guard<std::mutex> g(p->m);
    long long n = 0;
    for (LazyTree* t : p->idle) n += (long long)t->nodes.size();
    return n;
}

// Independent first-hit finder over the PREDICATE (no octree, no grid): Amanatides & Woo in binary64 over unit cells of
// grid space g = 2 * world (a voxel is the world cube [c/2, c/2 + 1/2)^3), clipped to [0, side)^3.
// Per ray: hit flag, world-space t of the entry, entry axis (-1: the origin is inside a solid cell), the cell.
void orc_dda_menger(uint32_t level, uint32_t clip, const float* origins, const float* dirs, size_t n, uint8_t* hit, double* time,
                    int32_t* axis, int32_t* cell, int nthreads) {
    const uint8_t c0[4] = {0, 0, 0, 0};
    const Sponge sp(level, clip, c0, 0);
    const double lim = double(sp.side);
    auto work = [&](size_t r0, size_t r1) {
        for (size_t r = r0; r < r1; r++) {
            double o[3], d[3];
            for (int a = 0; a < 3; a++) { o[a] = 2.0 * double(origins[3 * r + a]); d[a] = 2.0 * double(dirs[3 * r + a]); }
            double t0 = 0.0, t1 = INFINITY;
            int enter_axis = -1;
            bool miss = false;
            for (int a = 0; a < 3; a++) {
                if (d[a] == 0.0) { if (o[a] < 0.0 || o[a] >= lim) miss = true; continue; }
                double ta = (0.0 - o[a]) / d[a], tb = (lim - o[a]) / d[a];
                if (ta > tb) { const double s = ta; ta = tb; tb = s; }
                if (ta > t0) { t0 = ta; enter_axis = a; }
                if (tb < t1) t1 = tb;
            }
            hit[r] = 0; time[r] = 0.0; axis[r] = -1; cell[3 * r] = cell[3 * r + 1] = cell[3 * r + 2] = 0;
            if (miss || t0 >= t1) continue;
            int64_t c[3];
            int step[3];
            double tmax[3], tdelta[3];
            for (int a = 0; a < 3; a++) {
                const double p = o[a] + t0 * d[a];
                int64_t ci = int64_t(std::floor(p));
                if (a == enter_axis) ci = d[a] > 0.0 ? 0 : int64_t(lim) - 1;   // exactly on the entry face
                if (ci < 0) ci = 0;
                if (ci >= int64_t(lim)) ci = int64_t(lim) - 1;
                c[a] = ci;
                step[a] = d[a] > 0.0 ? 1 : -1;
                if (d[a] == 0.0) { tmax[a] = INFINITY; tdelta[a] = INFINITY; }
                else { tmax[a] = (double(ci + (d[a] > 0.0 ? 1 : 0)) - o[a]) / d[a]; tdelta[a] = std::fabs(1.0 / d[a]); }
            }
            double t = t0;
            int ax = enter_axis;
            for (;;) {
                if (sp.solid(c[0], c[1], c[2])) {
                    hit[r] = 1; time[r] = t; axis[r] = ax;
                    cell[3 * r] = int32_t(c[0]); cell[3 * r + 1] = int32_t(c[1]); cell[3 * r + 2] = int32_t(c[2]);
                    break;
                }
                const int a = tmax[0] < tmax[1] ? (tmax[0] < tmax[2] ? 0 : 2) : (tmax[1] < tmax[2] ? 1 : 2);
                t = tmax[a]; ax = a;
                c[a] += step[a];
                if (c[a] < 0 || c[a] >= int64_t(lim)) break;
                tmax[a] += tdelta[a];
            }
        }
    };
    if (nthreads <= 1 || n < 1024) { work(0, n); return; }
    std::vector<std::thread> pool;
    const size_t chunk = (n + size_t(nthreads) - 1) / size_t(nthreads);
    for (int t = 0; t < nthreads; t++) {
        const size_t a = size_t(t) * chunk, b = a + chunk < n ? a + chunk : n;
        if (a >= b) break;
        pool.emplace_back([=] { work(a, b); });
    }
    for (auto& th : pool) th.join();
}

}  // extern "C"
