// scene_procedural.cpp — procedural scenes built straight into the device scene format (SVO records + leaf
// words, kernels.h) without ever materialising a voxel list or the reference's 32-byte-per-node octree.
//
// BASELINE.json config 5 / SURVEY.md §8d: a level-7 Menger sponge (3^7 = 2187) clipped to [0, 2048)^3 is
// ~1.05e9 solid voxels; as a voxel list that is 10 GB and as a reference-layout octree > 4 GB with node indices
// close to the int limit of `nodes[8*node + octant]` (voxels.comp:175).  As SVO records it is ~1.2 GB of nodes
// plus 4.2 GB of leaf words.  The tree is the same tree create_octree_nodes (src/context.rs:710-773) would
// build for that voxel list — topology and leaf words are a function of the voxel set — so the traversal
// result is the one the reference's algorithm defines; tests/test_gpu_procedural.py checks that on sizes the
// generic path (voxel list -> octree -> SVO) can still handle.
#include <atomic>
#include <cstring>
#include <thread>

#include "kernels.h"
#include "scene_host.h"

namespace vxrt {

namespace {

uint32_t pow3(uint32_t k) {
    uint32_t p = 1;
    while (k--) p *= 3;
    return p;
}

struct Sponge {
    uint32_t level, clip;
    // does the sponge have a solid voxel inside [lo, hi) (per axis)?  Recursion over the base-3 cells,
    // pruning removed cells (two or more middle digits) and cells outside the box.
    bool any_solid(uint32_t k, uint32_t cx, uint32_t cy, uint32_t cz, const uint32_t lo[3], const uint32_t hi[3]) const {
        const uint32_t size = pow3(k);
        if (cx >= hi[0] || cy >= hi[1] || cz >= hi[2] || cx + size <= lo[0] || cy + size <= lo[1] || cz + size <= lo[2]) return false;
        if (k == 0) return true;
        const uint32_t t = size / 3;
        for (uint32_t i = 0; i < 3; i++)
            for (uint32_t j = 0; j < 3; j++)
                for (uint32_t l = 0; l < 3; l++) {
                    if ((i == 1) + (j == 1) + (l == 1) >= 2) continue;
                    if (any_solid(k - 1, cx + i * t, cy + j * t, cz + l * t, lo, hi)) return true;
                }
        return false;
    }
    bool cube_nonempty(uint32_t x, uint32_t y, uint32_t z, uint32_t side) const {
        const uint32_t lo[3] = {x, y, z};
        const uint32_t hi[3] = {x + side < clip ? x + side : clip, y + side < clip ? y + side : clip, z + side < clip ? z + side : clip};
        if (lo[0] >= hi[0] || lo[1] >= hi[1] || lo[2] >= hi[2]) return false;
        return any_solid(level, 0, 0, 0, lo, hi);
    }
    bool solid(uint32_t x, uint32_t y, uint32_t z) const {
        return x < clip && y < clip && z < clip && x < pow3(level) && y < pow3(level) && z < pow3(level) && menger_solid(level, x, y, z);
    }
};

template <class F>
void parallel_for(size_t n, F f) {
    unsigned threads = std::thread::hardware_concurrency();
    if (threads == 0) threads = 1;
    if (threads > 64) threads = 64;
    if (n < 4096 || threads == 1) { f(0, n); return; }
    std::vector<std::thread> pool;
    const size_t chunk = (n + threads - 1) / threads;
    for (unsigned t = 0; t < threads; t++) {
        const size_t a = size_t(t) * chunk, b = a + chunk < n ? a + chunk : n;
        if (a >= b) break;
        pool.emplace_back([=] { f(a, b); });
    }
    for (auto& th : pool) th.join();
}

}  // namespace

uint32_t procedural_hash(uint32_t x, uint32_t y, uint32_t z) {
    uint32_t h = x * 0x8DA6B343u ^ y * 0xD8163841u ^ z * 0xCB1AB31Fu;
    h ^= h >> 15; h *= 0x2C1B3C6Du;
    h ^= h >> 12; h *= 0x297A2D39u;
    h ^= h >> 15;
    return h;
}

int32_t procedural_leaf_word(uint32_t x, uint32_t y, uint32_t z, const uint8_t mrgb[4], uint32_t emissive_period) {
    uint32_t m = mrgb[0] & 0x7fu;
    if (emissive_period != 0 && procedural_hash(x, y, z) % emissive_period == 0) m |= 0x40u;
    return int32_t(0x80000000u | m << 24 | uint32_t(mrgb[1]) << 16 | uint32_t(mrgb[2]) << 8 | mrgb[3]);
}

// Builds the SVO of the level-`level` sponge clipped to [0, clip)^3, breadth first, children contiguous.
int build_menger_svo(uint32_t level, uint32_t clip, const uint8_t mrgb[4], uint32_t emissive_period,
                     std::vector<SvoRecord>* recs, std::vector<int32_t>* leaves, uint32_t* depth_out) {
    if (level > 9 || clip == 0 || clip > 32768) { set_error("menger: level <= 9 and 1 <= clip <= 32768"); return VXRT_E_INVALID; }
    const Sponge sp{level, clip};
    uint32_t depth = 0;
    while ((1u << depth) < clip) depth++;  // voxel_depth of coordinates 0 .. clip-1 (src/context.rs:813-834)
    *depth_out = depth;
    recs->clear();
    leaves->clear();

    struct Origin { uint16_t x, y, z; };
    // root: centre 0, extent 2^depth; everything lives in its slot 7 (x, y, z >= 0)
    const bool any = sp.cube_nonempty(0, 0, 0, 1u << depth);
    if (depth == 0) {  // a single voxel at the origin: the root's slot 7 is a leaf
        SvoRecord root{any ? 0x80u << 8 : 0u, 0u};
        if (any) leaves->push_back(procedural_leaf_word(0, 0, 0, mrgb, emissive_period));
        recs->push_back(root);
        return VXRT_OK;
    }
    recs->push_back(SvoRecord{any ? 0x80u : 0u, 1u});
    if (!any) return VXRT_OK;

    std::vector<Origin> cur{Origin{0, 0, 0}}, next;
    for (uint32_t side = 1u << depth; side >= 2; side >>= 1) {  // `cur` holds the nodes whose cubes have this side
        const bool terminal = side == 2;
        const uint32_t half = side / 2;
        const size_t n = cur.size();
        const size_t first = recs->size();
        recs->resize(first + n);
        std::vector<uint8_t> masks(n);
        parallel_for(n, [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                const Origin o = cur[i];
                uint32_t m = 0;
                for (uint32_t s = 0; s < 8; s++) {
                    const uint32_t x = o.x + ((s >> 2) & 1u) * half, y = o.y + ((s >> 1) & 1u) * half, z = o.z + (s & 1u) * half;
                    if (terminal ? sp.solid(x, y, z) : sp.cube_nonempty(x, y, z, half)) m |= 1u << s;
                }
                masks[i] = uint8_t(m);
            }
        });
        // bases: children (or leaf words) of node i start after those of nodes 0..i-1
        std::vector<uint64_t> base(n + 1, 0);
        for (size_t i = 0; i < n; i++) base[i + 1] = base[i] + uint64_t(__builtin_popcount(masks[i]));
        const uint64_t total = base[n];
        if (terminal) {
            const size_t lfirst = leaves->size();
            if (lfirst + total >= (uint64_t(1) << 32)) { set_error("menger: too many leaves"); return VXRT_E_SCENE; }
            leaves->resize(lfirst + total);
            parallel_for(n, [&](size_t a, size_t b) {
                for (size_t i = a; i < b; i++) {
                    (*recs)[first + i] = SvoRecord{uint32_t(masks[i]) << 8, uint32_t(lfirst + base[i])};
                    size_t w = lfirst + base[i];
                    for (uint32_t s = 0; s < 8; s++)
                        if (masks[i] >> s & 1u)
                            (*leaves)[w++] = procedural_leaf_word(cur[i].x + ((s >> 2) & 1u), cur[i].y + ((s >> 1) & 1u), cur[i].z + (s & 1u), mrgb, emissive_period);
                }
            });
        } else {
            const uint64_t child_first = first + n;  // the next level is appended right after this one
            if (child_first + total >= (uint64_t(1) << 32)) { set_error("menger: too many nodes"); return VXRT_E_SCENE; }
            next.resize(total);
            parallel_for(n, [&](size_t a, size_t b) {
                for (size_t i = a; i < b; i++) {
                    (*recs)[first + i] = SvoRecord{uint32_t(masks[i]), uint32_t(child_first + base[i])};
                    size_t w = base[i];
                    for (uint32_t s = 0; s < 8; s++)
                        if (masks[i] >> s & 1u)
                            next[w++] = Origin{uint16_t(cur[i].x + ((s >> 2) & 1u) * half), uint16_t(cur[i].y + ((s >> 1) & 1u) * half),
                                               uint16_t(cur[i].z + (s & 1u) * half)};
                }
            });
            cur.swap(next);
        }
    }
    return VXRT_OK;
}

}  // namespace vxrt
