// scene_device.hip — the procedural scene of BASELINE config 5 (level-L Menger sponge clipped to [0, clip)^3, sparse emissive
// seeds) built ON THE DEVICE, straight into the scene format the tracer reads (kernels.h: 8-byte SVO records + leaf words).
//
// The host builder (scene_procedural.cpp) needs ~9 s for the 2048^3 scene (261 M nodes, 1.05e9 leaf words) plus a 5.6 GB upload;
// here the same records — byte for byte (tests/test_gpu_procedural.py) — are made in two sweeps that are pure streaming work:
//   1. bottom-up, DENSE: for every 2x2x2 cell of the clip box the 8-bit leaf mask from the voxel predicate (8 table look-ups per
//      cell), then level by level the 8-bit "child is not empty" mask of every potential node from its eight children's masks:
//      (side/2)^3 + (side/4)^3 + ... bytes = 1.15 GB for side 2048, each byte written once and read once;
//   2. top-down, SPARSE: the nodes of a level in breadth-first order (a list of packed coordinates) look their masks up in the
//      dense arrays, an exclusive scan of the masks' popcounts gives every node the index of its first child (two-level scan:
//      block sums, one block over the sums, block-local scan), and the node writes its record, its children's coordinates for the
//      next level or — at the last level — its leaf words (src/context.rs:732-735 layout, emissive bit from the scene's hash).
// The level sizes are known after sweep 1 (non-zero bytes per dense level), so the final buffers are allocated exactly once.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "kernels.h"
#include "scene_host.h"

namespace vxrt {
uint32_t procedural_hash(uint32_t x, uint32_t y, uint32_t z);   // scene_procedural.cpp: the scene's specification (host side)

namespace {

constexpr int kScanBlock = 256, kScanItems = 8, kScanTile = kScanBlock * kScanItems;   // elements per block of the scan

struct SpongeDev {
    const uint16_t* ones;   // ones[c] bit k: base-3 digit k of coordinate c is 1  (c < side)
    uint32_t side;          // voxels live in [0, side)^3, side = min(3^level, clip)
    uint32_t m, r, g, b, emissive_period;
};

__device__ __forceinline__ bool sponge_solid(const SpongeDev& s, uint32_t x, uint32_t y, uint32_t z) {
    if (x >= s.side || y >= s.side || z >= s.side) return false;
    const uint32_t a = s.ones[x], b = s.ones[y], c = s.ones[z];
    return ((a & b) | (a & c) | (b & c)) == 0u;   // removed iff two coordinates share a digit position holding a 1
}

__device__ __forceinline__ int32_t sponge_leaf_word(const SpongeDev& s, uint32_t x, uint32_t y, uint32_t z) {
    uint32_t h = x * 0x8DA6B343u ^ y * 0xD8163841u ^ z * 0xCB1AB31Fu;   // = procedural_hash (scene_procedural.cpp)
    h ^= h >> 15; h *= 0x2C1B3C6Du;
    h ^= h >> 12; h *= 0x297A2D39u;
    h ^= h >> 15;
    uint32_t m = s.m & 0x7fu;
    if (s.emissive_period != 0u && h % s.emissive_period == 0u) m |= 0x40u;
    return int32_t(0x80000000u | m << 24 | s.r << 16 | s.g << 8 | s.b);
}

// slot s of a node: x bit = s>>2, y bit = s>>1, z bit = s  (src/context.rs:726-729)
// dense level of n^3 potential nodes with cubes of side 2: leaf masks from the predicate
__global__ __launch_bounds__(256) void dense_leaf_masks_kernel(SpongeDev sp, uint32_t n, uint8_t* out) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= size_t(n) * n * n) return;
    const uint32_t z = uint32_t(i % n), y = uint32_t((i / n) % n), x = uint32_t(i / (size_t(n) * n));
    uint32_t m = 0;
#pragma unroll
    for (uint32_t s = 0; s < 8; s++)
        if (sponge_solid(sp, 2 * x + ((s >> 2) & 1u), 2 * y + ((s >> 1) & 1u), 2 * z + (s & 1u))) m |= 1u << s;
    out[i] = uint8_t(m);
}

// dense level of n^3 potential nodes from the level below (2n)^3: bit s <=> child s has any occupied slot
__global__ __launch_bounds__(256) void dense_parent_masks_kernel(const uint8_t* child, uint32_t n, uint8_t* out) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= size_t(n) * n * n) return;
    const uint32_t z = uint32_t(i % n), y = uint32_t((i / n) % n), x = uint32_t(i / (size_t(n) * n));
    const uint32_t n2 = 2 * n;
    uint32_t m = 0;
#pragma unroll
    for (uint32_t s = 0; s < 8; s++) {
        const size_t c = (size_t(2 * x + ((s >> 2) & 1u)) * n2 + (2 * y + ((s >> 1) & 1u))) * n2 + (2 * z + (s & 1u));
        if (child[c] != 0) m |= 1u << s;
    }
    out[i] = uint8_t(m);
}

// non-empty entries of a dense level, and the sum of their popcounts -> totals[0], totals[1]
__global__ __launch_bounds__(256) void dense_count_kernel(const uint8_t* masks, size_t n, unsigned long long* totals) {
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    unsigned nodes = 0, bits = 0;
    for (; i < n; i += size_t(gridDim.x) * 256) {
        const unsigned m = masks[i];
        nodes += m != 0u;
        bits += unsigned(__popc(m));
    }
    for (int off = 32; off > 0; off >>= 1) { nodes += __shfl_down(nodes, off, 64); bits += __shfl_down(bits, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(totals, (unsigned long long)nodes); atomicAdd(totals + 1, (unsigned long long)bits); }
}

__device__ __forceinline__ uint32_t node_mask(const uint8_t* dense, uint32_t n, uint32_t coord) {
    const uint32_t x = coord & 0x3ffu, y = (coord >> 10) & 0x3ffu, z = (coord >> 20) & 0x3ffu;
    return dense[(size_t(x) * n + y) * n + z];
}

// scan, phase 1: children per tile of kScanTile nodes
__global__ __launch_bounds__(kScanBlock) void tile_sums_kernel(const uint32_t* coords, size_t count, const uint8_t* dense, uint32_t n, uint32_t* sums) {
    __shared__ unsigned wave_sum[kScanBlock / 64];
    const size_t base = size_t(blockIdx.x) * kScanTile;
    unsigned v = 0;
    for (int k = 0; k < kScanItems; k++) {
        const size_t i = base + size_t(k) * kScanBlock + threadIdx.x;
        if (i < count) v += unsigned(__popc(node_mask(dense, n, coords[i])));
    }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

// scan, phase 2: exclusive scan of the tile sums in place, one block
__global__ __launch_bounds__(1024) void scan_sums_kernel(uint32_t* sums, size_t tiles) {
    __shared__ unsigned part[1024];
    __shared__ unsigned carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (size_t start = 0; start < tiles; start += 1024) {
        const size_t i = start + threadIdx.x;
        const unsigned v = i < tiles ? sums[i] : 0u;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan in LDS
            const unsigned t = int(threadIdx.x) >= off ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < tiles) sums[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
}

// scan, phase 3 + emit: every node's record, and its children's coordinates (inner level) or leaf words (last level).
//   recs[first + i] = {mask, child_first + rank}  (inner)   or   {mask << 8, leaf_first + rank}  (last level: leaf parents)
// Thread t of a block owns kScanItems CONSECUTIVE nodes, so that ranks follow the breadth-first order.
__global__ __launch_bounds__(kScanBlock) void emit_level_kernel(const uint32_t* coords, size_t count, const uint8_t* dense, uint32_t n,
                                                               const uint32_t* tile_offsets, SvoRecord* recs, uint32_t child_first,
                                                               uint32_t* next_coords, int32_t* leaves, SpongeDev sp, int last_level) {
    __shared__ unsigned scan[kScanBlock];
    const size_t base = size_t(blockIdx.x) * kScanTile + size_t(threadIdx.x) * kScanItems;
    unsigned masks[kScanItems], mine = 0;
    for (int k = 0; k < kScanItems; k++) {
        masks[k] = base + k < count ? node_mask(dense, n, coords[base + k]) : 0u;
        mine += unsigned(__popc(masks[k]));
    }
    scan[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {
        const unsigned t = int(threadIdx.x) >= off ? scan[threadIdx.x - off] : 0u;
        __syncthreads();
        scan[threadIdx.x] += t;
        __syncthreads();
    }
    unsigned rank = tile_offsets[blockIdx.x] + scan[threadIdx.x] - mine;   // children before this thread's first node
    for (int k = 0; k < kScanItems; k++) {
        if (base + k >= count) break;
        const uint32_t c = coords[base + k];
        const uint32_t x = c & 0x3ffu, y = (c >> 10) & 0x3ffu, z = (c >> 20) & 0x3ffu;
        const unsigned m = masks[k];
        recs[base + k] = last_level ? SvoRecord{m << 8, child_first + rank} : SvoRecord{m, child_first + rank};
        for (uint32_t s = 0; s < 8; s++)
            if (m >> s & 1u) {
                const uint32_t cx = 2 * x + ((s >> 2) & 1u), cy = 2 * y + ((s >> 1) & 1u), cz = 2 * z + (s & 1u);
                if (last_level) leaves[rank] = sponge_leaf_word(sp, cx, cy, cz);
                else next_coords[rank] = cx | cy << 10 | cz << 20;
                rank++;
            }
    }
}

// ---- node order (VXRT_OPT_NODE_ORDER): depth-first TREELETS at the bottom of the tree ---------------------------------------------
// The builders emit the records breadth-first: level after level, each level in the order of its parents (a Morton order), the
// children of a node contiguous.  A descent through the last node levels then reads one far-apart array per level.  Node indices
// never reach an output and the walk only ever forms `base + popcount(...)`, so the order is free as long as THE CHILDREN OF A NODE
// STAY CONTIGUOUS.  A treelet therefore hangs below a node p of level depth - K (K = 2 or 3; p itself stays where it is, only its
// base changes): p's children as one block, then, child by child, that child's children as a block followed by their children's
// blocks — K levels, <= 8 + 64 (+ 512) records, all of p's descendants in one run of memory:
//   old: [.. level depth-K+1 ..] ... [.. level depth ..]                 one array per level (first[l]: where level l starts)
//   new: [treelet of p_0][treelet of p_1] ...                              in the same range [first[depth-K+1], n), p in level order
// Where p's treelet starts: every earlier p's descendants come before it, level by level — the index of p's first descendant within
// each level, read off the old bases (they are exclusive scans already).  The leaf words keep their order.
template <int K>
__global__ __launch_bounds__(256) void treelet_kernel(const SvoRecord* old, SvoRecord* out, uint32_t fp, uint32_t fa, uint32_t fb, uint32_t fc) {
    // K = 3: p in [fp, fa), levels A = [fa, fb), B = [fb, fc), C = [fc, n).   K = 2: p in [fa, fb) (passed as fp = fa), levels B, C.
    const uint32_t pi = fp + blockIdx.x * 256u + threadIdx.x;
    if (pi >= fa && K == 3) return;
    if (K == 2 && pi >= fb) return;
    const SvoRecord p = old[pi];
    if (K == 3) {
        const uint32_t a0 = p.base - fa;                                  // p's first child within level A
        const uint32_t b0 = old[fa + a0].base - fb;                       // ... its first descendant within level B
        const uint32_t c0 = old[fb + b0].base - fc;                       // ... and within level C
        const uint32_t pos0 = fa + a0 + b0 + c0;
        const uint32_t na = uint32_t(__popc(p.masks & 0xffu));
        out[pi] = SvoRecord{p.masks, pos0};
        uint32_t cursor = pos0 + na;
        for (uint32_t j = 0; j < na; j++) {
            const SvoRecord ra = old[fa + a0 + j];
            const uint32_t nb = uint32_t(__popc(ra.masks & 0xffu));
            out[pos0 + j] = SvoRecord{ra.masks, cursor};
            const uint32_t bpos = cursor;
            cursor += nb;
            for (uint32_t i = 0; i < nb; i++) {
                const SvoRecord rb = old[ra.base + i];
                const uint32_t nc = uint32_t(__popc(rb.masks & 0xffu));
                out[bpos + i] = SvoRecord{rb.masks, cursor};
                for (uint32_t c = 0; c < nc; c++) out[cursor + c] = old[rb.base + c];   // leaf parents: {leaf mask << 8, first leaf word}
                cursor += nc;
            }
        }
    } else {
        const uint32_t b0 = p.base - fb;
        const uint32_t c0 = old[fb + b0].base - fc;
        const uint32_t pos0 = fb + b0 + c0;
        const uint32_t nb = uint32_t(__popc(p.masks & 0xffu));
        out[pi] = SvoRecord{p.masks, pos0};
        uint32_t cursor = pos0 + nb;
        for (uint32_t i = 0; i < nb; i++) {
            const SvoRecord rb = old[fb + b0 + i];
            const uint32_t nc = uint32_t(__popc(rb.masks & 0xffu));
            out[pos0 + i] = SvoRecord{rb.masks, cursor};
            for (uint32_t c = 0; c < nc; c++) out[cursor + c] = old[rb.base + c];
            cursor += nc;
        }
    }
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

#define DEV_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) { set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); return VXRT_E_DEVICE; } \
    } while (0)

}  // namespace

// Can the device builder make this scene?  (dense sweep: (side/2)^3 bytes; packed 10-bit node coordinates)
bool menger_device_build_supported(uint32_t level, uint32_t clip) {
    uint32_t side = 1;
    for (uint32_t l = 0; l < level && l < 10; l++) side *= 3;
    if (clip != 0 && clip < side) side = clip;
    return level <= 9 && side >= 4 && side <= 2048;
}

// Builds the scene into freshly allocated device buffers (the caller owns them).  root: svo[0].
int build_menger_svo_device(uint32_t level, uint32_t clip, const uint8_t mrgb[4], uint32_t emissive_period, hipStream_t stream,
                            SvoRecord** d_svo, size_t* svo_count, int32_t** d_leaves, size_t* leaf_count, uint32_t* depth_out, SvoRecord* root) {
    uint32_t side = 1;
    for (uint32_t l = 0; l < level; l++) side *= 3;
    if (clip != 0 && clip < side) side = clip;
    uint32_t depth = 0;
    while ((1u << depth) < side) depth++;            // voxel_depth of coordinates 0 .. side-1 (src/context.rs:813-834)
    *depth_out = depth;
    // base-3 digit table of the coordinates
    std::vector<uint16_t> ones(side);
    for (uint32_t c = 0; c < side; c++) {
        uint32_t m = 0, v = c;
        for (uint32_t k = 0; k < level; k++) { if (v % 3 == 1) m |= 1u << k; v /= 3; }
        ones[c] = uint16_t(m);
    }
    DevBuf d_ones;
    DEV_TRY(d_ones.alloc(ones.size() * sizeof(uint16_t)));
    DEV_TRY(hipMemcpyAsync(d_ones.p, ones.data(), ones.size() * sizeof(uint16_t), hipMemcpyHostToDevice, stream));
    const SpongeDev sp{d_ones.as<uint16_t>(), side, mrgb[0], mrgb[1], mrgb[2], mrgb[3], emissive_period};

    // ---- sweep 1: dense masks, bottom-up.  Tree level l (1 .. depth) has cubes of side 2^(depth+1-l) and (2^(l-1))^3 potential nodes
    std::vector<DevBuf> dense(depth + 1);
    std::vector<uint32_t> dim(depth + 1, 0);
    for (uint32_t l = depth; l >= 1; l--) {
        dim[l] = 1u << (l - 1);
        const size_t n3 = size_t(dim[l]) * dim[l] * dim[l];
        DEV_TRY(dense[l].alloc(n3));
        const unsigned blocks = unsigned((n3 + 255) / 256);
        if (l == depth)
            hipLaunchKernelGGL(dense_leaf_masks_kernel, dim3(blocks), dim3(256), 0, stream, sp, dim[l], dense[l].as<uint8_t>());
        else
            hipLaunchKernelGGL(dense_parent_masks_kernel, dim3(blocks), dim3(256), 0, stream, dense[l + 1].as<uint8_t>(), dim[l], dense[l].as<uint8_t>());
        DEV_TRY(hipGetLastError());
    }
    // level sizes: nodes of level l = non-empty entries of dense level l; leaf words = set bits of the last level
    DevBuf d_totals;
    DEV_TRY(d_totals.alloc((depth + 1) * 2 * sizeof(unsigned long long)));
    DEV_TRY(hipMemsetAsync(d_totals.p, 0, (depth + 1) * 2 * sizeof(unsigned long long), stream));
    for (uint32_t l = 1; l <= depth; l++) {
        const size_t n3 = size_t(dim[l]) * dim[l] * dim[l];
        const unsigned blocks = unsigned(n3 / 256 < 1 ? 1 : (n3 / 256 > 4096 ? 4096 : n3 / 256));
        hipLaunchKernelGGL(dense_count_kernel, dim3(blocks), dim3(256), 0, stream, dense[l].as<uint8_t>(), n3, d_totals.as<unsigned long long>() + 2 * l);
    }
    std::vector<unsigned long long> totals((depth + 1) * 2);
    DEV_TRY(hipMemcpyAsync(totals.data(), d_totals.p, totals.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    DEV_TRY(hipStreamSynchronize(stream));
    size_t nodes = 1, max_level = 1;
    for (uint32_t l = 1; l <= depth; l++) { nodes += size_t(totals[2 * l]); max_level = size_t(totals[2 * l]) > max_level ? size_t(totals[2 * l]) : max_level; }
    const size_t nleaves = size_t(totals[2 * depth + 1]);
    const bool any = totals[2] != 0;
    if (nodes >= (size_t(1) << 32) || nleaves >= (size_t(1) << 32)) { set_error("menger: too many nodes"); return VXRT_E_SCENE; }

    DevBuf svo, leaves, coords_a, coords_b, tile_sums;
    DEV_TRY(svo.alloc(nodes * sizeof(SvoRecord)));
    DEV_TRY(leaves.alloc((nleaves ? nleaves : 1) * sizeof(int32_t)));
    DEV_TRY(coords_a.alloc(max_level * sizeof(uint32_t)));
    DEV_TRY(coords_b.alloc(max_level * sizeof(uint32_t)));
    DEV_TRY(tile_sums.alloc(((max_level + kScanTile - 1) / kScanTile + 1) * sizeof(uint32_t)));
    // root: centre 0, everything lives in its slot 7 (x, y, z >= 0); its one child is record 1
    *root = SvoRecord{any ? 0x80u : 0u, 1u};
    DEV_TRY(hipMemcpyAsync(svo.p, root, sizeof(SvoRecord), hipMemcpyHostToDevice, stream));
    if (nleaves == 0) DEV_TRY(hipMemsetAsync(leaves.p, 0, sizeof(int32_t), stream));

    // ---- sweep 2: top-down, level by level in breadth-first order
    if (any) {
        const uint32_t zero = 0;
        DEV_TRY(hipMemcpyAsync(coords_a.p, &zero, sizeof zero, hipMemcpyHostToDevice, stream));   // level 1: the node at (0, 0, 0)
        uint32_t* cur = coords_a.as<uint32_t>();
        uint32_t* next = coords_b.as<uint32_t>();
        size_t first = 1;
        for (uint32_t l = 1; l <= depth; l++) {
            const size_t count = size_t(totals[2 * l]);
            const unsigned tiles = unsigned((count + kScanTile - 1) / kScanTile);
            const int last = l == depth;
            const uint32_t child_first = last ? 0u : uint32_t(first + count);   // the next level follows this one; leaf words start at 0
            hipLaunchKernelGGL(tile_sums_kernel, dim3(tiles), dim3(kScanBlock), 0, stream, cur, count, dense[l].as<uint8_t>(), dim[l], tile_sums.as<uint32_t>());
            hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, stream, tile_sums.as<uint32_t>(), size_t(tiles));
            hipLaunchKernelGGL(emit_level_kernel, dim3(tiles), dim3(kScanBlock), 0, stream, cur, count, dense[l].as<uint8_t>(), dim[l],
                               tile_sums.as<uint32_t>(), svo.as<SvoRecord>() + first, child_first, next, leaves.as<int32_t>(), sp, last);
            DEV_TRY(hipGetLastError());
            first += count;
            uint32_t* t = cur; cur = next; next = t;
        }
    }
    DEV_TRY(hipStreamSynchronize(stream));
    *d_svo = svo.as<SvoRecord>(); svo.p = nullptr;
    *d_leaves = leaves.as<int32_t>(); leaves.p = nullptr;
    *svo_count = nodes;
    *leaf_count = nleaves ? nleaves : 1;
    return VXRT_OK;
}

// Reorders the last `levels` (2 or 3) node levels of a breadth-first record array into depth-first treelets (see above).
// level_first[l]: index of the first record of node level l (0 .. depth; level 0 is the root), n: records in all.  *d_svo is replaced.
int reorder_bottom_treelets(SvoRecord** d_svo, size_t n, const std::vector<size_t>& level_first, uint32_t depth, int levels, hipStream_t stream) {
    if ((levels != 2 && levels != 3) || depth < 4 || level_first.size() < size_t(depth) + 1 || n >= (size_t(1) << 32)) {
        set_error("treelets: 2 or 3 levels of a tree at least 4 levels deep");
        return VXRT_E_INVALID;
    }
    const uint32_t fp = uint32_t(level_first[depth - 3]), fa = uint32_t(level_first[depth - 2]), fb = uint32_t(level_first[depth - 1]), fc = uint32_t(level_first[depth]);
    if (!(fp < fa && fa < fb && fb < fc && fc < n)) { set_error("treelets: level starts out of order"); return VXRT_E_INVALID; }
    DevBuf out;
    DEV_TRY(out.alloc(n * sizeof(SvoRecord)));
    const uint32_t keep = levels == 3 ? fp : fa;        // the levels above the treelets' parents: as they are
    DEV_TRY(hipMemcpyAsync(out.p, *d_svo, size_t(keep) * sizeof(SvoRecord), hipMemcpyDeviceToDevice, stream));
    if (levels == 3)
        hipLaunchKernelGGL(treelet_kernel<3>, dim3((fa - fp + 255u) / 256u), dim3(256), 0, stream, *d_svo, out.as<SvoRecord>(), fp, fa, fb, fc);
    else
        hipLaunchKernelGGL(treelet_kernel<2>, dim3((fb - fa + 255u) / 256u), dim3(256), 0, stream, *d_svo, out.as<SvoRecord>(), fa, fa, fb, fc);
    DEV_TRY(hipGetLastError());
    DEV_TRY(hipStreamSynchronize(stream));
    (void)hipFree(*d_svo);
    *d_svo = out.as<SvoRecord>();
    out.p = nullptr;
    return VXRT_OK;
}

}  // namespace vxrt
