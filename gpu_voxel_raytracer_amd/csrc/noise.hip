// noise.hip — the noise table of shaders/voxels.comp:65-71 made on the device.
//
//  * noise_fill_kernel: the white stand-in table documented in vxrt.h (vxrt_noise_table);
//  * blue_noise_kernel: void-and-cluster blue noise as specified in include/vxrt_bluenoise.h — the kind of table the
//    reference loads from resources/blue-noise-128.zip (src/context.rs:1016-1116), which its repository does not ship.
//
// blue_noise_kernel: one 1024-thread block per layer, the whole layer resident in LDS (energy map 64 KB + ranks 32 KB
// + two bit patterns at N = 128), so a layer's 16 384 dependent "find the extreme cell, splat the kernel" rounds
// never touch memory; 512 layers = two waves of blocks on 256 CUs.  A round is: every thread scans its cells
// (stride 1024: conflict-free) for the extreme energy among cells of the wanted colour, packed as a 64-bit key
// (ordered energy bits : inverted cell index, so ties go to the lowest cell) -> wave64 shuffle reduction ->
// 16 partial keys through LDS -> the first 225 threads splat the 15x15 kernel around the winner.
#include "kernels.h"
#include "../../include/vxrt_bluenoise.h"

namespace vxrt {
namespace {

__global__ void noise_fill_kernel(float* dst, uint32_t seed, size_t n) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t z = uint32_t(i) * 0x9E3779B9u + seed;
        z ^= z >> 16; z *= 0x85EBCA6Bu;
        z ^= z >> 13; z *= 0xC2B2AE35u;
        z ^= z >> 16;
        dst[i] = float(z >> 8) * (1.0f / 16777216.0f);
    }
}

constexpr int kBnThreads = 1024;
constexpr int kBnTaps = VXBN_TAPS * VXBN_TAPS;

struct BnLayer {  // views into the block's LDS
    float* e;            // [cells]
    uint16_t* rank;      // [cells]
    uint32_t* bits;      // [cells / 32]  current pattern
    uint32_t* saved;     // [cells / 32]  relaxed pattern
    float* k;            // [kBnTaps]
    unsigned long long* part;  // [2][16] per-wave partial keys, double-buffered
    int n, cells, shift;
};

__device__ __forceinline__ bool bn_bit(const uint32_t* bits, int c) { return (bits[c >> 5] >> (c & 31)) & 1u; }

// larger key = better candidate; ties in energy -> lower cell index
__device__ __forceinline__ unsigned long long bn_key(float e, int c, bool want_max) {
    uint32_t u = __float_as_uint(e);
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;  // monotone map of binary32 onto unsigned
    if (!want_max) u = ~u;
    return (unsigned long long)u << 32 | (0xffffffffu - uint32_t(c));
}

// the cell of extreme energy among the cells whose bit equals `colour` (block-uniform result)
__device__ __forceinline__ int bn_find(const BnLayer& L, bool colour, bool want_max, int round) {
    const int tid = threadIdx.x;
    unsigned long long best = 0ull;
    for (int c = tid; c < L.cells; c += kBnThreads) {
        if (bn_bit(L.bits, c) == colour) {
            const unsigned long long key = bn_key(L.e[c], c, want_max);
            best = key > best ? key : best;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long other = __shfl_xor(best, off, 64);
        best = other > best ? other : best;
    }
    unsigned long long* part = L.part + (round & 1) * 16;
    if ((tid & 63) == 0) part[tid >> 6] = best;
    __syncthreads();
    unsigned long long all = part[0];
    for (int w = 1; w < kBnThreads / 64; w++) all = part[w] > all ? part[w] : all;
    return int(0xffffffffu - uint32_t(all));
}

// flip the winner's bit and add / subtract the kernel around it; ends with a barrier
__device__ __forceinline__ void bn_splat(const BnLayer& L, int c, bool add, bool new_bit) {
    const int tid = threadIdx.x;
    if (tid < kBnTaps) {
        const int dy = tid / VXBN_TAPS - VXBN_RADIUS, dx = tid % VXBN_TAPS - VXBN_RADIUS;
        const int cx = c & (L.n - 1), cy = c >> L.shift;
        const int cell = ((cx + dx) & (L.n - 1)) + (((cy + dy) & (L.n - 1)) << L.shift);
        const float v = L.e[cell], kk = L.k[tid];
        L.e[cell] = add ? v + kk : v - kk;
    } else if (tid == kBnTaps) {
        const uint32_t m = 1u << (c & 31);
        L.bits[c >> 5] = new_bit ? (L.bits[c >> 5] | m) : (L.bits[c >> 5] & ~m);
    }
    __syncthreads();
}

// E[c] = sum of K over the window cells whose bit equals `colour`, in the spec's tap order; ends with a barrier
__device__ __forceinline__ void bn_gather(const BnLayer& L, bool colour) {
    for (int c = threadIdx.x; c < L.cells; c += kBnThreads) {
        const int cx = c & (L.n - 1), cy = c >> L.shift;
        float s = 0.0f;
        for (int dy = -VXBN_RADIUS; dy <= VXBN_RADIUS; dy++) {
            const int row = ((cy + dy) & (L.n - 1)) << L.shift;
            for (int dx = -VXBN_RADIUS; dx <= VXBN_RADIUS; dx++)
                if (bn_bit(L.bits, ((cx + dx) & (L.n - 1)) + row) == colour) s = s + L.k[(dy + VXBN_RADIUS) * VXBN_TAPS + dx + VXBN_RADIUS];
        }
        L.e[c] = s;
    }
    __syncthreads();
}

__global__ __launch_bounds__(kBnThreads) void blue_noise_kernel(float* out, uint32_t seed, uint32_t first_layer, int n, int shift) {
    extern __shared__ unsigned char bn_lds[];
    BnLayer L;
    L.n = n; L.shift = shift; L.cells = n * n;
    L.e = reinterpret_cast<float*>(bn_lds);
    L.part = reinterpret_cast<unsigned long long*>(L.e + L.cells);
    L.k = reinterpret_cast<float*>(L.part + 32);
    L.bits = reinterpret_cast<uint32_t*>(L.k + 256);
    L.saved = L.bits + L.cells / 32;
    L.rank = reinterpret_cast<uint16_t*>(L.saved + L.cells / 32);
    const int tid = threadIdx.x;
    const uint32_t layer = first_layer + blockIdx.x;
    const int cells = L.cells, words = cells / 32, n0 = cells / 10, half = cells / 2;

    if (tid < kBnTaps) L.k[tid] = vxbn_kernel(tid % VXBN_TAPS - VXBN_RADIUS, tid / VXBN_TAPS - VXBN_RADIUS);
    for (int w = tid; w < words; w += kBnThreads) L.bits[w] = 0u;
    __syncthreads();
    if (tid == 0) {  // step 1: the initial pattern is a sequential draw (duplicates skipped)
        int placed = 0;
        for (uint32_t i = 0; placed < n0; i++) {
            const uint32_t c = vxbn_hash(seed, layer, i) % uint32_t(cells);
            const uint32_t m = 1u << (c & 31);
            if (!(L.bits[c >> 5] & m)) { L.bits[c >> 5] |= m; placed++; }
        }
    }
    __syncthreads();
    bn_gather(L, true);

    int round = 0;
    for (int it = 0; it < 4 * n0; it++) {  // step 2: relax
        const int c1 = bn_find(L, true, true, round++);
        bn_splat(L, c1, false, false);
        const int c0 = bn_find(L, false, false, round++);
        bn_splat(L, c0, true, true);
        if (c0 == c1) break;
    }
    for (int w = tid; w < words; w += kBnThreads) L.saved[w] = L.bits[w];
    __syncthreads();
    for (int r = n0 - 1; r >= 0; r--) {  // step 3
        const int c = bn_find(L, true, true, round++);
        if (tid == 0) L.rank[c] = uint16_t(r);
        bn_splat(L, c, false, false);
    }
    for (int w = tid; w < words; w += kBnThreads) L.bits[w] = L.saved[w];
    __syncthreads();
    bn_gather(L, true);
    for (int r = n0; r < half; r++) {  // step 4
        const int c = bn_find(L, false, false, round++);
        if (tid == 0) L.rank[c] = uint16_t(r);
        bn_splat(L, c, true, true);
    }
    bn_gather(L, false);
    for (int r = half; r < cells; r++) {  // step 5
        const int c = bn_find(L, false, true, round++);
        if (tid == 0) L.rank[c] = uint16_t(r);
        bn_splat(L, c, false, true);
    }
    float* dst = out + size_t(blockIdx.x) * cells;
    for (int c = tid; c < cells; c += kBnThreads) dst[c] = (float(L.rank[c]) + 0.5f) / float(cells);
}

}  // namespace

hipError_t launch_noise_fill(float* dst, uint32_t seed, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(noise_fill_kernel, dim3(2048), dim3(256), 0, s, dst, seed, n);
    return hipGetLastError();
}

size_t blue_noise_lds_bytes(int size) {
    const size_t cells = size_t(size) * size;
    return cells * 4 + 32 * 8 + 256 * 4 + 2 * (cells / 32) * 4 + cells * 2;
}

hipError_t launch_blue_noise(float* dst, uint32_t seed, uint32_t first_layer, uint32_t layers, int size, hipStream_t s) {
    int shift = 0;
    while ((1 << shift) < size) shift++;
    const size_t lds = blue_noise_lds_bytes(size);
    if (lds > 64 * 1024) {  // above the default dynamic-LDS limit: ask for it (gfx950: 160 KB per CU)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(blue_noise_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(blue_noise_kernel, dim3(layers), dim3(kBnThreads), lds, s, dst, seed, first_layer, size, shift);
    return hipGetLastError();
}

}  // namespace vxrt
