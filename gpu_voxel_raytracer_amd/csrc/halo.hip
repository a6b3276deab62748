// halo.hip — the multi-rank halo of libvxrt (gfx950): the rows a rank's temporal and denoise stages read of its neighbours' bands
// (shaders/denoise.comp:51-57: the (2r+1)^2 window; shaders/temporal.comp:85-113: the reprojected history texel), moved with ONE
// pack launch before the two messages leave and ONE unpack launch after the two have arrived.
//
//  halo_pack_kernel   : for every local band, its top `rows` rows -> the message for rank - 1 (they lie just below that rank's
//                       bands), its bottom `rows` rows -> the message for rank + 1.  Per pixel 48 bytes are read (accumulated
//                       colour, normal/depth, the leaf word of albedo/node) and 36 written: planes A = (rgb, depth), B = (normal,
//                       material id), C = blending factor (kernels.h: HaloView).  HBM-bound: 84 B per halo pixel.
//  halo_unpack_kernel : both received messages -> the context's halo store, a straight 16-byte-per-lane copy (the caller's
//                       buffers are borrowed, the store is what temporal_kernel / denoise_kernel read).
#include "halo_view.h"

namespace vxrt {
namespace {

__global__ __launch_bounds__(256) void halo_pack_kernel(const HaloPackArgs a) {
    const int x = blockIdx.y * 256 + threadIdx.x;
    const int lb = blockIdx.x / a.rows, k = blockIdx.x - lb * a.rows;
    const int side = blockIdx.z;   // 0: to the previous rank, 1: to the next rank
    if (x >= a.band.width) return;
    const BandMap& b = a.band;
    const int gb = lb * b.nranks + b.rank;
    const int y0 = band_first_row(b, gb), nominal = band_nominal_rows(b, gb);
    if (y0 >= b.height) return;
    const int rows_here = (y0 + nominal <= b.height) ? nominal : b.height - y0;
    int lrow, slot;
    if (side == 0) {   // my top rows are the rows below band gb - 1 (the previous rank's local band (gb - 1) / nranks)
        if (gb < 1 || k >= rows_here) return;
        lrow = local_band_first_row(b, lb) + k;
        slot = (gb - 1) / b.nranks;
    } else {           // my bottom rows are the rows above band gb + 1 (the next rank's local band (gb + 1) / nranks)
        if (rows_here != nominal || y0 + nominal >= b.height) return;
        lrow = local_band_first_row(b, lb) + (nominal - a.rows) + k;
        slot = (gb + 1) / b.nranks;
    }
    const size_t p = size_t(lrow) * b.width + x;
    const float4 c = a.color[p], nd = a.nd[p];
    const int32_t word = __float_as_int(a.albedo[p].w);
    float4* msg = side == 0 ? a.to_prev : a.to_next;
    const size_t i = halo_row_index(a.rows, b.width, slot, k) + x;
    msg[i] = make_float4(c.x, c.y, c.z, nd.w);
    msg[a.plane + i] = make_float4(nd.x, nd.y, nd.z, __int_as_float((word >> 24) & 0xff));   // denoise.comp:67
    reinterpret_cast<float*>(msg + 2 * a.plane)[i] = c.w;
}

__global__ __launch_bounds__(256) void halo_unpack_kernel(float4* store, const float4* from_prev, const float4* from_next, size_t message_f4) {
    const size_t stride = size_t(gridDim.x) * 256;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < 2 * message_f4; i += stride)
        store[i] = i < message_f4 ? from_prev[i] : from_next[i - message_f4];
}

}  // namespace

hipError_t launch_halo_pack(const HaloPackArgs& a, hipStream_t s) {
    if (a.local_bands <= 0 || a.rows <= 0) return hipSuccess;
    dim3 grid(unsigned(a.local_bands * a.rows), unsigned((a.band.width + 255) / 256), 2u);
    hipLaunchKernelGGL(halo_pack_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_halo_unpack(float4* store, const float4* from_prev, const float4* from_next, size_t message_f4, hipStream_t s) {
    if (message_f4 == 0) return hipSuccess;
    const size_t want = (2 * message_f4 + 255) / 256;
    hipLaunchKernelGGL(halo_unpack_kernel, dim3(unsigned(want < 4096 ? want : 4096)), dim3(256), 0, s, store, from_prev, from_next, message_f4);
    return hipGetLastError();
}

}  // namespace vxrt
