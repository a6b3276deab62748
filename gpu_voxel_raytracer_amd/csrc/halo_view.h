// halo_view.h — where a neighbour's pixel sits in a halo message / in the context's halo store (kernels.h: HaloView).
// Shared by the writer (halo.hip: halo_pack_kernel) and the readers (post.hip: temporal_kernel, denoise_kernel).
#pragma once
#include "kernels.h"

namespace vxrt {

// float4 index of (slot, row k, x = 0) in plane A of one message; plane B is `plane` further, plane C (floats) starts at 2 * plane
__device__ __forceinline__ size_t halo_row_index(int rows, int width, int slot, int k) { return (size_t(slot) * rows + k) * size_t(width); }

struct HaloRow {   // one row of a neighbour's pixels as this rank received it
    const float4* a;   // (r, g, b, depth)
    const float4* b;   // (nx, ny, nz, bits(material id))
    const float* c;    // blending factor of the accumulated colour
};
__device__ __forceinline__ HaloRow halo_row(const HaloView& h, int width, int side, int lband, int k) {
    const float4* msg = h.base + size_t(side) * h.message;
    const size_t i = halo_row_index(h.rows, width, lband, k);
    HaloRow r;
    r.a = msg + i;
    r.b = msg + h.plane + i;
    r.c = reinterpret_cast<const float*>(msg + 2 * h.plane) + i;
    return r;
}

// Frame row y, which this rank does not own, in its halo: false when the halo does not hold it (further than h.rows rows from
// every band of this rank, or no halo at all).
__device__ __forceinline__ bool halo_find(const BandMap& b, const HaloView& h, int y, HaloRow& out) {
    if (h.base == nullptr) return false;
    const int band = band_of_row(b, y), off = y - band_first_row(b, band), nominal = band_nominal_rows(b, band);
    if (band >= 1 && (band - 1) % b.nranks == b.rank && off < h.rows) {               // just below one of this rank's bands
        out = halo_row(h, b.width, 1, (band - 1) / b.nranks, off);
        return true;
    }
    if ((band + 1) % b.nranks == b.rank && off >= nominal - h.rows) {                 // just above one
        out = halo_row(h, b.width, 0, (band + 1) / b.nranks, off - (nominal - h.rows));
        return true;
    }
    return false;
}

}  // namespace vxrt
