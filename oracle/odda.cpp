// ORACLE — test infrastructure only.  Nothing in the product path may include, link or call this.
//
// odda.cpp: an INDEPENDENT first-hit finder used to cross-check the octree restatement
// (oshaders.cpp cast_bounded_ray): a textbook 3-D DDA (Amanatides & Woo 1987) over a dense
// occupancy grid, in binary64.  It shares no code and no data structure with the octree walk.
// It follows no reference file; it checks the *result contract* the octree defines:
// first non-empty unit voxel along the ray, entry t, entry face.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

extern "C" {

// grid: dense occupancy, dims (nx,ny,nz), cell (i,j,k) at grid[(i*ny + j)*nz + k] != 0, covering integer
// cells [base, base+n) per axis; a cell is the world cube [c/2, c/2 + 1/2)^3.
// For each ray: hit flag, world-space t of the entry, entry axis (0/1/2, or -1 when the origin is
// inside a solid cell), and the integer cell.
void orc_dda_cast(const uint8_t* grid, const int32_t* dims, const int32_t* base, const float* origins,
                  const float* dirs, size_t n, uint8_t* hit, double* time, int32_t* axis, int32_t* cell) {
    const int nx = dims[0], ny = dims[1], nz = dims[2];
    for (size_t r = 0; r < n; r++) {
        // grid space: unit cells, g = 2*world - base
        double o[3], d[3];
        for (int a = 0; a < 3; a++) { o[a] = 2.0 * (double)origins[3 * r + a] - base[a]; d[a] = 2.0 * (double)dirs[3 * r + a]; }
        const double lim[3] = {(double)nx, (double)ny, (double)nz};
        // clip to the grid box
        double t0 = 0.0, t1 = INFINITY; int enter_axis = -1; bool miss = false;
        for (int a = 0; a < 3; a++) {
            if (d[a] == 0.0) { if (o[a] < 0.0 || o[a] >= lim[a]) miss = true; continue; }
            double ta = (0.0 - o[a]) / d[a], tb = (lim[a] - o[a]) / d[a];
            if (ta > tb) { double s = ta; ta = tb; tb = s; }
            if (ta > t0) { t0 = ta; enter_axis = a; }
            if (tb < t1) t1 = tb;
        }
        hit[r] = 0; time[r] = 0.0; axis[r] = -1; cell[3 * r] = cell[3 * r + 1] = cell[3 * r + 2] = 0;
        if (miss || t0 >= t1) continue;
        int c[3], step[3]; double tmax[3], tdelta[3];
        for (int a = 0; a < 3; a++) {
            double p = o[a] + t0 * d[a];
            int ci = (int)std::floor(p);
            if (a == enter_axis) ci = d[a] > 0.0 ? 0 : (int)lim[a] - 1;  // exactly on the entry face
            if (ci < 0) ci = 0;
            if (ci >= (int)lim[a]) ci = (int)lim[a] - 1;
            c[a] = ci;
            step[a] = d[a] > 0.0 ? 1 : -1;
            if (d[a] == 0.0) { tmax[a] = INFINITY; tdelta[a] = INFINITY; }
            else { tmax[a] = ((double)(ci + (d[a] > 0.0 ? 1 : 0)) - o[a]) / d[a]; tdelta[a] = std::fabs(1.0 / d[a]); }
        }
        double t = t0; int ax = enter_axis;
        for (;;) {
            if (grid[((size_t)c[0] * ny + c[1]) * nz + c[2]]) {
                hit[r] = 1; time[r] = t; axis[r] = ax;
                cell[3 * r] = c[0] + base[0]; cell[3 * r + 1] = c[1] + base[1]; cell[3 * r + 2] = c[2] + base[2];
                break;
            }
            int a = tmax[0] < tmax[1] ? (tmax[0] < tmax[2] ? 0 : 2) : (tmax[1] < tmax[2] ? 1 : 2);
            t = tmax[a]; ax = a;
            c[a] += step[a];
            if (c[a] < 0 || c[a] >= (int)lim[a]) break;
            tmax[a] += tdelta[a];
        }
    }
}

}  // extern "C"
