// ORACLE — test infrastructure only.  Nothing in the product path may include, link or call this.
//
// Blue-noise table: CPU restatement of the void-and-cluster generator specified in include/vxrt_bluenoise.h,
// and of the reference's loader for its (missing) resources/blue-noise-128.zip image format
// (parse_raw_f32img, src/context.rs:1087-1116).  Parity status: UNPINNED — the reference ships neither the
// table nor a generator; this checks the HIP kernel against the written spec, nothing more.
#include <cstring>
#include <vector>

#include "../include/vxrt_bluenoise.h"
#include "oracle.h"

namespace {

struct Layer {
    int n, cells;
    std::vector<float> e;
    std::vector<uint8_t> one;
    float k[VXBN_TAPS][VXBN_TAPS];

    explicit Layer(int size) : n(size), cells(size * size), e(size_t(size) * size), one(size_t(size) * size) {
        for (int dy = -VXBN_RADIUS; dy <= VXBN_RADIUS; dy++)
            for (int dx = -VXBN_RADIUS; dx <= VXBN_RADIUS; dx++) k[dy + VXBN_RADIUS][dx + VXBN_RADIUS] = vxbn_kernel(dx, dy);
    }
    int wrap(int x, int y) const { return (x & (n - 1)) + n * (y & (n - 1)); }

    void gather(uint8_t minority) {  // E over the cells whose bit equals `minority`
        for (int c = 0; c < cells; c++) {
            int cx = c % n, cy = c / n;
            float s = 0.0f;
            for (int dy = -VXBN_RADIUS; dy <= VXBN_RADIUS; dy++)
                for (int dx = -VXBN_RADIUS; dx <= VXBN_RADIUS; dx++)
                    if (one[size_t(wrap(cx + dx, cy + dy))] == minority) s = s + k[dy + VXBN_RADIUS][dx + VXBN_RADIUS];
            e[size_t(c)] = s;
        }
    }
    void splat(int c, bool add) {
        int cx = c % n, cy = c / n;
        for (int dy = -VXBN_RADIUS; dy <= VXBN_RADIUS; dy++)
            for (int dx = -VXBN_RADIUS; dx <= VXBN_RADIUS; dx++) {
                float& v = e[size_t(wrap(cx + dx, cy + dy))];
                float kk = k[dy + VXBN_RADIUS][dx + VXBN_RADIUS];
                v = add ? v + kk : v - kk;
            }
    }
    // extreme of E over the cells whose bit equals `bit`; ties -> lowest index
    int find(uint8_t bit, bool want_max) const {
        int best = -1;
        for (int c = 0; c < cells; c++) {
            if (one[size_t(c)] != bit) continue;
            if (best < 0 || (want_max ? e[size_t(c)] > e[size_t(best)] : e[size_t(c)] < e[size_t(best)])) best = c;
        }
        return best;
    }
};

}  // namespace

extern "C" {

// One layer of the table of include/vxrt_bluenoise.h.  out: size*size floats.  0 ok, -1 bad size.
int orc_blue_noise_layer(uint32_t seed, uint32_t layer, int size, float* out) {
    if (size < 16 || size > VXBN_MAX_SIZE || (size & (size - 1)) != 0) return -1;
    Layer L(size);
    const int cells = L.cells, n0 = cells / 10, half = cells / 2;
    int placed = 0;
    for (uint32_t i = 0; placed < n0; i++) {
        uint32_t c = vxbn_hash(seed, layer, i) % uint32_t(cells);
        if (!L.one[c]) { L.one[c] = 1; placed++; }
    }
    L.gather(1);
    for (int round = 0; round < 4 * n0; round++) {  // relax
        int c1 = L.find(1, true);
        L.one[size_t(c1)] = 0; L.splat(c1, false);
        int c0 = L.find(0, false);
        L.one[size_t(c0)] = 1; L.splat(c0, true);
        if (c0 == c1) break;
    }
    std::vector<uint8_t> relaxed = L.one;
    std::vector<int> rank(size_t(cells), 0);
    for (int r = n0 - 1; r >= 0; r--) {
        int c = L.find(1, true);
        L.one[size_t(c)] = 0; L.splat(c, false);
        rank[size_t(c)] = r;
    }
    L.one = relaxed;
    L.gather(1);
    for (int r = n0; r < half; r++) {
        int c = L.find(0, false);
        L.one[size_t(c)] = 1; L.splat(c, true);
        rank[size_t(c)] = r;
    }
    L.gather(0);
    for (int r = half; r < cells; r++) {
        int c = L.find(0, true);
        L.one[size_t(c)] = 1; L.splat(c, false);
        rank[size_t(c)] = r;
    }
    for (int c = 0; c < cells; c++) out[c] = (float(rank[size_t(c)]) + 0.5f) / float(cells);
    return 0;
}

// parse_raw_f32img (src/context.rs:1087-1116): BE u32 width, BE u32 height, width*height BE f32.
// Returns the pixel count appended to out (<= cap) or -1 on a short buffer.
long orc_parse_raw_f32img(const uint8_t* bytes, size_t len, float* out, size_t cap, uint32_t* width, uint32_t* height) {
    auto be32 = [](const uint8_t* p) { return uint32_t(p[0]) << 24 | uint32_t(p[1]) << 16 | uint32_t(p[2]) << 8 | uint32_t(p[3]); };
    if (len < 8) return -1;
    *width = be32(bytes);
    *height = be32(bytes + 4);
    size_t count = size_t(*width) * *height;
    if (len < 8 + 4 * count || count > cap) return -1;
    for (size_t i = 0; i < count; i++) {
        uint32_t u = be32(bytes + 8 + 4 * i);
        memcpy(out + i, &u, 4);
    }
    return long(count);
}

}  // extern "C"
