// ORACLE — test infrastructure only.  Nothing in the product path may include, link or call this.
//
// ovox.cpp: CPU restatement of the reference's scene preparation:
//   * MagicaVoxel .vox parser            — follows src/vox.rs:11-101,193-312
//   * voxel-list adapter                 — follows src/context.rs:913-933 (voxels_from_vox)
//   * flat octree builder (GPU layout)   — follows src/context.rs:710-834
//   * camera basis                       — follows src/camera.rs:12-28
// Parity status: UNPINNED by the reference (it ships no tests / golden vectors for any of this);
// pinned here by the node-count table of SURVEY.md Appendix C and by self-consistency checks.
#include "oracle.h"

#include <cstdio>
#include <cmath>
#include <cctype>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace orc {

// `value.parse::<f32>()` (src/vox.rs:93-96): core::num::dec2flt accepts  [+-]? ( "inf" | "infinity" | "nan" )  in any case, or
// [+-]? digits* [ "." digits* ] [ (e|E) [+-]? digits+ ]  with at least one mantissa digit — and nothing else (no blanks, no hex).
static bool parses_as_rust_f32(const std::string& v) {
    std::string t = v;
    if (!t.empty() && (t[0] == '+' || t[0] == '-')) t.erase(0, 1);
    std::string low = t;
    for (char& ch : low) if (ch >= 'A' && ch <= 'Z') ch = char(ch - 'A' + 'a');
    if (low == "inf" || low == "infinity" || low == "nan") return true;
    size_t i = 0, mant = 0;
    for (; i < t.size() && isdigit((unsigned char)t[i]); i++) mant++;
    if (i < t.size() && t[i] == '.') for (i++; i < t.size() && isdigit((unsigned char)t[i]); i++) mant++;
    if (mant == 0) return false;
    if (i == t.size()) return true;
    if (t[i] != 'e' && t[i] != 'E') return false;
    i++;
    if (i < t.size() && (t[i] == '+' || t[i] == '-')) i++;
    if (i == t.size()) return false;
    for (; i < t.size(); i++) if (!isdigit((unsigned char)t[i])) return false;
    return true;
}


// ---- byte cursor (src/vox.rs:252-296: read / split / read_u32 / read_str, all little endian) ----
struct Cur {
    const uint8_t* p;
    size_t n;
    bool ok = true;
};
static bool take(Cur& c, size_t k, const uint8_t** out) {
    if (c.n < k) { c.ok = false; return false; }
    *out = c.p; c.p += k; c.n -= k; return true;
}
static uint32_t rd_u32(Cur& c) {
    const uint8_t* b;
    if (!take(c, 4, &b)) return 0;
    return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
}
struct Chunk {
    char id[4];
    Cur content;
};
// src/vox.rs:252-260 — a chunk's "content" is content_size + children_size bytes.
static bool rd_chunk(Cur& c, Chunk* out) {
    const uint8_t* id;
    if (!take(c, 4, &id)) return false;
    memcpy(out->id, id, 4);
    uint32_t content = rd_u32(c), children = rd_u32(c);
    if (!c.ok) return false;
    uint32_t total = content + children;  // u32 wrap like the reference's `content_size + children_size`
    const uint8_t* body;
    if (!take(c, total, &body)) return false;
    out->content = Cur{body, total, true};
    return true;
}
static bool starts_with(const Cur& c, const char* id) { return c.n >= 4 && memcmp(c.p, id, 4) == 0; }

// src/vox.rs:103-136 DEFAULT_PALETTE is the palette published in the MagicaVoxel .vox format
// description: index 0 transparent, then the 6x6x6 colour cube in steps of 0x33 without black
// (words are 0xAABBGGRR; blue steps fastest, then green, then red), then 10-step red, green, blue
// and grey ramps.
void default_palette(uint32_t pal[256]) {
    pal[0] = 0;
    int k = 1;
    for (int i = 0; i < 215; i++) {
        uint32_t b = 0xff - 0x33 * (i % 6), g = 0xff - 0x33 * ((i / 6) % 6), r = 0xff - 0x33 * (i / 36);
        pal[k++] = 0xff000000u | (b << 16) | (g << 8) | r;
    }
    static const uint32_t ramp[10] = {0xee, 0xdd, 0xbb, 0xaa, 0x88, 0x77, 0x55, 0x44, 0x22, 0x11};
    for (int i = 0; i < 10; i++) pal[k++] = 0xff000000u | ramp[i];
    for (int i = 0; i < 10; i++) pal[k++] = 0xff000000u | (ramp[i] << 8);
    for (int i = 0; i < 10; i++) pal[k++] = 0xff000000u | (ramp[i] << 16);
    for (int i = 0; i < 10; i++) pal[k++] = 0xff000000u | (ramp[i] << 16) | (ramp[i] << 8) | ramp[i];
}

struct Model {
    uint32_t sx, sy, sz;
    std::vector<uint8_t> xyzi;  // 4 bytes per voxel
};
struct Vox {
    std::vector<Model> models;
    uint32_t palette[256];
    std::map<uint32_t, int> materials;  // id -> 0 diffuse / 1 emit
};

// src/vox.rs:11-70.  Returns 0 or a negative ORC_E_* code.
static int parse_vox(const uint8_t* bytes, size_t len, Vox* vox) {
    if (len < 4 || memcmp(bytes, "VOX ", 4) != 0) return ORC_E_MAGIC;
    Cur c{bytes + 4, len - 4, true};
    int32_t version = (int32_t)rd_u32(c);
    if (!c.ok) return ORC_E_EOF;
    if (version != 150) return ORC_E_VERSION;
    if (!starts_with(c, "MAIN")) return ORC_E_NOMAIN;
    Chunk main;
    if (!rd_chunk(c, &main)) return ORC_E_EOF;

    Cur b = main.content;
    uint32_t model_count = 1;
    if (starts_with(b, "PACK")) {
        Chunk pack;
        if (!rd_chunk(b, &pack)) return ORC_E_EOF;
        model_count = rd_u32(pack.content);
        if (!pack.content.ok) return ORC_E_EOF;
    }
    for (uint32_t m = 0; m < model_count; m++) {
        Chunk size, xyzi;
        if (!rd_chunk(b, &size)) return ORC_E_EOF;
        if (memcmp(size.id, "SIZE", 4) != 0) return ORC_E_CHUNK;
        if (!rd_chunk(b, &xyzi)) return ORC_E_EOF;
        if (memcmp(xyzi.id, "XYZI", 4) != 0) return ORC_E_CHUNK;
        Model model;
        model.sx = rd_u32(size.content); model.sy = rd_u32(size.content); model.sz = rd_u32(size.content);
        if (!size.content.ok) return ORC_E_EOF;
        uint32_t count = rd_u32(xyzi.content);
        if (!xyzi.content.ok) return ORC_E_EOF;
        const uint8_t* data;
        if (!take(xyzi.content, (size_t)count * 4, &data)) return ORC_E_EOF;
        model.xyzi.assign(data, data + (size_t)count * 4);
        vox->models.push_back(std::move(model));
    }
    default_palette(vox->palette);
    while (b.n != 0) {
        Chunk ch;
        if (!rd_chunk(b, &ch)) return ORC_E_EOF;
        if (memcmp(ch.id, "RGBA", 4) == 0) {
            for (int i = 1; i < 256; i++) {  // src/vox.rs:50-54: 255 words land in palette[1..=255]
                uint32_t rgba = rd_u32(ch.content);
                if (!ch.content.ok) return ORC_E_EOF;
                vox->palette[i] = rgba;
            }
        } else if (memcmp(ch.id, "MATL", 4) == 0) {
            uint32_t id = rd_u32(ch.content);
            uint32_t entries = rd_u32(ch.content);  // read_dict, src/vox.rs:298-311
            if (!ch.content.ok) return ORC_E_EOF;
            int kind = 0;
            for (uint32_t e = 0; e < entries; e++) {
                uint32_t kl = rd_u32(ch.content);
                const uint8_t* k;
                if (!ch.content.ok || !take(ch.content, kl, &k)) return ORC_E_EOF;
                uint32_t vl = rd_u32(ch.content);
                const uint8_t* v;
                if (!ch.content.ok || !take(ch.content, vl, &v)) return ORC_E_EOF;
                std::string key((const char*)k, kl), val((const char*)v, vl);
                if (key == "_type") {  // src/vox.rs:82-92
                    if (val == "_emit") kind = 1;
                    else if (val == "_diffuse") kind = 0;
                    else return ORC_E_MATERIAL;
                } else if (key == "_flux") {  // src/vox.rs:93-96: must parse as f32
                    if (!parses_as_rust_f32(val)) return ORC_E_MATERIAL;
                }
            }
            vox->materials[id] = kind;
        }  // anything else: "unknown chunk", skipped (src/vox.rs:61)
    }
    return 0;
}

}  // namespace orc

using namespace orc;

extern "C" {

void orc_default_palette(uint32_t* out256) { default_palette(out256); }

// .vox bytes -> voxel list exactly as Context::voxels_from_vox builds it (src/context.rs:913-933):
// model 0 only, position (x, z_vox, y_vox), material byte 0x40 iff MATL[colour]._type == _emit,
// rgb = low three bytes of palette[colour].  A colour index with no MATL entry is the reference's
// `.unwrap()` panic -> ORC_E_NOMATL.  Returns the voxel count (also when cap is too small; nothing
// is written past cap) or a negative error.
long orc_voxels_from_vox(const uint8_t* bytes, size_t len, int16_t* pos, uint8_t* mrgb, size_t cap,
                         uint32_t* size_xyz) {
    Vox vox;
    int rc = parse_vox(bytes, len, &vox);
    if (rc < 0) return rc;
    if (vox.models.empty()) return ORC_E_NOMODEL;
    const Model& m = vox.models[0];
    if (size_xyz) { size_xyz[0] = m.sx; size_xyz[1] = m.sy; size_xyz[2] = m.sz; }
    size_t n = m.xyzi.size() / 4;
    for (size_t i = 0; i < n; i++) {
        uint8_t x = m.xyzi[4 * i], y = m.xyzi[4 * i + 1], z = m.xyzi[4 * i + 2], ci = m.xyzi[4 * i + 3];
        auto it = vox.materials.find((uint32_t)ci);
        if (it == vox.materials.end()) return ORC_E_NOMATL;
        if (i < cap) {
            uint32_t col = vox.palette[ci];
            pos[3 * i] = x; pos[3 * i + 1] = z; pos[3 * i + 2] = y;
            mrgb[4 * i] = it->second ? 0x40 : 0;
            mrgb[4 * i + 1] = col & 0xff; mrgb[4 * i + 2] = (col >> 8) & 0xff; mrgb[4 * i + 3] = (col >> 16) & 0xff;
        }
    }
    return (long)n;
}

// Context::voxel_depth (src/context.rs:813-834).
int orc_voxel_depth(const int16_t* pos, size_t n) {
    if (n == 0) return 0;
    int mn = pos[0], mx = pos[0];
    for (size_t i = 0; i < 3 * n; i++) { if (pos[i] < mn) mn = pos[i]; if (pos[i] > mx) mx = pos[i]; }
    auto npot_tz = [](uint32_t v) {  // next_power_of_two().trailing_zeros() on u16 semantics
        uint32_t p = 1; int tz = 0;
        while (p < v) { p <<= 1; tz++; }
        return tz;
    };
    int a = npot_tz((uint16_t)(mn < 0 ? -mn : mn));
    int b = npot_tz((uint16_t)((mx < 0 ? -mx : mx) + 1));
    return a > b ? a : b;
}

// Context::create_octree (src/context.rs:777-796) = 5-word header + create_octree_nodes (:710-773).
// Returns the number of int32 words (written only if <= cap).
long orc_create_octree(const int16_t* pos, const uint8_t* mrgb, size_t n, int32_t* out, size_t cap) {
    int depth = orc_voxel_depth(pos, n);
    std::vector<int32_t> nodes(8, 0);  // alloc_node for the root
    int extent0 = 1 << depth;
    for (size_t i = 0; i < n; i++) {
        size_t cur = 0;
        int cx = 0, cy = 0, cz = 0, extent = extent0;
        int px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
        for (;;) {
            int dx = cx <= px, dy = cy <= py, dz = cz <= pz;
            int oct = 4 * dx + 2 * dy + dz;
            if (extent == 1) {
                int32_t m = mrgb[4 * i], r = mrgb[4 * i + 1], g = mrgb[4 * i + 2], b = mrgb[4 * i + 3];
                nodes[8 * cur + oct] = (int32_t)(0x80000000u | ((uint32_t)(m & 0x7f) << 24) | (r << 16) | (g << 8) | b);
                break;
            }
            int32_t value = nodes[8 * cur + oct];
            size_t child;
            if (value == 0) {
                child = nodes.size() / 8;
                nodes.resize(nodes.size() + 8, 0);
                nodes[8 * cur + oct] = (int32_t)child;
            } else if (value > 0) {
                child = (size_t)value;
            } else {
                return ORC_E_SPLITLEAF;  // the reference's todo!() (src/context.rs:746)
            }
            cx = cx - extent / 2 + dx * extent;
            cy = cy - extent / 2 + dy * extent;
            cz = cz - extent / 2 + dz * extent;
            cur = child;
            extent /= 2;
        }
    }
    size_t total = 5 + nodes.size();
    if (total <= cap) {
        float hdr[5] = {0.0f, 0.0f, 0.0f, (float)(1 << depth), 1.0f};
        memcpy(out, hdr, sizeof hdr);
        memcpy(out + 5, nodes.data(), nodes.size() * sizeof(int32_t));
    }
    return (long)total;
}

// Camera::axis_scaled (src/camera.rs:12-28). out = right[3], up[3], forward_ray[3].
void orc_camera_axis_scaled(const float* position, const float* direction, float fov, uint32_t width,
                            uint32_t height, float* out9) {
    (void)position;
    V3 fwd = normalize(v3(direction[0], direction[1], direction[2]));
    V3 right = normalize(cross(v3(0.0f, 1.0f, 0.0f), fwd));
    V3 up = cross(fwd, right);
    float fov_scale = vx_tan(fov / 2.0f);
    float w = (float)width, h = (float)height;
    V3 fr = ((-w / 2.0f) * right + (h / 2.0f) * up) + ((h / 2.0f) / fov_scale) * fwd;
    out9[0] = right.x; out9[1] = right.y; out9[2] = right.z;
    out9[3] = up.x; out9[4] = up.y; out9[5] = up.z;
    out9[6] = fr.x; out9[7] = fr.y; out9[8] = fr.z;
}

// Context::create_voxels (src/context.rs:838-910): the start-up scene.  The reference draws its colours from
// rand::thread_rng(); draw k is defined here as the hash documented at vxrt_noise_table (include/vxrt.h) of
// (seed, k): gen_range(50..=255) = 50 + h % 206, gen_bool(p) = (h >> 8) * 2^-24 < p.  Returns the voxel count;
// writes at most cap entries.
long orc_default_scene(uint32_t seed, int16_t* pos, uint8_t* mrgb, size_t cap) {
    const int radius = 256;
    uint32_t k = 0;
    auto draw = [&]() {
        uint32_t z = k * 0x9E3779B9u + seed;
        k++;
        z ^= z >> 16; z *= 0x85EBCA6Bu; z ^= z >> 13; z *= 0xC2B2AE35u; z ^= z >> 16;
        return z;
    };
    size_t n = 0;
    auto push = [&](int x, int y, int z, uint8_t m, uint8_t r, uint8_t g, uint8_t b) {
        if (n < cap) {
            pos[3 * n] = (int16_t)x; pos[3 * n + 1] = (int16_t)y; pos[3 * n + 2] = (int16_t)z;
            mrgb[4 * n] = m; mrgb[4 * n + 1] = r; mrgb[4 * n + 2] = g; mrgb[4 * n + 3] = b;
        }
        n++;
    };
    // :861-876 — heights[x][z]: the lower half of a sphere inside the radius, 0 outside
    auto height_at = [&](int x, int z, bool* some) {
        *some = !(x < -radius || x > radius || z < -radius || z > radius);  // :879-881
        if (!*some) return 0;
        if (x * x + z * z <= radius * radius) return (int)(-sqrtf((float)(radius * radius) - (float)(x * x) - (float)(z * z)));
        return 0;
    };
    for (int x = -radius; x <= radius; x++)
        for (int z = -radius; z <= radius; z++) {
            bool some;
            int curr = height_at(x, z, &some);
            int low = curr;
            const int dx[4] = {-1, 1, 0, 0}, dz[4] = {0, 0, -1, 1};
            for (int i = 0; i < 4; i++) {  // :895-899
                int h = height_at(x + dx[i], z + dz[i], &some);
                if (!some) h = curr;
                if (h < low) low = h;
            }
            for (int y = low; y <= curr; y++) {  // :900-902, color(0.01, ..) :849-858
                uint8_t r = (uint8_t)(50 + draw() % 206), g = (uint8_t)(50 + draw() % 206), b = (uint8_t)(50 + draw() % 206);
                bool emissive = (float)(draw() >> 8) * (1.0f / 16777216.0f) < 0.01f;
                push(x, y, z, emissive ? 0x40 : 0x00, r, g, b);
            }
        }
    for (int x = -radius; x <= radius; x++) push(x, -10, 0, 0x40, 255, 255, 255);  // :907-910
    return (long)n;
}

}  // extern "C"
