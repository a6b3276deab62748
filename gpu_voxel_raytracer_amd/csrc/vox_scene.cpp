// vox_scene.cpp — wider scene input than the reference accepts; host only, no HIP calls.
//
//  * decode_vox_scene: whole MagicaVoxel v150 scenes.  The reference's parser (src/vox.rs:26-67) keeps every model of a
//    file but its adapter renders models[0] only, at its raw cell coordinates (src/context.rs:913-933), ignores the
//    scene graph chunks ("unknown chunk", src/vox.rs:61) and rejects every material type except _diffuse / _emit
//    (src/vox.rs:82-89).  Here, on request (flags), every shape instance of the nTRN / nGRP / nSHP graph is placed
//    with its translation and axis rotation, and other material types are taken as diffuse.  The output is the
//    same voxel list the renderer consumes (Vec<([i16;3],[u8;4])>, axes (x, z_file, y_file)).
//  * default_scene: Context::create_voxels (src/context.rs:838-910), the bowl-shaped height field with a light
//    strip the reference shows at start-up — with a seeded generator in place of rand::thread_rng().
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string_view>

#include "scene_host.h"

namespace vxrt {
namespace {

struct Reader {
    const uint8_t* p;
    size_t n;
    bool u32(uint32_t* v) {
        if (n < 4) return false;
        *v = uint32_t(p[0]) | uint32_t(p[1]) << 8 | uint32_t(p[2]) << 16 | uint32_t(p[3]) << 24;
        p += 4; n -= 4;
        return true;
    }
    bool i32(int32_t* v) { uint32_t u; if (!u32(&u)) return false; *v = int32_t(u); return true; }
    bool take(size_t k, Reader* out) {
        if (k > n) return false;
        *out = Reader{p, k};
        p += k; n -= k;
        return true;
    }
    bool text(std::string_view* out) {
        uint32_t len;
        Reader s{nullptr, 0};
        if (!u32(&len) || !take(len, &s)) return false;
        *out = std::string_view(reinterpret_cast<const char*>(s.p), s.n);
        return true;
    }
    bool dict(std::map<std::string, std::string>* out) {  // read_dict, src/vox.rs:298-310
        uint32_t pairs;
        if (!u32(&pairs)) return false;
        for (uint32_t i = 0; i < pairs; i++) {
            std::string_view k, v;
            if (!text(&k) || !text(&v)) return false;
            (*out)[std::string(k)] = std::string(v);
        }
        return true;
    }
};

int eof() { set_error("unexpected end of file"); return VXRT_E_VOX_EOF; }

struct Xform {  // p -> R p + t, R a signed permutation
    int r[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    int64_t t[3] = {0, 0, 0};
};

Xform compose(const Xform& parent, const Xform& child) {  // parent after child
    Xform o;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            o.r[i][j] = 0;
            for (int k = 0; k < 3; k++) o.r[i][j] += parent.r[i][k] * child.r[k][j];
        }
        o.t[i] = parent.t[i];
        for (int k = 0; k < 3; k++) o.t[i] += parent.r[i][k] * child.t[k];
    }
    return o;
}

// _r: bits 0-1 column of the non-zero entry of row 0, bits 2-3 of row 1 (row 2 takes the remaining column),
// bits 4/5/6 set: that row's entry is -1.
bool rotation_from_byte(uint32_t b, int r[3][3]) {
    const int c0 = int(b & 3u), c1 = int((b >> 2) & 3u);
    if (c0 > 2 || c1 > 2 || c0 == c1) return false;
    const int c2 = 3 - c0 - c1;
    memset(r, 0, 9 * sizeof(int));
    r[0][c0] = (b & 16u) ? -1 : 1;
    r[1][c1] = (b & 32u) ? -1 : 1;
    r[2][c2] = (b & 64u) ? -1 : 1;
    return true;
}

struct Model { uint32_t size[3]; const uint8_t* cells; uint32_t count; };
struct Node {
    char kind = 0;                  // 'T', 'G', 'S'
    Xform xf;                       // T
    int32_t child = -1;             // T
    std::vector<int32_t> children;  // G
    std::vector<int32_t> models;    // S
};

constexpr size_t kMaxPlacedVoxels = size_t(1) << 27;   // 1 GiB of Voxel records
constexpr long kMaxTranslation = 1L << 20;               // |_t| beyond this cannot place anything inside 16-bit coordinates

struct Placer {
    const std::vector<Model>& models;
    const std::map<int32_t, Node>& nodes;
    const uint32_t* palette;
    const uint8_t* kind_of;  // per colour index: 0 absent, 1 diffuse, 2 emit
    bool lenient;
    std::vector<Voxel>* out;
    int64_t lo[3], hi[3];
    size_t instances = 0;

    int place(const Model& m, const Xform& xf) {
        const int64_t pivot[3] = {m.size[0] / 2, m.size[1] / 2, m.size[2] / 2};
        // a shape graph may instance one model any number of times: bound what a file can make this process allocate
        if (out->size() + size_t(m.count) > kMaxPlacedVoxels) { set_error("scene places more than 2^27 voxels"); return VXRT_E_SCENE; }
        const uint8_t* c = m.cells;
        for (uint32_t i = 0; i < m.count; i++, c += 4) {
            // doubled coordinates of the cell centre relative to the pivot: odd integers, so R p + 2t is odd as well
            const int64_t p2[3] = {2 * int64_t(c[0]) + 1 - 2 * pivot[0], 2 * int64_t(c[1]) + 1 - 2 * pivot[1], 2 * int64_t(c[2]) + 1 - 2 * pivot[2]};
            int64_t w[3];
            for (int a = 0; a < 3; a++) {
                const int64_t w2 = xf.r[a][0] * p2[0] + xf.r[a][1] * p2[1] + xf.r[a][2] * p2[2] + 2 * xf.t[a];
                w[a] = (w2 - 1) / 2;  // floor of the centre: w2 is odd, so this division is exact
                lo[a] = w[a] < lo[a] ? w[a] : lo[a];
                hi[a] = w[a] > hi[a] ? w[a] : hi[a];
            }
            const uint8_t colour = c[3];
            uint8_t kind = kind_of[colour];
            if (kind == 0) {
                if (!lenient) { set_error("voxel colour index " + std::to_string(colour) + " has no MATL entry"); return VXRT_E_VOX_NOMATL; }
                kind = 1;
            }
            for (int a = 0; a < 3; a++)
                if (w[a] < -32768 || w[a] > 32767) { set_error("scene does not fit 16-bit voxel coordinates"); return VXRT_E_SCENE; }
            const uint32_t rgba = palette[colour];
            Voxel v;
            v.x = int16_t(w[0]); v.y = int16_t(w[2]); v.z = int16_t(w[1]);  // renderer axes, src/context.rs:927
            v.m = kind == 2 ? 0x40 : 0x00;
            v.r = uint8_t(rgba); v.g = uint8_t(rgba >> 8); v.b = uint8_t(rgba >> 16);
            out->push_back(v);
        }
        return VXRT_OK;
    }

    int walk(int32_t id, const Xform& xf, int depth) {
        if (depth > 64 || ++instances > 65536) { set_error("scene graph too deep or cyclic"); return VXRT_E_SCENE; }
        auto it = nodes.find(id);
        if (it == nodes.end()) { set_error("scene graph refers to a missing node " + std::to_string(id)); return VXRT_E_SCENE; }
        const Node& nd = it->second;
        if (nd.kind == 'T') return walk(nd.child, compose(xf, nd.xf), depth + 1);
        if (nd.kind == 'G') {
            for (int32_t ch : nd.children)
                if (int rc = walk(ch, xf, depth + 1)) return rc;
            return VXRT_OK;
        }
        for (int32_t m : nd.models) {
            if (m < 0 || size_t(m) >= models.size()) { set_error("shape refers to a missing model " + std::to_string(m)); return VXRT_E_SCENE; }
            if (int rc = place(models[size_t(m)], xf)) return rc;
        }
        return VXRT_OK;
    }
};

void builtin_palette(uint32_t* pal) {  // as in scene_host.cpp (the .vox format's default palette, src/vox.rs:103-136)
    static const uint8_t steps[10] = {0xee, 0xdd, 0xbb, 0xaa, 0x88, 0x77, 0x55, 0x44, 0x22, 0x11};
    pal[0] = 0;
    uint32_t* w = pal + 1;
    for (uint32_t lo = 0; lo < 6; lo++)
        for (uint32_t mid = 0; mid < 6; mid++)
            for (uint32_t hi = 0; hi < 6; hi++) {
                if (lo == 5 && mid == 5 && hi == 5) continue;
                *w++ = 0xff000000u | (0xffu - 0x33u * hi) << 16 | (0xffu - 0x33u * mid) << 8 | (0xffu - 0x33u * lo);
            }
    for (int shift = 0; shift <= 16; shift += 8)
        for (uint8_t s : steps) *w++ = 0xff000000u | uint32_t(s) << shift;
    for (uint8_t s : steps) *w++ = 0xff000000u | uint32_t(s) * 0x010101u;
}

}  // namespace

int decode_vox_scene(const uint8_t* bytes, size_t len, uint32_t flags, VoxScene* out, int32_t bounds_lo[3], int32_t bounds_hi[3]) {
    if (flags == 0) {  // no extension requested: exactly the reference's behaviour
        int rc = decode_vox(bytes, len, out);
        if (rc == VXRT_OK) for (int a = 0; a < 3; a++) { bounds_lo[a] = 0; bounds_hi[a] = int32_t(out->size[a == 1 ? 2 : (a == 2 ? 1 : 0)]) - 1; }
        return rc;
    }
    const bool lenient = (flags & VXRT_VOX_LENIENT_MATERIALS) != 0;
    Reader file{bytes, len};
    if (len < 4 || memcmp(bytes, "VOX ", 4) != 0) { set_error("invalid magic number"); return VXRT_E_VOX_MAGIC; }
    file.p += 4; file.n -= 4;
    int32_t version;
    if (!file.i32(&version)) return eof();
    if (version != 150 && !(lenient && version == 200)) {
        set_error("unsupported VOX-format: version " + std::to_string(version));
        return VXRT_E_VOX_VERSION;
    }
    if (file.n < 4 || memcmp(file.p, "MAIN", 4) != 0) { set_error("missing MAIN chunk"); return VXRT_E_VOX_NOMAIN; }
    file.p += 4; file.n -= 4;
    uint32_t content, children;
    Reader in{nullptr, 0};
    if (!file.u32(&content) || !file.u32(&children) || !file.take(size_t(uint64_t(content) + children > file.n ? file.n + 1 : uint64_t(content) + children), &in)) return eof();

    std::vector<Model> models;
    std::map<int32_t, Node> nodes;
    uint32_t palette[256];
    builtin_palette(palette);
    uint8_t kind_of[256] = {0};
    Model pending{};
    bool have_size = false;

    while (in.n != 0) {
        if (in.n < 12) return eof();
        char id[5] = {0};
        memcpy(id, in.p, 4);
        in.p += 4; in.n -= 4;
        uint32_t c_bytes, k_bytes;
        Reader body{nullptr, 0};
        if (!in.u32(&c_bytes) || !in.u32(&k_bytes)) return eof();
        if (uint64_t(c_bytes) + k_bytes > in.n || !in.take(size_t(c_bytes) + k_bytes, &body)) return eof();
        if (strcmp(id, "SIZE") == 0) {
            if (!body.u32(&pending.size[0]) || !body.u32(&pending.size[1]) || !body.u32(&pending.size[2])) return eof();
            have_size = true;
        } else if (strcmp(id, "XYZI") == 0) {
            if (!have_size) { set_error("expected chunk SIZE, found chunk XYZI"); return VXRT_E_VOX_CHUNK; }
            Reader cells{nullptr, 0};
            if (!body.u32(&pending.count) || !body.take(size_t(pending.count) * 4, &cells)) return eof();
            pending.cells = cells.p;
            models.push_back(pending);
            have_size = false;
        } else if (strcmp(id, "RGBA") == 0) {
            for (int i = 1; i <= 255; i++)
                if (!body.u32(&palette[i])) return eof();
        } else if (strcmp(id, "MATL") == 0) {
            uint32_t mid;
            std::map<std::string, std::string> d;
            if (!body.u32(&mid) || !body.dict(&d)) return eof();
            uint8_t kind = 1;
            auto type = d.find("_type");
            if (type != d.end()) {
                if (type->second == "_emit") kind = 2;
                else if (type->second != "_diffuse" && !lenient) {
                    set_error("unsupported material type: " + type->second);
                    return VXRT_E_VOX_MATERIAL;
                }
            }
            auto flux = d.find("_flux");
            if (flux != d.end() && !lenient) {
                if (!is_rust_f32_literal(flux->second)) {
                    set_error("failed to parse value of material key `_flux`");
                    return VXRT_E_VOX_MATERIAL;
                }
            }
            if (mid < 256) kind_of[mid] = kind;
        } else if (strcmp(id, "nTRN") == 0) {
            Node nd;
            nd.kind = 'T';
            int32_t nid, reserved, layer;
            uint32_t frames;
            std::map<std::string, std::string> attrs;
            if (!body.i32(&nid) || !body.dict(&attrs) || !body.i32(&nd.child) || !body.i32(&reserved) || !body.i32(&layer) || !body.u32(&frames)) return eof();
            for (uint32_t f = 0; f < frames; f++) {
                std::map<std::string, std::string> d;
                if (!body.dict(&d)) return eof();
                if (f != 0) continue;  // animation: the first key frame is the scene
                auto t = d.find("_t");
                if (t != d.end()) {
                    long v[3];
                    const char* s = t->second.c_str();
                    char* end = nullptr;
                    for (int a = 0; a < 3; a++) {
                        v[a] = strtol(s, &end, 10);
                        if (end == s) { set_error("bad _t in nTRN"); return VXRT_E_SCENE; }
                        if (v[a] > kMaxTranslation || v[a] < -kMaxTranslation) { set_error("nTRN translation out of range"); return VXRT_E_SCENE; }
                        s = end;
                    }
                    for (int a = 0; a < 3; a++) nd.xf.t[a] = v[a];
                }
                auto r = d.find("_r");
                if (r != d.end() && !rotation_from_byte(uint32_t(strtoul(r->second.c_str(), nullptr, 10)), nd.xf.r)) {
                    set_error("bad _r in nTRN");
                    return VXRT_E_SCENE;
                }
            }
            nodes[nid] = nd;
        } else if (strcmp(id, "nGRP") == 0) {
            Node nd;
            nd.kind = 'G';
            int32_t nid;
            uint32_t count;
            std::map<std::string, std::string> attrs;
            if (!body.i32(&nid) || !body.dict(&attrs) || !body.u32(&count)) return eof();
            for (uint32_t i = 0; i < count; i++) {
                int32_t ch;
                if (!body.i32(&ch)) return eof();
                nd.children.push_back(ch);
            }
            nodes[nid] = nd;
        } else if (strcmp(id, "nSHP") == 0) {
            Node nd;
            nd.kind = 'S';
            int32_t nid;
            uint32_t count;
            std::map<std::string, std::string> attrs;
            if (!body.i32(&nid) || !body.dict(&attrs) || !body.u32(&count)) return eof();
            for (uint32_t i = 0; i < count; i++) {
                int32_t m;
                std::map<std::string, std::string> d;
                if (!body.i32(&m) || !body.dict(&d)) return eof();
                nd.models.push_back(m);
            }
            nodes[nid] = nd;
        }
        // PACK (a count only), LAYR, rOBJ, rCAM, NOTE, IMAP, MATT ...: nothing the renderer uses
    }
    if (models.empty()) { set_error("file holds no model"); return VXRT_E_VOX_NOMODEL; }

    out->voxels.clear();
    memcpy(out->size, models[0].size, sizeof out->size);
    Placer pl{models, nodes, palette, kind_of, lenient, &out->voxels, {INT64_MAX, INT64_MAX, INT64_MAX}, {INT64_MIN, INT64_MIN, INT64_MIN}};
    if ((flags & VXRT_VOX_ALL_MODELS) == 0) {
        // models[0] at its raw cell coordinates, as the reference places it
        Xform raw;
        for (int a = 0; a < 3; a++) raw.t[a] = models[0].size[a] / 2;
        if (int rc = pl.place(models[0], raw)) return rc;
    } else if (nodes.empty()) {
        for (const Model& m : models) {  // pre-scene-graph files: every model at its raw coordinates
            Xform raw;
            for (int a = 0; a < 3; a++) raw.t[a] = m.size[a] / 2;
            if (int rc = pl.place(m, raw)) return rc;
        }
    } else {
        if (int rc = pl.walk(0, Xform(), 0)) return rc;
    }
    if (out->voxels.empty()) { for (int a = 0; a < 3; a++) pl.lo[a] = pl.hi[a] = 0; }
    if ((flags & VXRT_VOX_REBASE) && !out->voxels.empty()) {
        for (int a = 0; a < 3; a++)
            if (pl.hi[a] - pl.lo[a] > 32767) { set_error("re-based scene does not fit 16-bit voxel coordinates"); return VXRT_E_SCENE; }
        for (Voxel& v : out->voxels) { v.x = int16_t(v.x - pl.lo[0]); v.y = int16_t(v.y - pl.lo[2]); v.z = int16_t(v.z - pl.lo[1]); }
        for (int a = 0; a < 3; a++) { pl.hi[a] -= pl.lo[a]; pl.lo[a] = 0; }
    }
    // bounds in the renderer's axes
    bounds_lo[0] = int32_t(pl.lo[0]); bounds_lo[1] = int32_t(pl.lo[2]); bounds_lo[2] = int32_t(pl.lo[1]);
    bounds_hi[0] = int32_t(pl.hi[0]); bounds_hi[1] = int32_t(pl.hi[2]); bounds_hi[2] = int32_t(pl.hi[1]);
    return VXRT_OK;
}

// Context::create_voxels, src/context.rs:838-910.  The reference draws colours from rand::thread_rng(); here
// draw k of the scene is  h = vxrt noise hash of (seed, k)  (noise_value's integer, see vxrt.h), used as
// gen_range(50..=255) = 50 + h % 206 and gen_bool(p) = (h >> 8) * 2^-24 < p, in the reference's call order
// (red, green, blue, emissive) per voxel.
void default_scene(uint32_t seed, std::vector<Voxel>* out) {
    const int radius = 256;
    uint32_t draws = 0;
    auto next = [&]() {
        uint32_t z = (draws++) * 0x9E3779B9u + seed;
        z ^= z >> 16; z *= 0x85EBCA6Bu;
        z ^= z >> 13; z *= 0xC2B2AE35u;
        z ^= z >> 16;
        return z;
    };
    auto colour = [&](float p, Voxel* v) {  // :849-858
        v->r = uint8_t(50 + next() % 206);
        v->g = uint8_t(50 + next() % 206);
        v->b = uint8_t(50 + next() % 206);
        const bool emissive = float(next() >> 8) * (1.0f / 16777216.0f) < p;
        v->m = uint8_t(emissive ? 0x40 : 0x00);
    };
    const int width = 2 * (radius + 1);
    std::vector<int> heights(size_t(width) * width, 0);
    std::vector<uint8_t> known(size_t(width) * width, 0);
    for (int x = -radius; x <= radius; x++)        // height map, :861-876
        for (int z = -radius; z <= radius; z++) {
            const size_t at = size_t(x + radius) + size_t(z + radius) * width;
            known[at] = 1;
            if (x * x + z * z <= radius * radius) {
                const float inside = float(radius * radius) - float(x * x) - float(z * z);
                heights[at] = int(-sqrtf(inside));  // `-(..).sqrt() as i32`: truncation toward zero
            } else {
                heights[at] = 0;
            }
        }
    auto height = [&](int x, int z, int fallback) {  // get_height(..).unwrap_or(curr), :878-888
        if (x < -radius || x > radius || z < -radius || z > radius) return fallback;
        const size_t at = size_t(x + radius) + size_t(z + radius) * width;
        return known[at] ? heights[at] : fallback;
    };
    out->clear();
    for (int x = -radius; x <= radius; x++)        // columns, filling the gaps steep slopes leave, :892-905
        for (int z = -radius; z <= radius; z++) {
            const int curr = height(x, z, 0);
            int low = curr;
            const int nb[4] = {height(x - 1, z, curr), height(x + 1, z, curr), height(x, z - 1, curr), height(x, z + 1, curr)};
            for (int h : nb) low = h < low ? h : low;
            for (int y = low; y <= curr; y++) {
                Voxel v;
                v.x = int16_t(x); v.y = int16_t(y); v.z = int16_t(z);
                colour(0.01f, &v);
                out->push_back(v);
            }
        }
    for (int x = -radius; x <= radius; x++) {      // the strip of light, :907-910
        Voxel v;
        v.x = int16_t(x); v.y = -10; v.z = 0;
        v.m = 0x40; v.r = v.g = v.b = 255;
        out->push_back(v);
    }
}

}  // namespace vxrt
