// scene_host.cpp — see scene_host.h.  Host-only code of libvxrt (no HIP calls in this file).
#include "scene_host.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string_view>

#include "vx_vec.h"

namespace vxrt {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
const std::string& last_error() { return g_last_error; }

// ------------------------------------------------------------------------------------------------
// .vox decoding.  File grammar (MagicaVoxel v150, as consumed by src/vox.rs):
//   "VOX " i32 version | chunk MAIN { [PACK u32 models] (SIZE u32x3, XYZI u32 n, n x {x,y,z,i})* RGBA? MATL* ... }
//   chunk := id[4] u32 content_bytes u32 children_bytes payload[content_bytes + children_bytes]
// ------------------------------------------------------------------------------------------------
namespace {

class Span {
  public:
    Span(const uint8_t* p, size_t n) : p_(p), n_(n) {}
    size_t left() const { return n_; }
    bool has_prefix(std::string_view s) const { return n_ >= s.size() && memcmp(p_, s.data(), s.size()) == 0; }
    bool skip(size_t k) {
        if (k > n_) return false;
        p_ += k; n_ -= k;
        return true;
    }
    bool u32(uint32_t* v) {
        if (n_ < 4) return false;
        *v = uint32_t(p_[0]) | uint32_t(p_[1]) << 8 | uint32_t(p_[2]) << 16 | uint32_t(p_[3]) << 24;
        return skip(4);
    }
    bool sub(size_t k, Span* out) {
        if (k > n_) return false;
        *out = Span(p_, k);
        return skip(k);
    }
    bool text(std::string_view* out) {  // u32 length + bytes (read_str, src/vox.rs:293-296)
        uint32_t len;
        Span s(nullptr, 0);
        if (!u32(&len) || !sub(len, &s)) return false;
        *out = std::string_view(reinterpret_cast<const char*>(s.p_), s.n_);
        return true;
    }
    const uint8_t* data() const { return p_; }

  private:
    const uint8_t* p_;
    size_t n_;
};

struct ChunkView {
    char id[5] = {0, 0, 0, 0, 0};
    Span body{nullptr, 0};
};

int eof_error() {
    set_error("unexpected end of file");
    return VXRT_E_VOX_EOF;
}

int next_chunk(Span* in, ChunkView* out) {
    Span idbytes(nullptr, 0);
    uint32_t content = 0, children = 0;
    if (!in->sub(4, &idbytes)) return eof_error();
    memcpy(out->id, idbytes.data(), 4);
    if (!in->u32(&content) || !in->u32(&children)) return eof_error();
    // the reference adds the two u32 sizes in u32 arithmetic (src/vox.rs:256); a file crafted to
    // overflow that sum is rejected here instead of wrapping
    uint64_t total = uint64_t(content) + uint64_t(children);
    if (total > in->left() || !in->sub(size_t(total), &out->body)) return eof_error();
    return VXRT_OK;
}

// Palette of a file without an RGBA chunk: the one the .vox format description publishes
// (src/vox.rs:103-136 holds the same table): entry 0 = 0, a 6x6x6 cube of {ff,cc,99,66,33,00} levels
// minus black with the top colour byte running fastest, then four 10-step ramps.
void builtin_palette(uint32_t* pal) {
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

// What Rust's `str::parse::<f32>()` accepts (src/vox.rs:93-96 parses `_flux` with it): [+-] then "inf" | "infinity" | "nan"
// (any case) or a decimal number — digits with an optional point (at least one digit on either side) and an optional
// e/E[+-]digits exponent.  No blanks, no hex floats, no "nan(...)", which C's strtof would all accept.
bool is_rust_f32_literal(const std::string& t) {
    size_t i = 0;
    const size_t n = t.size();
    if (i < n && (t[i] == '+' || t[i] == '-')) i++;
    auto lower_eq = [&](const char* w) {
        size_t k = 0;
        for (; w[k]; k++)
            if (i + k >= n || (t[i + k] | 0x20) != w[k]) return false;
        return i + k == n;
    };
    if (lower_eq("inf") || lower_eq("infinity") || lower_eq("nan")) return true;
    size_t digits = 0;
    while (i < n && t[i] >= '0' && t[i] <= '9') { i++; digits++; }
    if (i < n && t[i] == '.') {
        i++;
        while (i < n && t[i] >= '0' && t[i] <= '9') { i++; digits++; }
    }
    if (digits == 0) return false;
    if (i < n && (t[i] == 'e' || t[i] == 'E')) {
        i++;
        if (i < n && (t[i] == '+' || t[i] == '-')) i++;
        size_t ed = 0;
        while (i < n && t[i] >= '0' && t[i] <= '9') { i++; ed++; }
        if (ed == 0) return false;
    }
    return i == n;
}

int decode_vox(const uint8_t* bytes, size_t len, VoxScene* out) {
    Span file(bytes, len);
    if (!file.has_prefix("VOX ")) { set_error("invalid magic number"); return VXRT_E_VOX_MAGIC; }
    file.skip(4);
    uint32_t version;
    if (!file.u32(&version)) return eof_error();
    if (int32_t(version) != 150) {
        set_error("unsupported VOX-format: version " + std::to_string(int32_t(version)));
        return VXRT_E_VOX_VERSION;
    }
    if (!file.has_prefix("MAIN")) { set_error("missing MAIN chunk"); return VXRT_E_VOX_NOMAIN; }
    ChunkView main_chunk;
    if (int rc = next_chunk(&file, &main_chunk)) return rc;
    Span in = main_chunk.body;

    uint32_t models = 1;
    if (in.has_prefix("PACK")) {
        ChunkView pack;
        if (int rc = next_chunk(&in, &pack)) return rc;
        if (!pack.body.u32(&models)) return eof_error();
    }

    // Only model 0 is rendered (src/context.rs:916) but every model must parse (src/vox.rs:35-42).
    Span model0(nullptr, 0);
    uint32_t model0_count = 0;
    for (uint32_t m = 0; m < models; m++) {
        ChunkView size_chunk, xyzi_chunk;
        if (int rc = next_chunk(&in, &size_chunk)) return rc;
        if (strcmp(size_chunk.id, "SIZE") != 0) {
            set_error(std::string("expected chunk SIZE, found chunk ") + size_chunk.id);
            return VXRT_E_VOX_CHUNK;
        }
        if (int rc = next_chunk(&in, &xyzi_chunk)) return rc;
        if (strcmp(xyzi_chunk.id, "XYZI") != 0) {
            set_error(std::string("expected chunk XYZI, found chunk ") + xyzi_chunk.id);
            return VXRT_E_VOX_CHUNK;
        }
        uint32_t dims[3], count;
        if (!size_chunk.body.u32(&dims[0]) || !size_chunk.body.u32(&dims[1]) || !size_chunk.body.u32(&dims[2])) return eof_error();
        if (!xyzi_chunk.body.u32(&count)) return eof_error();
        Span cells(nullptr, 0);
        if (!xyzi_chunk.body.sub(size_t(count) * 4, &cells)) return eof_error();
        if (m == 0) {
            memcpy(out->size, dims, sizeof dims);
            model0 = cells;
            model0_count = count;
        }
    }

    uint32_t palette[256];
    builtin_palette(palette);
    // material kind per MATL id: 0 = absent, 1 = diffuse, 2 = emit.  Ids are u32 in the file; only
    // ids <= 255 can ever be looked up by a u8 colour index (src/context.rs:919).
    uint8_t kind_of[256] = {0};

    while (in.left() != 0) {
        ChunkView ch;
        if (int rc = next_chunk(&in, &ch)) return rc;
        if (strcmp(ch.id, "RGBA") == 0) {
            for (int i = 1; i <= 255; i++)  // file entry i-1 -> palette[i] (src/vox.rs:50-54)
                if (!ch.body.u32(&palette[i])) return eof_error();
        } else if (strcmp(ch.id, "MATL") == 0) {
            uint32_t id, pairs;
            if (!ch.body.u32(&id) || !ch.body.u32(&pairs)) return eof_error();
            uint8_t kind = 1;
            for (uint32_t p = 0; p < pairs; p++) {
                std::string_view key, value;
                if (!ch.body.text(&key) || !ch.body.text(&value)) return eof_error();
                if (key == "_type") {
                    if (value == "_emit") kind = 2;
                    else if (value == "_diffuse") kind = 1;
                    else {
                        set_error("unsupported material type: " + std::string(value));
                        return VXRT_E_VOX_MATERIAL;
                    }
                } else if (key == "_flux") {
                    if (!is_rust_f32_literal(std::string(value))) {
                        set_error("failed to parse value of material key `_flux`");
                        return VXRT_E_VOX_MATERIAL;
                    }
                }
            }
            if (id < 256) kind_of[id] = kind;
        }
        // any other chunk id is skipped ("unknown chunk", src/vox.rs:61)
    }

    if (models == 0) { set_error("file holds no model"); return VXRT_E_VOX_NOMODEL; }
    out->voxels.clear();
    out->voxels.reserve(model0_count);
    const uint8_t* c = model0.data();
    for (uint32_t i = 0; i < model0_count; i++, c += 4) {
        uint8_t colour = c[3];
        if (kind_of[colour] == 0) {
            set_error("voxel colour index " + std::to_string(colour) + " has no MATL entry");
            return VXRT_E_VOX_NOMATL;
        }
        uint32_t rgba = palette[colour];
        Voxel v;
        v.x = c[0]; v.y = c[2]; v.z = c[1];  // renderer axes: (x, z_file, y_file), src/context.rs:927
        v.m = kind_of[colour] == 2 ? 0x40 : 0x00;
        v.r = uint8_t(rgba); v.g = uint8_t(rgba >> 8); v.b = uint8_t(rgba >> 16);
        out->voxels.push_back(v);
    }
    return VXRT_OK;
}

// ------------------------------------------------------------------------------------------------
// Octree
// ------------------------------------------------------------------------------------------------
static uint32_t ceil_log2_u16(uint32_t v) {  // u16::next_power_of_two().trailing_zeros()
    uint32_t bits = 0;
    while ((1u << bits) < v) bits++;
    return bits;
}

int build_octree(const Voxel* voxels, size_t n, Octree* out) {
    // depth (Context::voxel_depth, src/context.rs:813-834): the root cube [-2^d, 2^d)^3 must hold
    // every coordinate; min uses |min|, max uses |max|+1.
    uint32_t depth = 0;
    if (n != 0) {
        int lo = voxels[0].x, hi = voxels[0].x;
        for (size_t i = 0; i < n; i++) {
            const int c[3] = {voxels[i].x, voxels[i].y, voxels[i].z};
            for (int v : c) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
        }
        uint32_t dlo = ceil_log2_u16(uint32_t(abs(lo)) & 0xffffu);
        uint32_t dhi = ceil_log2_u16((uint32_t(abs(hi)) + 1u) & 0xffffu);
        depth = dlo > dhi ? dlo : dhi;
    }
    if (depth > 15) { set_error("octree depth > 15"); return VXRT_E_SCENE; }

    std::vector<int32_t>& w = out->words;
    w.clear();
    w.reserve(5 + 8 * (n / 2 + 16));
    const float header[5] = {0.0f, 0.0f, 0.0f, float(1u << depth), 1.0f};
    w.resize(5 + 8, 0);
    memcpy(w.data(), header, sizeof header);

    for (size_t i = 0; i < n; i++) {
        const Voxel& v = voxels[i];
        int32_t cx = 0, cy = 0, cz = 0;
        size_t node = 0;
        for (int32_t half = int32_t(1) << depth;; half >>= 1) {
            // slot bit set <=> coordinate on the >= side of the node centre (src/context.rs:726-729)
            const int32_t bx = v.x >= cx, by = v.y >= cy, bz = v.z >= cz;
            int32_t& slot = w[5 + 8 * node + size_t(4 * bx + 2 * by + bz)];
            if (half == 1) {
                slot = int32_t(0x80000000u | uint32_t(v.m & 0x7f) << 24 | uint32_t(v.r) << 16 | uint32_t(v.g) << 8 | v.b);
                break;
            }
            if (slot < 0) { set_error("voxel list would split a leaf"); return VXRT_E_SCENE; }
            size_t child;
            if (slot == 0) {
                child = (w.size() - 5) / 8;
                if (child >= (size_t(1) << 26)) { set_error("too many octree nodes"); return VXRT_E_SCENE; }
                slot = int32_t(child);   // `slot` dangles after the resize below; not used again
                w.resize(w.size() + 8, 0);
            } else {
                child = size_t(slot);
            }
            const int32_t q = half / 2;  // child centre = c -+ half/2 (src/context.rs:749-753)
            cx += bx ? q : -q; cy += by ? q : -q; cz += bz ? q : -q;
            node = child;
        }
    }
    out->depth = depth;
    return VXRT_OK;
}

// ------------------------------------------------------------------------------------------------
// Camera, noise, procedural scenes
// ------------------------------------------------------------------------------------------------
CameraBasis camera_axis_scaled(const float dir[3], float fov, uint32_t width, uint32_t height) {
    f3 forward = norm3(mk3(dir[0], dir[1], dir[2]));
    f3 right = norm3(cross3(mk3(0.0f, 1.0f, 0.0f), forward));
    f3 up = cross3(forward, right);
    float fov_scale = vx_tan(fov / 2.0f);
    float w = float(width), h = float(height);
    f3 fr = ((-w / 2.0f) * right + (h / 2.0f) * up) + ((h / 2.0f) / fov_scale) * forward;
    CameraBasis b;
    b.right[0] = right.x; b.right[1] = right.y; b.right[2] = right.z;
    b.up[0] = up.x; b.up[1] = up.y; b.up[2] = up.z;
    b.forward_ray[0] = fr.x; b.forward_ray[1] = fr.y; b.forward_ray[2] = fr.z;
    return b;
}

float noise_value(uint32_t seed, uint32_t index) {
    uint32_t z = index * 0x9E3779B9u + seed;
    z ^= z >> 16; z *= 0x85EBCA6Bu;
    z ^= z >> 13; z *= 0xC2B2AE35u;
    z ^= z >> 16;
    return float(z >> 8) * (1.0f / 16777216.0f);
}

bool menger_solid(uint32_t level, uint32_t x, uint32_t y, uint32_t z) {
    for (uint32_t l = 0; l < level; l++) {
        uint32_t ones = (x % 3 == 1) + (y % 3 == 1) + (z % 3 == 1);
        if (ones >= 2) return false;
        x /= 3; y /= 3; z /= 3;
    }
    return true;
}

}  // namespace vxrt
