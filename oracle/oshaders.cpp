// ORACLE — test infrastructure only.  Nothing in the product path may include, link or call this.
//
// oshaders.cpp: CPU restatement of the three compute shaders the reference dispatches per frame
// (src/context.rs:2014-2038):
//   * shaders/voxels.comp   — octree first-hit traversal + path tracing     (orc_trace)
//   * shaders/temporal.comp — reprojection + exponential blending           (orc_temporal)
//   * shaders/denoise.comp  — (2r+1)^2 cross-bilateral filter               (orc_denoise)
// Statement by statement, in the shaders' operation order, on the numeric contract of
// include/vxrt_detmath.h.  Parity status: PINNED BY THE REFERENCE'S COMPILED SHADERS, up to the driver-defined operations
// listed below.  The reference ships no tests or golden images and its host cannot be built here (no Rust, no Vulkan — SURVEY.md
// §8c), but it does ship the three shaders compiled (shaders/*.comp.spv, what it hands to the GPU): ospirv.cpp executes those
// modules instruction by instruction with U2-U8 bound to the functions below, and this restatement gives the same bits on every
// scene file, a moving camera, every denoise radius, NaN / inf G-buffers, 0 * inf rays and the trip cap
// (tests/test_oracle_spirv_exec.py; outputs of the modules as fixtures: tests/golden/spirv_exec/).  What stays open is what
// SPIR-V itself leaves to a driver — U1-U8 — whose width tests/test_oracle_builtin_sensitivity.py measures.
//
// Places where GLSL leaves behaviour undefined and this restatement picks one:
//   U1  cast_bounded_ray's iteration cap returns true without writing `normal`
//       (voxels.comp:166-169) -> normal = (0,0,0).
//   U2  pow(v, 2) (voxels.comp:380, denoise.comp:39,40,75) -> v*v (see vxrt_detmath.h).
//   U3  first frame: wgpu zero-initialised history + all-zero old camera make inverse() singular
//       and every comparison false (temporal.comp:82-92) -> "no history => blending = 1".
//   U4  texture() with the Linear sampler (src/context.rs:980-989): ideal bilinear with weights
//       quantised to 8 fractional bits (Vulkan subTexelPrecisionBits on the hardware the reference
//       ran on), clamp-to-edge; a texel whose quantised weight is 0 is not read.
//   U5  inverse(mat4) (temporal.comp:82): affine inverse by adjugate/determinant in binary64,
//       rounded once to binary32, hoisted out of the pixel loop (the matrix is per-frame).
//   U6  sin cos tan exp log pow sqrt normalize: include/vxrt_detmath.h.   U7  float -> int out of range: saturating.
//   U8  dot and matrix x vector: summed left to right.   U9  no contraction of a * b + c (a driver may fuse: the compiled
//       modules carry no NoContraction decoration).
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "oracle.h"

namespace orc {

#if ORC_ALT_BUILTINS
int g_alt_mask = 0;   // ovec.h: which driver-defined choices are made the other way (orc_set_alt)
#endif

static const float ALMOST_INFINITY = 1073741824.0f;  // float(1 << 30), voxels.comp:8
static const int32_t LEAF_BIT = (int32_t)0x80000000u;  // voxels.comp:10
static const int32_t EMMITANCE_BIT = 1 << 30;          // voxels.comp:11
static const int MAX_DEPTH = 16;                       // voxels.comp:3

// voxels.comp:73-90
static bool ray_cube_intersection(V3 origin, V3 inv_dir, V3 center, float half_size, float* entry, float* exit) {
    V3 signum = vsign(inv_dir);
    V3 entry_planes = center - half_size * signum;
    V3 exit_planes = center + half_size * signum;
    V3 entries = (entry_planes - origin) * inv_dir;
    V3 exits = (exit_planes - origin) * inv_dir;
    *entry = vx_max(vx_max(entries.x, entries.y), entries.z);
    *exit = vx_min(vx_min(exits.x, exits.y), exits.z);
    return *exit >= 0.0f && *entry < *exit;
}

// voxels.comp:92-95
static V3 octant_center(V3 center, float size, uint32_t octant) {
    V3 delta = v3((float)((octant >> 2) & 1), (float)((octant >> 1) & 1), (float)(octant & 1));
    return center + (0.5f * size) * (delta - v3s(0.5f));
}

// voxels.comp:119-125
static uint32_t current_octant(V3 position, V3 center) {
    V3 delta = position - center;
    uint32_t dx = delta.x > 0.0f ? 4 : 0;
    uint32_t dy = delta.y > 0.0f ? 2 : 0;
    uint32_t dz = delta.z > 0.0f ? 1 : 0;
    return dx + dy + dz;
}

thread_local long long g_iterations = 0;  // diagnostic: octree steps taken by this thread
thread_local int* g_phase_steps = nullptr;  // diagnostic: steps of each successive ray of the current pixel
thread_local float* g_ray_log = nullptr;    // diagnostic: 12 floats per cast of the current pixel (origin, dir, hit, time, node bits, normal)
thread_local int g_ray_log_n = 0;
static void log_ray(V3 o, V3 d, bool hit, const orc::Hit& h) {
    if (!g_ray_log || g_ray_log_n >= 32) return;
    float* r = g_ray_log + 12 * g_ray_log_n++;
    r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z; r[6] = hit ? 1.0f : 0.0f; r[7] = h.time;
    memcpy(&r[8], &h.node, 4); r[9] = h.normal.x; r[10] = h.normal.y; r[11] = h.normal.z;
}
thread_local int g_phase = 0;
thread_local uint8_t* g_branch_log = nullptr;  // diagnostic: one byte per trip of the walk — 0 advance, 1 descend, 2 pop, 3 the trip that ends the ray
static inline void log_branch(uint8_t kind) { if (g_branch_log) *g_branch_log++ = kind; }

// voxels.comp:134-247
// The octree buffer of voxels.comp:58-63 (header + nodes[]) as the walk sees it.  `lazy` != null: nodes[] is not stored but
// materialised slot by slot from a procedural voxel predicate on first touch (oprocedural.cpp) — the same words
// create_octree (src/context.rs:710-796) would have produced for that voxel set, up to node numbering.
Scene scene_of(const int32_t* octree) {
    float hdr[5];
    memcpy(hdr, octree, sizeof hdr);
    return Scene{v3(hdr[0], hdr[1], hdr[2]), hdr[3], octree + 5, nullptr};
}
static inline int32_t node_word(const Scene& sc, int32_t node, uint32_t octant) {
    return sc.lazy ? lazy_fetch(sc.lazy, node, octant) : sc.nodes[8 * node + octant];
}

bool cast_bounded_ray(const int32_t* octree, V3 ray_origin, V3 ray_dir, float max_distance, Hit* out) {
    return cast_bounded_ray(scene_of(octree), ray_origin, ray_dir, max_distance, out);
}

// voxels.comp:134-247
bool cast_bounded_ray(const Scene& sc, V3 ray_origin, V3 ray_dir, float max_distance, Hit* out) {
    const V3 root_center = sc.root_center;
    const float root_size = sc.root_size;
    struct Frame { int32_t node; uint32_t octant; } stack[MAX_DEPTH];

    out->normal = v3s(0.0f);  // U1
    out->node = 0;
    out->time = 0.0f;
    out->iterations = 0;

    const uint32_t dir_mask = (ray_dir.x < 0.0f ? 4 : 0) | (ray_dir.y < 0.0f ? 2 : 0) | (ray_dir.z < 0.0f ? 1 : 0);
    const V3 ray_inv_dir = 1.0f / ray_dir;

    float root_entry, root_exit;
    bool intersect = ray_cube_intersection(ray_origin, ray_inv_dir, root_center, 0.5f * root_size, &root_entry, &root_exit);
    if (!intersect) return false;

    int top = 0;
    float time = vx_max(0.0f, root_entry);
    int32_t node = 0;
    float exit = root_exit;
    V3 center = root_center;
    float size = root_size;
    uint32_t octant = current_octant(ray_origin + ray_dir * time, center);

    int iterations = 0;
    for (;;) {
        iterations++;
        g_iterations++;
        out->iterations = iterations;
        out->time = time;
        if (iterations >= 2048) {
            out->node = LEAF_BIT;
            log_branch(3);
            return true;
        }
        if (time > max_distance) { log_branch(3); return false; }

        int32_t value = node_word(sc, node, octant);

        if (value < 0) {
            out->node = value;
            V3 hit = ray_origin + time * ray_dir;
            V3 oc = octant_center(center, size, octant);
            V3 distances = vabs(hit - oc);
            float max_dist = vx_max(vx_max(distances.x, distances.y), distances.z);
            V3 mask = v3(distances.x == max_dist ? 1.0f : 0.0f, distances.y == max_dist ? 1.0f : 0.0f,
                         distances.z == max_dist ? 1.0f : 0.0f);
            out->normal = mask * (-vsign(ray_dir));
            log_branch(3);
            return true;
        }

        V3 t_mid = (center - ray_origin) * ray_inv_dir;
        uint32_t directional_octant = octant ^ dir_mask;
        V3 mid_intersect = v3((directional_octant & 4) ? ALMOST_INFINITY : t_mid.x,
                              (directional_octant & 2) ? ALMOST_INFINITY : t_mid.y,
                              (directional_octant & 1) ? ALMOST_INFINITY : t_mid.z);
        float next_time = vx_min(vx_min(mid_intersect.x, mid_intersect.y), mid_intersect.z);
        uint32_t transition = (mid_intersect.x == next_time) ? 4 : ((mid_intersect.y == next_time) ? 2 : ((mid_intersect.z == next_time) ? 1 : 0));
        uint32_t next_octant = octant ^ transition;
        bool has_next = next_time <= exit && transition != 0 && (directional_octant & transition) == 0;

        log_branch((uint8_t)((value > 0 ? 1 : (has_next ? 0 : 2)) | (top << 2)));   // bits 2..: the node's level (pushes so far)
        if (value > 0) {
            if (top >= MAX_DEPTH) return false;  // GLSL would write out of bounds; cannot happen for depth <= 15
            stack[top].node = has_next ? node : -1;
            stack[top].octant = octant | (next_octant << 3);
            top++;

            node = value;
            center = octant_center(center, size, octant);
            size *= 0.5f;
            octant = current_octant(ray_origin + ray_dir * time, center);

            float octant_entry, octant_exit;
            ray_cube_intersection(ray_origin, ray_inv_dir, center, 0.5f * size, &octant_entry, &octant_exit);
            time = vx_max(time, octant_entry);
            exit = octant_exit;
        } else if (has_next) {
            octant = next_octant;
            time = next_time;
        } else {
            do {
                if (top == 0) return false;
                top--;
                node = stack[top].node;
                size *= 2.0f;
                uint32_t parent_octant = stack[top].octant & 0x7;
                center = octant_center(center, size, ~parent_octant);
            } while (node == -1);

            time = exit;
            float new_entry;
            ray_cube_intersection(ray_origin, ray_inv_dir, center, 0.5f * size, &new_entry, &exit);
            octant = (stack[top].octant >> 3) & 0x7;
        }
    }
}

// voxels.comp:253-266
static V3 node_color(int32_t node) {
    float r = (float)((node >> 16) & 0xff), g = (float)((node >> 8) & 0xff), b = (float)(node & 0xff);
    return v3(r, g, b) / 255.0f;
}
static V3 node_emmitance(int32_t node, float emit_strength) {
    float e = (node & EMMITANCE_BIT) != 0 ? 1.0f : 0.0f;
    float r = (float)((node >> 16) & 0xff), g = (float)((node >> 8) & 0xff), b = (float)(node & 0xff);
    return ((e * emit_strength) * v3(r, g, b)) / 255.0f;
}

static const uint32_t BLUE_NOISE_SIZE = 128, BLUE_NOISE_COUNT = 512;
static const uint32_t BLUE_NOISE_BUFFER_SIZE = BLUE_NOISE_SIZE * BLUE_NOISE_SIZE * BLUE_NOISE_COUNT;

struct Rng {  // voxels.comp:268-275
    uint32_t index;
    const float* noise;
    float rand() {
        index = (index + BLUE_NOISE_SIZE * BLUE_NOISE_SIZE) % BLUE_NOISE_BUFFER_SIZE;
        return noise[index];
    }
};

// voxels.comp:277-287
static V3 random_hemisphere(V3 normal, Rng& rng) {
    float phi = (2.0f * 3.14159265358979f) * rng.rand();
    V3 d;
    d.x = 2.0f * rng.rand() - 1.0f;
    float plane_radius = vx_sqrt(1.0f - d.x * d.x);
    d.y = plane_radius * vx_cos(phi);
    d.z = plane_radius * vx_sin(phi);
    d = d - normal * vx_min(0.0f, 2.0f * dot(normal, d));
    return d;
}

static V3 pixel_ray_dir(const float* right, const float* up, const float* fwd, int x, int y) {
    // normalize(coord.x * camera_right - coord.y * camera_up + camera_forward)  voxels.comp:299-303
    V3 r = v3(right[0], right[1], right[2]), u = v3(up[0], up[1], up[2]), f = v3(fwd[0], fwd[1], fwd[2]);
    return normalize(((float)x * r - (float)y * u) + f);
}

// voxels.comp:289-397 for one pixel; out_* are rgba32f texels.  Returns the number of
// cast_bounded_ray invocations (the "ray" of the Mrays/s metric).
static int trace_pixel(const Scene& octree, const float* noise, const OrcUniforms& u, int max_bounces, int px,
                       int py, float* out_color, float* out_nd, float* out_albedo) {
    Rng rng;
    rng.noise = noise;
    rng.index = (uint32_t)px % BLUE_NOISE_SIZE + ((uint32_t)py % BLUE_NOISE_SIZE) * BLUE_NOISE_SIZE +
                (u.frame_number % BLUE_NOISE_COUNT) * BLUE_NOISE_SIZE * BLUE_NOISE_SIZE;
    int rays = 0;

    V3 first_normal = v3s(ALMOST_INFINITY);
    float first_time = -1.0f;
    int32_t first_node = 0xffffff;

    V3 sun_dir = v3(vx_cos(u.sun_yaw) * vx_cos(u.sun_pitch), -vx_sin(u.sun_pitch), vx_sin(u.sun_yaw) * vx_cos(u.sun_pitch));
    V3 sun_color = u.sun_strength * v3(u.sun_color[0], u.sun_color[1], u.sun_color[2]);  // SUN_COLOR, voxels.comp:6
    V3 sky = v3(u.sky_color[0], u.sky_color[1], u.sky_color[2]);

    V3 ray_origin = v3(u.camera_origin[0], u.camera_origin[1], u.camera_origin[2]);
    V3 ray_dir = pixel_ray_dir(u.camera_right, u.camera_up, u.camera_forward, px, py);

    V3 sample_color = v3s(0.0f);
    V3 blending_factor = v3s(1.0f);
    uint32_t ambient_rays = 1;

    for (int bounce = 0; bounce < max_bounces; bounce++) {
        Hit h;
        rays++;
        bool primary_hit = cast_bounded_ray(octree, ray_origin, ray_dir, ALMOST_INFINITY, &h);
        if (g_phase_steps && g_phase < 32) g_phase_steps[g_phase++] = h.iterations;
        log_ray(ray_origin, ray_dir, primary_hit, h);
        if (primary_hit) {
            V3 normal = h.normal;
            V3 hit_pos = ray_origin + ray_dir * h.time;
            V3 color = bounce == 0 ? v3s(1.0f) : node_color(h.node);
            V3 emmitance = node_emmitance(h.node, u.emit_strength);
            if (bounce == 0) {
                first_node = h.node;
                first_normal = normal;
                first_time = h.time;
            }
            if (rng.rand() < u.specularity) {
                V3 reflect_dir = normalize(reflect(ray_dir, normal));
                sample_color = sample_color + emmitance * blending_factor;
                blending_factor = blending_factor * ((2.0f * color) * dot(reflect_dir, normal));
                ray_origin = hit_pos + 1e-5f * normal;
                ray_dir = reflect_dir;
            } else {
                if (u.sun_strength > 0.0f) {
                    float r0 = rng.rand(), r1 = rng.rand(), r2 = rng.rand();
                    V3 rand_dir = v3(r0, r1, r2);
                    V3 up_dir = normalize(cross(rand_dir, sun_dir));
                    V3 right_dir = normalize(cross(sun_dir, up_dir));
                    float dx = 2.0f * rng.rand() - 1.0f;
                    float dy = 2.0f * rng.rand() - 1.0f;
                    V3 light_dir = normalize(sun_dir) + (dx * right_dir + dy * up_dir) * u.sun_size;
                    Hit sh;
                    rays++;
                    bool sun_obstructed = cast_bounded_ray(octree, hit_pos + 1e-5f * normal, normalize(-light_dir), ALMOST_INFINITY, &sh);
                    if (g_phase_steps && g_phase < 32) g_phase_steps[g_phase++] = sh.iterations;
                    log_ray(hit_pos + 1e-5f * normal, normalize(-light_dir), sun_obstructed, sh);
                    ambient_rays++;
                    if (!sun_obstructed) {
                        sample_color = sample_color + ((sun_color * color) * blending_factor) * vx_max(0.0f, dot(normal, normalize(-light_dir)));
                    }
                }
                V3 reflect_dir = random_hemisphere(normal, rng);
                sample_color = sample_color + emmitance * blending_factor;
                blending_factor = blending_factor * (color * dot(normal, reflect_dir));
                ray_origin = hit_pos + 1e-5f * normal;
                ray_dir = reflect_dir;
            }
        } else {
            if (bounce == 0) {
                blending_factor = v3s(1.0f);
                float sun_power = vx_pow(vx_max(0.0f, dot(ray_dir, normalize(-sun_dir))), 1.0f / (u.sun_size * u.sun_size));  // U2
                sample_color = sample_color + (sky + sun_color * sun_power) * blending_factor;
            } else {
                sample_color = sample_color + sky * blending_factor;
            }
            break;
        }
    }

    V3 out = sample_color / (float)ambient_rays;
    V3 albedo = (first_node & EMMITANCE_BIT) == 0 ? node_color(first_node) : v3s(1.0f);
    out_color[0] = out.x; out_color[1] = out.y; out_color[2] = out.z; out_color[3] = 1.0f;
    out_nd[0] = first_normal.x; out_nd[1] = first_normal.y; out_nd[2] = first_normal.z; out_nd[3] = first_time;
    out_albedo[0] = albedo.x; out_albedo[1] = albedo.y; out_albedo[2] = albedo.z;
    memcpy(&out_albedo[3], &first_node, 4);  // intBitsToFloat(first_node)
    return rays;
}

template <class F>
static void parallel_rows(int y0, int y1, int nthreads, F f) {
    if (nthreads <= 1) { for (int y = y0; y < y1; y++) f(y); return; }
    std::atomic<int> next(y0);
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++)
        pool.emplace_back([&] { for (;;) { int y = next.fetch_add(1); if (y >= y1) break; f(y); } });
    for (auto& th : pool) th.join();
}

// affine inverse of [R U F O; 0 0 0 1] (U5): rows of A^-1 and t = -A^-1 O, as 12 floats.
void affine_inverse(const float* R, const float* U, const float* F, const float* O, float inv[12]) {
#if ORC_ALT_BUILTINS
    if (g_alt_mask & 8) {   // U5 the other way: the same adjugate / determinant, every operation in binary32
        float a = R[0], b = U[0], c = F[0], d = R[1], e = U[1], f = F[1], g = R[2], h = U[2], i = F[2];
        float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
        float det = (a * A + b * B) + c * C;
        float m[9] = {A, -(b * i - c * h), b * f - c * e, B, a * i - c * g, -(a * f - c * d), C, -(a * h - b * g), a * e - b * d};
        for (int r = 0; r < 3; r++) {
            float r0 = m[3 * r] / det, r1 = m[3 * r + 1] / det, r2 = m[3 * r + 2] / det;
            inv[4 * r] = r0; inv[4 * r + 1] = r1; inv[4 * r + 2] = r2;
            inv[4 * r + 3] = -((r0 * O[0] + r1 * O[1]) + r2 * O[2]);
        }
        return;
    }
#endif
    double a = R[0], b = U[0], c = F[0], d = R[1], e = U[1], f = F[1], g = R[2], h = U[2], i = F[2];
    double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    double det = a * A + b * B + c * C;
    double m[9] = {A, -(b * i - c * h), b * f - c * e, B, a * i - c * g, -(a * f - c * d), C, -(a * h - b * g), a * e - b * d};
    for (int r = 0; r < 3; r++) {
        double r0 = m[3 * r] / det, r1 = m[3 * r + 1] / det, r2 = m[3 * r + 2] / det;
        inv[4 * r] = (float)r0; inv[4 * r + 1] = (float)r1; inv[4 * r + 2] = (float)r2;
        inv[4 * r + 3] = (float)(-(r0 * (double)O[0] + r1 * (double)O[1] + r2 * (double)O[2]));
    }
}

// struct Tex (oracle.h): rgba32f image + the Linear/ClampToEdge sampler of src/context.rs:980-989 (U4)
void Tex::fetch(int x, int y, float* o) const {
    x = x < 0 ? 0 : (x >= w ? w - 1 : x);
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    memcpy(o, data + 4 * ((size_t)y * w + x), 16);
}
void Tex::sample(float u, float v, float* o) const {
    {
        float fx = u * (float)w - 0.5f, fy = v * (float)h - 0.5f;
        float x0 = vx_floor(fx), y0 = vx_floor(fy);
        float ax = vx_floor((fx - x0) * 256.0f + 0.5f) / 256.0f, ay = vx_floor((fy - y0) * 256.0f + 0.5f) / 256.0f;
#if ORC_ALT_BUILTINS
        if (g_alt_mask & 4) { ax = fx - x0; ay = fy - y0; }   // U4 the other way: weights at full precision
#endif
        float t00[4], t10[4], t01[4], t11[4];
        int ix0 = vx_f2i(x0), iy0 = vx_f2i(y0);
        ix0 = ix0 > w ? w : (ix0 < -2 ? -2 : ix0);
        iy0 = iy0 > h ? h : (iy0 < -2 ? -2 : iy0);
        fetch(ix0, iy0, t00); fetch(ix0 + 1, iy0, t10);
        fetch(ix0, iy0 + 1, t01); fetch(ix0 + 1, iy0 + 1, t11);
        for (int k = 0; k < 4; k++) {
            // a texel whose quantised weight is 0 is not read at all (so a NaN/inf neighbour cannot leak in)
            float top = ax == 0.0f ? t00[k] : (ax == 1.0f ? t10[k] : (t00[k] * (1.0f - ax) + t10[k] * ax));
            float bot = ax == 0.0f ? t01[k] : (ax == 1.0f ? t11[k] : (t01[k] * (1.0f - ax) + t11[k] * ax));
            o[k] = ay == 0.0f ? top : (ay == 1.0f ? bot : (top * (1.0f - ay) + bot * ay));
        }
    }
}

}  // namespace orc

using namespace orc;

extern "C" {

#if ORC_ALT_BUILTINS
// liboracle_alt.so only: bits of ovec.h's table; returns the previous mask
int orc_set_alt(int mask) { int old = g_alt_mask; g_alt_mask = mask; return old; }
#endif

// Path-trace rows [y0,y1) x columns [x0,x1) of a width x height frame (voxels.comp main(), one
// thread per pixel; pixel coordinates are frame-absolute so a crop equals the same region of the
// full frame).  Output images are crop-sized, row-major rgba32f.  Returns total rays cast.
long long orc_trace(const int32_t* octree, const float* noise, const OrcUniforms* u, int max_bounces, int x0, int y0,
                    int x1, int y1, float* color, float* normal_depth, float* albedo, int nthreads) {
    int cw = x1 - x0;
    std::atomic<long long> rays(0);
    parallel_rows(y0, y1, nthreads, [&](int y) {
        long long r = 0;
        for (int x = x0; x < x1; x++) {
            size_t o = 4 * ((size_t)(y - y0) * cw + (x - x0));
            r += trace_pixel(scene_of(octree), noise, *u, max_bounces, x, y, color + o, normal_depth + o, albedo + o);
        }
        rays += r;
    });
    return rays.load();
}

// Diagnostic: per pixel 17 ints = total octree steps, then the steps of each successive ray (<= 16) — for a crop — used to reason about
// the GPU kernels' critical path, not by any parity test.
void orc_trace_steps(const int32_t* octree, const float* noise, const OrcUniforms* u, int max_bounces, int x0, int y0,
                     int x1, int y1, int32_t* steps, int nthreads) {
    int cw = x1 - x0;
    std::vector<float> scratch((size_t)12);
    parallel_rows(y0, y1, nthreads, [&](int y) {
        float c[4], n[4], a[4];
        for (int x = x0; x < x1; x++) {
            long long before = g_iterations;
            int32_t* row = steps + ((size_t)(y - y0) * cw + (x - x0)) * 17;
            for (int k = 0; k < 17; k++) row[k] = 0;
            g_phase_steps = row + 1;
            g_phase = 0;
            trace_pixel(scene_of(octree), noise, *u, max_bounces, x, y, c, n, a);
            g_phase_steps = nullptr;
            row[0] = (int32_t)(g_iterations - before);
        }
    });
}

// Diagnostic (tests/sim_schedule.py: branch_coherence): the branch every trip of every ray of a pixel takes, one byte per trip
// (bits 0-1: 0 advance, 1 descend, 2 pop, 3 the trip that ends the ray; bits 2-7: the level of the node the trip is in), the pixel's rays one after another: pixel (x, y) writes its
// orc_trace_steps total (row[0]) bytes at flat + offsets[pixel].
void orc_trace_branches(const int32_t* octree, const float* noise, const OrcUniforms* u, int max_bounces, int x0, int y0, int x1, int y1,
                        const int64_t* offsets, uint8_t* flat, int nthreads) {
    int cw = x1 - x0;
    parallel_rows(y0, y1, nthreads, [&](int y) {
        float c[4], n[4], a[4];
        for (int x = x0; x < x1; x++) {
            g_branch_log = flat + offsets[(size_t)(y - y0) * cw + (x - x0)];
            trace_pixel(scene_of(octree), noise, *u, max_bounces, x, y, c, n, a);
            g_branch_log = nullptr;
        }
    });
}

// Diagnostic: every cast of one pixel's path: 12 floats per ray (origin, dir, hit, time, node bits, normal); returns the ray count.
int orc_trace_pixel_log(const int32_t* octree, const float* noise, const OrcUniforms* u, int max_bounces, int x, int y, float* log) {
    float c[4], n[4], a[4];
    g_ray_log = log;
    g_ray_log_n = 0;
    trace_pixel(scene_of(octree), noise, *u, max_bounces, x, y, c, n, a);
    g_ray_log = nullptr;
    return g_ray_log_n;
}

// Batch of single rays through cast_bounded_ray — for traversal unit tests and the DDA cross-check.
void orc_cast_rays(const int32_t* octree, const float* origins, const float* dirs, size_t n, float max_distance,
                   uint8_t* hit, float* time, int32_t* node, float* normal, int32_t* iterations) {
    for (size_t i = 0; i < n; i++) {
        Hit h;
        bool ok = cast_bounded_ray(octree, v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]),
                                   v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), max_distance, &h);
        hit[i] = ok; time[i] = h.time; node[i] = h.node;
        normal[3 * i] = h.normal.x; normal[3 * i + 1] = h.normal.y; normal[3 * i + 2] = h.normal.z;
        if (iterations) iterations[i] = h.iterations;
    }
}

// ---- the same two entry points over the implicit octree of a procedural scene (oprocedural.cpp; BASELINE config 5) ----
// orc_trace for the level-`level` Menger sponge clipped to [0, clip)^3 with leaf words orc_procedural_leaf_word(...).
long long orc_trace_menger(uint32_t level, uint32_t clip, const uint8_t* mrgb, uint32_t emissive_period, const float* noise,
                           const OrcUniforms* u, int max_bounces, int x0, int y0, int x1, int y1, float* color, float* normal_depth,
                           float* albedo, int nthreads) {
    int cw = x1 - x0;
    std::atomic<long long> rays(0);
    LazyPool* pool = lazy_pool(level, clip, mrgb, emissive_period);
    parallel_rows(y0, y1, nthreads, [&](int y) {
        LazyTree* tree = lazy_acquire(pool);
        const Scene sc = lazy_scene(tree);
        long long r = 0;
        for (int x = x0; x < x1; x++) {
            size_t o = 4 * ((size_t)(y - y0) * cw + (x - x0));
            r += trace_pixel(sc, noise, *u, max_bounces, x, y, color + o, normal_depth + o, albedo + o);
        }
        rays += r;
        lazy_release(pool, tree);
    });
    return rays.load();
}

void orc_cast_rays_menger(uint32_t level, uint32_t clip, const uint8_t* mrgb, uint32_t emissive_period, const float* origins,
                          const float* dirs, size_t n, float max_distance, uint8_t* hit, float* time, int32_t* node, float* normal,
                          int32_t* iterations, int nthreads) {
    LazyPool* pool = lazy_pool(level, clip, mrgb, emissive_period);
    const int chunks = (int)((n + 4095) / 4096);
    parallel_rows(0, chunks, nthreads, [&](int c) {
        LazyTree* tree = lazy_acquire(pool);
        const Scene sc = lazy_scene(tree);
        const size_t a = (size_t)c * 4096, b = a + 4096 < n ? a + 4096 : n;
        for (size_t i = a; i < b; i++) {
            Hit h;
            bool ok = cast_bounded_ray(sc, v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]),
                                       v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), max_distance, &h);
            hit[i] = ok; time[i] = h.time; node[i] = h.node;
            normal[3 * i] = h.normal.x; normal[3 * i + 1] = h.normal.y; normal[3 * i + 2] = h.normal.z;
            if (iterations) iterations[i] = h.iterations;
        }
        lazy_release(pool, tree);
    });
}

// temporal.comp main() over a full width x height frame.
//   sampled_color, new_nd : this frame's trace outputs;  old_color (rgb + blending in .a), old_nd : history.
//   cam / old_cam : 16 floats each = origin, right, up, forward as vec4 (first 64 bytes of Uniforms).
//   has_history == 0 restates the reference's first frame (U3).
void orc_temporal(const float* sampled_color, const float* new_nd, const float* old_color, const float* old_nd,
                  const float* cam, const float* old_cam, const OrcTemporal* tu, int has_history, int width,
                  int height, float* new_color, int nthreads) {
    float inv[12] = {0};
    if (has_history) affine_inverse(old_cam + 4, old_cam + 8, old_cam + 12, old_cam, inv);
    Tex tex_color{old_color, width, height}, tex_nd{old_nd, width, height};
    V3 cam_o = v3(cam[0], cam[1], cam[2]);
    V3 ocr = v3(old_cam[4], old_cam[5], old_cam[6]), ocu = v3(old_cam[8], old_cam[9], old_cam[10]),
       ocf = v3(old_cam[12], old_cam[13], old_cam[14]), oco = v3(old_cam[0], old_cam[1], old_cam[2]);
    parallel_rows(0, height, nthreads, [&](int y) {
        for (int x = 0; x < width; x++) {
            size_t o = 4 * ((size_t)y * width + x);
            V3 color = v3(sampled_color[o], sampled_color[o + 1], sampled_color[o + 2]);
            V3 normal = v3(new_nd[o], new_nd[o + 1], new_nd[o + 2]);
            float depth = new_nd[o + 3];
            V3 ray_dir = pixel_ray_dir(cam + 4, cam + 8, cam + 12, x, y);
            V3 world_pos = cam_o + depth * ray_dir;

            float old_c[4] = {0, 0, 0, 0};
            float blending = 1.0f;
            if (depth >= 0.0f && has_history) {
                // old_screen = inverse(old_screen_to_world) * vec4(world_pos, 1)   temporal.comp:75-84
                float sx = ((inv[0] * world_pos.x + inv[1] * world_pos.y) + inv[2] * world_pos.z) + inv[3];
                float sy = ((inv[4] * world_pos.x + inv[5] * world_pos.y) + inv[6] * world_pos.z) + inv[7];
                float sz = ((inv[8] * world_pos.x + inv[9] * world_pos.y) + inv[10] * world_pos.z) + inv[11];
                sx = sx / sz; sy = sy / sz;  // temporal.comp:85
                float tu_ = (sx + 0.5f) * (1.0f / (float)width);       // temporal.comp:89
                float tv_ = (sy + -0.5f) * (-1.0f / (float)height);
                if (0.0f <= tu_ && tu_ <= 1.0f && 0.0f <= tv_ && tv_ <= 1.0f) {
                    float ond[4];
                    tex_nd.sample(tu_, tv_, ond);
                    float old_depth = ond[3];
                    // int(old_screen.x + 0.5), int(old_screen.y - 0.5): truncation   temporal.comp:99-103
                    V3 old_ray_dir = normalize(((float)vx_f2i(sx + 0.5f) * ocr + (float)vx_f2i(sy - 0.5f) * ocu) + ocf);
                    V3 old_position = oco + old_depth * old_ray_dir;
                    V3 camera_dir = normalize(cam_o - world_pos);
                    float bias = vx_max(0.0f, dot(camera_dir, normal));
                    float dist = length(old_position - world_pos);
                    bool same_position = dist < (bias * tu->blending_distance_cutoff) * depth;
                    if (same_position) {
                        tex_color.sample(tu_, tv_, old_c);
                        blending = old_c[3];
                    }
                }
            }
            V3 blended = depth >= 0.0f ? vmix(v3(old_c[0], old_c[1], old_c[2]), color, blending) : color;
            float next_blending = vx_clamp((1.0f - tu->sample_blending) * blending, 1.0f - tu->maximum_blending, 1.0f);
            new_color[o] = blended.x; new_color[o + 1] = blended.y; new_color[o + 2] = blended.z; new_color[o + 3] = next_blending;
        }
    });
}

// denoise.comp main() over a full width x height frame.
void orc_denoise(const float* colors, const float* normals_depths, const float* albedo, const float* cam,
                 const OrcDenoise* du, int width, int height, float* output, int nthreads) {
    float sigma_distance_2 = 2.0f * (du->sigma_distance * du->sigma_distance);  // U2
    float sigma_range_2 = 2.0f * (du->sigma_range * du->sigma_range);
    int r = (int)du->radius;
    parallel_rows(0, height, nthreads, [&](int y) {
        for (int x = 0; x < width; x++) {
            size_t o = 4 * ((size_t)y * width + x);
            V3 ray_dir = pixel_ray_dir(cam + 4, cam + 8, cam + 12, x, y);
            float normalization = 0.0f;
            V3 sum = v3s(0.0f);
            V3 cc = v3(colors[o], colors[o + 1], colors[o + 2]);
            V3 cn = v3(normals_depths[o], normals_depths[o + 1], normals_depths[o + 2]);
            float cd = normals_depths[o + 3];
            V3 calb = v3(albedo[o], albedo[o + 1], albedo[o + 2]);
            int32_t cmat; memcpy(&cmat, &albedo[o + 3], 4);
            float depth_bias = vx_max(0.0f, dot(cn, -ray_dir));
            for (int dy = -r; dy <= r; dy++) {
                for (int dx = -r; dx <= r; dx++) {
                    int nx = x + dx, ny = y + dy;
                    if (0 <= nx && nx < width && 0 <= ny && ny < height) {
                        size_t w = 4 * ((size_t)ny * width + nx);
                        V3 wc = v3(colors[w], colors[w + 1], colors[w + 2]);
                        V3 wn = v3(normals_depths[w], normals_depths[w + 1], normals_depths[w + 2]);
                        float wd = normals_depths[w + 3];
                        int32_t wmat; memcpy(&wmat, &albedo[w + 3], 4);
                        V3 color_delta = cc - wc;
                        V3 normal_delta = cn - wn;
                        float depth_delta = vx_log(vx_abs(cd)) - vx_log(vx_abs(wd));
                        float material_delta = (cmat >> 24) != (wmat >> 24) ? 1.0f : 0.0f;
                        float bd = depth_bias * depth_delta;
                        float factor_range = (((dot(color_delta, color_delta) + 1e4f * dot(normal_delta, normal_delta)) +
                                               1e4f * (bd * bd)) + 1e4f * material_delta) / sigma_range_2;
                        float factor_distance = (float)(dx * dx + dy * dy) / sigma_distance_2;
                        float factor = vx_exp(-factor_range - factor_distance);
                        normalization += factor;
                        sum = sum + wc * factor;
                    }
                }
            }
            V3 out = du->radius == 0 ? cc : sum / normalization;
            out = vmix(out, calb * out, du->albedo_factor);
            output[o] = out.x; output[o + 1] = out.y; output[o + 2] = out.z; output[o + 3] = 1.0f;
        }
    });
}

// The build's deterministic stand-in for resources/blue-noise-128.zip, which the reference does not
// ship (.MISSING_LARGE_BLOBS; loader format src/context.rs:1087-1116): 512 x 128 x 128 uniform
// floats in [0,1) from a counter-based hash of (seed, index).  Spec: see vxrt.h vxrt_noise_value().
void orc_noise_table(uint32_t seed, float* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        uint32_t z = (uint32_t)i * 0x9E3779B9u + seed;
        z ^= z >> 16; z *= 0x85EBCA6Bu; z ^= z >> 13; z *= 0xC2B2AE35u; z ^= z >> 16;
        out[i] = (float)(z >> 8) * (1.0f / 16777216.0f);
    }
}

// detmath probes, so tests can compare the contract with libm/numpy and with the device.
void orc_detmath(int fn, const float* x, const float* y, float* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        switch (fn) {
            case 0: out[i] = vx_sin(x[i]); break;
            case 1: out[i] = vx_cos(x[i]); break;
            case 2: out[i] = vx_exp(x[i]); break;
            case 3: out[i] = vx_log(x[i]); break;
            case 4: out[i] = vx_pow(x[i], y[i]); break;
            case 5: out[i] = vx_sqrt(x[i]); break;
            case 6: out[i] = x[i] / y[i]; break;
            case 7: out[i] = vx_tan(x[i]); break;
            case 9: case 10: {  // y / z of random_hemisphere before the flip
                float phi = (2.0f * 3.14159265358979f) * x[i];
                float rx = 2.0f * y[i] - 1.0f;
                float plane_radius = vx_sqrt(1.0f - rx * rx);
                out[i] = fn == 9 ? plane_radius * vx_cos(phi) : plane_radius * vx_sin(phi);
                break;
            }
            case 11: out[i] = x[i] * y[i]; break;
            case 12: out[i] = x[i] - y[i]; break;
            case 13: out[i] = x[i] - y[i] * vx_min(0.0f, 2.0f * x[i]); break;
            case 14: out[i] = vx_min(x[i], y[i]); break;
            case 15: out[i] = vx_max(x[i], y[i]); break;
            case 16: out[i] = vx_max(0.0f, x[i]) * y[i]; break;
            case 17: out[i] = vx_sign(x[i]) * y[i]; break;
            case 18: out[i] = vx_clamp(x[i], y[i], 1.0f); break;
            case 19: out[i] = vx_min(0.0f, x[i]) * y[i]; break;  // the GLSL wording; the device's vx_min0 must equal it
            case 20: out[i] = vx_exp_unfused(x[i]); break;       // round 1-3's exp (every product and sum rounded)
            default: out[i] = 0.0f;
        }
    }
}

}  // extern "C"
