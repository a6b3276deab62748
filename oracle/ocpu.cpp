// ORACLE — test infrastructure only.  Nothing in the product path may include, link or call this.
//
// ocpu.cpp: restatement of the reference's orphaned CPU ray caster — src/cpu.rs (CpuBackend) and
// src/cpu/octree.rs (pointer octree with sorted mid-plane crossings).  That code is not part of the
// reference's compiled crate (src/main.rs:7-12 has no `mod cpu`) and imports symbols that exist
// nowhere (Backend, Coord, Ray, Camera::cast_rays — src/cpu.rs:5), so it pins BASELINE.json's
// config 1 ("plumbing") only.  Definitions the reference lacks and this file supplies:
//   Coord{x,y,z: u16}, Ray{origin, direction}, and cast_rays(size) := for each pixel in row-major
//   order, origin = camera.position, direction = norm(x*right - y*up + forward_ray) with the
//   basis of Camera::axis_scaled (src/camera.rs:19-28) — the same rays shaders/voxels.comp casts.
// Parity status: UNPINNED by the reference.
#include <cmath>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>

#include "oracle.h"

namespace orc {

// Rust f32::min / f32::max ignore a NaN operand.
static inline float rmin(float a, float b) { return a != a ? b : (b != b ? a : (a < b ? a : b)); }
static inline float rmax(float a, float b) { return a != a ? b : (b != b ? a : (a > b ? a : b)); }
static inline float idx(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

// Octree<Color> (src/cpu/octree.rs:5-16) flattened: slot 0 = None, >0 = Branch(index), <0 = Leaf(~value index).
struct CpuOctree {
    int depth = 0;
    int32_t root = 0;
    std::vector<int32_t> branches;  // 8 slots per branch, branch k at [8k, 8k+8); index 0 unused
    std::vector<uint32_t> colors;   // leaf values (0x00RRGGBB)

    int32_t new_branch() {
        if (branches.empty()) branches.resize(8, 0);
        int32_t k = (int32_t)(branches.size() / 8);
        branches.resize(branches.size() + 8, 0);
        return k;
    }
    // Octree::insert (src/cpu/octree.rs:77-93) with get_or_insert_octant (:95-107)
    void insert(uint16_t x, uint16_t y, uint16_t z, uint32_t rgb) {
        uint16_t cx = (uint16_t)((1u << depth) / 2), cy = cx, cz = cx;  // root_center, :340-347
        int32_t* current = &root;
        size_t current_branch = 0; int current_slot = -1;  // re-derive the pointer after vector growth
        uint32_t size = 1u << depth;
        while (size > 1) {
            int oct = 4 * (x >= cx) + 2 * (y >= cy) + (z >= cz);
            int32_t v = current_slot < 0 ? root : branches[8 * current_branch + current_slot];
            if (v == 0) {
                v = new_branch();
                if (current_slot < 0) root = v; else branches[8 * current_branch + current_slot] = v;
            }
            current_branch = (size_t)v; current_slot = oct;
            size /= 2;
            uint16_t amount = (uint16_t)(size / 2);  // octant_center(octant, center, size), :360-375
            cx = (oct & 4) ? cx + amount : cx - amount;
            cy = (oct & 2) ? cy + amount : cy - amount;
            cz = (oct & 1) ? cz + amount : cz - amount;
        }
        (void)current;
        colors.push_back(rgb);
        int32_t leaf = -(int32_t)colors.size();
        if (current_slot < 0) root = leaf; else branches[8 * current_branch + current_slot] = leaf;
    }
};

struct RayExt { V3 origin, direction, inv_direction; };
struct RayHit { uint32_t value; V3 normal; float time; };

// ray_cube_intersection (src/cpu/octree.rs:385-427)
static bool cube(const RayExt& ray, V3 center, float size, float* entry, float* exit, int* plane) {
    float half_size = size / 2.0f;
    V3 delta = center - ray.origin;
    float en[3], ex[3];
    for (int p = 0; p < 3; p++) {
        float dir = idx(ray.direction, p), dist = idx(delta, p);
        if (dir == 0.0f) {
            if (std::fabs(dist) <= half_size) { en[p] = -INFINITY; ex[p] = INFINITY; }
            else { en[p] = INFINITY; ex[p] = -INFINITY; }
        } else {
            float inv = idx(ray.inv_direction, p);
            en[p] = (dist - std::copysign(half_size, inv)) * inv;
            ex[p] = (dist + std::copysign(half_size, inv)) * inv;
        }
    }
    int m = 0;
    if (en[1] > en[m]) m = 1;
    if (en[2] > en[m]) m = 2;
    float min_exit = rmin(rmin(ex[0], ex[1]), ex[2]);
    if (min_exit > 0.0f && en[m] < min_exit) { *entry = en[m]; *exit = min_exit; *plane = m; return true; }
    return false;
}

struct OctInt { int count; uint8_t octants[4], planes[4]; float times[5]; };

// octant_intersections (src/cpu/octree.rs:247-320)
static OctInt octant_intersections(const RayExt& ray, V3 center, float entry, float exit, int entry_plane) {
    V3 delta = center - ray.origin;
    V3 pe = delta * ray.inv_direction;
    float plane_entry[3] = {pe.x, pe.y, pe.z};
    int order[3] = {0, 1, 2};
    auto swp = [&](int a, int b) { int t = order[a]; order[a] = order[b]; order[b] = t; };
    if (plane_entry[order[0]] > plane_entry[order[1]]) swp(0, 1);
    if (plane_entry[order[1]] > plane_entry[order[2]]) {
        swp(1, 2);
        if (plane_entry[order[0]] > plane_entry[order[1]]) swp(0, 1);
    }
    int i = 0;
    while (i < 3 && plane_entry[order[i]] < 0.0f) i++;
    uint8_t octant = (uint8_t)(4 * (delta.x < 0.0f) + 2 * (delta.y < 0.0f) + (delta.z < 0.0f));
    while (i < 3 && plane_entry[order[i]] < entry) { octant ^= 4 >> order[i]; i++; }
    OctInt r;
    memset(&r, 0, sizeof r);
    int n = 1;
    r.octants[0] = octant; r.planes[0] = (uint8_t)entry_plane; r.times[0] = entry;
    while (i < 3 && plane_entry[order[i]] < exit) {
        octant ^= 4 >> order[i];
        i++;
        r.octants[n] = octant; r.planes[n] = (uint8_t)order[i - 1]; r.times[n] = plane_entry[order[i - 1]];
        n++;
    }
    r.times[n] = exit;
    r.count = n;
    return r;
}

// plane_normal (src/cpu/octree.rs:322-326)
static V3 plane_normal(int plane, V3 dir) {
    float n = -std::copysign(1.0f, idx(dir, plane));
    return v3(plane == 0 ? n : 0.0f, plane == 1 ? n : 0.0f, plane == 2 ? n : 0.0f);
}
// child_octant_center (src/cpu/octree.rs:328-338)
static V3 child_octant_center(V3 pc, float child_size, int octant) {
    V3 s = v3((octant & 4) == 0 ? -0.5f : 0.5f, (octant & 2) == 0 ? -0.5f : 0.5f, (octant & 1) == 0 ? -0.5f : 0.5f);
    return pc + child_size * s;
}

// Octree::cast_ray + cast_ray_children_iterative (src/cpu/octree.rs:109-198)
static bool cast_ray(const CpuOctree& t, V3 origin, V3 direction, RayHit* hit) {
    float c = (float)((1u << t.depth) / 2);
    V3 center = v3(c, c, c);
    RayExt ray{origin, direction, v3(1.0f / direction.x, 1.0f / direction.y, 1.0f / direction.z)};
    float size = (float)(1u << t.depth);
    float entry, exit; int entry_plane;
    if (!cube(ray, center, size, &entry, &exit, &entry_plane)) return false;
    if (t.root == 0) return false;
    if (t.root < 0) {
        hit->value = t.colors[(size_t)(-t.root) - 1]; hit->normal = plane_normal(entry_plane, direction); hit->time = entry;
        return true;
    }
    struct Frame { int step; V3 center; float size; int32_t children; OctInt ints; };
    Frame stack[16];
    stack[0] = Frame{0, center, size, t.root, octant_intersections(ray, center, entry, exit, entry_plane)};
    int top = 0;
    for (;;) {
        Frame& f = stack[top];
        int i = f.step;
        if (i >= f.ints.count) {
            if (top == 0) break;
            top--;
            continue;
        }
        f.step++;
        int octant = f.ints.octants[i];
        int32_t slot = t.branches[8 * (size_t)f.children + octant];
        if (slot == 0) continue;
        if (slot > 0) {
            int plane = f.ints.planes[i];
            float en = f.ints.times[i], ex = f.ints.times[i + 1];
            float child_size = f.size / 2.0f;
            V3 cc = child_octant_center(f.center, child_size, octant);
            OctInt ci = octant_intersections(ray, cc, en, ex, plane);
            if (top + 1 >= 16) return false;
            top++;
            stack[top] = Frame{0, cc, child_size, slot, ci};
        } else {
            hit->value = t.colors[(size_t)(-slot) - 1];
            hit->normal = plane_normal(f.ints.planes[i], direction);
            hit->time = f.ints.times[i];
            return true;
        }
    }
    return false;
}

static uint8_t round_u8(float v) {  // f32::round() as u8 : half away from zero, saturating, NaN -> 0
    float r = std::round(v);
    if (!(r == r)) return 0;
    if (r <= 0.0f) return 0;
    if (r >= 255.0f) return 255;
    return (uint8_t)r;
}

}  // namespace orc

using namespace orc;

extern "C" {

// CpuBackend::from_voxels (src/cpu.rs:13-30): the octree, kept between frames as the reference keeps its backend.
struct CpuBackend { CpuOctree tree; };

void* orc_cpu_rs_create(const uint16_t* coords, const uint8_t* rgb, size_t n) {
    uint16_t max_coord = 0;
    for (size_t i = 0; i < 3 * n; i++) if (coords[i] > max_coord) max_coord = coords[i];
    int max_depth = 0;
    if (max_coord != 0) { uint32_t p = 1; while (p < (uint32_t)max_coord + 1) { p <<= 1; max_depth++; } }
    CpuBackend* b = new CpuBackend();
    b->tree.depth = max_depth;
    for (size_t i = 0; i < n; i++)
        b->tree.insert(coords[3 * i], coords[3 * i + 1], coords[3 * i + 2],
                       ((uint32_t)rgb[3 * i] << 16) | ((uint32_t)rgb[3 * i + 1] << 8) | rgb[3 * i + 2]);
    return b;
}
void orc_cpu_rs_destroy(void* backend) { delete static_cast<CpuBackend*>(backend); }

// CpuBackend::render (src/cpu.rs:32-72).  basis9 = right, up, forward_ray from Camera::axis_scaled.  pixels: width*height*3 u8
// (row-major).  The reference renders the pixels with rayon's par_iter_mut over all host cores (src/cpu.rs:43-46); here the rows
// are dealt to `nthreads` std::threads (pixels are independent: the image does not depend on the thread count).
// Also returns, per pixel, the primary hit (time, normal, 0x00RRGGBB or -1 for a miss) so the
// shader-path traversal can be cross-checked against this independent octree (hit_time may be null).
void orc_cpu_rs_render_frame(const void* backend, const float* cam_pos, const float* basis9, int width, int height, float time,
                             uint8_t* pixels, float* hit_time, float* hit_normal, int32_t* hit_value, int nthreads) {
    const CpuOctree& tree = static_cast<const CpuBackend*>(backend)->tree;
    float c = 127.0f / 2.0f;
    V3 light_pos = v3(c - 10.0f * std::cos(0.3f * time), 15.0f + 8.0f * std::sin(3.0f * time), c - 13.0f * std::sin(0.3f * time));
    V3 origin = v3(cam_pos[0], cam_pos[1], cam_pos[2]);
    V3 R = v3(basis9[0], basis9[1], basis9[2]), U = v3(basis9[3], basis9[4], basis9[5]), F = v3(basis9[6], basis9[7], basis9[8]);
    auto render_row = [&](int y) {
        for (int x = 0; x < width; x++) {
            size_t p = (size_t)y * width + x;
            V3 dir = normalize(((float)x * R - (float)y * U) + F);
            RayHit hit;
            uint8_t r = 0, g = 0, b = 0;  // Color::BLACK on a miss, src/cpu.rs:69
            if (cast_ray(tree, origin, dir, &hit)) {
                V3 hit_point = origin + dir * hit.time;
                V3 light_delta = light_pos - hit_point;
                float light_distance = length(light_delta);
                V3 light_dir = light_delta / light_distance;
                RayHit sh;
                bool in_shadow = cast_ray(tree, hit_point + 0.001f * hit.normal, light_dir, &sh) && sh.time < light_distance;
                float attenuation = 60.0f * std::pow(light_distance, -2.0f);
                float shadow = 0.4f + 0.6f * (in_shadow ? 0.0f : 1.0f);
                float brightness = 0.0f + (shadow * rmax(dot(light_dir, hit.normal), 0.0f)) * attenuation;
                r = round_u8((float)((hit.value >> 16) & 0xff) * brightness);
                g = round_u8((float)((hit.value >> 8) & 0xff) * brightness);
                b = round_u8((float)(hit.value & 0xff) * brightness);
                if (hit_time) { hit_time[p] = hit.time; hit_value[p] = (int32_t)hit.value;
                                hit_normal[3 * p] = hit.normal.x; hit_normal[3 * p + 1] = hit.normal.y; hit_normal[3 * p + 2] = hit.normal.z; }
            } else if (hit_time) {
                hit_time[p] = -1.0f; hit_value[p] = -1; hit_normal[3 * p] = hit_normal[3 * p + 1] = hit_normal[3 * p + 2] = 0.0f;
            }
            pixels[3 * p] = r; pixels[3 * p + 1] = g; pixels[3 * p + 2] = b;
        }
    };
    if (nthreads <= 1) { for (int y = 0; y < height; y++) render_row(y); return; }
    std::atomic<int> next(0);
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++)
        pool.emplace_back([&] { for (;;) { int y = next.fetch_add(1); if (y >= height) break; render_row(y); } });
    for (auto& th : pool) th.join();
}

// from_voxels + one render (the form the parity tests use)
void orc_cpu_rs_render(const uint16_t* coords, const uint8_t* rgb, size_t n, const float* cam_pos, const float* basis9,
                       int width, int height, float time, uint8_t* pixels, float* hit_time, float* hit_normal,
                       int32_t* hit_value) {
    void* b = orc_cpu_rs_create(coords, rgb, n);
    orc_cpu_rs_render_frame(b, cam_pos, basis9, width, height, time, pixels, hit_time, hit_normal, hit_value, 1);
    orc_cpu_rs_destroy(b);
}

}  // extern "C"
