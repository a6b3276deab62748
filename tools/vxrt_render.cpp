// vxrt_render — headless render loop in C++ over include/vxrt.hpp: what the reference's src/main.rs does with a
// window (load a model, place the camera, render frames) ending in a PPM of the last denoised frame and, optionally,
// a raw float dump.
//
//   vxrt_render <scene.vox | menger:<level>[:clip[:emissive_period]] | default[:seed]> <width> <height> <frames> <bounces> <radius> <out.ppm> [out.f32]
//
// `default` is the reference's start-up scene and camera (src/context.rs:838-910, 618-622).  VXRT_NOISE=blue generates the
// blue-noise table on the GPU, VXRT_NOISE=<file.zip> loads one in the reference's resource format (src/context.rs:1016-1116).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include "../include/vxrt.hpp"

static unsigned char srgb8(float x) {  // what a Bgra8UnormSrgb swap chain stores (src/context.rs:696-706)
    if (!(x == x)) x = 0.0f;
    x = std::min(1.0f, std::max(0.0f, x));
    float y = x <= 0.0031308f ? 12.92f * x : 1.055f * std::pow(x, 1.0f / 2.4f) - 0.055f;
    return static_cast<unsigned char>(y * 255.0f + 0.5f);
}

int main(int argc, char** argv) {
    if (argc < 8) {
        std::fprintf(stderr, "usage: %s <scene.vox|menger:L[:clip[:period]]> <width> <height> <frames> <bounces> <radius> <out.ppm> [out.f32]\n", argv[0]);
        return 2;
    }
    try {
        const std::string scene = argv[1];
        const uint32_t width = std::atoi(argv[2]), height = std::atoi(argv[3]);
        const int frames = std::atoi(argv[4]);
        vxrt::Context ctx(width, height, std::atoi(argv[5]));
        ctx.denoise_uniforms.radius = std::atoi(argv[6]);
        if (const char* noise = std::getenv("VXRT_NOISE")) {
            if (std::strcmp(noise, "blue") == 0) ctx.generate_blue_noise();
            else ctx.load_blue_noise(noise);
        }
        float extent[3];
        bool start_camera = false;
        if (scene.rfind("default", 0) == 0) {
            unsigned seed = 1;
            std::sscanf(scene.c_str() + 7, ":%u", &seed);
            ctx.recreate_octree(vxrt::create_voxels(seed));
            extent[0] = extent[1] = extent[2] = 0.0f;
            start_camera = true;  // Camera{} is the reference's start camera
        } else if (scene.rfind("menger:", 0) == 0) {
            unsigned level = 0, clip = 0, period = 0;
            std::sscanf(scene.c_str() + 7, "%u:%u:%u", &level, &clip, &period);
            ctx.set_menger(level, clip, {0, 0x7b, 0xa2, 0x3f}, period);
            unsigned side = 1;
            for (unsigned l = 0; l < level; l++) side *= 3;
            if (clip && clip < side) side = clip;
            extent[0] = extent[1] = extent[2] = side * 0.5f;
        } else {
            std::ifstream f(scene, std::ios::binary);
            if (!f) throw std::runtime_error("cannot open " + scene);
            std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            vxrt::VoxelList voxels = vxrt::voxels_from_vox(bytes);
            ctx.recreate_octree(voxels);
            // file axes (x, y, z) -> renderer axes (x, z, y), half a world unit per voxel
            extent[0] = voxels.size[0] * 0.5f; extent[1] = voxels.size[2] * 0.5f; extent[2] = voxels.size[1] * 0.5f;
        }
        // the fixed outside view of SURVEY.md §8d: position = c + e * (-0.9, 0.6, -1.2), looking at c
        const float e = std::max(extent[0], std::max(extent[1], extent[2]));
        const float k[3] = {-0.9f, 0.6f, -1.2f};
        for (int i = 0; i < 3 && !start_camera; i++) {
            const float c = extent[i] * 0.5f;
            ctx.camera.position[i] = c + e * k[i];
            ctx.camera.direction[i] = c - ctx.camera.position[i];
        }
        // the reference presents every frame (src/context.rs:2046-2070); here every frame is taken to the host without stalling the
        // loop: frame f travels into one of two pinned buffers while frame f + 1 renders (vxrt_read_async / vxrt_read_wait)
        vxrt::PinnedImage shown[2] = {vxrt::PinnedImage(ctx.image_floats()), vxrt::PinnedImage(ctx.image_floats())};
        for (int f = 0; f < frames; f++) {
            ctx.render(VXRT_ALL);
            ctx.read_async(VXRT_DENOISED, shown[f & 1], uint32_t(f & 1));
            if (f > 0) ctx.read_wait(uint32_t((f + 1) & 1));        // frame f - 1 has arrived: this is where a viewer would show it
        }
        const int last = (frames - 1) & 1;
        if (frames > 0) ctx.read_wait(uint32_t(last));
        std::vector<float> img = frames > 0 ? std::vector<float>(shown[last].data(), shown[last].data() + shown[last].size()) : ctx.read(VXRT_DENOISED);
        const vxrt_stats st = ctx.stats();
        std::ofstream ppm(argv[7], std::ios::binary);
        ppm << "P6\n" << width << " " << height << "\n255\n";
        for (size_t p = 0; p < size_t(width) * height; p++) {
            const unsigned char rgb[3] = {srgb8(img[4 * p]), srgb8(img[4 * p + 1]), srgb8(img[4 * p + 2])};
            ppm.write(reinterpret_cast<const char*>(rgb), 3);
        }
        if (argc > 8) {
            std::ofstream raw(argv[8], std::ios::binary);
            raw.write(reinterpret_cast<const char*>(img.data()), std::streamsize(img.size() * sizeof(float)));
        }
        std::printf("%s: %llu frames, %llu rays, %ux%u -> %s\n", scene.c_str(), (unsigned long long)st.frames, (unsigned long long)st.rays, width, height, argv[7]);
        return 0;
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "vxrt_render: %s\n", ex.what());
        return 1;
    }
}
