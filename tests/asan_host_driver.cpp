// Test infrastructure: the host-only decoders of libvxrt (csrc/scene_host.cpp, vox_scene.cpp, noise_zip.cpp — no HIP in them) compiled
// by g++ with AddressSanitizer + UBSan and fed damaged inputs (tests/test_fuzz_parsers.py::test_decoders_under_sanitizers).
// usage: asan_host_driver <file> ...   (".zip" -> the blue-noise archive reader, anything else -> both .vox decoders)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../gpu_voxel_raytracer_amd/csrc/scene_host.h"

int main(int argc, char** argv) {
    long ok = 0, bad = 0;
    for (int i = 1; i < argc; i++) {
        const std::string path = argv[i];
        if (path.size() > 4 && path.compare(path.size() - 4, 4, ".zip") == 0) {
            std::vector<float> px;
            uint32_t size = 0, layers = 0;
            (vxrt::noise_zip_read(path.c_str(), &px, &size, &layers) == 0 ? ok : bad)++;
            continue;
        }
        std::ifstream f(path, std::ios::binary);
        std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        vxrt::VoxScene scene;
        int32_t lo[3], hi[3];
        (vxrt::decode_vox(bytes.data(), bytes.size(), &scene) == 0 ? ok : bad)++;
        for (uint32_t flags : {1u, 3u, 5u, 7u}) {
            vxrt::VoxScene s2;
            (vxrt::decode_vox_scene(bytes.data(), bytes.size(), flags, &s2, lo, hi) == 0 ? ok : bad)++;
            if (!s2.voxels.empty()) {
                vxrt::Octree tree;
                (void)vxrt::build_octree(s2.voxels.data(), s2.voxels.size() < 4096 ? s2.voxels.size() : 4096, &tree);
            }
        }
    }
    std::printf("ok %ld rejected %ld\n", ok, bad);
    return 0;
}
