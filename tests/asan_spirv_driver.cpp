// Test infrastructure (tests/test_oracle_spirv_exec.py builds this with -fsanitize=address,undefined together with oracle/ospirv.cpp):
// the SPIR-V interpreter on word streams it must not trust — the module given on the command line with a few words flipped per trial,
// bound to small synthetic buffers.  Every trial must end in a status code (0 or -1); a sanitizer report or a crash fails the test.
// usage: asan_spirv_driver <module.spv> <trials> <seed> <kinds>     kinds: one digit per binding 0, 1, ...: 0 buffer, 1 storage image, 2 sampled image, 3 sampler
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct OrcSpvBinding { uint32_t set, binding, kind, pad; void* data; uint64_t bytes; uint32_t width, height, ox, oy, cw, ch; };
extern "C" int orc_spirv_dispatch(const uint32_t* words, size_t nwords, const OrcSpvBinding* bind, int nbind, uint32_t x0, uint32_t y0,
                                  uint32_t x1, uint32_t y1, uint32_t flags, int nthreads, uint64_t* executed);
extern "C" const char* orc_spirv_error(void);

static uint64_t rng_state;
static uint32_t rnd() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return uint32_t(rng_state >> 33); }

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const std::string kinds = argv[4];
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint32_t> words;
    uint32_t w;
    while (fread(&w, 4, 1, f) == 1) words.push_back(w);
    fclose(f);
    const int trials = atoi(argv[2]);
    rng_state = strtoull(argv[3], nullptr, 10);
    // bindings 0-8 as both kinds of thing the three shaders expect there: images of 16 x 16 and buffers of a few KB
    std::vector<float> img[9];
    std::vector<uint32_t> buf[9];
    int refused = 0, ran = 0;
    for (int t = 0; t < trials; t++) {
        std::vector<uint32_t> m = words;
        const int flips = t == 0 ? 0 : 1 + int(rnd() % 4);
        for (int k = 0; k < flips; k++) {
            const size_t at = 5 + rnd() % (m.size() - 5);
            switch (rnd() % 4) {
                case 0: m[at] ^= 1u << (rnd() % 32); break;
                case 1: m[at] = rnd(); break;
                case 2: m[at] = rnd() % 1200; break;        // another id
                default: m[at] = (m[at] & 0xffff0000u) | (rnd() % 256); break;   // another opcode, same length
            }
        }
        if (t != 0 && rnd() % 16 == 0) m.resize(5 + rnd() % (m.size() - 5));
        std::vector<OrcSpvBinding> b;
        for (uint32_t k = 0; k < kinds.size() && k < 9; k++) {
            img[k].assign(16 * 16 * 4, 0.25f * float(k + 1));
            buf[k].assign(4096, 0u);
            for (size_t j = 0; j < buf[k].size(); j++) {
                // floats a camera / an octree header can live with in the first words, small node words (children, leaves, empties) after them
                if (j == 0) buf[k][j] = 2u;      // a denoise radius; as a float (a camera's or the root's x) a denormal
                else if (j < 5) { const float v = 0.1f * float(j % 7) + (j % 4 == 3 ? 4.0f : 0.5f); memcpy(&buf[k][j], &v, 4); }
                else buf[k][j] = rnd() % 8 == 0 ? (0x80000000u | (rnd() & 0xffffff)) : (rnd() % 3 == 0 ? rnd() % 400 : 0u);
            }
            uint32_t kind = uint32_t(kinds[k] - '0');
            if (rnd() % 64 == 0) kind = rnd() % 4;     // now and then the wrong kind of thing
            if (kind == 1 || kind == 2) b.push_back(OrcSpvBinding{0, k, kind, 0, img[k].data(), 16 * 16 * 16, 16, 16, 0, 0, 16, 16});
            else b.push_back(OrcSpvBinding{0, k, kind, 0, buf[k].data(), 4096 * 4, 0, 0, 0, 0, 0, 0});
        }
        uint64_t n = 0;
        const int rc = orc_spirv_dispatch(m.data(), m.size(), b.data(), int(b.size()), 0, 0, 3, 3, (rnd() % 2) | (1u << 8), 1 + int(rnd() % 2), &n);
        if (t == 0 && rc != 0) { printf("the unmodified module does not run on the synthetic bindings: %s\n", orc_spirv_error()); return 4; }
        if (rc == 0) ran++; else if (rc == -1 && orc_spirv_error()[0] != 0) refused++; else return 3;
    }
    {   // a module cut off right behind an OpExtInstImport whose one name word has no NUL ("GLSL"): the name compare must stay inside the instruction
        std::vector<uint32_t> m(words.begin(), words.begin() + 5);
        m.push_back(3u << 16 | 11u);          // OpExtInstImport, 3 words: result id, one word of name
        m.push_back(1u);
        m.push_back(0x4c534c47u);             // "GLSL", no terminator, last word of the buffer
        std::vector<uint32_t> exact(m);        // a heap block of exactly this size: ASan sees any read past it
        exact.shrink_to_fit();
        uint64_t n = 0;
        const int rc = orc_spirv_dispatch(exact.data(), exact.size(), nullptr, 0, 0, 0, 1, 1, 0, 1, &n);
        if (rc == 0) { printf("a module that ends inside its header ran\n"); return 5; }
        refused++;
    }
    printf("trials %d: ran to the end %d, refused %d\n", trials, ran, refused);
    return 0;
}
