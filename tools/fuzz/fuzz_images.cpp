// Mutation fuzz of the image decoders (host code only): random bit flips, byte overwrites, truncations and
// insertions on the repository's test images; the decoders must return (true or false) without touching memory
// they do not own.  Built with -fsanitize=address,undefined.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <random>
#include <string>
#include <vector>
#include "../../simple-ray-tracer_amd/host/srt_scene_io.cpp" /* its PNG decoder has internal linkage */
int main(int argc, char** argv)
{
    const int iters = atoi(argv[1]);
    std::mt19937 rng(12345);
    long ok = 0, bad = 0;
    for (int a = 2; a < argc; ++a) {
        std::ifstream f(argv[a], std::ios::binary);
        const std::vector<uint8_t> orig((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const bool png = std::string(argv[a]).find(".png") != std::string::npos;
        for (int it = 0; it < iters; ++it) {
            std::vector<uint8_t> m = orig;
            const int kind = rng() % 5;
            const int nmut = 1 + rng() % 8;
            for (int k = 0; k < nmut && !m.empty(); ++k) {
                const size_t p = rng() % m.size();
                if (kind == 0) m[p] ^= (uint8_t)(1u << (rng() % 8));
                else if (kind == 1) m[p] = (uint8_t)rng();
                else if (kind == 2) { m.resize(p); break; }
                else if (kind == 3) m.insert(m.begin() + p, (uint8_t)rng());
                else { const uint8_t v[4] = { 0xFF, 0x00, 0x7F, 0x80 }; m[p] = v[rng() % 4]; }
            }
            int w = 0, h = 0;
            std::vector<uint8_t> rgb;
            const bool r = png ? srt::decodePng(m, w, h, rgb) : srt::decodeJpeg(m, w, h, rgb);
            if (r) {
                if (w <= 0 || h <= 0 || rgb.size() != (size_t)w * h * 3) { printf("inconsistent result on %s iter %d\n", argv[a], it); return 1; }
                ++ok;
            } else ++bad;
        }
    }
    printf("decoded %ld, refused %ld\n", ok, bad);
    return 0;
}
