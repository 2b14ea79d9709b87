// Mutation fuzz of the text parsers (host code only): TOML reader + configuration, OBJ / MTL loader.  Parsers may throw
// std::exception on bad input; they must not crash or touch foreign memory.  -fsanitize=address,undefined.
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iterator>
#include <random>
#include <string>
#include <vector>
#include "srt_host.h"
#include "srt_toml.h"
#include "srt_config.h"
static std::string slurp(const char* p) { std::ifstream f(p, std::ios::binary); return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
static std::string mutate(const std::string& orig, std::mt19937& rng)
{
    std::string m = orig;
    static const char* tokens[] = { "[", "]", "[[", "]]", "=", "\"", "'", "#", "\n", ",", "{", "}", "-", "1e999", "nan", "inf", "0x", "/", "//", " ", "f ", "v ", "vt ", "vn ", "usemtl ", "mtllib ", "g ", "-1", "99999999999", "\\", "map_Kd ", "newmtl ", "Kd ", "true", "false", "." };
    const int nmut = 1 + rng() % 6;
    for (int k = 0; k < nmut && !m.empty(); ++k) {
        const size_t p = rng() % m.size();
        switch (rng() % 6) {
        case 0: m[p] = (char)rng(); break;
        case 1: m.erase(p, 1 + rng() % 8); break;
        case 2: m.insert(p, tokens[rng() % (sizeof tokens / sizeof *tokens)]); break;
        case 3: m.resize(p); break;
        case 4: { const size_t q = rng() % m.size(); m.insert(p, m.substr(q, rng() % 40)); break; }
        default: m[p] = "0123456789-+.eE"[rng() % 15]; break;
        }
    }
    return m;
}
int main(int argc, char** argv)
{
    const int iters = atoi(argv[1]);
    const std::filesystem::path dir = argv[2];
    std::mt19937 rng(777);
    long ok = 0, thrown = 0;
    for (int a = 3; a < argc; ++a) {
        const std::string orig = slurp(argv[a]);
        const std::string name = argv[a];
        const bool toml = name.find(".toml") != std::string::npos;
        const bool mtl = name.find(".mtl") != std::string::npos;
        for (int it = 0; it < iters; ++it) {
            const std::string m = mutate(orig, rng);
            try {
                if (toml) {
                    const auto path = dir / "f.toml";
                    { std::ofstream o(path, std::ios::binary); o << m; }
                    (void)srt::toml::parse(m);
                    (void)srt::readConfigFile(path);
                } else {
                    const auto path = dir / (mtl ? "quads_groups.mtl" : "f.obj");
                    { std::ofstream o(path, std::ios::binary); o << m; }
                    (void)srt::loadMesh(mtl ? dir / "base.obj" : path, (it & 1) != 0);
                }
                ++ok;
            } catch (const std::exception&) {
                ++thrown;
            }
        }
    }
    printf("parsed %ld, refused %ld\n", ok, thrown);
    return 0;
}
