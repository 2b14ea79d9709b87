/* srt_config.cpp -- see srt_config.h.  Key names, defaults and quirks follow the reference's reader. */
#include "srt_config.h"
#include "srt_toml.h"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>

namespace srt {

std::filesystem::path defaultDataDir()
{
    if (const char* env = std::getenv("SRT_DATA_DIR"))
        return env;
    return std::filesystem::current_path() / "data";
}

/* The prebuilt scenes: enumerator name as the reference's printer writes it, and the key of a configuration file.  The
 * reader takes the key as written, with hyphens for the underscores, or with the underscores left out, in any letter case
 * (reference src/config.cpp:404-431); the writer emits the key (reference src/config.cpp:376-402). */
static const struct SceneName {
    SceneType type;
    const char* enumerator;
    const char* key;
} kSceneNames[] = {
    { SingleTriangle, "SingleTriangle", "single_triangle" },
    { Cube, "Cube", "cube" },
    { CubeTextured, "CubeTextured", "cube_textured" },
    { CornellBox, "CornellBox", "cornell_box" },
    { CornellBoxParallelogramLight, "CornellBoxParallelogramLight", "cornell_box_parallelogram_light" },
    { Monkey, "Monkey", "monkey" },
    { Teapot, "Teapot", "teapot" },
    { Dragon, "Dragon", "dragon" },
    { Spheres, "Spheres", "spheres" },
    { Custom, "Custom", "custom" },
};

/* The feature switches of a configuration file, in the order the report lists them.  `required`: the reader insists on
 * the key (the reference dereferences it unchecked); `read`: false for switches the reference's reader never looks at
 * although its report prints them. */
struct Switch {
    const char* key;
    bool Features::*core;
    bool ExtraFeatures::*extra;
    bool read;
};
static const Switch kCoreSwitches[] = {
    { "enable_shading", &Features::enableShading, nullptr, true },
    { "enable_recursive", &Features::enableRecursive, nullptr, true },
    { "enable_hard_shadow", &Features::enableHardShadow, nullptr, true },
    { "enable_soft_shadow", &Features::enableSoftShadow, nullptr, false },
    { "enable_normal_interp", &Features::enableNormalInterp, nullptr, true },
    { "enable_texture_mapping", &Features::enableTextureMapping, nullptr, true },
    { "enable_accel_structure", &Features::enableAccelStructure, nullptr, true },
};
static const Switch kExtraSwitches[] = {
    { "enable_bloom_effect", nullptr, &ExtraFeatures::enableBloomEffect, true },
    { "enable_multiple_rays_per_pixel", nullptr, &ExtraFeatures::enableMultipleRaysPerPixel, true },
    { "enable_motion_blur", nullptr, &ExtraFeatures::enableMotionBlur, true },
    { "enable_depth_of_field", nullptr, &ExtraFeatures::enableDepthOfField, true },
    { "enable_glossy_reflection", nullptr, &ExtraFeatures::enableGlossyReflection, true },
    { "enable_transparency", nullptr, &ExtraFeatures::enableTransparency, false },
    { "enable_bvh_sah_binning", nullptr, &ExtraFeatures::enableBvhSahBinning, false },
    { "enable_environment_mapping", nullptr, &ExtraFeatures::enableEnvironmentMapping, true },
    { "enable_bilinear_texture_filtering", nullptr, &ExtraFeatures::enableBilinearTextureFiltering, true },
    { "enable_mipmap_texture_filtering", nullptr, &ExtraFeatures::enableMipmapTextureFiltering, true },
};

namespace {
/* The report is an indented list: "  + " opens a group, "    - " an entry, six blanks continue an entry. */
struct Report {
    std::ostream& os;
    template <class T> void line(const char* lead, const char* name, const T& value) { os << lead << name << ": " << value << std::endl; }
    void group(const char* name) { os << "  + " << name << ": " << std::endl; }
    template <class T> void top(const char* name, const T& value) { line("  + ", name, value); }
    template <class T> void entry(const char* name, const T& value) { line("    - ", name, value); }
    template <class T> void more(const char* name, const T& value) { line("      ", name, value); }
};
struct Bracketed {
    const vec3& v;
};
std::ostream& operator<<(std::ostream& os, const Bracketed& b) { return os << "[" << b.v.x << ", " << b.v.y << ", " << b.v.z << "]"; }
/* two bracketed vectors on one line: "a: [..], b: [..]" after the first name */
struct Pair {
    const vec3& first;
    const char* secondName;
    const vec3& second;
};
std::ostream& operator<<(std::ostream& os, const Pair& p) { return os << Bracketed { p.first } << ", " << p.secondName << ": " << Bracketed { p.second }; }
} // namespace

/* Prints what the reference's report prints, line for line (reference src/config.cpp:71-145: tools that read its
 * console output keep working), then the keys of the [mi355x] table. */
std::ostream& operator<<(std::ostream& os, const Config& config)
{
    Report r { os };
    os << "Final Project Configurations: " << std::endl << std::boolalpha;
    r.top("command_line_rendering", config.cliRenderingEnabled);
    os << "  + window_size: " << config.windowSize.x << ", " << config.windowSize.y << std::endl;
    r.top("data_path", config.dataPath);
    if (const SceneType* type = std::get_if<SceneType>(&config.scene)) {
        std::string name = "SceneType::?";
        for (const SceneName& n : kSceneNames)
            if (n.type == *type)
                name = std::string("SceneType::") + n.enumerator;
        r.top("scene", name);
    } else {
        r.top("scene", std::get<std::filesystem::path>(config.scene));
    }
    r.top("output_filepath", config.outputDir);
    r.group("features");
    for (const Switch& sw : kCoreSwitches)
        r.entry(sw.key, config.features.*sw.core);
    r.group("extra_features");
    for (const Switch& sw : kExtraSwitches)
        r.entry(sw.key, config.features.extra.*sw.extra);
    r.group("cameras");
    for (const CameraConfig& camera : config.cameras) {
        r.entry("field_of_view", camera.fieldOfView);
        r.more("distance_from_look_at", camera.distanceFromLookAt);
        r.more("look_at", Bracketed { camera.lookAt });
        r.more("rotation", Bracketed { camera.rotation });
    }
    r.group("lights");
    for (const Light& light : config.lights) {
        if (const auto* p = std::get_if<PointLight>(&light)) {
            r.entry("type", "point");
            r.more("position", Pair { p->position, "color", p->color });
        } else if (const auto* s = std::get_if<SegmentLight>(&light)) {
            r.entry("type", "segment");
            r.more("endpoint0", Pair { s->endpoint0, "endpoint1", s->endpoint1 });
            r.more("color0", Pair { s->color0, "color1", s->color1 });
        } else if (const auto* q = std::get_if<ParallelogramLight>(&light)) {
            r.entry("type", "parallelogram");
            r.more("v0", Bracketed { q->v0 });
            r.more("edge01", Pair { q->edge01, "edge02", q->edge02 });
            r.more("color0", Pair { q->color0, "color1", q->color1 });
            r.more("color2", Pair { q->color2, "color3", q->color3 });
        }
    }
    const BackendConfig& b = config.backend;
    r.group("mi355x");
    r.entry("gpu", b.gpu);
    if (!b.gpus.empty()) {
        os << "    - gpus:";
        for (int g : b.gpus)
            os << " " << g;
        os << ", tile " << (b.tile.x > 0 ? b.tile.x : 480) << " x " << (b.tile.y > 0 ? b.tile.y : 270) << std::endl;
    }
    r.entry("num_rays", b.numRays);
    r.entry("enable_soft_shadow", b.enableSoftShadow);
    r.entry("enable_transparency", b.enableTransparency);
    r.entry("traversal", b.referenceTraversal ? "reference" : b.exactShadowBoxes ? "miss-cull-exact" : "miss-cull");
    r.entry("max_depth", b.maxDepth);
    r.entry("seed", b.seed);
    os << "    - motion blur: samples " << b.mbSamples << ", time0 " << b.time0 << ", time1 " << b.time1 << ", direction " << Bracketed { b.direction } << std::endl
       << "    - depth of field: samples " << b.dofSamples << ", focal_length " << b.focalLength << ", aperture " << b.aperture << std::endl;
    return os;
}

/* tomlArrayToVec3, src/config.cpp:150-169: the first three numbers of the array; a missing array gives zeros (the
 * reference's optional is always engaged, so its value_or defaults never apply). */
static vec3 arrayToVec3(const toml::Value& a)
{
    vec3 out { 0.0f, 0.0f, 0.0f };
    int i = 0;
    for (size_t k = 0; k < a.size(); ++k) {
        const toml::Value& e = a.at(k);
        if (e.isNumber()) {
            if (i > 2)
                continue;
            out[i] = e.kind == toml::Value::Float ? (float)e.f : (float)e.i;
            i += 1;
        } else {
            std::cerr << "Error: Expected a number in array" << std::endl;
        }
    }
    return out;
}

/* tomlArrayToIVec2, src/config.cpp:172-190 */
static ivec2 arrayToIVec2(const toml::Value& a)
{
    ivec2 out { 0, 0 };
    int i = 0;
    for (size_t k = 0; k < a.size(); ++k) {
        const toml::Value& e = a.at(k);
        if (e.kind == toml::Value::Int) {
            if (i > 1)
                continue;
            out[i] = (int)e.i;
            i += 1;
        } else {
            std::cerr << "Error: Expected an integer in array" << std::endl;
        }
    }
    return out;
}

static bool mandatoryBool(const toml::Value& v, const char* key)
{
    if (v.kind != toml::Value::Bool)
        throw std::runtime_error(std::string("configuration: '") + key + "' must be present and a boolean (the reference dereferences it unchecked)");
    return v.b;
}

static void optionalBool(const toml::Value& v, bool& out)
{
    if (v.kind == toml::Value::None)
        return;
    if (v.kind != toml::Value::Bool)
        throw std::runtime_error("configuration: a [features.extra] switch is not a boolean");
    out = v.b;
}

static float number(const toml::Value& v, float fallback)
{
    if (v.kind == toml::Value::Float)
        return (float)v.f;
    if (v.kind == toml::Value::Int)
        return (float)v.i;
    return fallback;
}

/* An integer key of the [mi355x] table: absent -> fallback; anything but a TOML integer inside [lo, hi] is an error. */
static int64_t integerKey(const toml::Value& v, const char* key, int64_t fallback, int64_t lo, int64_t hi)
{
    if (v.kind == toml::Value::None)
        return fallback;
    if (v.kind != toml::Value::Int || v.i < lo || v.i > hi)
        throw std::runtime_error(std::string("configuration: mi355x.") + key + " must be an integer in [" + std::to_string(lo) + ", " + std::to_string(hi) + "]");
    return v.i;
}

static Config fromTable(const toml::Value& table)
{
    Config config = {};
    config.cliRenderingEnabled = mandatoryBool(table["command_line_rendering"], "command_line_rendering");
    config.windowSize = arrayToIVec2(table["window_size"]);

    std::string dataPath = table["data_path"].kind == toml::Value::String ? table["data_path"].s : std::string("default");
    config.dataPath = dataPath == "default" ? defaultDataDir() : std::filesystem::path(dataPath);

    const toml::Value& scene = table["scene"];
    if (scene.isNumber()) {
        config.scene = static_cast<SceneType>(scene.kind == toml::Value::Int ? scene.i : (int64_t)scene.f);
    } else {
        const std::string scenePath = scene.kind == toml::Value::String ? scene.s : std::string("none");
        if (const auto type = deserialize(scenePath)) {
            config.scene = *type;
        } else {
            const std::filesystem::path path = config.dataPath / scenePath;
            if (!std::filesystem::exists(path))
                throw std::runtime_error("Error: Scene file " + path.string() + " does not exist.");
            config.scene = path;
        }
    }

    std::string outputDir = table["output_dir"].kind == toml::Value::String ? table["output_dir"].s : std::string();
    if (outputDir.empty()) {
        std::cout << "Warning: No output directory specified, using current directory." << std::endl;
        config.outputDir = std::filesystem::current_path();
    } else {
        const char* home = std::getenv("HOME");
        if (home && outputDir[0] == '~')
            outputDir.replace(0, 1, home);
        if (home && outputDir.rfind("$HOME", 0) == 0)
            outputDir.replace(0, 5, home);
        config.outputDir = std::filesystem::absolute(std::filesystem::path(outputDir));
    }

    const toml::Value& features = table["features"];
    for (const Switch& sw : kCoreSwitches)
        if (sw.read)
            config.features.*sw.core = mandatoryBool(features[sw.key], (std::string("features.") + sw.key).c_str());
    const toml::Value& extra = features["extra"];
    for (const Switch& sw : kExtraSwitches)
        if (sw.read)
            optionalBool(extra[sw.key], config.features.extra.*sw.extra);

    const toml::Value& cameras = table["cameras"];
    for (size_t k = 0; k < cameras.size(); ++k) {
        const toml::Value& camera = cameras.at(k);
        CameraConfig c;
        c.fieldOfView = number(camera["field_of_view"], 50.0f);
        c.distanceFromLookAt = number(camera["distance_from_look_at"], 3.0f);
        c.lookAt = arrayToVec3(camera["look_at"]);
        c.rotation = arrayToVec3(camera["rotation"]);
        config.cameras.push_back(c);
    }

    const toml::Value& lights = table["lights"];
    if (lights.kind == toml::Value::ArrayKind) {
        for (size_t k = 0; k < lights.size(); ++k) {
            const toml::Value& light = lights.at(k);
            const std::string type = light["type"].kind == toml::Value::String ? light["type"].s : std::string("none");
            if (type == "point") {
                config.lights.emplace_back(PointLight { arrayToVec3(light["position"]), arrayToVec3(light["color"]) });
            } else if (type == "segment") {
                const toml::Value &e = light["endpoints"], &c = light["colors"];
                config.lights.emplace_back(SegmentLight { arrayToVec3(e.at(0)), arrayToVec3(e.at(1)), arrayToVec3(c.at(0)), arrayToVec3(c.at(1)) });
            } else if (type == "parallelogram") {
                const toml::Value &e = light["edges"], &c = light["colors"];
                config.lights.emplace_back(ParallelogramLight { arrayToVec3(light["corner"]), arrayToVec3(e.at(0)), arrayToVec3(e.at(1)), arrayToVec3(c.at(0)),
                    arrayToVec3(c.at(1)), arrayToVec3(c.at(2)), arrayToVec3(c.at(3)) });
            } else {
                std::cerr << "Unknown light type: " << type << " -- Skip" << std::endl;
            }
        }
    } else {
        std::cerr << "WARN: No lights found in config file.\n";
    }

    const toml::Value& backend = table["mi355x"];
    if (backend.kind == toml::Value::TableKind) {
        BackendConfig& b = config.backend;
        b.gpu = (int)integerKey(backend["gpu"], "gpu", 0, 0, 1023);
        const toml::Value& gpus = backend["gpus"];
        if (gpus.kind != toml::Value::None) {
            if (gpus.kind != toml::Value::ArrayKind || gpus.size() == 0 || gpus.size() > 64)
                throw std::runtime_error("configuration: mi355x.gpus must be an array of 1 to 64 GPU numbers");
            for (size_t k = 0; k < gpus.size(); ++k)
                b.gpus.push_back((int)integerKey(gpus.at(k), "gpus[]", 0, 0, 1023));
            b.gpu = b.gpus[0];
        }
        const toml::Value& tile = backend["tile"];
        if (tile.kind != toml::Value::None) {
            if (tile.kind != toml::Value::ArrayKind || tile.size() != 2)
                throw std::runtime_error("configuration: mi355x.tile must be an array of two positive integers");
            b.tile.x = (int)integerKey(tile.at(0), "tile[0]", 0, 8, 1 << 16);
            b.tile.y = (int)integerKey(tile.at(1), "tile[1]", 0, 8, 1 << 16);
        }
        b.numRays = (int)integerKey(backend["num_rays"], "num_rays", 1, 1, 64);
        b.maxDepth = (int)integerKey(backend["max_depth"], "max_depth", 5, 0, 30);
        b.seed = (uint32_t)integerKey(backend["seed"], "seed", 0, 0, 0xFFFFFFFFll);
        optionalBool(backend["enable_soft_shadow"], b.enableSoftShadow);
        optionalBool(backend["enable_transparency"], b.enableTransparency);
        b.mbSamples = (int)integerKey(backend["mb_samples"], "mb_samples", 0, 0, 4096);
        b.dofSamples = (int)integerKey(backend["dof_samples"], "dof_samples", 0, 0, 4096);
        b.focalLength = (int)integerKey(backend["focal_length"], "focal_length", 0, 0, 1000000);
        b.time0 = number(backend["time0"], 0.0f), b.time1 = number(backend["time1"], 0.0f);
        b.aperture = number(backend["aperture"], 0.0f);
        if (backend["direction"].kind == toml::Value::ArrayKind)
            b.direction = arrayToVec3(backend["direction"]);
        if (backend["traversal"].kind == toml::Value::String) {
            if (backend["traversal"].s == "reference")
                b.referenceTraversal = true;
            else if (backend["traversal"].s == "miss-cull-exact")
                b.exactShadowBoxes = true;
            else if (backend["traversal"].s != "miss-cull")
                throw std::runtime_error("configuration: mi355x.traversal must be \"reference\", \"miss-cull\" or \"miss-cull-exact\"");
        }
    }
    return config;
}

Config readConfigText(const std::string& text) { return fromTable(toml::parse(text)); }

Config readConfigFile(const std::filesystem::path& configPath)
{
    try {
        return fromTable(toml::parseFile(configPath.string()));
    } catch (const std::runtime_error& e) {
        throw std::runtime_error("Failed parsing " + configPath.string() + ": " + e.what());
    }
}

std::string serialize(const SceneType& sceneType)
{
    for (const SceneName& n : kSceneNames)
        if (n.type == sceneType)
            return n.key;
    return "unknown";
}

std::optional<SceneType> deserialize(const std::string& sceneTypeStr)
{
    std::string wanted;
    for (unsigned char c : sceneTypeStr)
        wanted.push_back((char)std::tolower(c));
    for (const SceneName& n : kSceneNames) {
        std::string hyphens, joined;
        for (const char* c = n.key; *c; ++c) {
            hyphens.push_back(*c == '_' ? '-' : *c);
            if (*c != '_')
                joined.push_back(*c);
        }
        if (wanted == n.key || wanted == hyphens || wanted == joined)
            return n.type;
    }
    return std::nullopt;
}

} // namespace srt
