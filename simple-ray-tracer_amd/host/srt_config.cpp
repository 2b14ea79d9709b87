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

static const char* sceneTypeName(SceneType t)
{
    switch (t) {
    case SingleTriangle: return "SceneType::SingleTriangle";
    case Cube: return "SceneType::Cube";
    case CubeTextured: return "SceneType::CubeTextured";
    case CornellBox: return "SceneType::CornellBox";
    case CornellBoxParallelogramLight: return "SceneType::CornellBoxParallelogramLight";
    case Monkey: return "SceneType::Monkey";
    case Teapot: return "SceneType::Teapot";
    case Dragon: return "SceneType::Dragon";
    case Spheres: return "SceneType::Spheres";
    case Custom: return "SceneType::Custom";
    }
    return "SceneType::?";
}

static std::ostream& operator<<(std::ostream& os, const vec3& v) { return os << "[" << v.x << ", " << v.y << ", " << v.z << "]"; }

/* Layout of the reference's printer (src/config.cpp:71-148), followed by the back-end keys. */
std::ostream& operator<<(std::ostream& os, const Config& config)
{
    const Features& f = config.features;
    os << "Final Project Configurations: " << std::endl
       << std::boolalpha << "  + command_line_rendering: " << config.cliRenderingEnabled << std::endl
       << "  + window_size: " << config.windowSize.x << ", " << config.windowSize.y << std::endl
       << "  + data_path: " << config.dataPath << std::endl
       << "  + scene: ";
    if (std::holds_alternative<SceneType>(config.scene))
        os << sceneTypeName(std::get<SceneType>(config.scene)) << std::endl;
    else
        os << std::get<std::filesystem::path>(config.scene) << std::endl;
    os << "  + output_filepath: " << config.outputDir << std::endl
       << "  + features: " << std::endl
       << "    - enable_shading: " << f.enableShading << std::endl
       << "    - enable_recursive: " << f.enableRecursive << std::endl
       << "    - enable_hard_shadow: " << f.enableHardShadow << std::endl
       << "    - enable_soft_shadow: " << f.enableSoftShadow << std::endl
       << "    - enable_normal_interp: " << f.enableNormalInterp << std::endl
       << "    - enable_texture_mapping: " << f.enableTextureMapping << std::endl
       << "    - enable_accel_structure: " << f.enableAccelStructure << std::endl
       << "  + extra_features: " << std::endl
       << "    - enable_bloom_effect: " << f.extra.enableBloomEffect << std::endl
       << "    - enable_multiple_rays_per_pixel: " << f.extra.enableMultipleRaysPerPixel << std::endl
       << "    - enable_motion_blur: " << f.extra.enableMotionBlur << std::endl
       << "    - enable_depth_of_field: " << f.extra.enableDepthOfField << std::endl
       << "    - enable_glossy_reflection: " << f.extra.enableGlossyReflection << std::endl
       << "    - enable_transparency: " << f.extra.enableTransparency << std::endl
       << "    - enable_bvh_sah_binning: " << f.extra.enableBvhSahBinning << std::endl
       << "    - enable_environment_mapping: " << f.extra.enableEnvironmentMapping << std::endl
       << "    - enable_bilinear_texture_filtering: " << f.extra.enableBilinearTextureFiltering << std::endl
       << "    - enable_mipmap_texture_filtering: " << f.extra.enableMipmapTextureFiltering << std::endl;
    os << "  + cameras: " << std::endl;
    for (const CameraConfig& camera : config.cameras)
        os << "    - field_of_view: " << camera.fieldOfView << std::endl
           << "      distance_from_look_at: " << camera.distanceFromLookAt << std::endl
           << "      look_at: " << camera.lookAt << std::endl
           << "      rotation: " << camera.rotation << std::endl;
    os << "  + lights: " << std::endl;
    for (const Light& elem : config.lights) {
        if (const auto* p = std::get_if<PointLight>(&elem)) {
            os << "    - type: point" << std::endl << "      position: " << p->position << ", color: " << p->color << std::endl;
        } else if (const auto* s = std::get_if<SegmentLight>(&elem)) {
            os << "    - type: segment" << std::endl
               << "      endpoint0: " << s->endpoint0 << ", endpoint1: " << s->endpoint1 << std::endl
               << "      color0: " << s->color0 << ", color1: " << s->color1 << std::endl;
        } else if (const auto* q = std::get_if<ParallelogramLight>(&elem)) {
            os << "    - type: parallelogram" << std::endl
               << "      v0: " << q->v0 << std::endl
               << "      edge01: " << q->edge01 << ", edge02: " << q->edge02 << std::endl
               << "      color0: " << q->color0 << ", color1: " << q->color1 << std::endl
               << "      color2: " << q->color2 << ", color3: " << q->color3 << std::endl;
        }
    }
    const BackendConfig& b = config.backend;
    os << "  + mi355x: " << std::endl
       << "    - gpu: " << b.gpu << std::endl
       << "    - num_rays: " << b.numRays << std::endl
       << "    - enable_soft_shadow: " << b.enableSoftShadow << std::endl
       << "    - enable_transparency: " << b.enableTransparency << std::endl
       << "    - traversal: " << (b.referenceTraversal ? "reference" : "miss-cull") << std::endl
       << "    - max_depth: " << b.maxDepth << std::endl
       << "    - seed: " << b.seed << std::endl
       << "    - motion blur: samples " << b.mbSamples << ", time0 " << b.time0 << ", time1 " << b.time1 << ", direction [" << b.direction.x << ", "
       << b.direction.y << ", " << b.direction.z << "]" << std::endl
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
    config.features.enableShading = mandatoryBool(features["enable_shading"], "features.enable_shading");
    config.features.enableRecursive = mandatoryBool(features["enable_recursive"], "features.enable_recursive");
    config.features.enableHardShadow = mandatoryBool(features["enable_hard_shadow"], "features.enable_hard_shadow");
    config.features.enableNormalInterp = mandatoryBool(features["enable_normal_interp"], "features.enable_normal_interp");
    config.features.enableTextureMapping = mandatoryBool(features["enable_texture_mapping"], "features.enable_texture_mapping");
    config.features.enableAccelStructure = mandatoryBool(features["enable_accel_structure"], "features.enable_accel_structure");
    const toml::Value& extra = features["extra"];
    optionalBool(extra["enable_bloom_effect"], config.features.extra.enableBloomEffect);
    optionalBool(extra["enable_multiple_rays_per_pixel"], config.features.extra.enableMultipleRaysPerPixel);
    optionalBool(extra["enable_motion_blur"], config.features.extra.enableMotionBlur);
    optionalBool(extra["enable_depth_of_field"], config.features.extra.enableDepthOfField);
    optionalBool(extra["enable_glossy_reflection"], config.features.extra.enableGlossyReflection);
    optionalBool(extra["enable_environment_mapping"], config.features.extra.enableEnvironmentMapping);
    optionalBool(extra["enable_bilinear_texture_filtering"], config.features.extra.enableBilinearTextureFiltering);
    optionalBool(extra["enable_mipmap_texture_filtering"], config.features.extra.enableMipmapTextureFiltering);

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
            else if (backend["traversal"].s != "miss-cull")
                throw std::runtime_error("configuration: mi355x.traversal must be \"reference\" or \"miss-cull\"");
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
    switch (sceneType) {
    case SingleTriangle: return "single_triangle";
    case Cube: return "cube";
    case CubeTextured: return "cube_textured";
    case CornellBox: return "cornell_box";
    case CornellBoxParallelogramLight: return "cornell_box_parallelogram_light";
    case Monkey: return "monkey";
    case Teapot: return "teapot";
    case Dragon: return "dragon";
    case Spheres: return "spheres";
    case Custom: return "custom";
    default: return "unknown";
    }
}

std::optional<SceneType> deserialize(const std::string& sceneTypeStr)
{
    std::string s;
    std::transform(sceneTypeStr.begin(), sceneTypeStr.end(), std::back_inserter(s), [](unsigned char c) { return (char)std::tolower(c); });
    if (s == "single_triangle" || s == "singletriangle" || s == "single-triangle") return SingleTriangle;
    if (s == "cube") return Cube;
    if (s == "cube-textured" || s == "cube_textured" || s == "cubetextured") return CubeTextured;
    if (s == "cornell_box" || s == "cornellbox" || s == "cornell-box") return CornellBox;
    if (s == "cornell_box_parallelogram_light" || s == "cornellboxparallelogramlight" || s == "cornell-box-parallelogram-light") return CornellBoxParallelogramLight;
    if (s == "monkey") return Monkey;
    if (s == "teapot") return Teapot;
    if (s == "dragon") return Dragon;
    if (s == "spheres") return Spheres;
    if (s == "custom") return Custom;
    return std::nullopt;
}

} // namespace srt
