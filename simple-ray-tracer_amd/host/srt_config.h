/*
 * srt_config.h -- configuration of the command-line front end: the reference's Config / CameraConfig /
 * readConfigFile / serialize / deserialize (src/config.h:16-41, src/config.cpp:192-432) with the same TOML keys.
 */
#pragma once
#include "srt_host.h"

#include <filesystem>
#include <iosfwd>
#include <optional>
#include <string>
#include <variant>
#include <vector>

namespace srt {

struct CameraConfig {
    float fieldOfView = 50.0f; /* degrees */
    float distanceFromLookAt = 3.0f;
    vec3 lookAt = { 0.0f, 0.0f, 0.0f };
    vec3 rotation = { 20.0f, 20.0f, 0.0f }; /* degrees */
};

/* Keys the reference does not have, all under the table [mi355x]; every one is optional. */
struct BackendConfig {
    int gpu = 0;
    std::vector<int> gpus;           /* "gpus = [0, 1, ...]": the GPUs of this node that share the frames' tiles (overrides gpu); the
                                        same GPU may be listed more than once */
    ivec2 tile = { 0, 0 };           /* "tile = [w, h]": tile size when several GPUs share a frame; 0: 480 x 270 */
    int numRays = 1;                 /* the reference's command-line path always renders with numRays = 1 (src/main.cpp:45,476) */
    bool enableSoftShadow = false;   /* the reference's reader never sets Features::enableSoftShadow */
    bool enableTransparency = false; /* nor extra.enableTransparency (only its viewer's check box does) */
    bool referenceTraversal = false; /* "traversal" = "reference" | "miss-cull" | "miss-cull-exact" */
    bool exactShadowBoxes = false;   /* "miss-cull-exact": shadow queries on full-precision boxes */
    int maxDepth = 5;
    uint32_t seed = 0;
    /* The Scene's motion-blur / depth-of-field variables, which the reference only sets from sliders of its viewer
     * (src/main.cpp:254-262); its command-line path leaves DOF_samples at 0 and MB_samples uninitialised. */
    int mbSamples = 0, dofSamples = 0, focalLength = 0;
    float time0 = 0.0f, time1 = 0.0f, aperture = 0.0f;
    vec3 direction = { 0.0f, 0.0f, 0.0f };
};

struct Config {
    Features features = {};
    bool cliRenderingEnabled = false;
    ivec2 windowSize = { 800, 800 };
    std::filesystem::path dataPath;
    std::variant<SceneType, std::filesystem::path> scene = SingleTriangle;
    std::filesystem::path outputDir = "";
    std::vector<CameraConfig> cameras;
    std::vector<Light> lights;
    BackendConfig backend;
};

/* Directory of the prebuilt scenes' files: $SRT_DATA_DIR, else "data/" under the working directory (the reference
 * compiles its source tree's data directory in as DATA_DIR). */
std::filesystem::path defaultDataDir();

std::ostream& operator<<(std::ostream& os, const Config& config);

/* Throws std::runtime_error where the reference prints an error and exits, and also where it would dereference a
 * missing mandatory key (command_line_rendering and the six [features] switches). */
Config readConfigFile(const std::filesystem::path& configPath);
Config readConfigText(const std::string& text);

std::string serialize(const SceneType& sceneType);
std::optional<SceneType> deserialize(const std::string& sceneTypeStr);

} // namespace srt
