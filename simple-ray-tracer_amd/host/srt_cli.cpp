/*
 * srt_cli.cpp -- command-line rendering: `srt_render config.toml`, the flow of the reference's
 * `FinalProject config.toml` with command_line_rendering = true (src/main.cpp:434-491): read the configuration,
 * load the scene, build the BVH, render one image per [[cameras]] entry and write
 * <output_dir>/<scene>_<start time>_cam_<index>.bmp.  Rendering happens on the GPU; there is no interactive mode.
 *
 *   srt_render config.toml            render
 *   srt_render --check config.toml    parse and print the configuration only (needs no GPU)
 */
#include "srt_config.h"

#include <chrono>
#include <cstdio>
#include <ctime>
#include <iostream>

using namespace srt;

int main(int argc, char** argv)
{
    try {
        bool checkOnly = false;
        const char* configPath = nullptr;
        for (int i = 1; i < argc; ++i) {
            if (std::string(argv[i]) == "--check")
                checkOnly = true;
            else
                configPath = argv[i];
        }
        if (!configPath) {
            std::cerr << "usage: srt_render [--check] config.toml   (without a configuration the reference opens its interactive viewer, "
                         "which is not part of this program)\n";
            return 2;
        }
        const Config config = readConfigFile(configPath);
        std::cout << config;
        if (checkOnly)
            return 0;
        if (!config.cliRenderingEnabled) {
            std::cerr << "command_line_rendering = false asks for the interactive viewer, which is not part of this program\n";
            return 2;
        }
        if (config.windowSize.x <= 0 || config.windowSize.y <= 0) {
            std::cerr << "window_size must be an array of two positive integers\n";
            return 2;
        }
        setDevice(config.backend.gpu);

        Scene scene;
        std::string sceneName;
        if (const auto* path = std::get_if<std::filesystem::path>(&config.scene)) {
            scene = loadSceneFromFile(*path, config.lights);
            sceneName = path->stem().string();
        } else {
            const SceneType type = std::get<SceneType>(config.scene);
            scene = loadScenePrebuilt(type, config.dataPath);
            sceneName = serialize(type);
        }
        scene.MB_samples = config.backend.mbSamples, scene.time0 = config.backend.time0, scene.time1 = config.backend.time1;
        scene.directionVector = config.backend.direction;
        scene.DOF_samples = config.backend.dofSamples, scene.focalLength = config.backend.focalLength, scene.aperture = config.backend.aperture;
        /* the reference reads the environment map from its compiled-in data directory (src/render.cpp:38) */
        if (config.features.extra.enableEnvironmentMapping)
            scene.environment = loadEnvironment(config.dataPath.empty() ? defaultDataDir() : config.dataPath);
        Features features = config.features;
        features.enableSoftShadow = config.backend.enableSoftShadow;
        features.extra.enableTransparency = config.backend.enableTransparency;
        BvhInterface bvh { &scene, features };

        using clock = std::chrono::high_resolution_clock;
        if (!std::filesystem::exists(config.outputDir))
            std::filesystem::create_directories(config.outputDir);
        const auto start = clock::now();
        char startTime[64];
        const std::time_t now = std::time(nullptr);
        std::strftime(startTime, sizeof startTime, "%Y-%m-%d-%H:%M:%S", std::localtime(&now));

        RenderOptions options;
        options.referenceTraversal = config.backend.referenceTraversal;
        options.exactShadowBoxes = config.backend.exactShadowBoxes;
        options.maxDepth = config.backend.maxDepth;
        options.seed = config.backend.seed;
        options.tileWidth = config.backend.tile.x, options.tileHeight = config.backend.tile.y;
        /* [mi355x] gpus = [...]: one more copy of the scene per further entry */
        for (size_t k = 1; k < config.backend.gpus.size(); ++k)
            bvh.addReplica(config.backend.gpus[k]);
        /* The reference starts one host thread per camera (src/main.cpp:468-487).  Here all cameras' frames go to the
         * back end in one call: on one GPU they are rendered one after the other (its launches are stream ordered), with
         * several GPUs the tiles of all frames form one work list that one host thread per GPU draws from. */
        std::vector<Screen> screens;
        std::vector<Trackball> cameras;
        screens.reserve(config.cameras.size()), cameras.reserve(config.cameras.size());
        for (const CameraConfig& cameraConfig : config.cameras) {
            screens.emplace_back(config.windowSize, false);
            screens.back().clear(vec3(0.0f));
            cameras.emplace_back(float(config.windowSize.x) / float(config.windowSize.y), radians(cameraConfig.fieldOfView), cameraConfig.distanceFromLookAt);
            cameras.back().setCamera(cameraConfig.lookAt, radians(cameraConfig.rotation), cameraConfig.distanceFromLookAt);
        }
        std::vector<FrameJob> frames;
        for (size_t k = 0; k < screens.size(); ++k)
            frames.push_back({ &cameras[k], &screens[k] });
        renderFrames(scene, bvh, frames, features, config.backend.numRays, options);
        for (size_t index = 0; index < screens.size(); ++index) {
            const std::filesystem::path filepath = config.outputDir / (sceneName + "_" + startTime + "_cam_" + std::to_string(index) + ".bmp");
            std::printf("Image %zu saved to %s\n", index, filepath.string().c_str());
            screens[index].writeBitmapToFile(filepath);
        }
        const auto duration = std::chrono::duration_cast<std::chrono::milliseconds>(clock::now() - start).count();
        std::printf("Rendering took %lld ms, %zu images rendered.\n", (long long)duration, config.cameras.size());
        return 0;
    } catch (const std::exception& e) {
        std::cerr << e.what() << "\n";
        return 1;
    }
}
