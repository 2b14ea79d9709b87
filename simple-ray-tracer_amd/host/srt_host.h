/*
 * srt_host.h -- C++ host mirror of the reference's Scene / BVH / Screen / camera / render interface for the
 * MI355X back end.  Same type and function names, argument meaning and error behaviour as the reference, in
 * namespace srt; the bodies build the BVH on the host exactly as the reference does, flatten it and call
 * libsrt_hip.so through the C ABI of include/srt_hip.h.  There is no CPU rendering path: without a GPU every
 * call that would trace a ray throws.
 *
 * Mirrors (reference file:line):
 *   Vertex / Material / Mesh / Image     framework/include/framework/mesh.h:14-43, image.h:11-18
 *   Ray                                  framework/include/framework/ray.h:9-13
 *   HitInfo / AxisAlignedBox / lights / Features / ExtraFeatures      src/common.h:14-80
 *   Scene / SceneType / loadScene*       src/scene.h:15-48
 *   BoundingVolumeHierarchy              src/bounding_volume_hierarchy.h:15-72, .cpp:14-94
 *   BvhInterface                         src/bvh_interface.h:12-42
 *   Screen                               src/screen.h, src/screen.cpp:80-104,153-171
 *   Trackball (window-free)              framework/include/framework/trackball.h, framework/src/trackball.cpp
 *   renderRayTracing / getFinalColor     src/render.h:17-20
 */
#pragma once
#include <cstddef>
#include <cstdint>
#include <filesystem>
#include <functional>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "../../include/srt_hip.h"

namespace srt {

struct vec2 {
    float x = 0, y = 0;
};
struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
struct uvec3 {
    uint32_t x = 0, y = 0, z = 0;
    uint32_t operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct ivec2 {
    int x = 0, y = 0;
    int operator[](int i) const { return i == 0 ? x : y; }
    int& operator[](int i) { return i == 0 ? x : y; }
};

struct Image {
    Image() = default;
    explicit Image(const std::filesystem::path& filePath); /* PNG or JPEG, decoded to the pixels stb_image gives; throws std::exception like the reference */
    int width = 0, height = 0;
    std::vector<vec3> pixels; /* u8 / 255.0f, row-major, top row first */
};

struct Vertex {
    vec3 position;
    vec3 normal;
    vec2 texCoord;
};

struct Material {
    vec3 kd;
    vec3 ks { 0.0f };
    float shininess { 1.0f };
    float transparency { 1.0f };
    std::shared_ptr<Image> kdTexture;
};

struct Mesh {
    std::vector<Vertex> vertices;
    std::vector<uvec3> triangles;
    Material material;
};

[[nodiscard]] std::vector<Mesh> loadMesh(const std::filesystem::path& file, bool normalize = false);

struct Ray {
    vec3 origin { 0.0f };
    vec3 direction { 0.0f, 0.0f, -1.0f };
    float t { std::numeric_limits<float>::max() };
};

struct HitInfo {
    vec3 normal;
    vec3 barycentricCoord;
    vec2 texCoord;
    Material material;
    int depthOfRecursion = 0;
};

struct AxisAlignedBox {
    vec3 lower { 0.0f };
    vec3 upper { 1.0f };
};

struct Sphere {
    vec3 center { 0.0f };
    float radius = 1.0f;
    Material material;
};

struct PointLight {
    vec3 position;
    vec3 color;
};
struct SegmentLight {
    vec3 endpoint0, endpoint1;
    vec3 color0, color1;
};
struct ParallelogramLight {
    vec3 v0;
    vec3 edge01, edge02;
    vec3 color0, color1, color2, color3;
};

struct ExtraFeatures {
    bool enableEnvironmentMapping = false;
    bool enableBvhSahBinning = false;
    bool enableMotionBlur = false;
    bool enableBloomEffect = false;
    bool enableBilinearTextureFiltering = false;
    bool enableMipmapTextureFiltering = false;
    bool enableMultipleRaysPerPixel = false;
    bool enableGlossyReflection = false;
    bool enableTransparency = false;
    bool enableDepthOfField = false;
};

struct Features {
    bool enableShading = false;
    bool enableRecursive = false;
    bool enableHardShadow = false;
    bool enableSoftShadow = false;
    bool enableNormalInterp = false;
    bool enableTextureMapping = false;
    bool enableAccelStructure = false;
    bool debugOptimisedNodes = false;
    bool enableDraw = false;
    ExtraFeatures extra;
};

enum SceneType {
    SingleTriangle,
    Cube,
    CubeTextured,
    CornellBox,
    CornellBoxParallelogramLight,
    Monkey,
    Teapot,
    Dragon,
    Spheres,
    Custom,
};

using Light = std::variant<PointLight, SegmentLight, ParallelogramLight>;

struct Scene {
    SceneType type = SingleTriangle;
    std::vector<Mesh> meshes;
    std::vector<Sphere> spheres;
    std::vector<Light> lights;
    int MB_samples = 0;
    float time0 = 0.0, time1 = 0.0;
    vec3 directionVector = { 0, 0, 0 };
    int focalLength = 0;
    float aperture = 0;
    int DOF_samples = 0;
    /* The image extra.enableEnvironmentMapping looks misses up in.  The reference keeps it in a global loaded from
     * DATA_DIR/cube_map.jpg when the program starts (src/render.cpp:38); here it belongs to the scene and must be
     * set (loadEnvironment, or any Image) before the BvhInterface is constructed. */
    std::shared_ptr<Image> environment;
};

/* Image of <dataDir>/cube_map.jpg, the file the reference's environment map comes from. */
std::shared_ptr<Image> loadEnvironment(const std::filesystem::path& dataDir);

Scene loadScenePrebuilt(SceneType type, const std::filesystem::path& dataDir);
Scene loadSceneFromFile(const std::filesystem::path& path, const std::vector<Light>& lights);

/* ---- back-end selection and policy (no counterpart in the reference) */

/* GPU used by objects created afterwards on this thread (default 0).  context selects one of several independent
 * back-end handles on that GPU (each with its own work buffers and scene copies): renders through BVHs of
 * different contexts may be in flight at once on different streams, e.g. so that one tile's tail overlaps the
 * next tile's start. */
void setDevice(int gpuId, int context = 0);
int currentDevice();
/* Lazily created back-end handle for a value of currentDevice(); throws std::runtime_error when no GPU is usable. */
srt_device* deviceFor(int deviceKey);

struct RenderOptions {
    bool referenceTraversal = false; /* true: the reference's exact visiting rule (full-tree shadow walks) */
    bool exactShadowBoxes = false;   /* true: shadow queries test the full-precision child boxes (SRT_F_MISS_CULL alone) */
    int maxDepth = 5;                /* the reference hard-codes 5 (src/render.cpp:107) */
    int samplesX = 0, samplesY = 0;  /* 0: numRays x numRays as in the reference */
    uint32_t seed = 0;
    float bloomThreshold = 0.5f;     /* what the reference's callers pass (src/main.cpp:39-42,476): threshold 0.5, */
    int bloomBoxSize = 1;            /* boxSize 2 * 0 + 1; used only with extra.enableBloomEffect */
    int rect[4] = { 0, 0, 0, 0 };    /* x0,y0,x1,y1; all zero = whole frame */
    bool stats = false;
    bool timing = false;
    void* stream = nullptr;
    int tileWidth = 0, tileHeight = 0; /* tiles a frame is cut into when the BVH has replicas on several GPUs; 0: 480 x 270 */
};

/* ---- BVH */

struct TriangleOrNode {
    size_t meshIndex;
    size_t triangleIndex;
    size_t nodeIndex;
};

struct Node {
    bool isLeaf = false;
    int level = 0;
    std::vector<TriangleOrNode> triangles;
    AxisAlignedBox box;
};

/* Flattened arrays handed to srt_scene_upload. */
struct FlatScene {
    std::vector<srt_node> nodes;
    std::vector<srt_tri> tris;
    std::vector<srt_tri_attr> attrs;
    std::vector<uint32_t> sceneOrder;
    std::vector<srt_material> materials;
    std::vector<srt_texture> textures;
    std::vector<float> texels;
    std::vector<srt_light> lights;
    float rootBox[6] = { 0, 0, 0, 0, 0, 0 };
    uint32_t rootRef = SRT_LEAF_FLAG;
    uint32_t numLevels = 0;
    uint32_t environment = 0; /* 1 + index of the environment image in textures, 0 for none */
    srt_scene_desc desc() const;
};

class BoundingVolumeHierarchy {
public:
    BoundingVolumeHierarchy(Scene* pScene, const Features& features);
    /* The reference's two recursive builders.  `out` receives the subtree's nodes in post-order (indices local to
     * out); the top `forkLevels` levels build their two subtrees as parallel tasks and splice them together. */
    size_t constructorHelper(std::vector<Node>& out, std::vector<TriangleOrNode>& triangles, size_t left, size_t right, int whichAxis, int level, int forkLevels);
    size_t sahConstructorHelper(std::vector<Node>& out, std::vector<TriangleOrNode>& triangles, size_t left, size_t right, int whichAxis, int level, int forkLevels);
    [[nodiscard]] int numLevels() const { return m_numLevels; }
    [[nodiscard]] int numLeaves() const { return m_numLeaves; }
    const std::vector<Node>& nodeTable() const { return nodes; } /* post-order, root last, as in the reference */
    /* bfsLevels: how many top levels are numbered breadth first before the depth-first (right child first) rest */
    FlatScene flatten(int bfsLevels = 0) const;

private:
    int m_numLevels = 0;
    int m_numLeaves = 0;
    Scene* m_pScene;
    std::vector<Node> nodes;
    /* per triangle, in scene order (mesh after mesh): centroid and corner positions, computed once */
    std::vector<size_t> m_meshOffset;
    std::vector<vec3> m_centroid;
    std::vector<vec3> m_corners;
    [[nodiscard]] size_t globalIndex(const TriangleOrNode& t) const { return m_meshOffset[t.meshIndex] + t.triangleIndex; }
    void sortByCentroid(std::vector<TriangleOrNode>& triangles, size_t left, size_t right, int whichAxis) const;
    [[nodiscard]] AxisAlignedBox getBox(const std::vector<TriangleOrNode>& triangles, size_t begin, size_t end) const;
    size_t splice(std::vector<Node>& out, std::vector<Node>& left, std::vector<Node>& right, Node&& parent) const;
};

class BvhInterface {
public:
    BvhInterface(Scene* pScene, const Features& features);
    ~BvhInterface();
    BvhInterface(const BvhInterface&) = delete;
    BvhInterface& operator=(const BvhInterface&) = delete;
    [[nodiscard]] int numLevels() const;
    [[nodiscard]] int numLeaves() const;
    /* One query on the GPU; updates ray.t and fills hitInfo as the reference does. */
    bool intersect(Ray& ray, HitInfo& hitInfo, const Features& features) const;
    /* Many queries in one launch. rays7: origin, direction, t per ray. */
    void intersectBatch(float* rays7, int64_t n, const Features& features, srt_hit* hits, srt_counters* counters = nullptr,
                        bool referenceTraversal = false) const;

    /* The host-side node table: only present when the tree was built on the host (SAH binning, or SRT_HOST_BUILD=1 in
     * the environment); the median-split tree is built on the device. */
    const BoundingVolumeHierarchy* impl() const { return m_impl; }
    const Scene& scene() const { return *m_pScene; }
    srt_scene* deviceScene() const { return m_dev; }
    srt_device* device() const;

    /* Several GPUs of one node (no counterpart in the reference, which spreads a frame's rows over the OpenMP threads of
     * one process, src/render.cpp:184-187): addReplica builds one more copy of the scene, on GPU gpuId, behind a back-end
     * handle of its own.  renderRayTracing / renderFrames then cut every frame into tiles and one host thread per copy
     * draws tiles from a shared counter.  The same GPU may be named more than once (its copies render side by side). */
    void addReplica(int gpuId);
    [[nodiscard]] size_t numRenderDevices() const { return 1 + m_replicas.size(); }
    /* copy k (0: the one the constructor built) */
    srt_device* renderDevice(size_t k) const;
    srt_scene* renderScene(size_t k) const { return k == 0 ? m_dev : m_replicas[k - 1].scene; }

private:
    struct Replica {
        int deviceKey;
        srt_scene* scene;
    };
    srt_scene* buildOn(int deviceKey);
    BoundingVolumeHierarchy* m_impl = nullptr;
    Scene* m_pScene;
    Features m_features;
    srt_scene* m_dev = nullptr;
    int m_gpu = 0;
    std::vector<Replica> m_replicas;
};

/* ---- framebuffer */

class Screen {
public:
    Screen(const ivec2& resolution, bool presentable = false);
    void clear(const vec3& color);
    void applyBloomFilter(const float threshold, const int boxSize);
    void setPixel(int x, int y, const vec3& color);
    void writeBitmapToFile(const std::filesystem::path& filePath);
    [[nodiscard]] ivec2 resolution() const { return m_resolution; }
    [[nodiscard]] const std::vector<vec3>& pixels() const { return m_textureData; }
    [[nodiscard]] std::vector<vec3>& pixels() { return m_textureData; }
    [[nodiscard]] int indexAt(int x, int y) const;

private:
    ivec2 m_resolution;
    std::vector<vec3> m_textureData;
};

/* ---- camera: the reference's Trackball without its Window; the aspect ratio is given explicitly */

class Trackball {
public:
    Trackball(float aspectRatio, float fovy, float distFromLookAt = 4.0f, float rotationX = 0.0f, float rotationY = 0.0f);
    void setCamera(const vec3 lookAt, const vec3 rotations, const float dist);
    [[nodiscard]] vec3 position() const;
    [[nodiscard]] vec3 lookAt() const { return m_lookAt; }
    [[nodiscard]] vec3 rotationEulerAngles() const { return m_rotationEulerAngles; }
    [[nodiscard]] float distanceFromLookAt() const { return m_distanceFromLookAt; }
    /* pixel in normalized coordinates (-1..1); evaluated on the GPU like every other ray */
    [[nodiscard]] Ray generateRay(const vec2& pixel) const;
    /* constants the kernels take */
    srt_camera deviceCamera(int width, int height) const;

private:
    float m_fovy, m_halfScreenSpaceHeight, m_halfScreenSpaceWidth;
    vec3 m_lookAt { 0.0f };
    float m_distanceFromLookAt;
    vec3 m_rotationEulerAngles { 0.0f };
};

float radians(float degrees);
vec3 radians(const vec3& degrees);

/* ---- frame */

uint32_t featureMask(const Features& features, bool referenceTraversal, bool exactShadowBoxes = false);

void renderRayTracing(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features,
                      const float& threshold = 0, const int& boxSize = 0, const int& numRays = 5);
void renderRayTracing(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features,
                      const int& numRays, const RenderOptions& options);
vec3 getFinalColor(const Scene& scene, const BvhInterface& bvh, Ray ray, const Features& features, int rayDepth = 0);

/* Several frames of one scene in one call: what the reference's command-line mode does with one host thread per camera
 * (src/main.cpp:468-487).  The tiles of all frames form one work list, frame after frame, each frame's tiles from its
 * centre outwards; one host thread per copy of the scene (BvhInterface::addReplica) draws from it until it is empty, so
 * the GPUs stay busy whatever the number of cameras.  With a single copy every frame is one rectangle. */
struct FrameJob {
    const Trackball* camera;
    Screen* screen;
};
void renderFrames(const Scene& scene, const BvhInterface& bvh, const std::vector<FrameJob>& frames, const Features& features, const int& numRays,
                  const RenderOptions& options);

/* The work list of renderFrames and its hand-out, exposed for tests: tilesOf lists a frame's tiles (x0, y0, x1, y1; y up)
 * centre first; runWorkers calls work(worker, item) for every item in [0, items) exactly once, items handed out in order
 * from one atomic counter to `workers` host threads, and rethrows the first exception a worker threw. */
struct TileRect {
    int x0, y0, x1, y1;
};
std::vector<TileRect> tilesOf(int width, int height, int tileWidth, int tileHeight);
void runWorkers(size_t workers, size_t items, const std::function<void(size_t, size_t)>& work);

} // namespace srt
