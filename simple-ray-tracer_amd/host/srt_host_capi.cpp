/*
 * srt_host_capi.cpp -- flat C entry points over the C++ host mirror, for ctypes / cgo-style callers
 * (tests, bench.py).  Every function returns 0 or a negative status; srth_last_error() has the message.
 */
#include <atomic>
#include <chrono>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <sstream>

#include "srt_config.h"
#include "srt_host.h"

using namespace srt;

static thread_local std::string g_err;

#define SRTH_TRY(...)                  \
    try {                              \
        __VA_ARGS__;                   \
        return 0;                      \
    } catch (const std::exception& e) { \
        g_err = e.what();              \
        return -1;                     \
    } catch (...) {                    \
        g_err = "unknown error";       \
        return -1;                     \
    }

#define SRTH_F_SAH (1u << 8) /* extra.enableBvhSahBinning; a host-side build option, not a kernel flag */
#define SRTH_F_BLOOM (1u << 9) /* extra.enableBloomEffect; a host-side pass over the finished frame */

static Features featuresFromMask(uint32_t m)
{
    Features f;
    f.enableShading = m & SRT_F_SHADING;
    f.enableRecursive = m & SRT_F_RECURSIVE;
    f.enableHardShadow = m & SRT_F_HARD_SHADOW;
    f.enableSoftShadow = m & SRT_F_SOFT_SHADOW;
    f.enableNormalInterp = m & SRT_F_NORMAL_INTERP;
    f.enableTextureMapping = m & SRT_F_TEXTURE;
    f.enableAccelStructure = m & SRT_F_ACCEL;
    f.extra.enableMultipleRaysPerPixel = m & SRT_F_MULTISAMPLE;
    f.extra.enableBvhSahBinning = m & SRTH_F_SAH;
    f.extra.enableBloomEffect = m & SRTH_F_BLOOM;
    f.extra.enableTransparency = m & SRT_F_TRANSPARENCY;
    f.extra.enableMotionBlur = m & SRT_F_MOTION_BLUR;
    f.extra.enableDepthOfField = m & SRT_F_DEPTH_OF_FIELD;
    f.extra.enableEnvironmentMapping = m & SRT_F_ENVIRONMENT;
    return f;
}

struct srth_scene {
    Scene scene;
    std::vector<std::shared_ptr<Image>> textures;
};

struct srth_bvh {
    srth_scene* owner;
    BvhInterface* bvh;
};

extern "C" {

const char* srth_last_error(void) { return g_err.c_str(); }

int srth_set_device(int gpu) { SRTH_TRY(setDevice(gpu); (void)deviceFor(currentDevice())) }
int srth_set_device_context(int gpu, int context) { SRTH_TRY(setDevice(gpu, context); (void)deviceFor(currentDevice())) }

srth_scene* srth_scene_create(void) { return new srth_scene(); }
void srth_scene_destroy(srth_scene* s) { delete s; }

int srth_scene_add_texture(srth_scene* s, int w, int h, const float* rgb)
{
    auto im = std::make_shared<Image>();
    im->width = w, im->height = h;
    im->pixels.resize((size_t)w * h);
    std::memcpy(im->pixels.data(), rgb, (size_t)w * h * 12);
    s->textures.push_back(im);
    return (int)s->textures.size() - 1;
}

/* Scene::environment from float RGB rows (top row first). */
int srth_scene_set_environment(srth_scene* s, int w, int h, const float* rgb)
{
    SRTH_TRY(if (!s || !rgb || w <= 0 || h <= 0) throw std::invalid_argument("srth_scene_set_environment: null argument or non-positive size");
             auto im = std::make_shared<Image>(); im->width = w; im->height = h; im->pixels.resize((size_t)w * (size_t)h);
             std::memcpy(im->pixels.data(), rgb, (size_t)w * (size_t)h * 12); s->scene.environment = im)
}
/* Scene::environment = loadEnvironment(dataDir), i.e. <dataDir>/cube_map.jpg */
int srth_scene_load_environment(srth_scene* s, const char* dataDir) { SRTH_TRY(s->scene.environment = loadEnvironment(dataDir)) }
/* The Scene's motion-blur and depth-of-field variables (reference src/scene.h:32-40). */
int srth_scene_set_lens(srth_scene* s, int mbSamples, float time0, float time1, const float* direction3, int focalLength, float aperture, int dofSamples)
{
    if (!s || !direction3) {
        g_err = "srth_scene_set_lens: null argument";
        return -1;
    }
    s->scene.MB_samples = mbSamples, s->scene.time0 = time0, s->scene.time1 = time1;
    s->scene.directionVector = { direction3[0], direction3[1], direction3[2] };
    s->scene.focalLength = focalLength, s->scene.aperture = aperture, s->scene.DOF_samples = dofSamples;
    return 0;
}
/* Image(path) as the loader decodes it: size query with rgb == NULL, then the float RGB rows. */
int srth_image_load(const char* path, int* w, int* h, float* rgb)
{
    SRTH_TRY(Image im { std::filesystem::path(path) }; *w = im.width; *h = im.height;
             if (rgb) std::memcpy(rgb, im.pixels.data(), im.pixels.size() * 12))
}

int srth_scene_add_mesh(srth_scene* s, const float* verts8, int nv, const uint32_t* tris3, int nt, const float* material8, int tex)
{
    Mesh m;
    m.vertices.resize((size_t)nv);
    static_assert(sizeof(Vertex) == 32, "Vertex is 8 floats");
    std::memcpy(m.vertices.data(), verts8, (size_t)nv * 32);
    m.triangles.resize((size_t)nt);
    std::memcpy(m.triangles.data(), tris3, (size_t)nt * 12);
    m.material.kd = { material8[0], material8[1], material8[2] };
    m.material.ks = { material8[3], material8[4], material8[5] };
    m.material.shininess = material8[6];
    m.material.transparency = material8[7];
    if (tex >= 0 && tex < (int)s->textures.size())
        m.material.kdTexture = s->textures[(size_t)tex];
    s->scene.meshes.push_back(std::move(m));
    return (int)s->scene.meshes.size() - 1;
}

/* light22: type, then the struct's floats in declaration order */
int srth_scene_add_light(srth_scene* s, const float* l)
{
    auto v = [&](int i) { return vec3(l[1 + i], l[2 + i], l[3 + i]); };
    const int type = (int)l[0];
    if (type == 0)
        s->scene.lights.emplace_back(PointLight { v(0), v(3) });
    else if (type == 1)
        s->scene.lights.emplace_back(SegmentLight { v(0), v(3), v(6), v(9) });
    else if (type == 2)
        s->scene.lights.emplace_back(ParallelogramLight { v(0), v(3), v(6), v(9), v(12), v(15), v(18) });
    else {
        g_err = "unknown light type";
        return -1;
    }
    return 0;
}

int srth_scene_load_obj(srth_scene* s, const char* path, int normalize)
{
    SRTH_TRY(auto sub = loadMesh(path, normalize != 0); for (auto& m : sub) s->scene.meshes.push_back(std::move(m)))
}

int srth_scene_load_prebuilt(srth_scene* s, int type, const char* dataDir) { SRTH_TRY(s->scene = loadScenePrebuilt((SceneType)type, dataDir)) }

int srth_scene_num_meshes(const srth_scene* s) { return (int)s->scene.meshes.size(); }
int srth_scene_num_lights(const srth_scene* s) { return (int)s->scene.lights.size(); }
void srth_scene_mesh_sizes(const srth_scene* s, int m, int* nv, int* nt)
{
    *nv = (int)s->scene.meshes[(size_t)m].vertices.size();
    *nt = (int)s->scene.meshes[(size_t)m].triangles.size();
}
/* returns 1 when the mesh has a texture; its size goes to tex_wh */
int srth_scene_mesh_get(const srth_scene* s, int m, float* verts8, uint32_t* tris3, float* material8, int* tex_wh)
{
    const Mesh& mesh = s->scene.meshes[(size_t)m];
    std::memcpy(verts8, mesh.vertices.data(), mesh.vertices.size() * 32);
    std::memcpy(tris3, mesh.triangles.data(), mesh.triangles.size() * 12);
    const Material& k = mesh.material;
    const float mat[8] = { k.kd.x, k.kd.y, k.kd.z, k.ks.x, k.ks.y, k.ks.z, k.shininess, k.transparency };
    std::memcpy(material8, mat, sizeof mat);
    if (k.kdTexture) {
        tex_wh[0] = k.kdTexture->width, tex_wh[1] = k.kdTexture->height;
        return 1;
    }
    return 0;
}
void srth_scene_mesh_texture(const srth_scene* s, int m, float* rgb)
{
    const Image& im = *s->scene.meshes[(size_t)m].material.kdTexture;
    std::memcpy(rgb, im.pixels.data(), im.pixels.size() * 12);
}
void srth_scene_light_get(const srth_scene* s, int i, float* out22)
{
    std::memset(out22, 0, 22 * sizeof(float));
    const Light& l = s->scene.lights[(size_t)i];
    if (std::holds_alternative<PointLight>(l)) {
        out22[0] = 0;
        std::memcpy(out22 + 1, &std::get<PointLight>(l), sizeof(PointLight));
    } else if (std::holds_alternative<SegmentLight>(l)) {
        out22[0] = 1;
        std::memcpy(out22 + 1, &std::get<SegmentLight>(l), sizeof(SegmentLight));
    } else {
        out22[0] = 2;
        std::memcpy(out22 + 1, &std::get<ParallelogramLight>(l), sizeof(ParallelogramLight));
    }
}

/* ---- BVH build / flatten without touching the GPU (host logic is testable on a CPU-only machine) */

struct srth_hostbvh {
    BoundingVolumeHierarchy* bvh;
};

srth_hostbvh* srth_hostbvh_create(srth_scene* s, uint32_t features)
{
    try {
        return new srth_hostbvh { new BoundingVolumeHierarchy(&s->scene, featuresFromMask(features)) };
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void srth_hostbvh_destroy(srth_hostbvh* b)
{
    if (b)
        delete b->bvh;
    delete b;
}
int srth_hostbvh_num_levels(const srth_hostbvh* b) { return b->bvh->numLevels(); }
int srth_hostbvh_num_leaves(const srth_hostbvh* b) { return b->bvh->numLeaves(); }
int srth_hostbvh_num_nodes(const srth_hostbvh* b) { return (int)b->bvh->nodeTable().size(); }
/* Node table in the reference's post-order: per node is_leaf, level, a, b (leaf: first entry of the ordered
 * triangle list and count; inner: left and right child) and the box as 6 floats. */
void srth_hostbvh_dump(const srth_hostbvh* b, int32_t* meta4, float* box6)
{
    int first = 0;
    const auto& nodes = b->bvh->nodeTable();
    for (size_t i = 0; i < nodes.size(); ++i) {
        const Node& n = nodes[i];
        meta4[4 * i] = n.isLeaf, meta4[4 * i + 1] = n.level;
        if (n.isLeaf) {
            meta4[4 * i + 2] = first, meta4[4 * i + 3] = (int)n.triangles.size();
            first += (int)n.triangles.size();
        } else {
            meta4[4 * i + 2] = (int)n.triangles[0].nodeIndex, meta4[4 * i + 3] = (int)n.triangles[1].nodeIndex;
        }
        const float bx[6] = { n.box.lower.x, n.box.lower.y, n.box.lower.z, n.box.upper.x, n.box.upper.y, n.box.upper.z };
        std::memcpy(box6 + 6 * i, bx, sizeof bx);
    }
}
int srth_hostbvh_tri_order(const srth_hostbvh* b, int32_t* mesh_tri2)
{
    int k = 0;
    for (const Node& n : b->bvh->nodeTable())
        if (n.isLeaf)
            for (const TriangleOrNode& t : n.triangles) {
                if (mesh_tri2)
                    mesh_tri2[2 * k] = (int)t.meshIndex, mesh_tri2[2 * k + 1] = (int)t.triangleIndex;
                ++k;
            }
    return k;
}
/* Flattened arrays: call with null pointers to get the sizes (n_nodes, n_tris), then with buffers. */
int srth_hostbvh_flatten(const srth_hostbvh* b, int bfs_levels, uint32_t* sizes2, srt_node* nodes, srt_tri* tris, srt_tri_attr* attrs,
                         uint32_t* scene_order, float* root_box6, uint32_t* root_ref)
{
    SRTH_TRY(const FlatScene f = b->bvh->flatten(bfs_levels); sizes2[0] = (uint32_t)f.nodes.size(); sizes2[1] = (uint32_t)f.tris.size();
             if (nodes) std::memcpy(nodes, f.nodes.data(), f.nodes.size() * sizeof(srt_node));
             if (tris) std::memcpy(tris, f.tris.data(), f.tris.size() * sizeof(srt_tri));
             if (attrs) std::memcpy(attrs, f.attrs.data(), f.attrs.size() * sizeof(srt_tri_attr));
             if (scene_order) std::memcpy(scene_order, f.sceneOrder.data(), f.sceneOrder.size() * 4);
             if (root_box6) std::memcpy(root_box6, f.rootBox, 24); if (root_ref) * root_ref = f.rootRef)
}

/* ---- BvhInterface on the GPU */

srth_bvh* srth_bvh_create(srth_scene* s, uint32_t features)
{
    try {
        return new srth_bvh { s, new BvhInterface(&s->scene, featuresFromMask(features)) };
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void srth_bvh_destroy(srth_bvh* b)
{
    if (b)
        delete b->bvh;
    delete b;
}
int srth_bvh_num_levels(const srth_bvh* b) { return b->bvh->numLevels(); }
int srth_bvh_num_leaves(const srth_bvh* b) { return b->bvh->numLeaves(); }

int srth_bvh_intersect(const srth_bvh* b, float* rays7, int64_t n, uint32_t features, srt_hit* hits, srt_counters* counters)
{
    SRTH_TRY(b->bvh->intersectBatch(rays7, n, featuresFromMask(features), hits, counters, !(features & SRT_F_MISS_CULL)))
}

int srth_final_color(const srth_bvh* b, const float* rays7, int64_t n, uint32_t features, int max_depth, uint32_t seed, float* rgb)
{
    if (srt_final_color_batch(b->bvh->device(), b->bvh->deviceScene(), rays7, n, features, max_depth, seed, rgb) != SRT_OK) {
        g_err = srt_last_error();
        return -1;
    }
    return 0;
}

/* Trackball set-up as the reference's CLI does it (src/main.cpp:474-475): fov and rotation in degrees. */
int srth_camera_make(float fovy_deg, float dist, const float* look_at3, const float* rot_deg3, int width, int height, srt_camera* out)
{
    const float aspect = (width == 0 || height == 0) ? 1.0f : float(width) / float(height);
    Trackball cam(aspect, radians(fovy_deg), dist);
    cam.setCamera({ look_at3[0], look_at3[1], look_at3[2] }, radians(vec3(rot_deg3[0], rot_deg3[1], rot_deg3[2])), dist);
    *out = cam.deviceCamera(width, height);
    return 0;
}

int srth_generate_rays(const srt_camera* cam, const float* ndc2, int64_t n, float* rays7)
{
    try {
        if (srt_generate_rays(deviceFor(currentDevice()), cam, ndc2, n, rays7) != SRT_OK) {
            g_err = srt_last_error();
            return -1;
        }
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

/* renderRayTracing for a rectangle.  Output arrays follow srt_output; `stream` is a hipStream_t. */
int srth_render(const srth_bvh* b, const srt_render_params* params, const srt_output* out, void* stream)
{
    if (srt_render(b->bvh->device(), b->bvh->deviceScene(), params, out, stream) != SRT_OK) {
        g_err = srt_last_error();
        return -1;
    }
    return 0;
}

/* The same through srt_render_noalloc (device output only, bounded rectangle, no allocation). */
int srth_render_noalloc(const srth_bvh* b, const srt_render_params* params, const srt_output* out, void* stream)
{
    if (srt_render_noalloc(b->bvh->device(), b->bvh->deviceScene(), params, out, stream) != SRT_OK) {
        g_err = srt_last_error();
        return -1;
    }
    return 0;
}

int srth_probe_gather(const srth_bvh* b, uint64_t window_bytes, uint32_t loads_per_lane, uint32_t quad, double* gbytes_per_s, uint64_t* window_used)
{
    if (srt_probe_gather(b->bvh->device(), b->bvh->deviceScene(), window_bytes, loads_per_lane, quad, gbytes_per_s, window_used) != SRT_OK) {
        g_err = srt_last_error();
        return -1;
    }
    return 0;
}

/* Sizes and arrays of the scene as it sits on the device (built there, or uploaded): sizes4 = nodes, triangles,
 * levels, leaves; arrays may be NULL. */
int srth_bvh_download(const srth_bvh* b, uint32_t* sizes4, void* nodes, void* tris, void* attrs, uint32_t* sceneOrder, float* rootBox6, uint32_t* rootRef)
{
    if (srt_scene_info(b->bvh->deviceScene(), sizes4, sizes4 + 1, sizes4 + 2, sizes4 + 3, rootRef, rootBox6) != SRT_OK
        || srt_scene_download(b->bvh->deviceScene(), (srt_node*)nodes, (srt_tri*)tris, (srt_tri_attr*)attrs, sceneOrder) != SRT_OK) {
        g_err = srt_last_error();
        return -1;
    }
    return 0;
}

int srth_read_counters(const srth_bvh* b, void* stream, srt_counters* out, int reset)
{
    if (srt_device_read_counters(b->bvh->device(), stream, out, reset) != SRT_OK) {
        g_err = srt_last_error();
        return -1;
    }
    return 0;
}

int srth_read_kernel_counters(const srth_bvh* b, void* stream, srt_counters* out4, int reset)
{
    if (srt_device_read_kernel_counters(b->bvh->device(), stream, out4, reset) != SRT_OK) {
        g_err = srt_last_error();
        return -1;
    }
    return 0;
}

int srth_kernel_times(const srth_bvh* b, float* ms4, int32_t* launches4, int reset)
{
    if (srt_device_kernel_times(b->bvh->device(), ms4, launches4, reset) != SRT_OK) {
        g_err = srt_last_error();
        return -1;
    }
    return 0;
}

int srth_synchronize(const srth_bvh* b, void* stream)
{
    if (srt_device_synchronize(b->bvh->device(), stream) != SRT_OK) {
        g_err = srt_last_error();
        return -1;
    }
    return 0;
}

int srth_selftest_division(uint32_t seed, uint64_t n, uint64_t* mismatches)
{
    try {
        if (srt_selftest_division(deviceFor(currentDevice()), seed, n, mismatches) != SRT_OK) {
            g_err = srt_last_error();
            return -1;
        }
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

/* ---- the C++ mirror exactly as a reference call site would use it (src/main.cpp:470-480): Screen + Trackball +
 * renderRayTracing, BvhInterface::intersect on one ray, getFinalColor, Trackball::generateRay */
int srth_mirror_render(const srth_bvh* b, uint32_t features, int width, int height, float fovy_deg, float dist, const float* look_at3,
                       const float* rot_deg3, int num_rays, float* rgb)
{
    SRTH_TRY(Screen screen({ width, height }); screen.clear(vec3(0.0f));
             Trackball camera(float(width) / float(height), radians(fovy_deg), dist);
             camera.setCamera({ look_at3[0], look_at3[1], look_at3[2] }, radians(vec3(rot_deg3[0], rot_deg3[1], rot_deg3[2])), dist);
             renderRayTracing(b->bvh->scene(), camera, *b->bvh, screen, featuresFromMask(features), 0.5f, 1, num_rays);
             std::memcpy(rgb, &screen.pixels()[0].x, (size_t)width * height * 12))
}

/* One more copy of the scene on GPU gpu (BvhInterface::addReplica); frames rendered through the mirror are then cut
 * into tiles that one host thread per copy draws from a shared counter. */
int srth_bvh_add_replica(srth_bvh* b, int gpu) { SRTH_TRY(b->bvh->addReplica(gpu)) }
int srth_bvh_num_render_devices(const srth_bvh* b) { return (int)b->bvh->numRenderDevices(); }

/* renderFrames: n_frames cameras (fovy, distance, look_at xyz, rotation xyz in degrees: 8 floats each) of one scene in one
 * call, as the command-line front end issues them; rgb receives the frames one after the other. */
int srth_mirror_render_frames(const srth_bvh* b, uint32_t features, int width, int height, int n_frames, const float* cameras8, int num_rays, int tile_w,
                              int tile_h, float* rgb)
{
    SRTH_TRY(std::vector<Screen> screens; std::vector<Trackball> cameras; screens.reserve((size_t)n_frames); cameras.reserve((size_t)n_frames);
             for (int k = 0; k < n_frames; ++k) {
                 const float* c = cameras8 + 8 * k;
                 screens.emplace_back(ivec2 { width, height });
                 screens.back().clear(vec3(0.0f));
                 cameras.emplace_back(float(width) / float(height), radians(c[0]), c[1]);
                 cameras.back().setCamera({ c[2], c[3], c[4] }, radians(vec3(c[5], c[6], c[7])), c[1]);
             }
             std::vector<FrameJob> frames;
             for (int k = 0; k < n_frames; ++k) frames.push_back({ &cameras[(size_t)k], &screens[(size_t)k] });
             RenderOptions options; options.tileWidth = tile_w; options.tileHeight = tile_h;
             renderFrames(b->bvh->scene(), *b->bvh, frames, featuresFromMask(features), num_rays, options);
             for (int k = 0; k < n_frames; ++k)
                 std::memcpy(rgb + (size_t)k * width * height * 3, &screens[(size_t)k].pixels()[0].x, (size_t)width * height * 12))
}

/* tilesOf: the tiles of a frame in hand-out order (4 ints each); returns their number (rects may be NULL). */
int srth_tiles_of(int width, int height, int tile_w, int tile_h, int* rects4, int cap)
{
    const std::vector<TileRect> t = tilesOf(width, height, tile_w, tile_h);
    for (size_t k = 0; rects4 && k < t.size() && (int)k < cap; ++k)
        rects4[4 * k] = t[k].x0, rects4[4 * k + 1] = t[k].y0, rects4[4 * k + 2] = t[k].x1, rects4[4 * k + 3] = t[k].y1;
    return (int)t.size();
}

/* runWorkers with a work function that only records who ran what (no GPU involved): done[i] counts how often item i ran,
 * by_worker[w] how many items worker w took; fail_at >= 0 makes that item throw, and the call then returns -1 after all
 * workers have stopped. */
int srth_run_workers_selftest(int workers, int items, int fail_at, int* done, int* by_worker)
{
    SRTH_TRY(std::vector<std::atomic<int>> ran((size_t)items); std::vector<std::atomic<int>> took((size_t)workers);
             for (auto& a : ran) a.store(0);
             for (auto& a : took) a.store(0);
             std::exception_ptr err;
             try {
                 runWorkers((size_t)workers, (size_t)items, [&](size_t w, size_t i) {
                     if ((int)i == fail_at) throw std::runtime_error("item " + std::to_string(i) + " failed (as asked)");
                     std::this_thread::sleep_for(std::chrono::microseconds(100 + 50 * (i % 7))); /* items of unequal length */
                     ran[i].fetch_add(1); took[w].fetch_add(1); });
             } catch (...) { err = std::current_exception(); }
             for (int i = 0; i < items; ++i) done[i] = ran[(size_t)i].load();
             for (int w = 0; w < workers; ++w) by_worker[w] = took[(size_t)w].load();
             if (err) std::rethrow_exception(err))
}

int srth_mirror_intersect_one(const srth_bvh* b, float* ray7, uint32_t features, float* normal3, float* bary3, float* uv2, float* kd3, float* ks3,
                              float* shininess, int* has_texture)
{
    try {
        Ray ray { { ray7[0], ray7[1], ray7[2] }, { ray7[3], ray7[4], ray7[5] }, ray7[6] };
        HitInfo hi;
        const bool hit = b->bvh->intersect(ray, hi, featuresFromMask(features));
        ray7[6] = ray.t;
        if (hit) {
            normal3[0] = hi.normal.x, normal3[1] = hi.normal.y, normal3[2] = hi.normal.z;
            bary3[0] = hi.barycentricCoord.x, bary3[1] = hi.barycentricCoord.y, bary3[2] = hi.barycentricCoord.z;
            uv2[0] = hi.texCoord.x, uv2[1] = hi.texCoord.y;
            kd3[0] = hi.material.kd.x, kd3[1] = hi.material.kd.y, kd3[2] = hi.material.kd.z;
            ks3[0] = hi.material.ks.x, ks3[1] = hi.material.ks.y, ks3[2] = hi.material.ks.z;
            *shininess = hi.material.shininess;
            *has_texture = hi.material.kdTexture ? 1 : 0;
        }
        return hit ? 1 : 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

int srth_mirror_final_color(const srth_bvh* b, const float* ray7, uint32_t features, float* rgb)
{
    SRTH_TRY(Ray ray { { ray7[0], ray7[1], ray7[2] }, { ray7[3], ray7[4], ray7[5] }, ray7[6] };
             const vec3 c = getFinalColor(b->bvh->scene(), *b->bvh, ray, featuresFromMask(features)); rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z)
}

int srth_mirror_generate_ray(float aspect, float fovy_deg, float dist, const float* look_at3, const float* rot_deg3, const float* ndc2, float* ray7)
{
    SRTH_TRY(Trackball camera(aspect, radians(fovy_deg), dist);
             camera.setCamera({ look_at3[0], look_at3[1], look_at3[2] }, radians(vec3(rot_deg3[0], rot_deg3[1], rot_deg3[2])), dist);
             const Ray r = camera.generateRay({ ndc2[0], ndc2[1] }); ray7[0] = r.origin.x; ray7[1] = r.origin.y; ray7[2] = r.origin.z;
             ray7[3] = r.direction.x; ray7[4] = r.direction.y; ray7[5] = r.direction.z; ray7[6] = r.t)
}

/* Screen::writeBitmapToFile for a W*H*3 float framebuffer in Screen layout */
int srth_write_bmp(const char* path, const float* rgb, int w, int h)
{
    SRTH_TRY(Screen sc({ w, h }); std::memcpy(&sc.pixels()[0].x, rgb, (size_t)w * h * 12); sc.writeBitmapToFile(path))
}

} // extern "C"

/* ---- tile stealing between the per-GPU processes of one node: an array of 64-bit counters in POSIX shared
 * memory.  Every process draws the next tile index of the current frame with one fetch-add; there is no other
 * coordination on the data path.  (The reference distributes rows over OpenMP threads with schedule(guided),
 * src/render.cpp:184-187; this is the same idea across processes.) */
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>

#define SRTH_TILE_SLOTS 64

struct srth_tilecounter {
    std::atomic<int64_t>* slots;
    std::string name;
    bool owner;
};

extern "C" {

srth_tilecounter* srth_tilecounter_open(const char* name, int create)
{
    const size_t bytes = SRTH_TILE_SLOTS * sizeof(std::atomic<int64_t>);
    static_assert(sizeof(std::atomic<int64_t>) == 8 && std::atomic<int64_t>::is_always_lock_free, "plain 8-byte atomics");
    int fd = shm_open(name, create ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) {
        g_err = std::string("shm_open failed for ") + name;
        return nullptr;
    }
    if (create && ftruncate(fd, (off_t)bytes) != 0) {
        close(fd);
        g_err = "ftruncate failed on the tile counter";
        return nullptr;
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        g_err = "mmap failed on the tile counter";
        return nullptr;
    }
    auto* t = new srth_tilecounter { (std::atomic<int64_t>*)p, name, create != 0 };
    if (create)
        for (int i = 0; i < SRTH_TILE_SLOTS; ++i)
            t->slots[i].store(0);
    return t;
}
int64_t srth_tilecounter_next(srth_tilecounter* t, int slot) { return t->slots[slot % SRTH_TILE_SLOTS].fetch_add(1); }
void srth_tilecounter_reset(srth_tilecounter* t, int slot) { t->slots[slot % SRTH_TILE_SLOTS].store(0); }
void srth_tilecounter_close(srth_tilecounter* t)
{
    if (!t)
        return;
    munmap((void*)t->slots, SRTH_TILE_SLOTS * sizeof(std::atomic<int64_t>));
    if (t->owner)
        shm_unlink(t->name.c_str());
    delete t;
}

/* ---- the frame board: who rendered which tile, published through shared memory.
 * At the end of a frame every process posts the tiles it rendered and what each cost; all of them then read all records
 * (rank 0 to know how many tiles to expect from whom, everybody to derive the next frame's hand-out order).  That used to
 * be a collective on device tensors followed by a copy to the host; the processes of one node share memory, so it is a
 * few cache lines and one counter here, and the GPU is not involved.  Slot = frame % SRTH_BOARD_SLOTS; a slot's `posted`
 * counter only grows (world posts per use), so nothing is ever reset, and a process can be at most one frame ahead of
 * the slowest one because everybody waits for everybody's post. */
#define SRTH_BOARD_SLOTS 4
#define SRTH_BOARD_MAX_RANKS 64

struct srth_board_header {
    std::atomic<int64_t> posted[SRTH_BOARD_SLOTS];
    int64_t world, n_tiles;
};

struct srth_frameboard {
    srth_board_header* head;
    int32_t* records; /* [slot][rank][1 + 2 * n_tiles]: count, tile ids, tile costs */
    size_t bytes;
    std::string name;
    bool owner;
    size_t stride() const { return 1 + 2 * (size_t)head->n_tiles; }
    int32_t* record(int slot, int rank) const { return records + ((size_t)slot * SRTH_BOARD_MAX_RANKS + (size_t)rank) * stride(); }
};

srth_frameboard* srth_frameboard_open(const char* name, int create, int world, int n_tiles)
{
    if (world < 1 || world > SRTH_BOARD_MAX_RANKS || n_tiles < 1 || n_tiles > (1 << 20)) {
        g_err = "frame board: world must be 1..64 and n_tiles 1..2^20";
        return nullptr;
    }
    const size_t head = (sizeof(srth_board_header) + 63) & ~(size_t)63;
    const size_t bytes = head + (size_t)SRTH_BOARD_SLOTS * SRTH_BOARD_MAX_RANKS * (1 + 2 * (size_t)n_tiles) * sizeof(int32_t);
    int fd = shm_open(name, create ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) {
        g_err = std::string("shm_open failed for ") + name;
        return nullptr;
    }
    if (create && ftruncate(fd, (off_t)bytes) != 0) {
        close(fd);
        g_err = "ftruncate failed on the frame board";
        return nullptr;
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        g_err = "mmap failed on the frame board";
        return nullptr;
    }
    auto* b = new srth_frameboard { (srth_board_header*)p, (int32_t*)((char*)p + head), bytes, name, create != 0 };
    if (create) {
        for (auto& c : b->head->posted)
            c.store(0);
        b->head->world = world, b->head->n_tiles = n_tiles;
    } else if (b->head->world != world || b->head->n_tiles != n_tiles) {
        g_err = "frame board: opened with another world size or tile count than it was created with";
        munmap(p, bytes);
        delete b;
        return nullptr;
    }
    return b;
}

/* This process's record of frame `frame`: count tiles with their ids and costs. */
int srth_frameboard_post(srth_frameboard* b, int64_t frame, int rank, int count, const int32_t* ids, const int32_t* costs)
{
    if (!b || rank < 0 || rank >= b->head->world || count < 0 || count > b->head->n_tiles || (count && (!ids || !costs))) {
        g_err = "frame board: bad post";
        return -1;
    }
    const int slot = (int)(frame % SRTH_BOARD_SLOTS);
    int32_t* r = b->record(slot, rank);
    r[0] = count;
    std::memcpy(r + 1, ids, (size_t)count * 4);
    std::memcpy(r + 1 + b->head->n_tiles, costs, (size_t)count * 4);
    b->head->posted[slot].fetch_add(1, std::memory_order_release);
    return 0;
}

/* Waits until every process has posted frame `frame` (0), or timeout_ms have passed (-1). */
int srth_frameboard_wait(srth_frameboard* b, int64_t frame, int timeout_ms)
{
    const int slot = (int)(frame % SRTH_BOARD_SLOTS);
    const int64_t want = b->head->world * (frame / SRTH_BOARD_SLOTS + 1);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        if (b->head->posted[slot].load(std::memory_order_acquire) >= want)
            return 0;
        if (spin > 2000) /* a peer is still rendering: stop burning the core it may need */
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms)) {
            g_err = "frame board: timed out waiting for the other processes' tile lists";
            return -1;
        }
    }
}

/* Record of `rank` for frame `frame` (after srth_frameboard_wait); returns its tile count. */
int srth_frameboard_read(const srth_frameboard* b, int64_t frame, int rank, int32_t* ids, int32_t* costs)
{
    const int32_t* r = b->record((int)(frame % SRTH_BOARD_SLOTS), rank);
    const int count = r[0];
    std::memcpy(ids, r + 1, (size_t)count * 4);
    std::memcpy(costs, r + 1 + b->head->n_tiles, (size_t)count * 4);
    return count;
}

void srth_frameboard_close(srth_frameboard* b)
{
    if (!b)
        return;
    munmap((void*)b->head, b->bytes);
    if (b->owner)
        shm_unlink(b->name.c_str());
    delete b;
}

/* Screen::applyBloomFilter on a frame given top row first (Screen's storage order), in place. */
int srth_bloom(float* rgb, int width, int height, float threshold, int box_size)
{
    SRTH_TRY(Screen screen({ width, height }); std::memcpy(&screen.pixels()[0].x, rgb, (size_t)width * height * 12);
             screen.applyBloomFilter(threshold, box_size); std::memcpy(rgb, &screen.pixels()[0].x, (size_t)width * height * 12))
}

/* Parses a configuration text as the command-line front end does and writes its printed form (the reference's
 * `std::cout << config`) into out; returns the length needed (terminator included) or -1. */
long srth_config_describe(const char* toml_text, char* out, long cap)
{
    try {
        std::ostringstream os;
        os << readConfigText(toml_text);
        const std::string s = os.str();
        if (out && cap > 0) {
            const size_t n = std::min<size_t>(s.size(), (size_t)cap - 1);
            std::memcpy(out, s.data(), n);
            out[n] = 0;
        }
        return (long)s.size() + 1;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

} // extern "C"
