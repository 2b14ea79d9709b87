/*
 * ref_harness.cpp -- C entry points over the reference's OWN object code, for pinning the oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is compiled together with unmodified reference translation units
 * taken from where they lie under /root/reference (see oracle/Makefile); the result goes to oracle/_ref/
 * and is never committed.  Only the reference files that compile and link from their own sources in this
 * image are used:
 *   src/intersect.cpp  src/interpolate.cpp  src/texture.cpp  src/shading.cpp  src/scene.cpp
 *   framework/src/mesh.cpp  framework/src/image.cpp  third_party/tinyobjloader/src/tiny_obj_loader.cc
 * The BVH, light, render, screen and trackball units need the reference's OpenGL drawing / window code
 * (GL/glu.h, GLU, GLFW are not in this image); no substitute for those is written, so they are not linked.
 * For the camera, the glm header-only arithmetic that trackball.cpp uses is exercised directly; for the bitmap
 * file, the vendored stb_image_write.h and the glm conversion expression of src/screen.cpp:96-103.
 */
#include "scene.h"
#include "intersect.h"
#include "interpolate.h"
#include "shading.h"
#include "texture.h"
#include <framework/disable_all_warnings.h>
DISABLE_WARNINGS_PUSH()
#include <glm/gtc/quaternion.hpp>
#include <glm/glm.hpp>
DISABLE_WARNINGS_POP()
#include <cstring>
#include <limits>
#include <vector>
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include <stb/stb_image_write.h>

static glm::vec3 g3(const float* p) { return glm::vec3(p[0], p[1], p[2]); }
static Ray mkray(const float* r) { return Ray { g3(r), g3(r + 3), r[6] }; }

extern "C" {

/* ---- scenes: loadScenePrebuilt (src/scene.cpp:5-94) -> loadMesh (framework/src/mesh.cpp:52-149) */
struct ref_scene {
    Scene scene;
    std::vector<std::shared_ptr<Image>> textures; /* distinct texture objects, first-seen order */
    std::vector<int> mesh_tex;
};

ref_scene* ref_scene_load_prebuilt(int type, const char* data_dir)
{
    ref_scene* s = new ref_scene();
    try {
        s->scene = loadScenePrebuilt((SceneType)type, data_dir);
    } catch (...) {
        delete s;
        return nullptr;
    }
    for (const Mesh& m : s->scene.meshes) {
        int id = -1;
        if (m.material.kdTexture) {
            for (size_t i = 0; i < s->textures.size(); ++i)
                if (s->textures[i] == m.material.kdTexture)
                    id = (int)i;
            if (id < 0) {
                id = (int)s->textures.size();
                s->textures.push_back(m.material.kdTexture);
            }
        }
        s->mesh_tex.push_back(id);
    }
    return s;
}
ref_scene* ref_scene_load_file(const char* path, int normalize)
{
    ref_scene* s = new ref_scene();
    try {
        auto sub = loadMesh(path, normalize != 0);
        for (auto& m : sub)
            s->scene.meshes.push_back(std::move(m));
    } catch (...) {
        delete s;
        return nullptr;
    }
    for (const Mesh& m : s->scene.meshes) {
        int id = -1;
        if (m.material.kdTexture) {
            id = (int)s->textures.size();
            s->textures.push_back(m.material.kdTexture);
        }
        s->mesh_tex.push_back(id);
    }
    return s;
}
void ref_scene_free(ref_scene* s) { delete s; }
int ref_scene_num_meshes(const ref_scene* s) { return (int)s->scene.meshes.size(); }
int ref_scene_num_textures(const ref_scene* s) { return (int)s->textures.size(); }
int ref_scene_num_lights(const ref_scene* s) { return (int)s->scene.lights.size(); }
void ref_scene_mesh_sizes(const ref_scene* s, int m, int* nv, int* nt)
{
    *nv = (int)s->scene.meshes[m].vertices.size();
    *nt = (int)s->scene.meshes[m].triangles.size();
}
/* verts: 8 floats each; tris: 3 u32 each; material: kd, ks, shininess, transparency; returns texture id */
int ref_scene_mesh_get(const ref_scene* s, int m, float* verts, unsigned* tris, float* material8)
{
    const Mesh& mesh = s->scene.meshes[m];
    static_assert(sizeof(Vertex) == 32, "Vertex is 8 floats");
    std::memcpy(verts, mesh.vertices.data(), mesh.vertices.size() * sizeof(Vertex));
    std::memcpy(tris, mesh.triangles.data(), mesh.triangles.size() * sizeof(glm::uvec3));
    const Material& k = mesh.material;
    const float mat[8] = { k.kd.x, k.kd.y, k.kd.z, k.ks.x, k.ks.y, k.ks.z, k.shininess, k.transparency };
    std::memcpy(material8, mat, sizeof mat);
    return s->mesh_tex[m];
}
void ref_scene_texture_size(const ref_scene* s, int t, int* w, int* h)
{
    *w = s->textures[t]->width;
    *h = s->textures[t]->height;
}
void ref_scene_texture_get(const ref_scene* s, int t, float* rgb)
{
    const Image& im = *s->textures[t];
    std::memcpy(rgb, im.pixels.data(), im.pixels.size() * sizeof(glm::vec3));
}
/* 22 floats: type then the light struct's floats in declaration order (src/common.h:38-53) */
void ref_scene_light_get(const ref_scene* s, int i, float* out22)
{
    std::memset(out22, 0, 22 * sizeof(float));
    const auto& l = s->scene.lights[i];
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

/* ---- leaf math, batched: each call runs the reference function n times */

/* intersectRayWithTriangle (src/intersect.cpp:63-77). tri9 = v0,v1,v2; ray7 updated in place */
void ref_ray_triangle(const float* tri9, float* rays7, int* hit, long n)
{
    HitInfo hi;
    for (long i = 0; i < n; ++i) {
        Ray r = mkray(rays7 + 7 * i);
        hit[i] = intersectRayWithTriangle(g3(tri9 + 9 * i), g3(tri9 + 9 * i + 3), g3(tri9 + 9 * i + 6), r, hi);
        rays7[7 * i + 6] = r.t;
    }
}
/* intersectRayWithShape(AxisAlignedBox) (src/intersect.cpp:134-160) */
void ref_ray_aabb(const float* box6, float* rays7, int* hit, long n)
{
    for (long i = 0; i < n; ++i) {
        Ray r = mkray(rays7 + 7 * i);
        AxisAlignedBox b { g3(box6 + 6 * i), g3(box6 + 6 * i + 3) };
        hit[i] = intersectRayWithShape(b, r);
        rays7[7 * i + 6] = r.t;
    }
}
/* computeBarycentricCoord / interpolateNormal / interpolateTexCoord (src/interpolate.cpp:4-25) */
void ref_barycentric(const float* tri9, const float* p3, float* out3, long n)
{
    for (long i = 0; i < n; ++i) {
        glm::vec3 b = computeBarycentricCoord(g3(tri9 + 9 * i), g3(tri9 + 9 * i + 3), g3(tri9 + 9 * i + 6), g3(p3 + 3 * i));
        std::memcpy(out3 + 3 * i, &b, 12);
    }
}
void ref_interp_normal(const float* n9, const float* bary3, float* out3, long n)
{
    for (long i = 0; i < n; ++i) {
        glm::vec3 b = interpolateNormal(g3(n9 + 9 * i), g3(n9 + 9 * i + 3), g3(n9 + 9 * i + 6), g3(bary3 + 3 * i));
        std::memcpy(out3 + 3 * i, &b, 12);
    }
}
void ref_interp_texcoord(const float* t6, const float* bary3, float* out2, long n)
{
    for (long i = 0; i < n; ++i) {
        const float* t = t6 + 6 * i;
        glm::vec2 b = interpolateTexCoord({ t[0], t[1] }, { t[2], t[3] }, { t[4], t[5] }, g3(bary3 + 3 * i));
        std::memcpy(out2 + 2 * i, &b, 8);
    }
}
/* acquireTexel (src/texture.cpp:5-15) on texture t of a loaded scene; caller keeps the index in range */
void ref_acquire_texel(const ref_scene* s, int t, const float* uv2, float* out3, long n)
{
    Features f;
    for (long i = 0; i < n; ++i) {
        glm::vec3 c = acquireTexel(*s->textures[t], { uv2[2 * i], uv2[2 * i + 1] }, f);
        std::memcpy(out3 + 3 * i, &c, 12);
    }
}
/* computeShading (src/shading.cpp:6-30). in: lightPos3, lightCol3, ray7, normal3, kd3, ks3, shininess */
void ref_compute_shading(const float* light_pos3, const float* light_col3, const float* rays7, const float* normal3,
                         const float* kd3, const float* ks3, const float* shininess, float* out3, long n)
{
    Features f;
    for (long i = 0; i < n; ++i) {
        HitInfo hi;
        hi.normal = g3(normal3 + 3 * i);
        hi.material.kd = g3(kd3 + 3 * i);
        hi.material.ks = g3(ks3 + 3 * i);
        hi.material.shininess = shininess[i];
        glm::vec3 c = computeShading(g3(light_pos3 + 3 * i), g3(light_col3 + 3 * i), f, mkray(rays7 + 7 * i), hi);
        std::memcpy(out3 + 3 * i, &c, 12);
    }
}
/* computeReflectionRay (src/shading.cpp:33-51) */
void ref_reflection_ray(const float* rays7, const float* normal3, float* out7, long n)
{
    for (long i = 0; i < n; ++i) {
        HitInfo hi;
        hi.normal = g3(normal3 + 3 * i);
        Ray r = computeReflectionRay(mkray(rays7 + 7 * i), hi);
        const float o[7] = { r.origin.x, r.origin.y, r.origin.z, r.direction.x, r.direction.y, r.direction.z, r.t };
        std::memcpy(out7 + 7 * i, o, sizeof o);
    }
}

/* ---- camera arithmetic.  Trackball itself is not linkable here; this composes the same glm calls in the
 * order of framework/src/trackball.cpp:26-27 (half extents), :71-74 (position), :101-110 (generateRay),
 * with the aspect ratio of framework/src/window.cpp:379-384 and the CLI set-up of src/main.cpp:474-475. */
void ref_glm_camera(float fovy_deg, float dist, const float* look_at3, const float* rot_deg3, int w, int h,
                    float* origin3, float* quat_wxyz, float* half_wh)
{
    float fovy = glm::radians(fovy_deg);
    glm::vec3 e = glm::radians(g3(rot_deg3));
    glm::quat q(e);
    float halfH = std::tan(fovy / 2.0f);
    float aspect = (w == 0 || h == 0) ? 1.0f : float(w) / float(h);
    float halfW = aspect * halfH;
    glm::vec3 pos = g3(look_at3) + q * glm::vec3(0, 0, -dist);
    std::memcpy(origin3, &pos, 12);
    quat_wxyz[0] = q.w, quat_wxyz[1] = q.x, quat_wxyz[2] = q.y, quat_wxyz[3] = q.z;
    half_wh[0] = halfW, half_wh[1] = halfH;
}
void ref_glm_camera_rays(float fovy_deg, float dist, const float* look_at3, const float* rot_deg3, int w, int h,
                         const float* ndc2, float* out7, long n)
{
    float fovy = glm::radians(fovy_deg);
    glm::vec3 e = glm::radians(g3(rot_deg3));
    float halfH = std::tan(fovy / 2.0f);
    float aspect = (w == 0 || h == 0) ? 1.0f : float(w) / float(h);
    float halfW = aspect * halfH;
    for (long i = 0; i < n; ++i) {
        glm::vec2 pixel(ndc2[2 * i], ndc2[2 * i + 1]);
        const glm::vec3 cameraSpaceDirection = glm::normalize(glm::vec3(-pixel.x * halfW, pixel.y * halfH, 1.0f));
        glm::vec3 o = g3(look_at3) + glm::quat(e) * glm::vec3(0, 0, -dist);
        glm::vec3 d = glm::quat(e) * cameraSpaceDirection;
        const float r[7] = { o.x, o.y, o.z, d.x, d.y, d.z, std::numeric_limits<float>::max() };
        std::memcpy(out7 + 7 * i, r, sizeof r);
    }
}

/* Image::Image (framework/src/image.cpp:13-35: stb_image, three channels, u8 / 255.0f).  Size query with rgb == NULL;
 * returns 0 on success, -1 when the reference throws. */
int ref_image_load(const char* path, int* w, int* h, float* rgb)
{
    try {
        Image im { std::filesystem::path(path) };
        *w = im.width, *h = im.height;
        if (rgb)
            std::memcpy(rgb, im.pixels.data(), im.pixels.size() * sizeof(glm::vec3));
        return 0;
    } catch (...) {
        return -1;
    }
}

/* What Screen::writeBitmapToFile (src/screen.cpp:93-104) hands to stb: clamp, scale by 255, convert to bytes. */
int ref_write_bmp(const char* path, int w, int h, const float* rgb)
{
    std::vector<glm::u8vec4> px((size_t)w * h);
    for (size_t i = 0; i < px.size(); ++i)
        px[i] = glm::u8vec4(glm::vec4(glm::clamp(g3(rgb + 3 * i), 0.0f, 1.0f), 1.0f) * 255.0f);
    return stbi_write_bmp(path, w, h, 4, px.data());
}

} // extern "C"
