/*
 * srt_oracle.h -- C interface of the CPU restatement ("oracle") of the reference ray tracer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/, the smoke check
 * and bench.py's cpu_baseline leg may load it, and only as the checker or the reported CPU baseline.
 *
 * Every function restates, operation for operation (IEEE binary32, no FMA contraction), one function of
 * the reference; the reference location is cited next to each definition in srt_oracle.cpp.
 */
#ifndef SRT_ORACLE_H
#define SRT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Feature bits (mirror of struct Features / ExtraFeatures, reference src/common.h:55-80). */
enum {
    ORC_F_SHADING        = 1u << 0,
    ORC_F_RECURSIVE      = 1u << 1,
    ORC_F_HARD_SHADOW    = 1u << 2,
    ORC_F_SOFT_SHADOW    = 1u << 3,
    ORC_F_NORMAL_INTERP  = 1u << 4,
    ORC_F_TEXTURE        = 1u << 5,
    ORC_F_ACCEL          = 1u << 6,
    ORC_F_MULTISAMPLE    = 1u << 7,  /* extra.enableMultipleRaysPerPixel */
    ORC_F_SAH            = 1u << 8,  /* extra.enableBvhSahBinning (build only) */
    ORC_F_TRANSPARENCY   = 1u << 10, /* extra.enableTransparency */
    ORC_F_MOTION_BLUR    = 1u << 11, /* extra.enableMotionBlur */
    ORC_F_DEPTH_OF_FIELD = 1u << 12, /* extra.enableDepthOfField */
    ORC_F_ENVIRONMENT    = 1u << 13, /* extra.enableEnvironmentMapping */
    /* traversal policy: 0 = reference-faithful, 1 = additionally cull every box whose slab test fails */
    ORC_F_MISS_CULL      = 1u << 16,
    ORC_F_COARSE_SHADOW  = 1u << 17, /* with ORC_F_MISS_CULL: shadow queries test child boxes on a 16-bit grid (see srt_oracle.cpp) */
};

/* Light record: 22 floats. [0] = type (0 point, 1 segment, 2 parallelogram) then the struct's floats
 * in declaration order (reference src/common.h:38-53). */
#define ORC_LIGHT_STRIDE 22

typedef struct {
    uint32_t n_meshes;
    const uint32_t* mesh_vert_off; /* n_meshes+1 prefix offsets into verts (in vertices) */
    const uint32_t* mesh_tri_off;  /* n_meshes+1 prefix offsets into tris (in triangles) */
    const float* verts;            /* 8 floats per vertex: position, normal, texCoord */
    const uint32_t* tris;          /* 3 mesh-local vertex indices per triangle */
    const float* materials;        /* 8 floats per mesh: kd, ks, shininess, transparency */
    const int32_t* mesh_tex;       /* texture id per mesh or -1 */
    uint32_t n_textures;
    const uint32_t* tex_w;
    const uint32_t* tex_h;
    const uint64_t* tex_off;       /* offset of each texture in texels, in pixels */
    const float* texels;           /* RGB float, row-major, top row first */
    uint32_t n_lights;
    const float* lights;           /* ORC_LIGHT_STRIDE floats per light */
} orc_scene;

typedef struct {
    float origin[3];
    float quat[4];  /* w, x, y, z */
    float half_w, half_h;
    int32_t width, height;
} orc_camera;

typedef struct {
    uint64_t path_rays;    /* camera + reflection rays (one per getFinalColor call) */
    uint64_t shadow_rays;  /* shadow rays actually traced */
    uint64_t node_visits;  /* nodes that survive the pop test (the reference's drawAABB call site) */
    uint64_t inner_visits; /* of which inner nodes */
    uint64_t box_tests;
    uint64_t tri_tests;
    uint64_t shaded_hits;  /* path rays that hit */
} orc_counters;

typedef struct {
    int32_t hit;
    float t;
    int32_t mesh, tri;     /* winning triangle (scene mesh index, triangle index in mesh); -1 on miss */
    float normal[3];
    float bary[3];
    float uv[2];
    float kd[3];
} orc_hit;

typedef struct orc_bvh orc_bvh;

/* Camera set-up: reference framework/src/trackball.cpp:23-31,64-74 and window.cpp:379-384. */
void orc_camera_make(orc_camera* cam, float fovy_deg, float dist, const float look_at[3],
                     const float rot_deg[3], int32_t width, int32_t height);
/* Trackball::generateRay, trackball.cpp:101-110.  ray = origin[3], direction[3], t. */
void orc_camera_ray(const orc_camera* cam, float ndc_x, float ndc_y, float ray[7]);

orc_bvh* orc_bvh_build(const orc_scene* scene, uint32_t features);
void orc_bvh_free(orc_bvh*);
int32_t orc_bvh_num_levels(const orc_bvh*);
int32_t orc_bvh_num_leaves(const orc_bvh*);
int32_t orc_bvh_num_nodes(const orc_bvh*);
/* Node table in the reference's post-order.  Per node: is_leaf, level, a (leaf: first entry in the
 * ordered triangle list; inner: left child), b (leaf: entry count; inner: right child); box = 6 floats. */
void orc_bvh_dump(const orc_bvh*, int32_t* meta4, float* box6);
/* Ordered (mesh, tri) list that leaves index into; returns its length. */
int32_t orc_bvh_tri_order(const orc_bvh*, int32_t* mesh_tri2);

/* BvhInterface::intersect for n rays (7 floats each: o, d, t); ray t is updated in place. */
void orc_intersect_batch(const orc_bvh*, float* rays7, int64_t n, uint32_t features, orc_hit* out,
                         orc_counters* counters);

/* Leaf math, exposed so that it can be pinned one function at a time. */
int32_t orc_ray_triangle(const float v0[3], const float v1[3], const float v2[3], float ray[7]);
float orc_boxval(const float box[6], const float ray[7]);
int32_t orc_ray_aabb(const float box[6], float ray[7]);
void orc_barycentric(const float v0[3], const float v1[3], const float v2[3], const float p[3], float out[3]);
void orc_compute_shading(const float light_pos[3], const float light_col[3], const float ray[7],
                         const float normal[3], const float kd[3], const float ks[3], float shininess,
                         float out[3]);
void orc_reflection_ray(const float ray[7], const float normal[3], float out[7]);
void orc_texel(const float* texels, int32_t w, int32_t h, float u, float v, float out[3]);

/* getFinalColor for n rays; rgb = 3 floats per ray.  path_ids feed the soft-shadow sampler. */
void orc_final_color_batch(const orc_bvh*, const float* rays7, int64_t n, uint32_t features,
                           int32_t max_depth, uint32_t seed, float* rgb, orc_counters* counters);

/* renderRayTracing into a W*H*3 float framebuffer laid out like Screen (top row first).
 * rect = x0, y0, x1, y1 in pixels (y up, as in the reference's loops); pixels outside stay untouched.
 * nsx*nsy samples per pixel when ORC_F_MULTISAMPLE is set (the reference has nsx == nsy == numRays).
 * Optional per-pixel primary-hit outputs (W*H, Screen layout) of sample 0: tri id (mesh<<24 is NOT used;
 * two arrays instead), t. n_threads <= 0 means all cores. */
void orc_render(const orc_bvh*, const orc_camera* cam, uint32_t features, int32_t nsx, int32_t nsy,
                int32_t max_depth, uint32_t seed, const int32_t rect[4], float* rgb,
                int32_t* prim_mesh, int32_t* prim_tri, float* prim_t, int32_t n_threads,
                orc_counters* counters);

/* The Scene's motion-blur / depth-of-field variables (reference src/scene.h:32-40) and the environment image
 * (the reference's global envImage, src/render.cpp:38), for orc_render_ex. */
typedef struct {
    int32_t samples;       /* MB_samples or DOF_samples, whichever mode is active */
    float time0, time1;
    float direction[3];
    float focal_length;
    float aperture;
    const float* env_texels; /* RGB float, row-major, top row first; NULL for none */
    int32_t env_w, env_h;
} orc_extras;

/* orc_render plus the branches of reference src/render.cpp:135-178,218-223: environment map on misses, motion blur
 * and depth of field (jitter from the counter-based generator; the reference uses rand()). */
void orc_render_ex(const orc_bvh*, const orc_camera* cam, uint32_t features, int32_t nsx, int32_t nsy,
                   int32_t max_depth, uint32_t seed, const int32_t rect[4], float* rgb,
                   int32_t* prim_mesh, int32_t* prim_tri, float* prim_t, int32_t n_threads,
                   orc_counters* counters, const orc_extras* extras);
/* orc_final_color_batch with an environment image. */
void orc_final_color_batch_ex(const orc_bvh*, const float* rays7, int64_t n, uint32_t features,
                              int32_t max_depth, uint32_t seed, float* rgb, orc_counters* counters, const orc_extras* extras);

/* Counter-based uniform sample in [0,1) shared with the device code. */
float orc_rand01(uint32_t seed, uint32_t path, uint32_t depth, uint32_t light, uint32_t sample, uint32_t dim);
/* Screen::applyBloomFilter (reference src/screen.cpp:32-78) on a frame stored top row first, in place.
 * Parity unpinned: screen.cpp needs the OpenGL headers and cannot be built here. */
void orc_bloom(float* rgb, int32_t width, int32_t height, float threshold, int32_t box_size);

#ifdef __cplusplus
}
#endif
#endif
