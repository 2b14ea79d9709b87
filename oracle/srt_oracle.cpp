/*
 * srt_oracle.cpp -- CPU restatement of the reference ray tracer's BVH / intersection / shading path.
 *
 * TEST INFRASTRUCTURE ONLY (see srt_oracle.h).  Build with -ffp-contract=off and without -ffast-math:
 * the reference is compiled for baseline x86-64 (no FMA), so every +,-,*,/ and sqrt here must round once.
 *
 * Pinning status.  The leaf functions (ray/triangle, ray/box, barycentrics, texel fetch, Phong shading,
 * reflection ray) and the glm arithmetic used by the camera are checked bit for bit against the
 * reference's own object code (oracle/_ref, built from /root/reference by oracle/Makefile).  The BVH
 * build, the traversal state machine, the light loop and the frame loop cannot be linked from the
 * reference in this image (they need its OpenGL draw/window translation units); they are pinned by the
 * known answers recorded from a reference run: node/level/leaf counts per scene, one primary hit
 * (t, mesh, triangle), and ray / node-visit totals of two frames.  tests/test_oracle_*.py hold both.
 *
 * All helpers below follow glm 0.9.9.9 scalar code paths (reference framework/third_party/glm):
 *   dot   = (a.x*b.x + a.y*b.y) + a.z*b.z            glm/detail/func_geometric.inl:48-55
 *   cross = (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)        func_geometric.inl:68-79
 *   normalize(v) = v * (1.0f / sqrt(dot(v,v)))        func_geometric.inl:82-90, func_exponential.inl:134-139
 *   length(v) = sqrt(dot(v,v))                        func_geometric.inl:8-14
 *   min(x,y) = (y < x) ? y : x ; max(x,y) = (x < y) ? y : x   (std:: and glm:: agree)
 */
#include "srt_oracle.h"

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct V3 {
    float x, y, z;
};
struct V2 {
    float x, y;
};

inline V3 v3(const float* p) { return { p[0], p[1], p[2] }; }
inline V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 operator-(V3 a) { return { -a.x, -a.y, -a.z }; }
inline V3 operator*(V3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline V3 operator*(float s, V3 a) { return { s * a.x, s * a.y, s * a.z }; }
inline V3 operator*(V3 a, V3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
inline V3 operator/(V3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
inline bool operator==(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 x, V3 y) { return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
inline float length(V3 v) { return std::sqrt(dot(v, v)); }
inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline float fmin_std(float a, float b) { return (b < a) ? b : a; }
inline float fmax_std(float a, float b) { return (a < b) ? b : a; }
inline float clamp01(float x) { return fmin_std(fmax_std(x, 0.0f), 1.0f); }

struct Ray {
    V3 o, d;
    float t;
};

struct Material {
    V3 kd, ks;
    float shininess, transparency;
    int tex;
};

struct HitInfo {
    V3 normal { 0, 0, 0 };
    V3 bary { 0, 0, 0 }; /* the reference leaves this uninitialised (src/common.h:16); defined as 0 here */
    V2 uv { 0, 0 };
    Material material { { 0, 0, 0 }, { 0, 0, 0 }, 1.0f, 1.0f, -1 };
    int mesh = -1, tri = -1;
};

struct Box {
    V3 lo, hi;
};

struct Node {
    bool leaf;
    int level;
    int a, b; /* leaf: first entry / count in order[]; inner: left / right child */
    Box box;
};

struct TriRef {
    uint32_t mesh, tri;
};

} // namespace

struct orc_bvh {
    orc_scene s;
    std::vector<Node> nodes;
    std::vector<TriRef> order; /* leaf payload, concatenated in node creation order */
    int levels = 0, leaves = 0;
    /* shadow queries on coarse boxes (ORC_F_COARSE_SHADOW), filled by coarse_prepare at the end of the build */
    bool coarse_ok = false;
    float q_scale[3] = { 1.0f, 1.0f, 1.0f };
    std::vector<std::array<uint32_t, 3>> cq; /* per node: its box on the 16-bit grid, lower | upper << 16 per axis */
    std::vector<std::array<uint32_t, 3>> tq; /* per entry of order[]: that triangle's own box on the grid */
    size_t wide_cap = 0; /* pending siblings a coarse shadow walk may stack; a query that wants more takes the exact walk */
};

namespace {

/* ---------------------------------------------------------------- leaf math */

/* intersectRayWithShape(AxisAlignedBox), reference src/intersect.cpp:134-160.  Returns pass/fail and the
 * value getClosestIntersectionWithBox (src/bounding_volume_hierarchy.cpp:305-310) would return. */
inline bool slab(const Box& b, const Ray& r, float& val)
{
    float txmin = (b.lo.x - r.o.x) / r.d.x;
    float tymin = (b.lo.y - r.o.y) / r.d.y;
    float tzmin = (b.lo.z - r.o.z) / r.d.z;
    float txmax = (b.hi.x - r.o.x) / r.d.x;
    float tymax = (b.hi.y - r.o.y) / r.d.y;
    float tzmax = (b.hi.z - r.o.z) / r.d.z;
    float tinx = fmin_std(txmin, txmax), toutx = fmax_std(txmin, txmax);
    float tiny = fmin_std(tymin, tymax), touty = fmax_std(tymin, tymax);
    float tinz = fmin_std(tzmin, tzmax), toutz = fmax_std(tzmin, tzmax);
    float tin = fmax_std(tinx, fmax_std(tiny, tinz));
    float tout = fmin_std(toutx, fmin_std(touty, toutz));
    if (tin > tout || tout < 0 || tin > r.t) {
        val = r.t;
        return false;
    }
    val = tin;
    if (tin <= 0)
        val = tout;
    return true;
}

/* sameSide / pointInTriangle, src/intersect.cpp:13-26 */
inline bool same_side(V3 p1, V3 p2, V3 a, V3 b)
{
    V3 c1 = cross(b - a, p1 - a);
    V3 c2 = cross(b - a, p2 - a);
    return dot(c1, c2) >= 0;
}

/* intersectRayWithTriangle with trianglePlane and intersectRayWithPlane inlined, src/intersect.cpp:28-77 */
inline bool ray_triangle(V3 v0, V3 v1, V3 v2, Ray& r)
{
    float prev_t = r.t;
    V3 n;
    float D;
    V3 c = cross(v1 - v0, v2 - v0);
    if (length(c) == 0.0f) {
        n = { 0, 0, 0 };
        D = 0;
    } else {
        n = normalize(c);
        D = dot(v0, n);
    }
    if (dot(r.d, n) == 0.0f || dot(r.o, n) == D)
        return false;
    float new_t = (D - dot(r.o, n)) / dot(r.d, n);
    if (new_t < 0)
        return false;
    if (!(new_t <= prev_t))
        return false;
    r.t = new_t;
    V3 p = r.o + r.d * r.t;
    if (same_side(p, v0, v1, v2) && same_side(p, v1, v0, v2) && same_side(p, v2, v0, v1))
        return true;
    r.t = prev_t;
    return false;
}

/* computeBarycentricCoord, src/interpolate.cpp:4-15 */
inline V3 barycentric(V3 v0, V3 v1, V3 v2, V3 p)
{
    V3 d0 = v1 - v0, d1 = v2 - v1, d2 = v0 - v2;
    float a0 = length(cross(d1, v1 - p));
    float a1 = length(cross(d2, v2 - p));
    float a2 = length(cross(d0, v0 - p));
    float sum = a0 + a1 + a2;
    if (sum == 0)
        return { 1, 0, 0 };
    return { a0 / sum, a1 / sum, a2 / sum };
}

/* acquireTexel, src/texture.cpp:5-15: truncating fetch, v flipped, no filtering.  The reference does
 * not bound the index (u == 1 or v == 0 run past the row / the image); here the linear index is
 * clamped into the image, which is identical wherever the reference's behaviour is defined. */
inline V3 texel(const float* px, int w, int h, V2 uv)
{
    int i = (int)(uv.x * (float)w);
    int j = (int)((float)h - uv.y * (float)h);
    long idx = (long)j * w + i;
    long last = (long)w * h - 1;
    if (idx < 0)
        idx = 0;
    if (idx > last)
        idx = last;
    return { px[3 * idx], px[3 * idx + 1], px[3 * idx + 2] };
}

/* computeShading, src/shading.cpp:6-30 */
inline V3 compute_shading(V3 lightPos, V3 lightCol, const Ray& ray, V3 hitNormal, V3 kd, V3 ks, float shininess)
{
    V3 rayPosition = ray.o + ray.d * ray.t;
    V3 L = normalize(lightPos - rayPosition);
    V3 N = normalize(hitNormal);
    V3 camV = rayPosition - ray.o;
    V3 R = ((2.0f * dot(L, N)) * N) - L;
    V3 lambert = (lightCol * kd) * clamp01(dot(N, L));
    V3 spec = (lightCol * ks) * std::pow(clamp01(dot(R, normalize(camV))), shininess);
    return lambert + spec;
}

/* computeReflectionRay, src/shading.cpp:33-51.  glm::pow(10, -5) is std::pow(int,int) -> double,
 * narrowed to float: 9.99999975e-06f (bits 0x3727C5AC). */
inline Ray reflection_ray(const Ray& ray, V3 hitNormal)
{
    float offset = (float)std::pow(10.0, -5.0);
    Ray out;
    V3 dn = normalize(ray.d);
    V3 nn = normalize(hitNormal);
    out.d = dn - ((2.0f * dot(dn, nn)) * nn);
    out.o = (ray.o + ray.d * ray.t) + (offset * out.d);
    out.t = FLT_MAX;
    return out;
}

/* ---------------------------------------------------------------- scene access */

inline const float* vert(const orc_scene& s, uint32_t mesh, uint32_t vi)
{
    return s.verts + 8ull * (s.mesh_vert_off[mesh] + vi);
}
inline const uint32_t* tri(const orc_scene& s, uint32_t mesh, uint32_t ti)
{
    return s.tris + 3ull * (s.mesh_tri_off[mesh] + ti);
}
inline Material material_of(const orc_scene& s, uint32_t mesh)
{
    const float* m = s.materials + 8ull * mesh;
    return { { m[0], m[1], m[2] }, { m[3], m[4], m[5] }, m[6], m[7], s.mesh_tex[mesh] };
}

/* getMedian, src/bounding_volume_hierarchy.cpp:37-45 */
inline V3 centroid(const orc_scene& s, TriRef t)
{
    const uint32_t* tr = tri(s, t.mesh, t.tri);
    V3 p1 = v3(vert(s, t.mesh, tr[0])), p2 = v3(vert(s, t.mesh, tr[1])), p3 = v3(vert(s, t.mesh, tr[2]));
    return ((p1 + p2) + p3) / 3.0f;
}
inline float axis_of(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

/* getBox, src/bounding_volume_hierarchy.cpp:47-68 */
Box get_box(const orc_scene& s, const TriRef* begin, const TriRef* end)
{
    V3 lower = { FLT_MAX, FLT_MAX, FLT_MAX };
    V3 upper = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (const TriRef* it = begin; it != end; ++it) {
        const uint32_t* tr = tri(s, it->mesh, it->tri);
        V3 a = v3(vert(s, it->mesh, tr[0])), b = v3(vert(s, it->mesh, tr[1])), c = v3(vert(s, it->mesh, tr[2]));
        upper.x = fmax_std(fmax_std(upper.x, a.x), fmax_std(b.x, c.x));
        upper.y = fmax_std(fmax_std(upper.y, a.y), fmax_std(b.y, c.y));
        upper.z = fmax_std(fmax_std(upper.z, a.z), fmax_std(b.z, c.z));
        lower.x = fmin_std(fmin_std(lower.x, a.x), fmin_std(b.x, c.x));
        lower.y = fmin_std(fmin_std(lower.y, a.y), fmin_std(b.y, c.y));
        lower.z = fmin_std(fmin_std(lower.z, a.z), fmin_std(b.z, c.z));
    }
    return { lower, upper };
}

/* ---------------------------------------------------------------- build */

struct Builder {
    orc_bvh& bvh;
    std::vector<TriRef>& tris;

    size_t leaf(size_t left, size_t right, int level)
    {
        Node n;
        n.leaf = true;
        n.level = level;
        n.a = (int)bvh.order.size();
        n.b = (int)(right - left);
        bvh.order.insert(bvh.order.end(), tris.begin() + left, tris.begin() + right);
        n.box = get_box(bvh.s, tris.data() + left, tris.data() + right);
        bvh.nodes.push_back(n);
        bvh.leaves += 1;
        return bvh.nodes.size() - 1;
    }
    void sort_range(size_t left, size_t right, int axis)
    {
        const orc_scene& s = bvh.s;
        /* std::sort is not stable: tie order is whatever libstdc++'s introsort yields for this
         * comparator and input order, and it is part of the result (leaf order = test order). */
        std::sort(tris.begin() + left, tris.begin() + right, [&s, axis](const TriRef& t1, const TriRef& t2) {
            return axis_of(centroid(s, t1), axis) < axis_of(centroid(s, t2), axis);
        });
    }
    /* constructorHelper, src/bounding_volume_hierarchy.cpp:70-94 */
    size_t median_split(size_t left, size_t right, int axis, int level)
    {
        if (right - left <= 1 || level > 16)
            return leaf(left, right, level);
        sort_range(left, right, axis);
        size_t median = (left + right + 1) / 2;
        size_t li = median_split(left, median, (axis + 1) % 3, level + 1);
        size_t ri = median_split(median, right, (axis + 1) % 3, level + 1);
        Node n;
        n.leaf = false;
        n.level = level;
        n.a = (int)li;
        n.b = (int)ri;
        n.box = get_box(bvh.s, tris.data() + left, tris.data() + right);
        bvh.nodes.push_back(n);
        return bvh.nodes.size() - 1;
    }
    /* getProb, src/bounding_volume_hierarchy.cpp:96-113 */
    static float prob(const Box& in, const Box& out)
    {
        float ia = std::abs(in.lo.x - in.hi.x), ib = std::abs(in.lo.y - in.hi.y), ic = std::abs(in.lo.z - in.hi.z);
        float oa = std::abs(out.lo.x - out.hi.x), ob = std::abs(out.lo.y - out.hi.y), oc = std::abs(out.lo.z - out.hi.z);
        float areaInner = 2 * (ia * ib + ib * ic + ic * ia);
        float areaOuter = 2 * (oa * ob + ob * oc + oc * oa);
        if (areaOuter == 0)
            return 0;
        return areaInner / areaOuter;
    }
    /* sahConstructorHelper + calculateCostOfDivision, src/bounding_volume_hierarchy.cpp:116-178 */
    size_t sah_split(size_t left, size_t right, int axis, int level)
    {
        if (right - left <= 1 || level > 16)
            return leaf(left, right, level);
        sort_range(left, right, axis);
        const orc_scene& s = bvh.s;
        const size_t planes = std::min((size_t)5, right - left - 1);
        const float lb = axis_of(centroid(s, tris[left]), axis);
        const float rb = axis_of(centroid(s, tris[right - 1]), axis);
        const float step = (rb - lb) / (planes + 1);
        float minCost = FLT_MAX;
        size_t median = left;
        for (size_t i = 1; i <= planes; ++i) {
            const float boundary = lb + step * i;
            size_t middle = left;
            /* the reference dereferences before the bound check (:124); same order of tests, with the
             * dereference guarded so that middle == right is not read */
            while (middle != right && axis_of(centroid(s, tris[middle]), axis) < boundary)
                ++middle;
            Box outer = get_box(s, tris.data() + left, tris.data() + right);
            Box lbx = get_box(s, tris.data() + left, tris.data() + middle);
            Box rbx = get_box(s, tris.data() + middle, tris.data() + right);
            float cost = prob(lbx, outer) * (middle - left) + prob(rbx, outer) * (right - middle);
            if (cost < minCost) {
                minCost = cost;
                median = middle;
            }
        }
        size_t li = sah_split(left, median, (axis + 1) % 3, level + 1);
        size_t ri = sah_split(median, right, (axis + 1) % 3, level + 1);
        Node n;
        n.leaf = false;
        n.level = level;
        n.a = (int)li;
        n.b = (int)ri;
        n.box = get_box(bvh.s, tris.data() + left, tris.data() + right);
        bvh.nodes.push_back(n);
        return bvh.nodes.size() - 1;
    }
};

/* ---------------------------------------------------------------- traversal */

struct Ctx {
    const orc_bvh& bvh;
    uint32_t features;
    int max_depth;
    uint32_t seed;
    orc_counters cnt {};
    const orc_extras* ex = nullptr;
};

/* intersectWithLeafTriangle + shoudlBeReverted, src/bounding_volume_hierarchy.cpp:262-303 */
bool leaf_triangle(Ctx& c, Ray& ray, HitInfo& hit, uint32_t mesh, uint32_t ti)
{
    const orc_scene& s = c.bvh.s;
    const uint32_t* tr = tri(s, mesh, ti);
    const float *p0 = vert(s, mesh, tr[0]), *p1 = vert(s, mesh, tr[1]), *p2 = vert(s, mesh, tr[2]);
    V3 v0 = v3(p0), v1 = v3(p1), v2 = v3(p2);
    c.cnt.tri_tests += 1;
    if (!ray_triangle(v0, v1, v2, ray))
        return false;
    hit.material = material_of(s, mesh);
    hit.normal = normalize(cross(v1 - v0, v2 - v0));
    if (c.features & ORC_F_NORMAL_INTERP) {
        V3 ip = ray.o + ray.d * ray.t;
        hit.bary = barycentric(v0, v1, v2, ip);
        V3 n0 = v3(p0 + 3), n1 = v3(p1 + 3), n2 = v3(p2 + 3);
        hit.normal = normalize(((hit.bary.x * n0) + (hit.bary.y * n1)) + (hit.bary.z * n2));
    }
    {
        V3 p = ray.o + ray.d * ray.t;
        V3 toCam = normalize(ray.o - p);
        if (dot(hit.normal, toCam) <= 0)
            hit.normal = -hit.normal;
    }
    if (c.features & ORC_F_TEXTURE) {
        V2 t0 = { p0[6], p0[7] }, t1 = { p1[6], p1[7] }, t2 = { p2[6], p2[7] };
        hit.uv = { (t0.x * hit.bary.x + t1.x * hit.bary.y) + t2.x * hit.bary.z,
            (t0.y * hit.bary.x + t1.y * hit.bary.y) + t2.y * hit.bary.z };
        int tex = hit.material.tex;
        if (tex >= 0)
            hit.material.kd = texel(s.texels + 3ull * s.tex_off[tex], (int)s.tex_w[tex], (int)s.tex_h[tex], hit.uv);
    }
    hit.mesh = (int)mesh;
    hit.tri = (int)ti;
    return true;
}

/* BoundingVolumeHierarchy::intersect, src/bounding_volume_hierarchy.cpp:316-393.
 * shadow_stop: leave as soon as a leaf has been tested with the ray's t below 0.99.  Only for shadow rays (t = 1 at
 * the start) whose caller asks `hit && t < 0.99`: t never grows, so the answer is decided there.  That is how the HIP
 * path's exact shadow walk counts its work; the restated coarse walk falls back to it. */
bool intersect(Ctx& c, Ray& ray, HitInfo& hit, bool shadow_stop = false)
{
    const orc_bvh& bvh = c.bvh;
    const orc_scene& s = bvh.s;
    if (!(c.features & ORC_F_ACCEL)) {
        bool any = false;
        for (uint32_t m = 0; m < s.n_meshes; ++m) {
            uint32_t nt = s.mesh_tri_off[m + 1] - s.mesh_tri_off[m];
            for (uint32_t t = 0; t < nt; ++t)
                if (leaf_triangle(c, ray, hit, m, t))
                    any = true;
        }
        return any; /* spheres: the reference's sphere test always returns false (src/intersect.cpp:81-125) */
    }
    if (bvh.nodes.empty())
        return false;
    const bool cull = (c.features & ORC_F_MISS_CULL) != 0;
    float closest = FLT_MAX;
    /* Deliberately a fresh heap container per call, as the reference's std::deque (:340): this file is
     * also the timed CPU baseline. */
    std::vector<int> stack;
    float val;
    bool pass = slab(bvh.nodes.back().box, ray, val);
    c.cnt.box_tests += 1;
    if (val < closest && (pass || !cull))
        stack.push_back((int)bvh.nodes.size() - 1);
    while (!stack.empty()) {
        const Node& next = bvh.nodes[stack.back()];
        stack.pop_back();
        pass = slab(next.box, ray, val);
        c.cnt.box_tests += 1;
        if (val > closest || (cull && !pass))
            continue;
        c.cnt.node_visits += 1;
        if (next.leaf) {
            for (int k = 0; k < next.b; ++k) {
                TriRef t = bvh.order[next.a + k];
                if (leaf_triangle(c, ray, hit, t.mesh, t.tri))
                    closest = fmin_std(closest, ray.t);
            }
            if (shadow_stop && closest != FLT_MAX && ray.t < 0.99f)
                return true;
        } else {
            c.cnt.inner_visits += 1;
            const int ch[2] = { next.a, next.b };
            for (int k = 0; k < 2; ++k) {
                pass = slab(bvh.nodes[ch[k]].box, ray, val);
                c.cnt.box_tests += 1;
                if (val < closest && (pass || !cull))
                    stack.push_back(ch[k]);
            }
        }
    }
    return closest != FLT_MAX;
}

/* ---------------------------------------------------------------- shadow queries on coarse boxes
 * The policy ORC_F_COARSE_SHADOW of the HIP path, restated (it has no counterpart in the reference: like
 * ORC_F_MISS_CULL it only changes which boxes a query looks into).  A shadow query tests boxes on a 16-bit grid over
 * the root box, each rounded outwards by more than a grid step, with one fused multiply-add per slab value, and two
 * levels at a time: at an inner node of even level it tests the four grandchildren (a child that is a leaf stands for
 * itself) in the order the reference's walk would pop them - right-right, right-left, left-right, left-left - goes on
 * in the first one the ray meets and stacks the others.  The first leaf with an accepted triangle answers the query
 * when the hit lies below 0.99 and
 * hands it to the exact walk otherwise.  Every float operation is written out in the order the kernels use. */
inline bool fast_dir(float d)
{
    uint32_t b;
    std::memcpy(&b, &d, 4);
    const uint32_t e = (b >> 23) & 0xFFu;
    return e >= 107u && e <= 147u;
}
inline bool fast_coord(float c)
{
    uint32_t b;
    std::memcpy(&b, &c, 4);
    b &= 0x7FFFFFFFu;
    const uint32_t e = b >> 23;
    return b == 0u || (e >= 87u && e < 147u);
}
inline uint32_t coarse_axis(float lo, float hi, float rlo, float s, bool& bad)
{
    const float ul = (lo - rlo) * s, uh = (hi - rlo) * s;
    if (!(ul >= 0.0f && uh <= 65534.0f && ul <= uh))
        bad = true;
    const float ql = fmin_std(fmax_std(std::floor(ul), 0.0f), 65535.0f);
    const float qh = fmin_std(fmax_std(std::ceil(uh) + 2.0f, 0.0f), 65535.0f);
    return (uint32_t)ql | ((uint32_t)qh << 16);
}
void coarse_prepare(orc_bvh& b)
{
    b.coarse_ok = false;
    if (b.nodes.size() < 2) /* no inner node */
        return;
    const Box& root = b.nodes.back().box;
    const float rlo[3] = { root.lo.x, root.lo.y, root.lo.z }, rhi[3] = { root.hi.x, root.hi.y, root.hi.z };
    for (int k = 0; k < 3; ++k) {
        const float ext = rhi[k] - rlo[k];
        b.q_scale[k] = ext > 0.0f ? 65533.0f / ext : 1.0f;
    }
    bool bad = false;
    b.cq.assign(b.nodes.size(), { 0u, 0u, 0u });
    for (size_t i = 0; i < b.nodes.size(); ++i) {
        const Box& x = b.nodes[i].box;
        const float lo[3] = { x.lo.x, x.lo.y, x.lo.z }, hi[3] = { x.hi.x, x.hi.y, x.hi.z };
        for (int k = 0; k < 3; ++k) {
            if (!(fast_coord(lo[k]) && fast_coord(hi[k]) && lo[k] <= hi[k]))
                bad = true; /* outside the ranges of the kernels' fast slab test: they keep the plain policy for the scene */
            if (i + 1 < b.nodes.size())
                b.cq[i][k] = coarse_axis(lo[k], hi[k], rlo[k], b.q_scale[k], bad);
        }
    }
    /* boxes of single triangles, for the records whose leaves are listed triangle by triangle */
    b.tq.assign(b.order.size(), { 0u, 0u, 0u });
    for (size_t i = 0; i < b.order.size(); ++i) {
        const orc_scene& s = b.s;
        const uint32_t* t = s.tris + 3ull * (s.mesh_tri_off[b.order[i].mesh] + b.order[i].tri);
        const float* base = s.verts + 8ull * s.mesh_vert_off[b.order[i].mesh];
        const float *p0 = base + 8ull * t[0], *p1 = base + 8ull * t[1], *p2 = base + 8ull * t[2];
        for (int k = 0; k < 3; ++k) {
            const float lo = fmin_std(fmin_std(p0[k], p1[k]), p2[k]), hi = fmax_std(fmax_std(p0[k], p1[k]), p2[k]);
            b.tq[i][k] = coarse_axis(lo, hi, rlo[k], b.q_scale[k], bad);
        }
    }
    b.coarse_ok = !bad;
    /* the HIP path's stack capacity for this walk: what no walk of a shallow tree can overflow (three entries per wide
     * level and one), at most 19 and at least levels + 1; SRT_WIDE_CAP lowers it on both sides (tests) */
    {
        const size_t levels = (size_t)b.levels, never = 3 * (levels / 2) + 1;
        b.wide_cap = std::max(levels + 1, std::min<size_t>(never, 19));
    }
    if (const char* cap = std::getenv("SRT_WIDE_CAP")) {
        const long v = std::strtol(cap, nullptr, 10);
        if (v >= 0 && (size_t)v < b.wide_cap)
            b.wide_cap = (size_t)v;
    }
}
struct CoarseRay {
    float qi[3], qc[3];
};
inline bool coarse_meets(const std::array<uint32_t, 3>& q, const CoarseRay& r)
{
    float tl[3], th[3];
    for (int k = 0; k < 3; ++k) {
        const float a = std::fmaf((float)(q[k] & 0xFFFFu), r.qi[k], r.qc[k]), b = std::fmaf((float)(q[k] >> 16), r.qi[k], r.qc[k]);
        tl[k] = fmin_std(a, b), th[k] = fmax_std(a, b);
    }
    const float tin = fmax_std(tl[0], fmax_std(tl[1], tl[2]));
    const float tout = fmin_std(th[0], fmin_std(th[1], th[2]));
    return !(tin > tout || tout < 0.0f || tin > 1.0f);
}
/* Returns true when the coarse walk applies to this ray and leaves the answer in `occluded`; false: the caller walks the
 * exact way (rays outside the fast ranges, origins far from the root box, and a first hit in [0.99, 1]). */
bool shadow_coarse(Ctx& c, const Ray& ray, const HitInfo& like, bool& occluded)
{
    const orc_bvh& bvh = c.bvh;
    const float o[3] = { ray.o.x, ray.o.y, ray.o.z }, d[3] = { ray.d.x, ray.d.y, ray.d.z };
    for (int k = 0; k < 3; ++k)
        if (!fast_dir(d[k]) || !fast_coord(o[k]))
            return false;
    /* no test of the root box: a ray that misses it meets none of the root's four slots (or a rounded-out one) */
    const Box& root = bvh.nodes.back().box;
    const float rlo[3] = { root.lo.x, root.lo.y, root.lo.z };
    CoarseRay cr;
    for (int k = 0; k < 3; ++k) {
        const float og = (o[k] - rlo[k]) * bvh.q_scale[k] + 1.0f;
        cr.qi[k] = 1.0f / (d[k] * bvh.q_scale[k]);
        cr.qc[k] = -(og * cr.qi[k]);
        if (!(og >= -65536.0f && og <= 131072.0f))
            return false;
    }
    /* a walk position: a node (>= 0), or a single triangle given by its place in order[] (-(place + 1)) */
    std::vector<int> stack;
    int cur = (int)bvh.nodes.size() - 1;
    for (;;) {
        bool descended = false;
        c.cnt.node_visits += 1;
        if (cur < 0 || bvh.nodes[cur].leaf) {
            const int first = cur < 0 ? -(cur + 1) : bvh.nodes[cur].a, count = cur < 0 ? 1 : bvh.nodes[cur].b;
            Ray r = ray;
            HitInfo h = like;
            bool hit = false;
            for (int k = 0; k < count; ++k) {
                TriRef t = bvh.order[first + k];
                if (leaf_triangle(c, r, h, t.mesh, t.tri))
                    hit = true;
            }
            if (hit) {
                if (r.t < 0.99f) {
                    occluded = true;
                    return true;
                }
                return false;
            }
        } else {
            const Node& n = bvh.nodes[cur];
            c.cnt.inner_visits += 1;
            c.cnt.box_tests += 4;
            int slot[4], ns = 0;
            for (int child : { n.b, n.a }) {
                const Node& cn = bvh.nodes[child];
                if (cn.leaf) {
                    slot[ns++] = child;
                } else {
                    slot[ns++] = cn.b;
                    slot[ns++] = cn.a;
                }
            }
            /* leaves listed triangle by triangle when that fits the four slots */
            int singles = 0;
            bool anyLeaf = false;
            for (int k = 0; k < ns; ++k) {
                const Node& sn = bvh.nodes[slot[k]];
                anyLeaf = anyLeaf || sn.leaf;
                singles += sn.leaf ? sn.b : 1;
            }
            if (anyLeaf && singles <= 4) {
                int wide[4], nw = 0;
                for (int k = 0; k < ns; ++k) {
                    const Node& sn = bvh.nodes[slot[k]];
                    if (sn.leaf && sn.b != 1) {
                        for (int j = 0; j < sn.b; ++j)
                            wide[nw++] = -(sn.a + j + 1);
                    } else {
                        wide[nw++] = slot[k];
                    }
                }
                ns = nw;
                for (int k = 0; k < ns; ++k)
                    slot[k] = wide[k];
            }
            bool met[4];
            int first = -1;
            size_t pushes = 0;
            for (int k = 0; k < ns; ++k) {
                met[k] = coarse_meets(slot[k] < 0 ? bvh.tq[-(slot[k] + 1)] : bvh.cq[slot[k]], cr);
                if (met[k]) {
                    if (first < 0)
                        first = k;
                    else
                        pushes += 1;
                }
            }
            if (pushes > 0 && stack.size() + 3 >= bvh.wide_cap)
                return false; /* the HIP path wants three free entries and a spare one at any visit that stacks: the exact walk answers */
            for (int k = ns - 1; k > first && first >= 0; --k)
                if (met[k])
                    stack.push_back(slot[k]);
            if (first >= 0) {
                cur = slot[first];
                descended = true;
            }
        }
        if (descended)
            continue;
        if (stack.empty()) {
            occluded = false;
            return true;
        }
        cur = stack.back();
        stack.pop_back();
    }
}

/* ---------------------------------------------------------------- lights */

/* the arguments folded into one word, each times an odd constant of its own, then one two-multiplication finaliser */
inline uint32_t final32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
inline float rand01(uint32_t seed, uint32_t path, uint32_t depth, uint32_t light, uint32_t sample, uint32_t dim)
{
    const uint32_t x = seed ^ (path * 0x9E3779B1u) ^ (depth * 0x85EBCA77u) ^ (light * 0xC2B2AE3Du) ^ (sample * 0x27D4EB2Fu) ^ (dim * 0x165667B1u);
    return (float)(final32(x) >> 8) * (1.0f / 16777216.0f);
}

/* testVisibilityLightSample, src/light.cpp:43-74 */
float visibility(Ctx& c, V3 samplePos, const Ray& ray, const HitInfo& hit)
{
    if (!(c.features & (ORC_F_HARD_SHADOW | ORC_F_SOFT_SHADOW)))
        return 1;
    V3 ip = ray.o + ray.d * ray.t;
    if (ip == samplePos)
        return 1;
    Ray nr = { ip, samplePos - ip, 1.0f };
    nr.o = nr.o + normalize(nr.d) * 0.001f;
    HitInfo h2 = hit;
    c.cnt.shadow_rays += 1;
    if ((c.features & ORC_F_COARSE_SHADOW) && (c.features & ORC_F_MISS_CULL) && (c.features & ORC_F_ACCEL) && c.bvh.coarse_ok) {
        bool occluded = false;
        if (shadow_coarse(c, nr, hit, occluded))
            return occluded ? 0.0f : 1.0f;
        /* the coarse walk does not apply, or gave up: the exact walk, stopping where the HIP path's stops */
        return (intersect(c, nr, h2, true) && (double)nr.t < 1 - 0.01) ? 0.0f : 1.0f;
    }
    /* `newRay.t < 1 - 0.01` compares in double (:62) */
    if (intersect(c, nr, h2) && (double)nr.t < 1 - 0.01)
        return 0.0f;
    return 1.0f;
}

/* computeLightContribution, src/light.cpp:109-166; samplers :14-39.  The reference draws its
 * samples from libc rand(); here they come from the counter-based generator above. */
V3 light_contribution(Ctx& c, const Ray& ray, const HitInfo& hit, uint32_t path, uint32_t depth)
{
    const orc_scene& s = c.bvh.s;
    if (!(c.features & ORC_F_SHADING))
        return hit.material.kd;
    V3 res = { 0, 0, 0 };
    const V3 kd = hit.material.kd, ks = hit.material.ks;
    const float sh = hit.material.shininess;
    for (uint32_t li = 0; li < s.n_lights; ++li) {
        const float* L = s.lights + (size_t)ORC_LIGHT_STRIDE * li;
        int type = (int)L[0];
        if (type == 0) {
            V3 pos = v3(L + 1), col = v3(L + 4);
            if (c.features & ORC_F_HARD_SHADOW)
                res = res + compute_shading(pos, col, ray, hit.normal, kd, ks, sh) * visibility(c, pos, ray, hit);
            else
                res = res + compute_shading(pos, col, ray, hit.normal, kd, ks, sh);
        } else if (type == 1) {
            if (!(c.features & ORC_F_SOFT_SHADOW))
                continue;
            V3 e0 = v3(L + 1), e1 = v3(L + 4), c0 = v3(L + 7), c1 = v3(L + 10);
            const int N = 100;
            for (int t = 0; t < N; ++t) {
                float r = rand01(c.seed, path, depth, li, (uint32_t)t, 0);
                float trand = (float)t + r;
                float a = trand / (float)N;
                V3 pos = { e0.x * a + e1.x * (1 - a), e0.y * a + e1.y * (1 - a), e0.z * a + e1.z * (1 - a) };
                V3 col = { c0.x * a + c1.x * (1 - a), c0.y * a + c1.y * (1 - a), c0.z * a + c1.z * (1 - a) };
                res = res + (compute_shading(pos, col, ray, hit.normal, kd, ks, sh) / (float)N) * visibility(c, pos, ray, hit);
            }
        } else {
            if (!(c.features & ORC_F_SOFT_SHADOW))
                continue;
            V3 p0 = v3(L + 1), e01 = v3(L + 4), e02 = v3(L + 7);
            V3 c0 = v3(L + 10), c1 = v3(L + 13), c2 = v3(L + 16), c3 = v3(L + 19);
            const int N = 10;
            for (int i = 0; i < N; ++i) {
                for (int j = 0; j < N; ++j) {
                    float r1 = rand01(c.seed, path, depth, li, (uint32_t)(i * N + j), 0);
                    float r2 = rand01(c.seed, path, depth, li, (uint32_t)(i * N + j), 1);
                    float x = ((float)i + r1) / (float)N;
                    float y = ((float)j + r2) / (float)N;
                    V3 pos = (p0 + (x * e01)) + (y * e02);
                    V3 col = ((((c0 * (1 - x)) * (1 - y)) + ((c1 * x) * (1 - y))) + ((c2 * y) * (1 - x))) + ((c3 * x) * y);
                    res = res + (compute_shading(pos, col, ray, hit.normal, kd, ks, sh) / (float)(N * N)) * visibility(c, pos, ray, hit);
                }
            }
        }
    }
    return res;
}

/* getEnvironmentTexelCoords and the miss branch of getFinalColor, src/render.cpp:40-81,141-147 */
V3 environment_color(Ctx& c, Ray ray)
{
    if (!c.ex || !c.ex->env_texels)
        return { 0, 0, 0 };
    float val;
    if (!slab({ { -64, -64, -64 }, { 64, 64, 64 } }, ray, val))
        return { 0, 0, 0 };
    ray.t = val;
    const V3 p = (ray.o + ray.d * ray.t) / 64.0f;
    const float absX = std::fabs(p.x), absY = std::fabs(p.y), absZ = std::fabs(p.z);
    V2 center = { -1, -1 }, shift = { -1, -1 };
    if (absX >= absY && absX >= absZ) {
        if (p.x > 0) {
            center = { 2.5f / 4, 1.5f / 3 };
            shift = { -p.z / 8, p.y / 6 };
        } else {
            center = { 0.5f / 4, 1.5f / 3 };
            shift = { p.z / 8, p.y / 6 };
        }
    }
    if (absY >= absX && absY >= absZ) {
        if (p.y > 0) {
            center = { 1.5f / 4, 2.5f / 3 };
            shift = { p.x / 8, -p.z / 6 };
        } else {
            center = { 1.5f / 4, 0.5f / 3 };
            shift = { p.x / 8, p.z / 6 };
        }
    }
    if (absZ >= absX && absZ >= absY) {
        if (p.z > 0) {
            center = { 1.5f / 4, 1.5f / 3 };
            shift = { p.x / 8, p.y / 6 };
        } else {
            center = { 3.5f / 4, 1.5f / 3 };
            shift = { -p.x / 8, p.y / 6 };
        }
    }
    V2 ans = { center.x + shift.x, center.y + shift.y };
    if (!(0 <= ans.x && ans.x <= 1 && 0 <= ans.y && ans.y <= 1))
        ans = { 0, 0 };
    return texel(c.ex->env_texels, c.ex->env_w, c.ex->env_h, ans);
}

/* getFinalColor, src/render.cpp:83-150 (debug branches left out) */
V3 final_color(Ctx& c, Ray ray, int depth, uint32_t path, orc_hit* primary)
{
    HitInfo hit;
    c.cnt.path_rays += 1;
    bool h = intersect(c, ray, hit);
    if (primary) {
        primary->hit = h;
        primary->t = ray.t;
        primary->mesh = h ? hit.mesh : -1;
        primary->tri = h ? hit.tri : -1;
        primary->normal[0] = hit.normal.x, primary->normal[1] = hit.normal.y, primary->normal[2] = hit.normal.z;
        primary->bary[0] = hit.bary.x, primary->bary[1] = hit.bary.y, primary->bary[2] = hit.bary.z;
        primary->uv[0] = hit.uv.x, primary->uv[1] = hit.uv.y;
        primary->kd[0] = hit.material.kd.x, primary->kd[1] = hit.material.kd.y, primary->kd[2] = hit.material.kd.z;
    }
    if (!h)
        return (c.features & ORC_F_ENVIRONMENT) ? environment_color(c, ray) : V3 { 0, 0, 0 };
    c.cnt.shaded_hits += 1;
    V3 Lo = light_contribution(c, ray, hit, path, (uint32_t)depth);
    if (c.features & ORC_F_RECURSIVE) {
        Ray refl = reflection_ray(ray, hit.normal);
        V3 ks = hit.material.ks;
        if (!(ks == V3 { 0.0f, 0.0f, 0.0f }) && depth < c.max_depth)
            Lo = Lo + ks * final_color(c, refl, depth + 1, path, nullptr);
    }
    if (c.features & ORC_F_TRANSPARENCY) {
        /* :111-120, :129-130.  Neither branch taken (depth cap reached, or transparency above 1): black. */
        V3 finalColor = { 0, 0, 0 };
        const float tr = hit.material.transparency;
        Ray through = { ray.o + ray.d * (0.000001f + ray.t), ray.d, FLT_MAX };
        if (depth < c.max_depth && tr < 1.0f)
            finalColor = finalColor + (tr * Lo + (1 - tr) * final_color(c, through, depth + 1, path, nullptr));
        else if (depth < c.max_depth && tr == 1.0f)
            finalColor = finalColor + Lo;
        return finalColor;
    }
    return Lo;
}

void add_counters(orc_counters* dst, const orc_counters& src)
{
    if (!dst)
        return;
    dst->path_rays += src.path_rays;
    dst->shadow_rays += src.shadow_rays;
    dst->node_visits += src.node_visits;
    dst->inner_visits += src.inner_visits;
    dst->box_tests += src.box_tests;
    dst->tri_tests += src.tri_tests;
    dst->shaded_hits += src.shaded_hits;
}

Ray camera_ray(const orc_camera& cam, float nx, float ny)
{
    /* Trackball::generateRay, framework/src/trackball.cpp:101-110; quat * vec3 as glm
     * detail/type_quat.inl:347-354: v + ((uv * w) + uuv) * 2 */
    V3 d = normalize(V3 { -nx * cam.half_w, ny * cam.half_h, 1.0f });
    V3 q = { cam.quat[1], cam.quat[2], cam.quat[3] };
    V3 uv = cross(q, d);
    V3 uuv = cross(q, uv);
    Ray r;
    r.o = v3(cam.origin);
    r.d = d + ((uv * cam.quat[0]) + uuv) * 2.0f;
    r.t = FLT_MAX;
    return r;
}

} // namespace

extern "C" {

void orc_camera_make(orc_camera* cam, float fovy_deg, float dist, const float look_at[3], const float rot_deg[3],
                     int32_t width, int32_t height)
{
    /* glm::radians: degrees * 0.01745329251994329576923690768489f (func_trigonometric.inl:9-14) */
    const float k = 0.01745329251994329576923690768489f;
    float fovy = fovy_deg * k;
    float e[3] = { rot_deg[0] * k, rot_deg[1] * k, rot_deg[2] * k };
    /* qua(eulerAngle), glm detail/type_quat.inl:208-217 */
    float cx = std::cos(e[0] * 0.5f), cy = std::cos(e[1] * 0.5f), cz = std::cos(e[2] * 0.5f);
    float sx = std::sin(e[0] * 0.5f), sy = std::sin(e[1] * 0.5f), sz = std::sin(e[2] * 0.5f);
    cam->quat[0] = cx * cy * cz + sx * sy * sz;
    cam->quat[1] = sx * cy * cz - cx * sy * sz;
    cam->quat[2] = cx * sy * cz + sx * cy * sz;
    cam->quat[3] = cx * cy * sz - sx * sy * cz;
    /* half extents: trackball.cpp:26-27; aspect: framework/src/window.cpp:379-384 */
    float aspect = (width == 0 || height == 0) ? 1.0f : (float)width / (float)height;
    cam->half_h = std::tan(fovy / 2.0f);
    cam->half_w = aspect * cam->half_h;
    /* Trackball::position, trackball.cpp:71-74 */
    V3 q = { cam->quat[1], cam->quat[2], cam->quat[3] };
    V3 v = { 0, 0, -dist };
    V3 uv = cross(q, v);
    V3 uuv = cross(q, uv);
    V3 p = v3(look_at) + (v + ((uv * cam->quat[0]) + uuv) * 2.0f);
    cam->origin[0] = p.x, cam->origin[1] = p.y, cam->origin[2] = p.z;
    cam->width = width;
    cam->height = height;
}

void orc_camera_ray(const orc_camera* cam, float ndc_x, float ndc_y, float ray[7])
{
    Ray r = camera_ray(*cam, ndc_x, ndc_y);
    ray[0] = r.o.x, ray[1] = r.o.y, ray[2] = r.o.z, ray[3] = r.d.x, ray[4] = r.d.y, ray[5] = r.d.z, ray[6] = r.t;
}

/* BoundingVolumeHierarchy ctor, src/bounding_volume_hierarchy.cpp:14-35 */
orc_bvh* orc_bvh_build(const orc_scene* scene, uint32_t features)
{
    orc_bvh* b = new orc_bvh();
    b->s = *scene;
    std::vector<TriRef> tris;
    for (uint32_t m = 0; m < scene->n_meshes; ++m) {
        uint32_t nt = scene->mesh_tri_off[m + 1] - scene->mesh_tri_off[m];
        for (uint32_t t = 0; t < nt; ++t)
            tris.push_back({ m, t });
    }
    Builder bd { *b, tris };
    if (features & ORC_F_SAH)
        bd.sah_split(0, tris.size(), 0, 0);
    else
        bd.median_split(0, tris.size(), 0, 0);
    for (const Node& n : b->nodes)
        b->levels = std::max(b->levels, n.level + 1);
    coarse_prepare(*b);
    return b;
}
void orc_bvh_free(orc_bvh* b) { delete b; }
int32_t orc_bvh_num_levels(const orc_bvh* b) { return b->levels; }
int32_t orc_bvh_num_leaves(const orc_bvh* b) { return b->leaves; }
int32_t orc_bvh_num_nodes(const orc_bvh* b) { return (int32_t)b->nodes.size(); }
void orc_bvh_dump(const orc_bvh* b, int32_t* meta4, float* box6)
{
    for (size_t i = 0; i < b->nodes.size(); ++i) {
        const Node& n = b->nodes[i];
        meta4[4 * i + 0] = n.leaf, meta4[4 * i + 1] = n.level, meta4[4 * i + 2] = n.a, meta4[4 * i + 3] = n.b;
        const float bx[6] = { n.box.lo.x, n.box.lo.y, n.box.lo.z, n.box.hi.x, n.box.hi.y, n.box.hi.z };
        std::memcpy(box6 + 6 * i, bx, sizeof bx);
    }
}
int32_t orc_bvh_tri_order(const orc_bvh* b, int32_t* mesh_tri2)
{
    if (mesh_tri2)
        for (size_t i = 0; i < b->order.size(); ++i)
            mesh_tri2[2 * i] = (int32_t)b->order[i].mesh, mesh_tri2[2 * i + 1] = (int32_t)b->order[i].tri;
    return (int32_t)b->order.size();
}

void orc_intersect_batch(const orc_bvh* bvh, float* rays7, int64_t n, uint32_t features, orc_hit* out,
                         orc_counters* counters)
{
    Ctx c { *bvh, features, 5, 0 };
    for (int64_t i = 0; i < n; ++i) {
        float* r = rays7 + 7 * i;
        Ray ray = { v3(r), v3(r + 3), r[6] };
        HitInfo hit;
        bool h = intersect(c, ray, hit);
        r[6] = ray.t;
        orc_hit& o = out[i];
        o.hit = h;
        o.t = ray.t;
        o.mesh = h ? hit.mesh : -1;
        o.tri = h ? hit.tri : -1;
        o.normal[0] = hit.normal.x, o.normal[1] = hit.normal.y, o.normal[2] = hit.normal.z;
        o.bary[0] = hit.bary.x, o.bary[1] = hit.bary.y, o.bary[2] = hit.bary.z;
        o.uv[0] = hit.uv.x, o.uv[1] = hit.uv.y;
        o.kd[0] = hit.material.kd.x, o.kd[1] = hit.material.kd.y, o.kd[2] = hit.material.kd.z;
    }
    add_counters(counters, c.cnt);
}

int32_t orc_ray_triangle(const float v0[3], const float v1[3], const float v2[3], float ray[7])
{
    Ray r = { v3(ray), v3(ray + 3), ray[6] };
    bool h = ray_triangle(v3(v0), v3(v1), v3(v2), r);
    ray[6] = r.t;
    return h;
}
float orc_boxval(const float box[6], const float ray[7])
{
    Ray r = { v3(ray), v3(ray + 3), ray[6] };
    float val;
    slab({ v3(box), v3(box + 3) }, r, val);
    return val;
}
int32_t orc_ray_aabb(const float box[6], float ray[7])
{
    Ray r = { v3(ray), v3(ray + 3), ray[6] };
    float val;
    bool pass = slab({ v3(box), v3(box + 3) }, r, val);
    ray[6] = val;
    return pass;
}
void orc_barycentric(const float v0[3], const float v1[3], const float v2[3], const float p[3], float out[3])
{
    V3 b = barycentric(v3(v0), v3(v1), v3(v2), v3(p));
    out[0] = b.x, out[1] = b.y, out[2] = b.z;
}
void orc_compute_shading(const float light_pos[3], const float light_col[3], const float ray[7], const float normal[3],
                         const float kd[3], const float ks[3], float shininess, float out[3])
{
    Ray r = { v3(ray), v3(ray + 3), ray[6] };
    V3 c = compute_shading(v3(light_pos), v3(light_col), r, v3(normal), v3(kd), v3(ks), shininess);
    out[0] = c.x, out[1] = c.y, out[2] = c.z;
}
void orc_reflection_ray(const float ray[7], const float normal[3], float out[7])
{
    Ray r = { v3(ray), v3(ray + 3), ray[6] };
    Ray o = reflection_ray(r, v3(normal));
    out[0] = o.o.x, out[1] = o.o.y, out[2] = o.o.z, out[3] = o.d.x, out[4] = o.d.y, out[5] = o.d.z, out[6] = o.t;
}
void orc_texel(const float* texels, int32_t w, int32_t h, float u, float v, float out[3])
{
    V3 c = texel(texels, w, h, { u, v });
    out[0] = c.x, out[1] = c.y, out[2] = c.z;
}

void orc_final_color_batch(const orc_bvh* bvh, const float* rays7, int64_t n, uint32_t features, int32_t max_depth,
                           uint32_t seed, float* rgb, orc_counters* counters)
{
    orc_final_color_batch_ex(bvh, rays7, n, features, max_depth, seed, rgb, counters, nullptr);
}

void orc_final_color_batch_ex(const orc_bvh* bvh, const float* rays7, int64_t n, uint32_t features, int32_t max_depth,
                              uint32_t seed, float* rgb, orc_counters* counters, const orc_extras* extras)
{
    Ctx c { *bvh, features, max_depth, seed };
    c.ex = extras;
    for (int64_t i = 0; i < n; ++i) {
        const float* r = rays7 + 7 * i;
        Ray ray = { v3(r), v3(r + 3), r[6] };
        V3 col = final_color(c, ray, 0, (uint32_t)i, nullptr);
        rgb[3 * i] = col.x, rgb[3 * i + 1] = col.y, rgb[3 * i + 2] = col.z;
    }
    add_counters(counters, c.cnt);
}

/* renderRayTracing, src/render.cpp:180-234, with Screen::setPixel (src/screen.cpp:85-91).
 * Multisampling: the reference's per-sample offset dist(mt)/RAND_MAX is below 4.7e-10 and is absorbed
 * by the float additions of :204-205 for every pixel, so the samples sit on the regular grid. */
void orc_render(const orc_bvh* bvh, const orc_camera* cam, uint32_t features, int32_t nsx, int32_t nsy,
                int32_t max_depth, uint32_t seed, const int32_t rect[4], float* rgb, int32_t* prim_mesh,
                int32_t* prim_tri, float* prim_t, int32_t n_threads, orc_counters* counters)
{
    orc_render_ex(bvh, cam, features, nsx, nsy, max_depth, seed, rect, rgb, prim_mesh, prim_tri, prim_t, n_threads, counters, nullptr);
}

void orc_render_ex(const orc_bvh* bvh, const orc_camera* cam, uint32_t features, int32_t nsx, int32_t nsy,
                   int32_t max_depth, uint32_t seed, const int32_t rect[4], float* rgb, int32_t* prim_mesh,
                   int32_t* prim_tri, float* prim_t, int32_t n_threads, orc_counters* counters, const orc_extras* extras)
{
    const int W = cam->width, H = cam->height;
    /* the else-if chain of src/render.cpp:196-227 */
    const int lens_mode = (features & ORC_F_MULTISAMPLE) ? 0 : ((features & ORC_F_MOTION_BLUR) ? 1 : ((features & ORC_F_DEPTH_OF_FIELD) ? 2 : 0));
    const orc_extras none {};
    const orc_extras& ex = extras ? *extras : none;
    const int x0 = rect ? rect[0] : 0, y0 = rect ? rect[1] : 0, x1 = rect ? rect[2] : W, y1 = rect ? rect[3] : H;
    const bool ms = (features & ORC_F_MULTISAMPLE) != 0;
    if (!ms)
        nsx = nsy = 1;
    orc_counters total {};
#ifdef _OPENMP
    if (n_threads <= 0)
        n_threads = omp_get_max_threads();
#pragma omp parallel num_threads(n_threads)
#endif
    {
        Ctx c { *bvh, features, max_depth, seed };
        c.ex = extras;
#ifdef _OPENMP
#pragma omp for schedule(guided)
#endif
        for (int y = y0; y < y1; ++y) {
            for (int x = x0; x < x1; ++x) {
                V3 colour = { 0, 0, 0 };
                orc_hit prim {};
                const uint32_t pix = (uint32_t)y * (uint32_t)W + (uint32_t)x;
                if (ms) {
                    for (int i = 0; i < nsx; ++i) {
                        for (int j = 0; j < nsy; ++j) {
                            float a = ((float)i + 0.0f) / (float)nsx + (float)x;
                            float b = ((float)j + 0.0f) / (float)nsy + (float)y;
                            float nx = a / (float)W * 2.0f - 1.0f;
                            float ny = b / (float)H * 2.0f - 1.0f;
                            uint32_t sidx = (uint32_t)(i * nsy + j);
                            V3 col = final_color(c, camera_ray(*cam, nx, ny), 0, pix * (uint32_t)(nsx * nsy) + sidx,
                                sidx == 0 ? &prim : nullptr);
                            colour = colour + col;
                        }
                    }
                    colour = colour / (float)(nsx * nsy);
                } else if (lens_mode) {
                    /* motionBlur / DOF, src/render.cpp:152-178: one camera ray through the pixel, `samples` jittered copies */
                    const float nx = (float)x / (float)W * 2.0f - 1.0f;
                    const float ny = (float)y / (float)H * 2.0f - 1.0f;
                    const Ray cameraRay = camera_ray(*cam, nx, ny);
                    const int N = ex.samples;
                    for (int t = 0; t < N; ++t) {
                        const uint32_t key = pix * (uint32_t)N + (uint32_t)t;
                        Ray ray = cameraRay;
                        if (lens_mode == 1) {
                            const float random = rand01(seed, key, 0xFFFFFFFFu, 0, 0, 0);
                            const float k = (ex.time0 + random * (ex.time1 - ex.time0)) - ((ex.time1 - ex.time0) / 2);
                            ray.o = cameraRay.o + normalize(v3(ex.direction)) * k;
                        } else {
                            const V3 converge = cameraRay.o + cameraRay.d * ex.focal_length;
                            const float r1 = -1.0f + 2.0f * rand01(seed, key, 0xFFFFFFFFu, 0, 0, 0);
                            const float r2 = -1.0f + 2.0f * rand01(seed, key, 0xFFFFFFFFu, 0, 0, 1);
                            const float r3 = -1.0f + 2.0f * rand01(seed, key, 0xFFFFFFFFu, 0, 0, 2);
                            ray.o = cameraRay.o + V3 { r1 * ex.aperture, r2 * ex.aperture, r3 * ex.aperture };
                            ray.d = normalize(converge - ray.o);
                        }
                        colour = colour + final_color(c, ray, 0, key, t == 0 ? &prim : nullptr) / (float)N;
                    }
                    if (N == 0)
                        prim.mesh = prim.tri = -1;
                } else {
                    float nx = (float)x / (float)W * 2.0f - 1.0f;
                    float ny = (float)y / (float)H * 2.0f - 1.0f;
                    colour = final_color(c, camera_ray(*cam, nx, ny), 0, pix, &prim);
                }
                size_t idx = (size_t)(H - 1 - y) * W + x;
                rgb[3 * idx] = colour.x, rgb[3 * idx + 1] = colour.y, rgb[3 * idx + 2] = colour.z;
                if (prim_mesh)
                    prim_mesh[idx] = prim.mesh;
                if (prim_tri)
                    prim_tri[idx] = prim.tri;
                if (prim_t)
                    prim_t[idx] = prim.t;
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        add_counters(&total, c.cnt);
    }
    add_counters(counters, total);
}

float orc_rand01(uint32_t seed, uint32_t path, uint32_t depth, uint32_t light, uint32_t sample, uint32_t dim)
{
    return rand01(seed, path, depth, light, sample, dim);
}

/* reference src/screen.cpp:32-78, in its own terms: a [width][height] table filled in storage order, thresholded
 * (:45-55), box-filtered in place over the half-open window with the full box as divisor (:58-72) and added to the
 * frame in the same pass (:74). */
void orc_bloom(float* rgb, int32_t width, int32_t height, float threshold, int32_t box_size)
{
    std::vector<std::vector<std::array<float, 3>>> table((size_t)width, std::vector<std::array<float, 3>>((size_t)height));
    size_t i = 0;
    for (size_t w = 0; w < (size_t)width; w++)
        for (size_t h = 0; h < (size_t)height; h++, i++) {
            const float* px = rgb + 3 * i;
            const float m = std::max(px[0], std::max(px[1], px[2]));
            if (m > threshold)
                table[w][h] = { px[0], px[1], px[2] };
            else
                table[w][h] = { 0.0f, 0.0f, 0.0f };
        }
    i = 0;
    const int half = box_size / 2;
    for (size_t w = 0; w < (size_t)width; w++)
        for (size_t h = 0; h < (size_t)height; h++, i++) {
            if (w >= (size_t)half && w < (size_t)(width - half) && h >= (size_t)half && h < (size_t)(height - half)) {
                float sum[3] = { 0.0f, 0.0f, 0.0f };
                for (size_t a = w - half; a < w + half; a++)
                    for (size_t b = h - half; b < h + half; b++)
                        for (int c = 0; c < 3; c++)
                            sum[c] += table[a][b][c];
                for (int c = 0; c < 3; c++)
                    table[w][h][c] = sum[c] / float(box_size * box_size);
            }
            for (int c = 0; c < 3; c++)
                rgb[3 * i + c] += table[w][h][c];
        }
}

} // extern "C"
