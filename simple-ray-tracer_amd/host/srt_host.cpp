/*
 * srt_host.cpp -- host mirror of the reference's BVH / Screen / camera / render interface (see srt_host.h).
 * Compiled with -ffp-contract=off: the BVH boxes, the per-triangle precomputation and the camera constants
 * must carry exactly the bits the reference's x86-64 build produces.
 */
#include "srt_host.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <atomic>
#include <exception>
#include <mutex>
#include <queue>
#include <thread>

namespace srt {

namespace {

inline vec3 operator+(vec3 a, vec3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline vec3 operator-(vec3 a, vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline vec3 operator*(vec3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline vec3 operator/(vec3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 x, vec3 y) { return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
inline float length(vec3 v) { return std::sqrt(dot(v, v)); }
inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }

thread_local int t_gpu = 0;
std::mutex g_mu;
std::map<int, srt_device*> g_devices;

[[noreturn]] void raise(const std::string& what) { throw std::runtime_error(what + ": " + srt_last_error()); }

} // namespace

/* device key = gpu + 1024 * context */
srt_device* deviceFor(int key)
{
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_devices.find(key);
    if (it != g_devices.end())
        return it->second;
    srt_device* d = nullptr;
    if (srt_device_create(key & 1023, &d) != SRT_OK)
        raise("MI355X back end unavailable");
    g_devices[key] = d;
    return d;
}

void setDevice(int gpuId, int context) { t_gpu = (gpuId & 1023) + 1024 * context; }
int currentDevice() { return t_gpu; }

float radians(float degrees) { return degrees * 0.01745329251994329576923690768489f; }
vec3 radians(const vec3& d) { return { radians(d.x), radians(d.y), radians(d.z) }; }

uint32_t featureMask(const Features& f, bool referenceTraversal, bool exactShadowBoxes)
{
    uint32_t m = 0;
    if (f.enableShading) m |= SRT_F_SHADING;
    if (f.enableRecursive) m |= SRT_F_RECURSIVE;
    if (f.enableHardShadow) m |= SRT_F_HARD_SHADOW;
    if (f.enableSoftShadow) m |= SRT_F_SOFT_SHADOW;
    if (f.enableNormalInterp) m |= SRT_F_NORMAL_INTERP;
    if (f.enableTextureMapping) m |= SRT_F_TEXTURE;
    if (f.enableAccelStructure) m |= SRT_F_ACCEL;
    if (f.extra.enableMultipleRaysPerPixel) m |= SRT_F_MULTISAMPLE;
    if (f.extra.enableTransparency) m |= SRT_F_TRANSPARENCY;
    if (f.extra.enableMotionBlur) m |= SRT_F_MOTION_BLUR;
    if (f.extra.enableDepthOfField) m |= SRT_F_DEPTH_OF_FIELD;
    if (f.extra.enableEnvironmentMapping) m |= SRT_F_ENVIRONMENT;
    if (!referenceTraversal) m |= SRT_F_MISS_CULL | (exactShadowBoxes ? 0u : SRT_F_COARSE_SHADOW);
    return m;
}

/* ------------------------------------------------------------------------------------------ BVH build */

/* getMedian, reference src/bounding_volume_hierarchy.cpp:37-45 */
static vec3 getMedian(const TriangleOrNode& t, const Scene& scene)
{
    const Mesh& mesh = scene.meshes[t.meshIndex];
    const uvec3& tr = mesh.triangles[t.triangleIndex];
    const vec3 &p1 = mesh.vertices[tr.x].position, &p2 = mesh.vertices[tr.y].position, &p3 = mesh.vertices[tr.z].position;
    return ((p1 + p2) + p3) / 3.0f;
}

/* getBox, reference :47-68, over triangles[begin, end) with the corners taken from the per-triangle table (the
 * same values folded in the same order) */
AxisAlignedBox BoundingVolumeHierarchy::getBox(const std::vector<TriangleOrNode>& triangles, size_t begin, size_t end) const
{
    vec3 lower = { FLT_MAX, FLT_MAX, FLT_MAX };
    vec3 upper = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (size_t k = begin; k < end; ++k) {
        const vec3* c = &m_corners[3 * globalIndex(triangles[k])];
        const vec3 &v1 = c[0], &v2 = c[1], &v3 = c[2];
        upper.x = std::max(std::max(upper.x, v1.x), std::max(v2.x, v3.x));
        upper.y = std::max(std::max(upper.y, v1.y), std::max(v2.y, v3.y));
        upper.z = std::max(std::max(upper.z, v1.z), std::max(v2.z, v3.z));
        lower.x = std::min(std::min(lower.x, v1.x), std::min(v2.x, v3.x));
        lower.y = std::min(std::min(lower.y, v1.y), std::min(v2.y, v3.y));
        lower.z = std::min(std::min(lower.z, v1.z), std::min(v2.z, v3.z));
    }
    return { lower, upper };
}

BoundingVolumeHierarchy::BoundingVolumeHierarchy(Scene* pScene, const Features& features)
    : m_pScene(pScene)
{
    std::vector<TriangleOrNode> triangles;
    for (size_t i = 0; i < pScene->meshes.size(); ++i) {
        m_meshOffset.push_back(triangles.size());
        for (size_t j = 0; j < pScene->meshes[i].triangles.size(); ++j)
            triangles.push_back(TriangleOrNode { i, j, 0 });
    }
    m_centroid.resize(triangles.size());
    m_corners.resize(3 * triangles.size());
    for (size_t k = 0; k < triangles.size(); ++k) {
        const Mesh& mesh = pScene->meshes[triangles[k].meshIndex];
        const uvec3& tr = mesh.triangles[triangles[k].triangleIndex];
        m_centroid[k] = getMedian(triangles[k], *pScene);
        m_corners[3 * k] = mesh.vertices[tr.x].position, m_corners[3 * k + 1] = mesh.vertices[tr.y].position, m_corners[3 * k + 2] = mesh.vertices[tr.z].position;
    }
    /* Subtrees are independent once their triangle range is fixed, so the top levels fork; every node is built by
     * the reference's own steps and numbered as its recursion numbers it, whatever the number of threads. */
    const int forkLevels = triangles.size() >= 4096 ? 5 : 0;
#pragma omp parallel if (forkLevels > 0)
#pragma omp single
    {
        if (features.extra.enableBvhSahBinning)
            sahConstructorHelper(nodes, triangles, 0, triangles.size(), 0, 0, forkLevels);
        else
            constructorHelper(nodes, triangles, 0, triangles.size(), 0, 0, forkLevels);
    }
    for (const Node& node : nodes) {
        m_numLevels = std::max(m_numLevels, node.level + 1);
        m_numLeaves += node.isLeaf ? 1 : 0;
    }
    m_centroid = {};
    m_corners = {};
}

/* std::sort, not std::stable_sort: equal centroids end up in libstdc++ introsort order, as in the reference (:81-85).
 * The sort runs on (centroid coordinate, position) pairs: introsort's moves depend only on the outcomes of the
 * comparisons, which are those of the reference's comparator, so the permutation is the one sorting the
 * triangle list itself produces. */
void BoundingVolumeHierarchy::sortByCentroid(std::vector<TriangleOrNode>& triangles, size_t left, size_t right, int whichAxis) const
{
    struct Key {
        float value;
        uint32_t from;
    };
    std::vector<Key> keys(right - left);
    for (size_t k = 0; k < keys.size(); ++k)
        keys[k] = { m_centroid[globalIndex(triangles[left + k])][whichAxis], (uint32_t)k };
    std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.value < b.value; });
    std::vector<TriangleOrNode> sorted(keys.size());
    for (size_t k = 0; k < keys.size(); ++k)
        sorted[k] = triangles[left + keys[k].from];
    std::copy(sorted.begin(), sorted.end(), triangles.begin() + left);
}

/* Appends two finished subtrees and their parent to out; child indices become positions in out. */
size_t BoundingVolumeHierarchy::splice(std::vector<Node>& out, std::vector<Node>& left, std::vector<Node>& right, Node&& parent) const
{
    for (std::vector<Node>* part : { &left, &right }) {
        const size_t base = out.size();
        for (Node& n : *part) {
            if (!n.isLeaf)
                n.triangles[0].nodeIndex += base, n.triangles[1].nodeIndex += base;
            out.push_back(std::move(n));
        }
    }
    parent.triangles[0].nodeIndex = out.size() - right.size() - 1;
    parent.triangles[1].nodeIndex = out.size() - 1;
    out.push_back(std::move(parent));
    return out.size() - 1;
}

/* constructorHelper, reference :70-94: median split on the centroid, axis cycling with depth, one-triangle
 * leaves, depth cap 17 levels, children appended before their parent */
size_t BoundingVolumeHierarchy::constructorHelper(std::vector<Node>& out, std::vector<TriangleOrNode>& triangles, size_t left, size_t right, int whichAxis,
                                                  int level, int forkLevels)
{
    const auto beginIt = triangles.begin() + left, endIt = triangles.begin() + right;
    if (right - left <= 1 || level > 16) {
        out.push_back({ true, level, std::vector<TriangleOrNode>(beginIt, endIt), getBox(triangles, left, right) });
        return out.size() - 1;
    }
    sortByCentroid(triangles, left, right, whichAxis);
    const size_t median = (left + right + 1) / 2;
    if (forkLevels > 0) {
        std::vector<Node> l, r;
#pragma omp task shared(l, triangles)
        constructorHelper(l, triangles, left, median, (whichAxis + 1) % 3, level + 1, forkLevels - 1);
#pragma omp task shared(r, triangles)
        constructorHelper(r, triangles, median, right, (whichAxis + 1) % 3, level + 1, forkLevels - 1);
#pragma omp taskwait
        /* after the children, as in the reference: they re-sort their halves of the range in place, and reading
         * the range while they do would be a race */
        const AxisAlignedBox box = getBox(triangles, left, right);
        return splice(out, l, r, { false, level, { { 0, 0, 0 }, { 0, 0, 0 } }, box });
    }
    const size_t leftIndex = constructorHelper(out, triangles, left, median, (whichAxis + 1) % 3, level + 1, 0);
    const size_t rightIndex = constructorHelper(out, triangles, median, right, (whichAxis + 1) % 3, level + 1, 0);
    out.push_back({ false, level, { { 0, 0, leftIndex }, { 0, 0, rightIndex } }, getBox(triangles, left, right) });
    return out.size() - 1;
}

/* getProb, reference :96-113 */
static float getProb(const AxisAlignedBox& inner, const AxisAlignedBox& outer)
{
    float ia = std::abs(inner.lower.x - inner.upper.x), ib = std::abs(inner.lower.y - inner.upper.y), ic = std::abs(inner.lower.z - inner.upper.z);
    float oa = std::abs(outer.lower.x - outer.upper.x), ob = std::abs(outer.lower.y - outer.upper.y), oc = std::abs(outer.lower.z - outer.upper.z);
    float areaInner = 2 * (ia * ib + ib * ic + ic * ia);
    float areaOuter = 2 * (oa * ob + ob * oc + oc * oa);
    if (areaOuter == 0)
        return 0;
    return areaInner / areaOuter;
}

/* sahConstructorHelper with calculateCostOfDivision, reference :116-178 (five candidate planes) */
size_t BoundingVolumeHierarchy::sahConstructorHelper(std::vector<Node>& out, std::vector<TriangleOrNode>& triangles, size_t left, size_t right,
                                                     int whichAxis, int level, int forkLevels)
{
    const auto beginIt = triangles.begin() + left, endIt = triangles.begin() + right;
    if (right - left <= 1 || level > 16) {
        out.push_back({ true, level, std::vector<TriangleOrNode>(beginIt, endIt), getBox(triangles, left, right) });
        return out.size() - 1;
    }
    sortByCentroid(triangles, left, right, whichAxis);
    const size_t planesNumber = std::min(static_cast<size_t>(5), right - left - 1);
    const float leftBoundary = m_centroid[globalIndex(triangles[left])][whichAxis];
    const float rightBoundary = m_centroid[globalIndex(triangles[right - 1])][whichAxis];
    const float step = (rightBoundary - leftBoundary) / (planesNumber + 1);
    float minCost = FLT_MAX;
    size_t median = left;
    const AxisAlignedBox outerBox = getBox(triangles, left, right);
    for (size_t i = 1; i <= planesNumber; ++i) {
        const float boundary = leftBoundary + step * i;
        size_t middle = left;
        while (middle != right && m_centroid[globalIndex(triangles[middle])][whichAxis] < boundary)
            ++middle;
        const AxisAlignedBox leftBox = getBox(triangles, left, middle), rightBox = getBox(triangles, middle, right);
        const float cost = getProb(leftBox, outerBox) * (middle - left) + getProb(rightBox, outerBox) * (right - middle);
        if (cost < minCost) {
            minCost = cost;
            median = middle;
        }
    }
    if (forkLevels > 0) {
        std::vector<Node> l, r;
#pragma omp task shared(l, triangles)
        sahConstructorHelper(l, triangles, left, median, (whichAxis + 1) % 3, level + 1, forkLevels - 1);
#pragma omp task shared(r, triangles)
        sahConstructorHelper(r, triangles, median, right, (whichAxis + 1) % 3, level + 1, forkLevels - 1);
#pragma omp taskwait
        return splice(out, l, r, { false, level, { { 0, 0, 0 }, { 0, 0, 0 } }, outerBox });
    }
    const size_t leftIndex = sahConstructorHelper(out, triangles, left, median, (whichAxis + 1) % 3, level + 1, 0);
    const size_t rightIndex = sahConstructorHelper(out, triangles, median, right, (whichAxis + 1) % 3, level + 1, 0);
    out.push_back({ false, level, { { 0, 0, leftIndex }, { 0, 0, rightIndex } }, outerBox });
    return out.size() - 1;
}

/* ------------------------------------------------------------------------------------------ flatten */

srt_scene_desc FlatScene::desc() const
{
    srt_scene_desc d {};
    d.nodes = nodes.data();
    d.n_nodes = (uint32_t)nodes.size();
    std::memcpy(d.root_box, rootBox, sizeof rootBox);
    d.root_ref = rootRef;
    d.num_levels = numLevels;
    d.tris = tris.data();
    d.attrs = attrs.data();
    d.n_tris = (uint32_t)tris.size();
    d.scene_order = sceneOrder.data();
    d.materials = materials.data();
    d.n_materials = (uint32_t)materials.size();
    d.textures = textures.data();
    d.n_textures = (uint32_t)textures.size();
    d.texels = texels.data();
    d.n_texels = texels.size() / 3;
    d.lights = lights.data();
    d.n_lights = (uint32_t)lights.size();
    d.environment = environment;
    return d;
}

static void put3(float* dst, vec3 v) { dst[0] = v.x, dst[1] = v.y, dst[2] = v.z; }
static void putBox(float* dst, const AxisAlignedBox& b)
{
    put3(dst, b.lower);
    put3(dst + 3, b.upper);
}

/* materials (one per mesh), distinct textures (the environment image among them), lights */
static void packAppearance(const Scene& scene, FlatScene& f)
{
    std::vector<const Image*> seen;
    for (const Mesh& mesh : scene.meshes) {
        srt_material m {};
        put3(m.kd, mesh.material.kd);
        put3(m.ks, mesh.material.ks);
        m.shininess = mesh.material.shininess;
        m.transparency = mesh.material.transparency;
        m.texture = -1;
        if (const Image* im = mesh.material.kdTexture.get()) {
            auto it = std::find(seen.begin(), seen.end(), im);
            if (it == seen.end()) {
                srt_texture t { (uint32_t)im->width, (uint32_t)im->height, (uint64_t)(f.texels.size() / 3) };
                f.textures.push_back(t);
                for (const vec3& p : im->pixels)
                    f.texels.push_back(p.x), f.texels.push_back(p.y), f.texels.push_back(p.z);
                seen.push_back(im);
                m.texture = (int32_t)seen.size() - 1;
            } else {
                m.texture = (int32_t)(it - seen.begin());
            }
        }
        f.materials.push_back(m);
    }
    if (const Image* im = scene.environment.get()) {
        auto it = std::find(seen.begin(), seen.end(), im);
        if (it == seen.end()) {
            f.textures.push_back({ (uint32_t)im->width, (uint32_t)im->height, (uint64_t)(f.texels.size() / 3) });
            f.texels.reserve(f.texels.size() + 3 * im->pixels.size());
            for (const vec3& p : im->pixels)
                f.texels.push_back(p.x), f.texels.push_back(p.y), f.texels.push_back(p.z);
            seen.push_back(im);
            f.environment = (uint32_t)seen.size();
        } else {
            f.environment = (uint32_t)(it - seen.begin()) + 1;
        }
    }
    for (const Light& l : scene.lights) {
        srt_light o {};
        if (std::holds_alternative<PointLight>(l)) {
            const PointLight& p = std::get<PointLight>(l);
            o.type = 0;
            put3(o.f, p.position), put3(o.f + 3, p.color);
        } else if (std::holds_alternative<SegmentLight>(l)) {
            const SegmentLight& s = std::get<SegmentLight>(l);
            o.type = 1;
            put3(o.f, s.endpoint0), put3(o.f + 3, s.endpoint1), put3(o.f + 6, s.color0), put3(o.f + 9, s.color1);
        } else {
            const ParallelogramLight& p = std::get<ParallelogramLight>(l);
            o.type = 2;
            put3(o.f, p.v0), put3(o.f + 3, p.edge01), put3(o.f + 6, p.edge02);
            put3(o.f + 9, p.color0), put3(o.f + 12, p.color1), put3(o.f + 15, p.color2), put3(o.f + 18, p.color3);
        }
        f.lights.push_back(o);
    }
}

FlatScene BoundingVolumeHierarchy::flatten(int bfsLevels) const
{
    const Scene& scene = *m_pScene;
    FlatScene f;
    f.numLevels = (uint32_t)m_numLevels;

    /* leaf-ordered triangle streams: leaves in the order the builder created them (left to right) */
    std::vector<uint32_t> leafStart(nodes.size(), 0);
    std::vector<size_t> meshBase(scene.meshes.size() + 1, 0);
    for (size_t m = 0; m < scene.meshes.size(); ++m)
        meshBase[m + 1] = meshBase[m] + scene.meshes[m].triangles.size();
    f.sceneOrder.assign(meshBase.back(), 0);
    for (size_t ni = 0; ni < nodes.size(); ++ni) {
        const Node& n = nodes[ni];
        if (!n.isLeaf)
            continue;
        if (n.triangles.size() > SRT_MAX_LEAF_TRIS)
            throw std::runtime_error("BVH leaf holds more than 127 triangles: outside the back end's packed child reference");
        leafStart[ni] = (uint32_t)f.tris.size();
        for (const TriangleOrNode& t : n.triangles) {
            const Mesh& mesh = scene.meshes[t.meshIndex];
            const uvec3& tr = mesh.triangles[t.triangleIndex];
            const Vertex &a = mesh.vertices[tr.x], &b = mesh.vertices[tr.y], &c = mesh.vertices[tr.z];
            const vec3 v0 = a.position, v1 = b.position, v2 = c.position;
            srt_tri rec {};
            /* trianglePlane, reference src/intersect.cpp:45-59 */
            const vec3 cr = cross(v1 - v0, v2 - v0);
            vec3 n3 { 0, 0, 0 };
            float D = 0;
            if (!(length(cr) == 0.0f)) {
                n3 = normalize(cr);
                D = dot(v0, n3);
            }
            put3(rec.n, n3);
            rec.D = D;
            put3(rec.v0, v0), put3(rec.v1, v1), put3(rec.v2, v2);
            rec.mesh = (uint32_t)t.meshIndex;
            rec.tri = (uint32_t)t.triangleIndex;
            /* second operands of the three sameSide tests, reference src/intersect.cpp:13-26 */
            put3(rec.c0, cross(v2 - v1, v0 - v1));
            put3(rec.c1, cross(v2 - v0, v1 - v0));
            put3(rec.c2, cross(v1 - v0, v2 - v0));
            srt_tri_attr at {};
            put3(at.n0, a.normal), put3(at.n1, b.normal), put3(at.n2, c.normal);
            at.uv0[0] = a.texCoord.x, at.uv0[1] = a.texCoord.y;
            at.uv1[0] = b.texCoord.x, at.uv1[1] = b.texCoord.y;
            at.uv2[0] = c.texCoord.x, at.uv2[1] = c.texCoord.y;
            at.mesh = (uint32_t)t.meshIndex;
            f.sceneOrder[meshBase[t.meshIndex] + t.triangleIndex] = (uint32_t)f.tris.size();
            f.tris.push_back(rec);
            f.attrs.push_back(at);
        }
    }
    if (f.tris.size() > SRT_MAX_TRIS)
        throw std::runtime_error("scene has more than 2^24 triangles: outside the back end's packed child reference");

    auto childRef = [&](size_t ni, const std::vector<int>& newId) -> uint32_t {
        const Node& n = nodes[ni];
        if (n.isLeaf)
            return SRT_LEAF_FLAG | ((uint32_t)n.triangles.size() << 24) | leafStart[ni];
        return (uint32_t)newId[ni];
    };

    /* inner-node numbering: top bfsLevels levels breadth first, everything below depth first with the
     * right child (visited first by the traversal) directly after its parent */
    std::vector<int> newId(nodes.size(), -1);
    std::vector<size_t> orderList;
    if (!nodes.empty()) {
        const size_t root = nodes.size() - 1;
        putBox(f.rootBox, nodes[root].box);
        if (!nodes[root].isLeaf) {
            std::vector<size_t> frontier { root }, dfsRoots;
            while (!frontier.empty()) {
                std::vector<size_t> next;
                for (size_t ni : frontier) {
                    if (nodes[ni].level >= bfsLevels) {
                        dfsRoots.push_back(ni);
                        continue;
                    }
                    newId[ni] = (int)orderList.size();
                    orderList.push_back(ni);
                    for (const TriangleOrNode& ch : nodes[ni].triangles)
                        if (!nodes[ch.nodeIndex].isLeaf)
                            next.push_back(ch.nodeIndex);
                }
                frontier.swap(next);
            }
            for (size_t r : dfsRoots) {
                std::vector<size_t> stack { r };
                while (!stack.empty()) {
                    const size_t ni = stack.back();
                    stack.pop_back();
                    newId[ni] = (int)orderList.size();
                    orderList.push_back(ni);
                    const size_t l = nodes[ni].triangles[0].nodeIndex, rr = nodes[ni].triangles[1].nodeIndex;
                    if (!nodes[l].isLeaf)
                        stack.push_back(l);
                    if (!nodes[rr].isLeaf)
                        stack.push_back(rr); /* popped first */
                }
            }
        }
        f.rootRef = childRef(root, newId);
    }
    f.nodes.resize(orderList.size());
    for (size_t k = 0; k < orderList.size(); ++k) {
        const Node& n = nodes[orderList[k]];
        const size_t l = n.triangles[0].nodeIndex, r = n.triangles[1].nodeIndex;
        srt_node& o = f.nodes[k];
        putBox(o.lbox, nodes[l].box);
        putBox(o.rbox, nodes[r].box);
        o.left = childRef(l, newId);
        o.right = childRef(r, newId);
        o.left_level = (uint32_t)nodes[l].level;
        o.pad = 0;
    }

    packAppearance(scene, f);
    return f;
}

/* ------------------------------------------------------------------------------------------ BvhInterface */

BvhInterface::BvhInterface(Scene* pScene, const Features& features)
    : m_pScene(pScene)
    , m_features(features)
    , m_gpu(currentDevice())
{
    const char* hostBuild = std::getenv("SRT_HOST_BUILD");
    if (features.extra.enableBvhSahBinning || (hostBuild && hostBuild[0] == '1'))
        m_impl = new BoundingVolumeHierarchy(pScene, features);
    try {
        m_dev = buildOn(m_gpu);
    } catch (...) {
        delete m_impl;
        throw;
    }
}

/* One copy of the scene on the GPU behind deviceKey. */
srt_scene* BvhInterface::buildOn(int deviceKey)
{
    srt_device* dev = deviceFor(deviceKey);
    srt_scene* out = nullptr;
    if (m_impl) {
        /* the SAH-binned tree's shape depends on the geometry: built and flattened on the host, then uploaded.
         * Depth-first node order; SRT_EXPERIMENT_BFS_LEVELS=k stores the top k levels breadth first instead (layout
         * experiments: the measured difference is within 1 %). */
        const char* lv = std::getenv("SRT_EXPERIMENT_BFS_LEVELS");
        const FlatScene flat = m_impl->flatten(lv ? std::atoi(lv) : 0);
        const srt_scene_desc d = flat.desc();
        if (srt_scene_upload(dev, &d, &out) != SRT_OK)
            raise("scene upload failed");
        return out;
    }
    /* the reference's median-split tree: the back end builds it on the device from the triangle list */
    const Scene& scene = *m_pScene;
    FlatScene look;
    packAppearance(scene, look);
    std::vector<float> verts;
    std::vector<uint32_t> tris, triMesh, meshTriOff { 0 };
    size_t nVerts = 0, nTris = 0;
    for (const Mesh& mesh : scene.meshes)
        nVerts += mesh.vertices.size(), nTris += mesh.triangles.size();
    verts.reserve(8 * nVerts), tris.reserve(3 * nTris), triMesh.reserve(nTris);
    static_assert(sizeof(Vertex) == 32, "Vertex is 8 packed floats");
    uint32_t base = 0;
    for (size_t m = 0; m < scene.meshes.size(); ++m) {
        const Mesh& mesh = scene.meshes[m];
        const float* v = reinterpret_cast<const float*>(mesh.vertices.data());
        verts.insert(verts.end(), v, v + 8 * mesh.vertices.size());
        for (const uvec3& t : mesh.triangles) {
            tris.push_back(base + t.x), tris.push_back(base + t.y), tris.push_back(base + t.z);
            triMesh.push_back((uint32_t)m);
        }
        base += (uint32_t)mesh.vertices.size();
        meshTriOff.push_back((uint32_t)triMesh.size());
    }
    srt_build_desc b {};
    b.verts = verts.data(), b.n_verts = (uint32_t)nVerts;
    b.tris = tris.data(), b.tri_mesh = triMesh.data(), b.n_tris = (uint32_t)nTris;
    b.mesh_tri_off = meshTriOff.data(), b.n_meshes = (uint32_t)scene.meshes.size();
    b.materials = look.materials.data(), b.n_materials = (uint32_t)look.materials.size();
    b.textures = look.textures.data(), b.n_textures = (uint32_t)look.textures.size();
    b.texels = look.texels.data(), b.n_texels = look.texels.size() / 3;
    b.lights = look.lights.data(), b.n_lights = (uint32_t)look.lights.size();
    b.environment = look.environment;
    if (srt_scene_build(dev, &b, &out) != SRT_OK)
        raise("scene build failed");
    return out;
}

/* Replicas live behind back-end handles of their own (contexts from 512 upwards, away from the ones setDevice hands
 * out), so that their host threads never share a handle. */
void BvhInterface::addReplica(int gpuId)
{
    const int key = (gpuId & 1023) + 1024 * (512 + (int)m_replicas.size());
    m_replicas.push_back({ key, buildOn(key) });
}

srt_device* BvhInterface::renderDevice(size_t k) const { return deviceFor(k == 0 ? m_gpu : m_replicas[k - 1].deviceKey); }

BvhInterface::~BvhInterface()
{
    for (const Replica& r : m_replicas)
        srt_scene_destroy(r.scene);
    srt_scene_destroy(m_dev);
    delete m_impl;
}

int BvhInterface::numLevels() const
{
    uint32_t levels = 0;
    srt_scene_info(m_dev, nullptr, nullptr, &levels, nullptr, nullptr, nullptr);
    return (int)levels;
}
int BvhInterface::numLeaves() const
{
    uint32_t leaves = 0;
    srt_scene_info(m_dev, nullptr, nullptr, nullptr, &leaves, nullptr, nullptr);
    return (int)leaves;
}
srt_device* BvhInterface::device() const { return deviceFor(m_gpu); }

void BvhInterface::intersectBatch(float* rays7, int64_t n, const Features& features, srt_hit* hits, srt_counters* counters, bool referenceTraversal) const
{
    if (srt_intersect_batch(device(), m_dev, rays7, n, featureMask(features, referenceTraversal), hits, counters) != SRT_OK)
        raise("BvhInterface::intersect");
}

bool BvhInterface::intersect(Ray& ray, HitInfo& hitInfo, const Features& features) const
{
    float r[7] = { ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x, ray.direction.y, ray.direction.z, ray.t };
    srt_hit h {};
    intersectBatch(r, 1, features, &h);
    ray.t = r[6];
    if (!h.hit)
        return false;
    hitInfo.normal = { h.normal[0], h.normal[1], h.normal[2] };
    if (features.enableNormalInterp)
        hitInfo.barycentricCoord = { h.bary[0], h.bary[1], h.bary[2] };
    if (features.enableTextureMapping)
        hitInfo.texCoord = { h.uv[0], h.uv[1] };
    hitInfo.material = m_pScene->meshes[(size_t)h.mesh].material;
    hitInfo.material.kd = { h.kd[0], h.kd[1], h.kd[2] };
    return true;
}

/* ------------------------------------------------------------------------------------------ Screen */

Screen::Screen(const ivec2& resolution, bool)
    : m_resolution(resolution)
    , m_textureData(size_t(resolution.x) * size_t(resolution.y), vec3(0.0f))
{
}
void Screen::clear(const vec3& color) { std::fill(m_textureData.begin(), m_textureData.end(), color); }
void Screen::setPixel(int x, int y, const vec3& color) { m_textureData[(size_t)indexAt(x, y)] = color; }

/* applyBloomFilter, reference src/screen.cpp:32-78, with everything that makes its result what it is:
 *  - the pixel array is walked in storage order but addressed as table[w][h] with w = i / height, h = i % height,
 *    i.e. as a width x height array laid over the row-major height x width image;
 *  - the window of a pixel is [w - half, w + half) x [h - half, h + half) (upper bounds excluded, (2 half)^2 values)
 *    while the divisor is boxSize^2;
 *  - the table is filtered in place, so window entries before the pixel (in walking order) are already filtered;
 *  - pixels nearer than `half` to the table's border keep their thresholded value;
 *  - the comparisons mix size_t and int as the reference does (a negative bound wraps to a huge one). */
void Screen::applyBloomFilter(const float threshold, const int boxSize)
{
    const int W = m_resolution.x, H = m_resolution.y;
    std::vector<vec3> bright(m_textureData.size());
    for (size_t i = 0; i < bright.size(); ++i) {
        const vec3& c = m_textureData[i];
        const float maxColour = std::max(c.x, std::max(c.y, c.z));
        bright[i] = maxColour > threshold ? c : vec3(0.0f);
    }
    const int half = boxSize / 2;
    const float divisor = float(boxSize * boxSize);
    size_t i = 0;
    for (size_t w = 0; w < (size_t)W; ++w) {
        for (size_t h = 0; h < (size_t)H; ++h, ++i) {
            if (w >= (size_t)half && w < (size_t)(W - half) && h >= (size_t)half && h < (size_t)(H - half)) {
                vec3 sum(0.0f);
                for (size_t a = w - (size_t)half; a < w + (size_t)half; ++a)
                    for (size_t b = h - (size_t)half; b < h + (size_t)half; ++b) {
                        const vec3& t = bright[a * (size_t)H + b];
                        sum.x += t.x, sum.y += t.y, sum.z += t.z;
                    }
                bright[i] = vec3(sum.x / divisor, sum.y / divisor, sum.z / divisor);
            }
            m_textureData[i].x += bright[i].x, m_textureData[i].y += bright[i].y, m_textureData[i].z += bright[i].z;
        }
    }
}
int Screen::indexAt(int x, int y) const { return (m_resolution.y - 1 - y) * m_resolution.x + x; }

/* writeBitmapToFile, reference src/screen.cpp:93-104: clamp to [0,1], scale by 255, truncate, opaque alpha, and
 * the file stb_image_write makes of four-channel data (stb_image_write.h:501-509): BITMAPV4HEADER, 32 bits per
 * pixel, BI_BITFIELDS with the masks of B,G,R,A byte order, rows bottom-up, every other header field zero. */
void Screen::writeBitmapToFile(const std::filesystem::path& filePath)
{
    const int w = m_resolution.x, h = m_resolution.y;
    const uint32_t dataSize = (uint32_t)w * (uint32_t)h * 4u, headerSize = 14 + 108;
    std::vector<uint8_t> file(headerSize + dataSize, 0);
    auto put16 = [&](size_t at, uint16_t v) { file[at] = v & 255, file[at + 1] = v >> 8; };
    auto put32 = [&](size_t at, uint32_t v) { for (int k = 0; k < 4; ++k) file[at + k] = (v >> (8 * k)) & 255; };
    file[0] = 'B', file[1] = 'M';
    put32(2, headerSize + dataSize);
    put32(10, headerSize);
    put32(14, 108);
    put32(18, (uint32_t)w);
    put32(22, (uint32_t)h);
    put16(26, 1);
    put16(28, 32);
    put32(30, 3); /* BI_BITFIELDS */
    put32(54, 0x00ff0000u);
    put32(58, 0x0000ff00u);
    put32(62, 0x000000ffu);
    put32(66, 0xff000000u);
    auto to8 = [](float c) {
        const float cl = std::min(std::max(c, 0.0f), 1.0f);
        return (uint8_t)(cl * 255.0f);
    };
    for (int row = 0; row < h; ++row) {
        const vec3* src = m_textureData.data() + (size_t)(h - 1 - row) * w; /* BMP stores the bottom row first */
        uint8_t* dst = file.data() + headerSize + (size_t)row * w * 4;
        for (int x = 0; x < w; ++x) {
            dst[4 * x + 0] = to8(src[x].z);
            dst[4 * x + 1] = to8(src[x].y);
            dst[4 * x + 2] = to8(src[x].x);
            dst[4 * x + 3] = 255;
        }
    }
    std::ofstream out(filePath, std::ios::binary);
    out.write((const char*)file.data(), (std::streamsize)file.size());
}

/* ------------------------------------------------------------------------------------------ Trackball */

Trackball::Trackball(float aspectRatio, float fovy, float distFromLookAt, float rotationX, float rotationY)
    : m_fovy(fovy)
    , m_halfScreenSpaceHeight(std::tan(m_fovy / 2.0f))
    , m_halfScreenSpaceWidth(aspectRatio * m_halfScreenSpaceHeight)
    , m_distanceFromLookAt(distFromLookAt)
    , m_rotationEulerAngles(rotationX, rotationY, 0)
{
}

void Trackball::setCamera(const vec3 lookAt, const vec3 rotations, const float dist)
{
    m_lookAt = lookAt;
    m_rotationEulerAngles = rotations;
    m_distanceFromLookAt = dist;
}

/* glm::quat(eulerAngles), glm/detail/type_quat.inl:208-217 */
static void eulerQuat(const vec3& e, float q[4])
{
    const float cx = std::cos(e.x * 0.5f), cy = std::cos(e.y * 0.5f), cz = std::cos(e.z * 0.5f);
    const float sx = std::sin(e.x * 0.5f), sy = std::sin(e.y * 0.5f), sz = std::sin(e.z * 0.5f);
    q[0] = cx * cy * cz + sx * sy * sz;
    q[1] = sx * cy * cz - cx * sy * sz;
    q[2] = cx * sy * cz + sx * cy * sz;
    q[3] = cx * cy * sz - sx * sy * cz;
}
/* quat * vec3, glm/detail/type_quat.inl:347-354 */
static vec3 rotate(const float q[4], vec3 v)
{
    const vec3 qv { q[1], q[2], q[3] };
    const vec3 uv = cross(qv, v), uuv = cross(qv, uv);
    return v + ((uv * q[0]) + uuv) * 2.0f;
}

vec3 Trackball::position() const
{
    float q[4];
    eulerQuat(m_rotationEulerAngles, q);
    return m_lookAt + rotate(q, vec3(0, 0, -m_distanceFromLookAt));
}

srt_camera Trackball::deviceCamera(int width, int height) const
{
    srt_camera c {};
    eulerQuat(m_rotationEulerAngles, c.quat);
    const vec3 p = position();
    c.origin[0] = p.x, c.origin[1] = p.y, c.origin[2] = p.z;
    c.half_w = m_halfScreenSpaceWidth;
    c.half_h = m_halfScreenSpaceHeight;
    c.width = width, c.height = height;
    return c;
}

Ray Trackball::generateRay(const vec2& pixel) const
{
    const srt_camera c = deviceCamera(1, 1);
    const float ndc[2] = { pixel.x, pixel.y };
    float r[7];
    if (srt_generate_rays(deviceFor(currentDevice()), &c, ndc, 1, r) != SRT_OK)
        raise("Trackball::generateRay");
    Ray ray;
    ray.origin = { r[0], r[1], r[2] };
    ray.direction = { r[3], r[4], r[5] };
    ray.t = r[6];
    return ray;
}

/* ------------------------------------------------------------------------------------------ frame */

/* One rectangle of one frame on copy k of the scene, into the Screen's host pixels. */
static void renderRect(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, size_t k, Screen& screen, const Features& features, int numRays,
                       const RenderOptions& opt, const int rect[4])
{
    const ivec2 res = screen.resolution();
    srt_render_params p {};
    p.camera = camera.deviceCamera(res.x, res.y);
    p.features = featureMask(features, opt.referenceTraversal, opt.exactShadowBoxes);
    p.samples_x = opt.samplesX > 0 ? opt.samplesX : numRays;
    p.samples_y = opt.samplesY > 0 ? opt.samplesY : numRays;
    p.max_depth = opt.maxDepth;
    p.seed = opt.seed;
    for (int i = 0; i < 4; ++i)
        p.rect[i] = rect[i];
    p.flags = (opt.stats ? SRT_RENDER_STATS : 0u) | (opt.timing ? SRT_RENDER_TIMING : 0u);
    /* the Scene's motion-blur / depth-of-field variables, reference src/scene.h:32-40 (motion blur wins, src/render.cpp:218-223) */
    p.lens.samples = features.extra.enableMotionBlur ? scene.MB_samples : scene.DOF_samples;
    p.lens.time0 = scene.time0, p.lens.time1 = scene.time1;
    p.lens.direction[0] = scene.directionVector.x, p.lens.direction[1] = scene.directionVector.y, p.lens.direction[2] = scene.directionVector.z;
    p.lens.focal_length = (float)scene.focalLength;
    p.lens.aperture = scene.aperture;
    srt_output out {};
    static_assert(sizeof(vec3) == 12, "Screen pixels are packed floats");
    out.rgb = &screen.pixels()[0].x;
    out.out_width = res.x, out.out_height = res.y, out.x0 = 0, out.y0 = 0;
    out.on_device = 0;
    if (srt_render(bvh.renderDevice(k), bvh.renderScene(k), &p, &out, k == 0 ? opt.stream : nullptr) != SRT_OK)
        raise("renderRayTracing");
}

std::vector<TileRect> tilesOf(int width, int height, int tileWidth, int tileHeight)
{
    std::vector<TileRect> tiles;
    if (width <= 0 || height <= 0 || tileWidth <= 0 || tileHeight <= 0)
        return tiles;
    for (int y = 0; y < height; y += tileHeight)
        for (int x = 0; x < width; x += tileWidth)
            tiles.push_back({ x, y, std::min(x + tileWidth, width), std::min(y + tileHeight, height) });
    /* centre first: the model usually sits there, and the dearest tiles should not be the last ones handed out */
    const double cx = width / 2.0, cy = height / 2.0;
    auto distance = [&](const TileRect& t) { return std::hypot((t.x0 + t.x1) / 2.0 - cx, (t.y0 + t.y1) / 2.0 - cy); };
    std::stable_sort(tiles.begin(), tiles.end(), [&](const TileRect& a, const TileRect& b) { return distance(a) < distance(b); });
    return tiles;
}

void runWorkers(size_t workers, size_t items, const std::function<void(size_t, size_t)>& work)
{
    if (workers == 0 || items == 0)
        return;
    std::atomic<size_t> next { 0 };
    std::atomic<bool> failed { false };
    std::exception_ptr error;
    std::mutex errorMutex;
    auto body = [&](size_t worker) {
        try {
            for (;;) {
                const size_t item = next.fetch_add(1);
                if (item >= items || failed.load())
                    break;
                work(worker, item);
            }
        } catch (...) {
            std::lock_guard<std::mutex> lock(errorMutex);
            if (!error)
                error = std::current_exception();
            failed.store(true);
        }
    };
    std::vector<std::thread> threads;
    for (size_t w = 1; w < workers; ++w)
        threads.emplace_back(body, w);
    body(0); /* the calling thread is worker 0 */
    for (std::thread& t : threads)
        t.join();
    if (error)
        std::rethrow_exception(error);
}

void renderFrames(const Scene& scene, const BvhInterface& bvh, const std::vector<FrameJob>& frames, const Features& features, const int& numRays,
                  const RenderOptions& opt)
{
    const bool whole = opt.rect[0] == 0 && opt.rect[1] == 0 && opt.rect[2] == 0 && opt.rect[3] == 0;
    const size_t copies = bvh.numRenderDevices();
    struct Item {
        size_t frame;
        int rect[4];
    };
    std::vector<Item> items;
    for (size_t f = 0; f < frames.size(); ++f) {
        const ivec2 res = frames[f].screen->resolution();
        if (!whole) {
            items.push_back({ f, { opt.rect[0], opt.rect[1], opt.rect[2], opt.rect[3] } });
        } else if (copies == 1) {
            items.push_back({ f, { 0, 0, res.x, res.y } });
        } else {
            for (const TileRect& t : tilesOf(res.x, res.y, opt.tileWidth > 0 ? opt.tileWidth : 480, opt.tileHeight > 0 ? opt.tileHeight : 270))
                items.push_back({ f, { t.x0, t.y0, t.x1, t.y1 } });
        }
    }
    /* tiles of one frame land in disjoint rows and columns of its Screen, so the workers need no lock */
    runWorkers(std::min(copies, items.size()), items.size(), [&](size_t worker, size_t i) {
        const Item& it = items[i];
        renderRect(scene, *frames[it.frame].camera, bvh, worker, *frames[it.frame].screen, features, numRays, opt, it.rect);
    });
    /* reference src/render.cpp:231-233; a host-side pass over the frame, as in the reference (sequential by construction) */
    if (features.extra.enableBloomEffect && whole)
        for (const FrameJob& f : frames)
            f.screen->applyBloomFilter(opt.bloomThreshold, 2 * opt.bloomBoxSize + 1);
}

void renderRayTracing(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features, const int& numRays,
                      const RenderOptions& opt)
{
    renderFrames(scene, bvh, { FrameJob { &camera, &screen } }, features, numRays, opt);
}

void renderRayTracing(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features,
                      const float& threshold, const int& boxSize, const int& numRays)
{
    RenderOptions options;
    options.bloomThreshold = threshold;
    options.bloomBoxSize = boxSize;
    renderRayTracing(scene, camera, bvh, screen, features, numRays, options);
}

vec3 getFinalColor(const Scene&, const BvhInterface& bvh, Ray ray, const Features& features, int rayDepth)
{
    const float r[7] = { ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x, ray.direction.y, ray.direction.z, ray.t };
    float rgb[3] = { 0, 0, 0 };
    /* a ray that starts at recursion depth d may bounce 5 - d more times */
    if (srt_final_color_batch(bvh.device(), bvh.deviceScene(), r, 1, featureMask(features, false), 5 - rayDepth, 0, rgb) != SRT_OK)
        raise("getFinalColor");
    return { rgb[0], rgb[1], rgb[2] };
}

} // namespace srt
