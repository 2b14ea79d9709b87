/*
 * srt_build.h -- device-side construction of the scene's BVH and triangle records (srt_scene_build); textually part of
 * srt_kernels.hip, which includes it at its end.
 *
 * The reference builds its hierarchy by recursive median splits (BoundingVolumeHierarchy's constructor,
 * reference src/bounding_volume_hierarchy.cpp:14-94): sort the node's triangles by centroid along axis level % 3, cut at
 * (left + right + 1) / 2, stop at one triangle or below level 16, and give every node the box of its triangles.  For that
 * builder the SHAPE of the tree follows from the triangle count alone, and a node's box is the min / max over its
 * triangles, so the work splits into
 *   order     the final left-to-right order of the triangles: the nested std::sort calls, tie order included.  Computed
 *             here on the host on (centroid coordinate, position) pairs, sub-trees of the top levels on their own threads
 *             (std::sort's moves depend only on comparison outcomes, so sorting keys gives the permutation sorting the
 *             triangle list itself gives), unless the caller supplies the order;
 *   shape     node numbering, child references and levels from the triangle count (host, microseconds);
 *   records   one thread per triangle: the 96-byte traversal record (plane, vertices, the three ray-independent cross
 *             products of the inside test) and the 64-byte shading record, with the reference's float operations;
 *   boxes     level by level from the leaves up, one thread per inner node: a leaf child's box is folded over its
 *             triangles exactly as getBox folds it, an inner child's box is the union of the two boxes its own record
 *             holds.  std::min / std::max keep their FIRST argument on a tie, so getBox's fold picks the earliest of equal
 *             extremes (which only shows in the sign of a zero); "keep the left operand on a tie" in the union does the
 *             same, since a left child's triangles precede its sibling's.
 * The result is the node array, the triangle streams and scene_order of srt_scene_upload, bit for bit what the host
 * mirror's BoundingVolumeHierarchy + flatten(0) produce (depth-first numbering, right child next to its parent).
 */

#include "srt_sort.h"

struct BuildNode {
    uint32_t left, right; /* child references, final encoding */
    uint32_t level;       /* of this node (root = 0) */
};

/* getBox's update for one triangle (reference :55-66) on three floats at a time */
SRT_DI void fold_triangle(f3& lower, f3& upper, f3 v1, f3 v2, f3 v3)
{
    upper.x = max_std(max_std(upper.x, v1.x), max_std(v2.x, v3.x));
    upper.y = max_std(max_std(upper.y, v1.y), max_std(v2.y, v3.y));
    upper.z = max_std(max_std(upper.z, v1.z), max_std(v2.z, v3.z));
    lower.x = min_std(min_std(lower.x, v1.x), min_std(v2.x, v3.x));
    lower.y = min_std(min_std(lower.y, v1.y), min_std(v2.y, v3.y));
    lower.z = min_std(min_std(lower.z, v1.z), min_std(v2.z, v3.z));
}

/* Box of the subtree behind a child reference: folded over a leaf's triangles, or the union of the two boxes an
 * inner node's finished record holds. */
SRT_DI void child_box(uint32_t ref, const float4* __restrict__ tris, const float4* __restrict__ nodes, f3& lower, f3& upper)
{
    if (ref & SRT_LEAF_FLAG) {
        lower = mk3(SRT_FLT_MAX, SRT_FLT_MAX, SRT_FLT_MAX);
        upper = mk3(-SRT_FLT_MAX, -SRT_FLT_MAX, -SRT_FLT_MAX);
        const uint32_t first = ref & 0x00FFFFFFu, cnt = (ref >> 24) & 0x7Fu;
        for (uint32_t k = 0; k < cnt; ++k) {
            const float4* rec = tris + 6ull * (first + k);
            fold_triangle(lower, upper, xyz(rec[1]), xyz(rec[2]), xyz(rec[3]));
        }
    } else {
        const float4* nd = nodes + 4ull * ref;
        const float4 a = nd[0], b = nd[1], c = nd[2];
        const f3 llo = mk3(a.x, a.y, a.z), lhi = mk3(a.w, b.x, b.y), rlo = mk3(b.z, b.w, c.x), rhi = mk3(c.y, c.z, c.w);
        lower = mk3(min_std(llo.x, rlo.x), min_std(llo.y, rlo.y), min_std(llo.z, rlo.z));
        upper = mk3(max_std(lhi.x, rhi.x), max_std(lhi.y, rhi.y), max_std(lhi.z, rhi.z));
    }
}

/* Triangle records in leaf order: position k holds triangle order[k] (an index into the scene's triangle list). */
__global__ __launch_bounds__(SRT_BLOCK) void k_build_triangles(const float* __restrict__ verts8, const uint32_t* __restrict__ tris3,
                                                                 const uint32_t* __restrict__ tri_mesh, const uint32_t* __restrict__ mesh_tri_off,
                                                                 const uint32_t* __restrict__ order, uint32_t n, srt_tri* out_tri, srt_tri_attr* out_attr,
                                                                 uint32_t* scene_order)
{
    const uint32_t k = blockIdx.x * SRT_BLOCK + threadIdx.x;
    if (k >= n)
        return;
    const uint32_t t = order[k];
    const uint32_t mesh = tri_mesh[t];
    const float* a = verts8 + 8ull * tris3[3ull * t], *b = verts8 + 8ull * tris3[3ull * t + 1], *c = verts8 + 8ull * tris3[3ull * t + 2];
    const f3 v0 = mk3(a[0], a[1], a[2]), v1 = mk3(b[0], b[1], b[2]), v2 = mk3(c[0], c[1], c[2]);
    /* trianglePlane, reference src/intersect.cpp:45-59 */
    const f3 cr = cross3(v1 - v0, v2 - v0);
    f3 n3 = mk3(0, 0, 0);
    float D = 0;
    if (!(length3(cr) == 0.0f)) {
        n3 = normalize3(cr);
        D = dot3(v0, n3);
    }
    /* second operands of the three sameSide tests, reference src/intersect.cpp:13-26 */
    const f3 c0 = cross3(v2 - v1, v0 - v1), c1 = cross3(v2 - v0, v1 - v0), c2 = cross3(v1 - v0, v2 - v0);
    float4* o = (float4*)(out_tri + k);
    o[0] = make_float4(n3.x, n3.y, n3.z, D);
    o[1] = make_float4(v0.x, v0.y, v0.z, __uint_as_float(mesh));
    o[2] = make_float4(v1.x, v1.y, v1.z, __uint_as_float(t - mesh_tri_off[mesh]));
    o[3] = make_float4(v2.x, v2.y, v2.z, c0.x);
    o[4] = make_float4(c0.y, c0.z, c1.x, c1.y);
    o[5] = make_float4(c1.z, c2.x, c2.y, c2.z);
    float4* q = (float4*)(out_attr + k);
    q[0] = make_float4(a[3], a[4], a[5], b[3]);
    q[1] = make_float4(b[4], b[5], c[3], c[4]);
    q[2] = make_float4(c[5], a[6], a[7], b[6]);
    q[3] = make_float4(b[7], c[6], c[7], __uint_as_float(mesh));
    scene_order[t] = k;
}

/* The inner nodes of one level (their children's levels are done): both child boxes, references, level. */
__global__ __launch_bounds__(SRT_BLOCK) void k_build_level(const BuildNode* __restrict__ shape, const uint32_t* __restrict__ level_nodes, uint32_t count,
                                                             const float4* __restrict__ tris, float4* nodes, uint32_t* not_fast)
{
    const uint32_t i = blockIdx.x * SRT_BLOCK + threadIdx.x;
    if (i >= count)
        return;
    const uint32_t id = level_nodes[i];
    const BuildNode s = shape[id];
    f3 llo, lhi, rlo, rhi;
    child_box(s.left, tris, nodes, llo, lhi);
    child_box(s.right, tris, nodes, rlo, rhi);
    float4* nd = nodes + 4ull * id;
    nd[0] = make_float4(llo.x, llo.y, llo.z, lhi.x);
    nd[1] = make_float4(lhi.y, lhi.z, rlo.x, rlo.y);
    nd[2] = make_float4(rlo.z, rhi.x, rhi.y, rhi.z);
    nd[3] = make_float4(__uint_as_float(s.left), __uint_as_float(s.right), __uint_as_float(s.level + 1u), 0.0f);
    /* operand ranges of the fast slab test (srt_device.h) */
    const float c12[12] = { llo.x, llo.y, llo.z, lhi.x, lhi.y, lhi.z, rlo.x, rlo.y, rlo.z, rhi.x, rhi.y, rhi.z };
    bool ok = llo.x <= lhi.x && llo.y <= lhi.y && llo.z <= lhi.z && rlo.x <= rhi.x && rlo.y <= rhi.y && rlo.z <= rhi.z;
    for (int k = 0; k < 12; ++k)
        ok = ok && coord_in_fast_range(c12[k]);
    if (!ok)
        *not_fast = 1u; /* a plain vector store: every writer stores the same value */
}

/* Root box: the union for an inner root, the fold for a leaf root (a scene of at most one triangle). */
__global__ void k_build_root(uint32_t root_ref, const float4* __restrict__ tris, const float4* __restrict__ nodes, float* root_box6)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    f3 lo, hi;
    child_box(root_ref, tris, nodes, lo, hi);
    root_box6[0] = lo.x, root_box6[1] = lo.y, root_box6[2] = lo.z, root_box6[3] = hi.x, root_box6[4] = hi.y, root_box6[5] = hi.z;
}

namespace srt_build {

/* Node numbering of the median-split tree over n triangles: depth first, right child directly after its parent
 * (the order the host mirror's flatten(0) emits).  Returns the reference of the subtree over [l, r). */
static uint32_t shape_rec(std::vector<BuildNode>& nodes, std::vector<std::vector<uint32_t>>& by_level, uint32_t l, uint32_t r, uint32_t level,
                          uint32_t& deepest, uint32_t& leaves, bool& leaf_too_big)
{
    if (level > deepest)
        deepest = level;
    if (r - l <= 1 || level > 16) {
        leaves += 1;
        if (r - l > SRT_MAX_LEAF_TRIS)
            leaf_too_big = true;
        return SRT_LEAF_FLAG | ((r - l) << 24) | l;
    }
    const uint32_t id = (uint32_t)nodes.size();
    nodes.push_back({ 0, 0, level });
    if (by_level.size() <= level)
        by_level.resize(level + 1);
    by_level[level].push_back(id);
    const uint32_t median = (uint32_t)(((uint64_t)l + r + 1) / 2);
    const uint32_t right = shape_rec(nodes, by_level, median, r, level + 1, deepest, leaves, leaf_too_big);
    const uint32_t left = shape_rec(nodes, by_level, l, median, level + 1, deepest, leaves, leaf_too_big);
    nodes[id].left = left, nodes[id].right = right;
    return id;
}

struct Key {
    float value;
    uint32_t from;
};

/* constructorHelper's sorting (reference :70-94) on [l, r) of idx, recursively; forks threads for the top levels. */
static void order_rec(std::vector<uint32_t>& idx, const std::vector<float>* centroid, size_t l, size_t r, int axis, int level, int fork_levels)
{
    if (r - l <= 1 || level > 16)
        return;
    std::vector<Key> keys(r - l);
    const std::vector<float>& c = centroid[axis];
    for (size_t k = 0; k < keys.size(); ++k)
        keys[k] = { c[idx[l + k]], (uint32_t)k };
    std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.value < b.value; });
    std::vector<uint32_t> sorted(keys.size());
    for (size_t k = 0; k < keys.size(); ++k)
        sorted[k] = idx[l + keys[k].from];
    std::copy(sorted.begin(), sorted.end(), idx.begin() + (long)l);
    std::vector<Key>().swap(keys);
    std::vector<uint32_t>().swap(sorted);
    const size_t median = (l + r + 1) / 2;
    if (fork_levels > 0) {
        std::thread other([&] { order_rec(idx, centroid, l, median, (axis + 1) % 3, level + 1, fork_levels - 1); });
        order_rec(idx, centroid, median, r, (axis + 1) % 3, level + 1, fork_levels - 1);
        other.join();
    } else {
        order_rec(idx, centroid, l, median, (axis + 1) % 3, level + 1, 0);
        order_rec(idx, centroid, median, r, (axis + 1) % 3, level + 1, 0);
    }
}

} // namespace srt_build

extern "C" {

int srt_scene_build(srt_device* dev, const srt_build_desc* desc, srt_scene** out)
{
    using namespace srt_build;
    if (!dev || !desc || !out)
        return fail(SRT_ERR_INVALID, "srt_scene_build: null argument");
    const bool trace = getenv("SRT_BUILD_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char* what) {
        if (trace) {
            const double t = now();
            fprintf(stderr, "[srt_scene_build] %-28s %7.2f ms\n", what, t - t_prev);
            t_prev = t;
        }
    };
    *out = nullptr;
    const uint32_t n = desc->n_tris;
    if (n > SRT_MAX_TRIS)
        return fail(SRT_ERR_INVALID, "scene has %u triangles; the packed child reference addresses at most %u", n, SRT_MAX_TRIS);
    if ((n && (!desc->tris || !desc->verts || !desc->tri_mesh)) || (desc->n_meshes && !desc->mesh_tri_off) || (desc->n_materials && !desc->materials)
        || (desc->n_lights && !desc->lights) || (desc->n_textures && (!desc->textures || !desc->texels)))
        return fail(SRT_ERR_INVALID, "srt_scene_build: a non-empty array is null");
    if (desc->n_meshes != desc->n_materials)
        return fail(SRT_ERR_INVALID, "srt_scene_build: %u meshes but %u materials (one per mesh)", desc->n_meshes, desc->n_materials);
    /* everything a kernel will index with is checked first */
    if (desc->n_meshes && (desc->mesh_tri_off[0] != 0 || desc->mesh_tri_off[desc->n_meshes] != n))
        return fail(SRT_ERR_INVALID, "srt_scene_build: mesh_tri_off does not span the %u triangles", n);
    for (uint32_t m = 0; m < desc->n_meshes; ++m)
        if (desc->mesh_tri_off[m] > desc->mesh_tri_off[m + 1])
            return fail(SRT_ERR_INVALID, "srt_scene_build: mesh_tri_off decreases at mesh %u", m);
    for (uint32_t t = 0; t < n; ++t) {
        const uint32_t m = desc->tri_mesh[t];
        if (m >= desc->n_meshes || t < desc->mesh_tri_off[m] || t >= desc->mesh_tri_off[m + 1])
            return fail(SRT_ERR_INVALID, "triangle %u: mesh %u does not own it", t, m);
        for (int k = 0; k < 3; ++k)
            if (desc->tris[3ull * t + k] >= desc->n_verts)
                return fail(SRT_ERR_INVALID, "triangle %u: vertex index outside %u vertices", t, desc->n_verts);
    }
    for (uint32_t i = 0; i < desc->n_materials; ++i)
        if (desc->materials[i].texture >= (int32_t)desc->n_textures)
            return fail(SRT_ERR_INVALID, "material %u: texture index outside %u textures", i, desc->n_textures);
    for (uint32_t i = 0; i < desc->n_textures; ++i) {
        const srt_texture& t = desc->textures[i];
        if (t.width == 0 || t.height == 0 || t.offset + (uint64_t)t.width * t.height > desc->n_texels)
            return fail(SRT_ERR_INVALID, "texture %u does not fit the texel array", i);
    }
    if (desc->environment > desc->n_textures)
        return fail(SRT_ERR_INVALID, "environment image %u outside %u textures", desc->environment, desc->n_textures);
    for (uint32_t i = 0; i < desc->n_lights; ++i)
        if (desc->lights[i].type > 2)
            return fail(SRT_ERR_INVALID, "light %u: unknown type %u", i, desc->lights[i].type);

    lap("argument checks");
    /* order: given, or worked out below (on the device; on the host if the device gives up) */
    std::vector<uint32_t> order;
    if (desc->order) {
        std::vector<uint8_t> seen(n, 0);
        for (uint32_t k = 0; k < n; ++k) {
            if (desc->order[k] >= n || seen[desc->order[k]])
                return fail(SRT_ERR_INVALID, "srt_scene_build: order is not a permutation of the %u triangles", n);
            seen[desc->order[k]] = 1;
        }
        order.assign(desc->order, desc->order + n);
    }
    auto host_order = [&]() {
        order.resize(n);
        std::vector<float> centroid[3];
        for (auto& c : centroid)
            c.resize(n);
        for (uint32_t t = 0; t < n; ++t) {
            order[t] = t;
            const float *a = desc->verts + 8ull * desc->tris[3ull * t], *b = desc->verts + 8ull * desc->tris[3ull * t + 1], *c = desc->verts + 8ull * desc->tris[3ull * t + 2];
            for (int k = 0; k < 3; ++k)
                centroid[k][t] = ((a[k] + b[k]) + c[k]) / 3.0f; /* getMedian, reference :37-45 */
        }
        order_rec(order, centroid, 0, n, 0, 0, n >= 4096 ? 4 : 0);
    };
    /* shape */
    std::vector<BuildNode> shape;
    std::vector<std::vector<uint32_t>> by_level;
    uint32_t deepest = 0, leaves = 0;
    bool leaf_too_big = false;
    const uint32_t root_ref = shape_rec(shape, by_level, 0, n, 0, deepest, leaves, leaf_too_big);
    if (leaf_too_big)
        return fail(SRT_ERR_INVALID, "a leaf of the depth-capped tree would hold more than %u triangles", SRT_MAX_LEAF_TRIS);
    std::vector<uint32_t> level_nodes;
    std::vector<uint32_t> level_start;
    for (const auto& lv : by_level) {
        level_start.push_back((uint32_t)level_nodes.size());
        level_nodes.insert(level_nodes.end(), lv.begin(), lv.end());
    }
    level_start.push_back((uint32_t)level_nodes.size());

    lap("shape");
    HIPCHK(hipSetDevice(dev->gpu));
    srt_scene* s = new srt_scene();
    s->dev = dev;
    s->num_levels = deepest + 1;
    s->num_leaves = leaves;
    for (uint32_t i = 0; i < desc->n_lights; ++i)
        s->light_types.push_back(desc->lights[i].type);
    SceneDev& v = s->view;
    /* inputs that only the build kernels read are freed at the end; the rest belongs to the scene */
    std::vector<void*> temp;
    auto temp_upload = [&](const void* host, size_t bytes, void** dev_out) -> int {
        void* p = nullptr;
        HIPCHK(hipMalloc(&p, bytes ? bytes : 4));
        temp.push_back(p);
        if (bytes)
            HIPCHK(hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, 0));
        *dev_out = p;
        return SRT_OK;
    };
    auto scene_alloc = [&](size_t bytes, void** dev_out) -> int {
        void* p = nullptr;
        HIPCHK(hipMalloc(&p, bytes ? bytes : 4));
        s->allocs.push_back(p);
        *dev_out = p;
        return SRT_OK;
    };
    void *d_verts, *d_tris3, *d_tri_mesh, *d_mesh_off, *d_order, *d_shape, *d_level_nodes, *d_result;
    void *d_nodes, *d_rec, *d_attr, *d_scene_order;
    const uint32_t zero_off[1] = { 0 };
    const uint32_t zeros[8] = {};
    int rc = temp_upload(desc->verts, (size_t)desc->n_verts * 32, &d_verts);
    if (rc == SRT_OK) rc = temp_upload(desc->tris, (size_t)n * 12, &d_tris3);
    if (rc == SRT_OK) rc = temp_upload(desc->tri_mesh, (size_t)n * 4, &d_tri_mesh);
    if (rc == SRT_OK) rc = temp_upload(desc->n_meshes ? desc->mesh_tri_off : zero_off, ((size_t)desc->n_meshes + 1) * 4, &d_mesh_off);
    lap("uploads of the triangle list");
    d_order = nullptr;
    const char* where = getenv("SRT_BUILD_SORT"); /* "host": skip the device sort (A/B and fallback testing) */
    if (rc == SRT_OK && order.empty() && n >= 2 && !(where && !strcmp(where, "host"))) {
        bool fell_back = false;
        uint32_t* dev_order = nullptr;
        rc = device_order((const float*)d_verts, (const uint32_t*)d_tris3, n, (uint32_t)by_level.size(), temp, &dev_order, &fell_back);
        if (rc == SRT_OK && !fell_back)
            d_order = dev_order;
        lap(fell_back ? "device sort (gave up)" : "device sort");
    }
    if (rc == SRT_OK && !d_order) {
        if (order.empty())
            host_order();
        lap("host sort");
        rc = temp_upload(order.data(), (size_t)n * 4, &d_order);
    }
    if (rc == SRT_OK) rc = temp_upload(shape.data(), shape.size() * sizeof(BuildNode), &d_shape);
    if (rc == SRT_OK) rc = temp_upload(level_nodes.data(), level_nodes.size() * 4, &d_level_nodes);
    if (rc == SRT_OK) rc = temp_upload(zeros, sizeof zeros, &d_result); /* [0]: some box is outside the fast ranges; [1..6]: the root box */
    if (rc == SRT_OK) rc = scene_alloc(shape.size() * sizeof(srt_node), &d_nodes);
    if (rc == SRT_OK) rc = scene_alloc((size_t)n * sizeof(srt_tri), &d_rec);
    if (rc == SRT_OK) rc = scene_alloc((size_t)n * sizeof(srt_tri_attr), &d_attr);
    if (rc == SRT_OK) rc = scene_alloc((size_t)n * 4, &d_scene_order);
    auto cleanup = [&]() {
        for (void* p : temp)
            (void)hipFree(p);
    };
    if (rc == SRT_OK)
        rc = upload(s, desc->materials, desc->n_materials, &v.materials);
    if (rc == SRT_OK) rc = upload(s, desc->textures, desc->n_textures, &v.textures);
    if (rc == SRT_OK) rc = upload(s, desc->texels, (size_t)desc->n_texels * 3, &v.texels);
    if (rc == SRT_OK) rc = upload(s, desc->lights, desc->n_lights, &v.lights);
    if (rc != SRT_OK) {
        cleanup();
        srt_scene_destroy(s);
        return rc;
    }
    lap("allocations and uploads");
    uint32_t* d_not_fast = (uint32_t*)d_result;
    float* d_root_box = (float*)d_result + 1;
    if (n)
        hipLaunchKernelGGL(k_build_triangles, dim3((n + SRT_BLOCK - 1) / SRT_BLOCK), dim3(SRT_BLOCK), 0, 0, (const float*)d_verts, (const uint32_t*)d_tris3,
            (const uint32_t*)d_tri_mesh, (const uint32_t*)d_mesh_off, (const uint32_t*)d_order, n, (srt_tri*)d_rec, (srt_tri_attr*)d_attr,
            (uint32_t*)d_scene_order);
    for (size_t lv = by_level.size(); lv-- > 0;) {
        const uint32_t count = level_start[lv + 1] - level_start[lv];
        if (count)
            hipLaunchKernelGGL(k_build_level, dim3((count + SRT_BLOCK - 1) / SRT_BLOCK), dim3(SRT_BLOCK), 0, 0, (const BuildNode*)d_shape,
                (const uint32_t*)d_level_nodes + level_start[lv], count, (const float4*)d_rec, (float4*)d_nodes, d_not_fast);
    }
    hipLaunchKernelGGL(k_build_root, dim3(1), dim3(64), 0, 0, root_ref, (const float4*)d_rec, (const float4*)d_nodes, d_root_box);
    uint32_t result[7] = {};
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpy(result, d_result, sizeof result, hipMemcpyDeviceToHost); /* also waits for the kernels */
    lap("kernels");
    cleanup();
    if (e != hipSuccess) {
        srt_scene_destroy(s);
        return fail(SRT_ERR_HIP, "srt_scene_build: %s", hipGetErrorString(e));
    }
    v.nodes = (const float4*)d_nodes;
    v.tris = (const float4*)d_rec;
    v.attrs = (const float4*)d_attr;
    v.scene_order = (const uint32_t*)d_scene_order;
    v.n_nodes = (uint32_t)shape.size(), v.n_tris = n, v.n_materials = desc->n_materials;
    v.n_textures = desc->n_textures, v.n_lights = desc->n_lights;
    v.root_ref = root_ref;
    float root_box[6];
    memcpy(root_box, result + 1, sizeof root_box);
    for (int k = 0; k < 3; ++k)
        v.root_lo[k] = root_box[k], v.root_hi[k] = root_box[3 + k];
    v.n_texels = desc->n_texels;
    v.env_tex = desc->environment ? (int32_t)desc->environment - 1 : -1;
    bool fast = result[0] == 0;
    for (int k = 0; k < 6 && fast; ++k) {
        uint32_t b;
        memcpy(&b, &root_box[k], 4);
        b &= 0x7FFFFFFFu;
        fast = b == 0u || ((b >> 23) >= 87u && (b >> 23) < 147u);
    }
    fast = fast && root_box[0] <= root_box[3] && root_box[1] <= root_box[4] && root_box[2] <= root_box[5];
    v.fast_boxes = fast ? 1u : 0u;
    s->host_children.resize(2 * shape.size());
    for (size_t i = 0; i < shape.size(); ++i)
        s->host_children[2 * i] = shape[i].left, s->host_children[2 * i + 1] = shape[i].right;
    rc = scene_grids(dev, s);
    lap("frees and launch shapes");
    if (rc != SRT_OK) {
        srt_scene_destroy(s);
        return rc;
    }
    *out = s;
    return SRT_OK;
}

int srt_scene_info(const srt_scene* s, uint32_t* n_nodes, uint32_t* n_tris, uint32_t* num_levels, uint32_t* num_leaves, uint32_t* root_ref, float root_box6[6])
{
    if (!s)
        return fail(SRT_ERR_INVALID, "srt_scene_info: null scene");
    if (n_nodes) *n_nodes = s->view.n_nodes;
    if (n_tris) *n_tris = s->view.n_tris;
    if (num_levels) *num_levels = s->num_levels;
    if (num_leaves) *num_leaves = s->num_leaves;
    if (root_ref) *root_ref = s->view.root_ref;
    if (root_box6)
        for (int k = 0; k < 3; ++k)
            root_box6[k] = s->view.root_lo[k], root_box6[3 + k] = s->view.root_hi[k];
    return SRT_OK;
}

int srt_scene_download(const srt_scene* s, srt_node* nodes, srt_tri* tris, srt_tri_attr* attrs, uint32_t* scene_order)
{
    if (!s)
        return fail(SRT_ERR_INVALID, "srt_scene_download: null scene");
    HIPCHK(hipSetDevice(s->dev->gpu));
    if (nodes && s->view.n_nodes) HIPCHK(hipMemcpy(nodes, s->view.nodes, (size_t)s->view.n_nodes * sizeof(srt_node), hipMemcpyDeviceToHost));
    if (tris && s->view.n_tris) HIPCHK(hipMemcpy(tris, s->view.tris, (size_t)s->view.n_tris * sizeof(srt_tri), hipMemcpyDeviceToHost));
    if (attrs && s->view.n_tris) HIPCHK(hipMemcpy(attrs, s->view.attrs, (size_t)s->view.n_tris * sizeof(srt_tri_attr), hipMemcpyDeviceToHost));
    if (scene_order && s->view.n_tris) HIPCHK(hipMemcpy(scene_order, s->view.scene_order, (size_t)s->view.n_tris * 4, hipMemcpyDeviceToHost));
    return SRT_OK;
}

} // extern "C"
