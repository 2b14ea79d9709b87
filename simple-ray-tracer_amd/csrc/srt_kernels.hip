/*
 * srt_kernels.hip -- hand-written gfx950 kernels and the C ABI of libsrt_hip.so (include/srt_hip.h).
 *
 * One frame rectangle is rendered as a wavefront pipeline, all launches stream ordered, no host round trip:
 *
 *   trace_primary   one lane per pixel sample: camera ray (Trackball::generateRay) + BVH query; lanes that
 *                   hit are compacted (wave ballot + one atomic per wave) into the hit queue
 *   trace_shadow    one lane per (queued hit, light sample): the shadow-ray BVH query of that sample (a walk of its
 *                   own, with idle lanes refilled from the queue); one visibility byte per pair
 *   shade           one lane per queued hit: hit attributes, light loop (Phong term of every sample times its
 *                   visibility byte, summed in the reference's order) into the sample's accumulator slot,
 *                   reflection ray (and, with transparency, the ray continued behind the surface) compacted
 *                   into the next path queue
 *   trace_secondary one lane per queued path ray: BVH query, hits compacted into the hit queue
 *   ... trace_shadow / shade / trace_secondary repeat up to max_depth times ...
 *   resolve         one lane per pixel: sum of the pixel's samples in the reference's loop order, division,
 *                   store at Screen::setPixel's position
 *
 * Queue kernels run as a persistent grid that strides over the queue; the queue length is read from
 * device memory.  A sample owns its accumulator slot and has at most one ray per depth, so no floating-point
 * atomics are used and a frame is bit-reproducible run to run (the exception is SRT_F_TRANSPARENCY, where a
 * sample's rays branch and their shares are added atomically).
 *
 * Traversal (BoundingVolumeHierarchy::intersect, reference src/bounding_volume_hierarchy.cpp:316-393).
 * The reference keeps a LIFO of node indices, tests a child's box against the running closest distance C
 * when the parent is visited (push if boxval < C) and again when the node is popped (skip if boxval > C),
 * pushes left then right, and never sorts children.  The same visiting order is produced here with a
 * per-lane stack in LDS that only ever holds pending LEFT siblings:
 *   - the right child is popped immediately after being pushed, with the same T and C, so its pop test
 *     repeats its push test and always passes: the lane simply continues into it (likewise a lone child);
 *   - for a pending left sibling the pop test needs boxval under the CURRENT T.  With t_in / t_out of the
 *     box fixed, boxval is T for a failed slab test, t_out when the origin is inside, else t_in, and after
 *     the first accepted hit C == T.  A node that was pushed is therefore skipped at pop exactly when
 *     it is an origin-inside node with t_out > C (reference rule), or - with SRT_F_MISS_CULL - when its
 *     stored boxval exceeds C.  Both are `v > C` on a float v stored next to the node reference
 *     (v = -1 for entries the reference never skips), so a pop costs one 8-byte LDS read and no box test.
 */
#include <hip/hip_runtime.h>
#include <link.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "srt_device.h"

/* ------------------------------------------------------------------------------------------ device views */

struct SceneDev {
    const float4* nodes; /* 4 float4 per srt_node */
    const float4* tris;  /* 6 float4 per srt_tri */
    const float4* attrs; /* 4 float4 per srt_tri_attr */
    const uint32_t* scene_order;
    const srt_material* materials;
    const srt_texture* textures;
    const float* texels;
    const srt_light* lights;
    uint32_t n_nodes, n_tris, n_materials, n_textures, n_lights;
    uint32_t root_ref;
    float root_lo[3], root_hi[3];
    int32_t env_tex;     /* texture index of the environment image, -1 for none */
    uint32_t fast_boxes; /* every box has lower <= upper and coordinates 0 or 2^-40 <= |c| < 2^20: slab_test_fast applies */
    uint64_t n_texels;
    /* four-wide coarse nodes for shadow queries (k_widen): one record of four uint4 per inner node of even depth, one
     * uint4 per slot: the box of a grandchild (or of a child that is a leaf) on a 16-bit grid over the root box, rounded
     * outwards (x: lower | upper << 16, y, z), and its reference (leaf reference, or index into this array); slots in
     * the order the reference's walk pops them, unused ones (SRT_WIDE_EMPTY) last */
    const uint4* wnodes;
    float q_scale[3];          /* grid coordinate of x: (x - root_lo) * q_scale + 1 */
    uint32_t wide_cap;         /* pending siblings a shadow query may stack on the wide nodes; beyond: the exact walk */
    uint32_t n_wide;           /* records in wnodes */
    uint32_t n_top;            /* the first n_top records are the top levels of the tree: the shadow kernel keeps them in LDS */
    const float4* stris;       /* 48-byte triangle records of the coarse shadow walk (normal + vertices), same order as tris; null: not derived */
    uint32_t coarse_ok;        /* wnodes is usable: every child box lies inside the root box */
};

struct DevCounters {
    unsigned long long path_rays, shadow_rays, shaded_hits, inner_visits, tri_tests, box_tests, wave_inner_steps, wave_leaf_steps;
};

struct Stats {
    uint32_t inner, tri, box;
    uint32_t wave_inner, wave_leaf; /* loop trips of the wave, tallied by its first active lane */
};

/* A queued path ray is three float4: (o.xyz, d.x) (d.y, d.z, thr.x, thr.y) (thr.z, sample, sampler key, -). */

struct HitQ {
    uint32_t* slot; /* primary: sample index; secondary: index into the path queue */
    float* t;
    uint32_t* tri; /* leaf-order triangle position */
    float4* ip;    /* hit point o + d * t and, in w, the bits of the path's sampler key: all a shadow query needs of its hit */
};

struct RenderDev {
    srt_camera cam;
    uint32_t features;
    int32_t nsx, nsy, S;
    int32_t max_depth;
    uint32_t seed;
    int32_t x0, y0, tw, th; /* rectangle origin and size */
    uint32_t n_samples;     /* sample slots: 8 x 8-pixel blocks covering tw x th, times S (srt_final_color_batch: ray count) */
    float* out_rgb;
    int32_t* out_mesh;
    int32_t* out_tri;
    float* out_t;
    int32_t out_w, out_h, out_x0, out_y0;
    const float* custom_rays; /* when set, sample s is the caller's ray s (7 floats) instead of a camera ray */
    srt_lens lens;
    int32_t lens_mode; /* 0: none; 1: motion blur; 2: depth of field (S = lens.samples rays per pixel) */
};

/* ------------------------------------------------------------------------------------------ traversal */

SRT_DI uint32_t wave_sum(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    return v;
}

/* Adds per-lane tallies to the device totals: one reduction per wave, one atomic per field. */
SRT_DI void flush_counters(DevCounters* c, uint32_t path, uint32_t shadow, uint32_t shaded, const Stats& st)
{
    uint32_t a = wave_sum(path), b = wave_sum(shadow), s = wave_sum(shaded);
    uint32_t i = wave_sum(st.inner), t = wave_sum(st.tri), x = wave_sum(st.box);
    uint32_t wi = wave_sum(st.wave_inner), wl = wave_sum(st.wave_leaf);
    if ((threadIdx.x & 63) == 0) {
        if (a) atomicAdd(&c->path_rays, (unsigned long long)a);
        if (b) atomicAdd(&c->shadow_rays, (unsigned long long)b);
        if (s) atomicAdd(&c->shaded_hits, (unsigned long long)s);
        if (i) atomicAdd(&c->inner_visits, (unsigned long long)i);
        if (t) atomicAdd(&c->tri_tests, (unsigned long long)t);
        if (x) atomicAdd(&c->box_tests, (unsigned long long)x);
        if (wi) atomicAdd(&c->wave_inner_steps, (unsigned long long)wi);
        if (wl) atomicAdd(&c->wave_leaf_steps, (unsigned long long)wl);
    }
}

/* Records of the traversal arrays by 32-bit byte offset from the array's base: the base is wave-uniform (a kernel
 * argument), so the load takes it from scalar registers and the lane supplies one shifted index, where 64-bit pointer
 * arithmetic costs the lane a 64-bit shift and a 64-bit add per visit (dragon shadow launch 98.4 -> 96.7 ms).  The
 * arrays stay below 4 GiB: at most 2^24 triangles of 96 bytes, fewer than 2^26 nodes of 64 bytes (srt_scene_upload
 * checks both).  The empty asm keeps the offset a 32-bit value of its own where a loop would otherwise turn it into a
 * 64-bit induction pointer. */
SRT_DI const float4* node_record(const SceneDev& sc, uint32_t i)
{
    uint32_t off = i << 6;
    asm volatile("" : "+v"(off));
    return (const float4*)((const char*)sc.nodes + off);
}
SRT_DI const uint4* wide_record(const SceneDev& sc, uint32_t i)
{
    uint32_t off = i << 6;
    asm volatile("" : "+v"(off));
    return (const uint4*)((const char*)sc.wnodes + off);
}
SRT_DI const float4* tri_record(const SceneDev& sc, uint32_t i)
{
    uint32_t off = i * 96u;
    asm volatile("" : "+v"(off));
    return (const float4*)((const char*)sc.tris + off);
}
SRT_DI const float4* shadow_tri_record(const SceneDev& sc, uint32_t i)
{
    uint32_t off = i * 48u;
    asm volatile("" : "+v"(off));
    return (const float4*)((const char*)sc.stris + off);
}
/* Lane mask of a predicate, straight from the compare (HIP's __ballot takes an int: the predicate would be turned into
 * 0 / 1 and compared again, two vector instructions per vote). */
SRT_DI unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

/* The accel-structure-off branch (reference :319-332): every triangle of every mesh, in scene order. */
template <bool COUNT>
SRT_DI bool traverse_linear(const SceneDev& sc, f3 o, f3 d, float& T, uint32_t& hitTri, Stats& st)
{
    bool any = false;
    for (uint32_t k = 0; k < sc.n_tris; ++k) {
        uint32_t pos = sc.scene_order[k];
        if (COUNT)
            st.tri += 1;
        if (tri_test(sc.tris + 6ull * pos, o, d, T)) {
            any = true;
            hitTri = pos;
        }
    }
    return any;
}

/* BVH query.  stk points at this lane's column of the block's LDS stack (entry k at stk[k * SRT_BLOCK]).
 * Returns whether anything was accepted; T is the final Ray::t, hitTri the last accepted triangle.
 *
 * Loop shape.  A lane stands either on an inner node (box-pair test, ~120 vector instructions) or on a leaf
 * (triangle tests).  Each trip the wave votes (ballot) and runs only the kind of step more of its lanes are
 * waiting for (weighted by cost); the others keep their node for a later trip.  A wave therefore never pays for
 * both kinds of work in one trip.  Which lane advances when does not change any
 * lane's own visiting order.  FAST selects div_by for the slab divisions.  (Shadow queries, whose caller only
 * asks "accepted something and final t < 0.99", have a walk of their own further down: swalk_*.) */
/* A walk that has ended stands on SRT_WALK_DONE, which no reference equals (as a leaf reference it would pass 2^24
 * triangles).  Read as a signed number, `cur >= 0` is an inner node, `cur < -1` a leaf and -1 the ended walk: the wave's
 * votes are then lane masks straight out of one compare each. */
#define SRT_WALK_DONE 0xFFFFFFFFu
/* One BVH query in flight on a lane. */
struct Walk {
    f3 o, d;
    RayRcp rr;
    f3 qi, qc; /* coarse shadow walk: slab value of grid coordinate q is q * qi + qc */
    uint32_t qsx, qsy, qsz; /* ... and per axis 16 where the ray runs towards smaller coordinates (its near bound is the upper one), else 0 */
    float T, C;
    uint32_t cur, hitTri;
    int sp;
    uint32_t* top; /* shadow walks: the lane's first free stack entry (entries are SRT_SHADOW_BLOCK words apart) */
    bool alive;
};

/* Starts a query: the root's push test (reference :344-346). */
template <bool CULL, bool COUNT, bool FAST>
SRT_DI void walk_begin(const SceneDev& sc, Walk& w, f3 o, f3 d, float T0, Stats& st)
{
    w.o = o, w.d = d, w.T = T0, w.C = SRT_FLT_MAX;
    w.rr.x = w.rr.y = w.rr.z = 0.0f;
    if (FAST)
        w.rr.x = refined_rcp(d.x), w.rr.y = refined_rcp(d.y), w.rr.z = refined_rcp(d.z);
    float val;
    bool inside;
    const bool pass = FAST ? slab_test_fast(sc.root_lo, sc.root_hi, o, d, w.rr, T0, val, inside) : slab_test(sc.root_lo, sc.root_hi, o, d, T0, val, inside);
    if (COUNT)
        st.box += 1;
    w.alive = val < w.C && (pass || !CULL);
    w.cur = w.alive ? sc.root_ref : SRT_WALK_DONE;
    w.hitTri = 0xFFFFFFFFu;
    w.sp = 0;
}

/* Next pending left sibling that the reference would not skip at pop; false when the stack is empty. */
SRT_DI bool walk_pop(Walk& w, const uint2* stk)
{
    while (w.sp > 0) {
        --w.sp;
        const uint2 en = stk[w.sp * SRT_BLOCK];
        if (!(__uint_as_float(en.y) > w.C)) {
            w.cur = en.x;
            return true;
        }
    }
    w.cur = SRT_WALK_DONE;
    return false;
}

/* Visit of the inner node the lane stands on: both children's push tests (reference :370-385). */
template <bool CULL, bool COUNT, bool FAST>
SRT_DI void walk_inner(const SceneDev& sc, Walk& w, uint2* stk, Stats& st)
{
    if (COUNT) {
        st.inner += 1;
        st.box += 2;
    }
    const float4* nd = node_record(sc, w.cur);
    const float4 a = nd[0], b = nd[1], c = nd[2], e = nd[3];
    const float llo[3] = { a.x, a.y, a.z }, lhi[3] = { a.w, b.x, b.y };
    const float rlo[3] = { b.z, b.w, c.x }, rhi[3] = { c.y, c.z, c.w };
    const uint32_t left = __float_as_uint(e.x), right = __float_as_uint(e.y);
    float lval, rval;
    bool lin, rin;
    const bool lpass = FAST ? slab_test_fast(llo, lhi, w.o, w.d, w.rr, w.T, lval, lin) : slab_test(llo, lhi, w.o, w.d, w.T, lval, lin);
    const bool rpass = FAST ? slab_test_fast(rlo, rhi, w.o, w.d, w.rr, w.T, rval, rin) : slab_test(rlo, rhi, w.o, w.d, w.T, rval, rin);
    const bool pushL = lval < w.C && (lpass || !CULL);
    const bool pushR = rval < w.C && (rpass || !CULL);
    /* one conditional store, one select, one branch for the pop (a four-way if / else chain compiles to four nested
     * exec-mask regions) */
    if (pushL && pushR) {
        /* pending left sibling; v as described at the top of the file */
        const float v = CULL ? lval : ((lpass && lin) ? lval : -1.0f);
        stk[w.sp * SRT_BLOCK] = make_uint2(left, __float_as_uint(v));
        ++w.sp;
    }
    w.cur = pushR ? right : left;
    if (!(pushL || pushR))
        w.alive = walk_pop(w, stk);
}

/* Visit of the leaf the lane stands on: its triangles in stored order (reference :359-368). */
template <bool COUNT>
SRT_DI void walk_leaf(const SceneDev& sc, Walk& w, const uint2* stk, Stats& st)
{
    const uint32_t first = w.cur & 0x00FFFFFFu;
    const uint32_t cnt = (w.cur >> 24) & 0x7Fu;
    for (uint32_t k = 0; k < cnt; ++k) {
        if (COUNT)
            st.tri += 1;
        if (tri_test_flat(tri_record(sc, first + k), w.o, w.d, w.T)) {
            w.C = min_std(w.C, w.T);
            w.hitTri = first + k;
        }
    }
    w.alive = walk_pop(w, stk);
}

/* Vote weights: a leaf step costs about 2.5 box steps, so it is worth taking once half as many lanes wait for it
 * (measured: same on the dragon workload, 8 % faster on teapot than equal weights). */
#ifndef SRT_VOTE_INNER_W
#define SRT_VOTE_INNER_W 1
#define SRT_VOTE_LEAF_W 2
#endif
#ifndef SRT_REFILL_MIN
#define SRT_REFILL_MIN 24
#endif
#ifndef SRT_STACK_SLACK
#define SRT_STACK_SLACK 1
#endif

/* One trip of the wave: vote, then the kind of step more live lanes are waiting for.  Returns false when no
 * lane of the wave has a query in flight. */
template <bool CULL, bool COUNT, bool FAST>
SRT_DI bool walk_trip(const SceneDev& sc, Walk& w, uint2* stk, Stats& st)
{
    const bool atLeaf = (int)w.cur < -1;
    const bool atInner = (int)w.cur >= 0;
    const unsigned long long mInner = ballot64(atInner), mLeaf = ballot64(atLeaf);
    if ((mInner | mLeaf) == 0)
        return false;
    if (__popcll(mInner) * SRT_VOTE_INNER_W >= __popcll(mLeaf) * SRT_VOTE_LEAF_W) {
        if (COUNT && (int)(threadIdx.x & 63u) == __ffsll((long long)mInner) - 1)
            st.wave_inner += 1;
        if (atInner)
            walk_inner<CULL, COUNT, FAST>(sc, w, stk, st);
    } else {
        if (COUNT && (int)(threadIdx.x & 63u) == __ffsll((long long)mLeaf) - 1)
            st.wave_leaf += 1;
        if (atLeaf)
            walk_leaf<COUNT>(sc, w, stk, st);
    }
    return true;
}

template <bool CULL, bool COUNT, bool FAST>
SRT_DI bool traverse_loop(const SceneDev& sc, f3 o, f3 d, float& T, uint32_t& hitTri, uint2* stk, Stats& st)
{
    Walk w;
    walk_begin<CULL, COUNT, FAST>(sc, w, o, d, T, st);
    while (walk_trip<CULL, COUNT, FAST>(sc, w, stk, st)) {
    }
    T = w.T;
    hitTri = w.hitTri;
    return w.C != SRT_FLT_MAX;
}

/* Operands for which slab_test_fast / div_by are exact (srt_device.h). */
SRT_DI bool ray_in_fast_range(const SceneDev& sc, f3 o, f3 d)
{
    return sc.fast_boxes && dir_in_fast_range(d.x) && dir_in_fast_range(d.y) && dir_in_fast_range(d.z) && coord_in_fast_range(o.x)
        && coord_in_fast_range(o.y) && coord_in_fast_range(o.z);
}

template <bool CULL, bool COUNT>
SRT_DI bool traverse(const SceneDev& sc, f3 o, f3 d, float& T, uint32_t& hitTri, uint2* stk, Stats& st)
{
    /* one decision per wave: every lane's operands in the ranges div_by is exact for, or plain division for all */
    if (ballot64(!ray_in_fast_range(sc, o, d)) == 0)
        return traverse_loop<CULL, COUNT, true>(sc, o, d, T, hitTri, stk, st);
    return traverse_loop<CULL, COUNT, false>(sc, o, d, T, hitTri, stk, st);
}

template <bool CULL, bool COUNT>
SRT_DI bool bvh_query(const SceneDev& sc, uint32_t features, f3 o, f3 d, float& T, uint32_t& hitTri, uint2* stk, Stats& st)
{
    hitTri = 0xFFFFFFFFu;
    if (features & SRT_F_ACCEL)
        return traverse<CULL, COUNT>(sc, o, d, T, hitTri, stk, st);
    return traverse_linear<COUNT>(sc, o, d, T, hitTri, st);
}

/* ------------------------------------------------------------------------------------------ hit attributes */

struct Surface {
    f3 normal, bary, kd, ks;
    f2 uv;
    float shininess, transparency;
    uint32_t mesh, tri;
};

/* acquireTexel, reference src/texture.cpp:5-15.  The reference does not bound the index; the linear index
 * is clamped into the image here, identical wherever the reference's behaviour is defined. */
SRT_DI f3 fetch_texel(const SceneDev& sc, int tex, f2 uv)
{
    const srt_texture tx = sc.textures[tex];
    int i = (int)(uv.x * (float)tx.width);
    int j = (int)((float)tx.height - uv.y * (float)tx.height);
    long long idx = (long long)j * (long long)tx.width + i;
    long long last = (long long)tx.width * tx.height - 1;
    idx = idx < 0 ? 0 : (idx > last ? last : idx);
    const float* p = sc.texels + 3ull * (tx.offset + (unsigned long long)idx);
    return mk3(p[0], p[1], p[2]);
}

/* What intersectWithLeafTriangle leaves in HitInfo for the accepted triangle (reference
 * src/bounding_volume_hierarchy.cpp:262-303).  bary stays 0 without normal interpolation (the reference
 * reads it uninitialised when only texture mapping is on). */
SRT_DI void surface_at(const SceneDev& sc, uint32_t features, f3 o, f3 d, float T, uint32_t pos, Surface& s)
{
    const float4* rec = sc.tris + 6ull * pos;
    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
    const f3 v0 = xyz(q1), v1 = xyz(q2), v2 = xyz(q3);
    s.mesh = __float_as_uint(q1.w);
    s.tri = __float_as_uint(q2.w);
    /* normalize(cross(v1 - v0, v2 - v0)) is the stored plane normal: same operations, and an accepted
     * triangle cannot be the zero-area case */
    s.normal = xyz(q0);
    s.bary = mk3(0, 0, 0);
    s.uv.x = 0, s.uv.y = 0;
    const srt_material m = sc.materials[s.mesh];
    s.kd = mk3(m.kd[0], m.kd[1], m.kd[2]);
    s.ks = mk3(m.ks[0], m.ks[1], m.ks[2]);
    s.shininess = m.shininess;
    s.transparency = m.transparency;
    const f3 p = o + d * T;
    const float4* at = sc.attrs + 4ull * pos;
    if (features & SRT_F_NORMAL_INTERP) {
        s.bary = barycentric3(v0, v1, v2, p);
        const float4 a0 = at[0], a1 = at[1], a2 = at[2];
        const f3 n0 = mk3(a0.x, a0.y, a0.z), n1 = mk3(a0.w, a1.x, a1.y), n2 = mk3(a1.z, a1.w, a2.x);
        s.normal = normalize3(((s.bary.x * n0) + (s.bary.y * n1)) + (s.bary.z * n2));
    }
    {
        const f3 toCam = normalize3(o - p);
        if (dot3(s.normal, toCam) <= 0)
            s.normal = -s.normal;
    }
    if (features & SRT_F_TEXTURE) {
        const float4 a2 = at[2], a3 = at[3];
        const f2 t0 = { a2.y, a2.z }, t1 = { a2.w, a3.x }, t2 = { a3.y, a3.z };
        s.uv.x = (t0.x * s.bary.x + t1.x * s.bary.y) + t2.x * s.bary.z;
        s.uv.y = (t0.y * s.bary.x + t1.y * s.bary.y) + t2.y * s.bary.z;
        if (m.texture >= 0)
            s.kd = fetch_texel(sc, m.texture, s.uv);
    }
}

/* ------------------------------------------------------------------------------------------ lights
 * computeLightContribution and its samplers (reference src/light.cpp:14-39,109-166) are split over two
 * kernels: trace_shadow answers testVisibilityLightSample (src/light.cpp:43-74) for every (hit, light sample)
 * pair, one lane each, and leaves one bit per pair; shade then walks the lights and samples in the
 * reference's order, computes the Phong term of each sample and multiplies it by that bit.  Both kernels
 * derive a sample's position from the same function below, so they agree bit for bit. */

/* Shadow queries the reference issues for one light at one hit. */
SRT_DI uint32_t shadow_samples(uint32_t type, uint32_t features)
{
    if (type == 0)
        return (features & SRT_F_HARD_SHADOW) ? 1u : 0u;
    return (features & SRT_F_SOFT_SHADOW) ? 100u : 0u;
}

/* Position and colour of sample s of a light: the light itself for a point light, sampleSegmentLight with
 * t = (s + r) / 100, or sampleParallelogramLight with (i + r1) / 10, (j + r2) / 10, s = 10 i + j.  The reference
 * draws r from libc rand(); here it comes from the counter-based generator keyed by (seed, path, depth, light,
 * sample). */
SRT_DI void light_sample(const srt_light* L, uint32_t seed, uint32_t path, uint32_t depth, uint32_t li, uint32_t s, f3& pos, f3& col)
{
    const uint32_t type = L->type;
    if (type == 0) {
        pos = mk3(L->f[0], L->f[1], L->f[2]);
        col = mk3(L->f[3], L->f[4], L->f[5]);
    } else if (type == 1) {
        const f3 e0 = mk3(L->f[0], L->f[1], L->f[2]), e1 = mk3(L->f[3], L->f[4], L->f[5]);
        const f3 c0 = mk3(L->f[6], L->f[7], L->f[8]), c1 = mk3(L->f[9], L->f[10], L->f[11]);
        const float r = rand01(seed, path, depth, li, s, 0);
        const float a = ((float)s + r) / 100.0f;
        pos = mk3(e0.x * a + e1.x * (1 - a), e0.y * a + e1.y * (1 - a), e0.z * a + e1.z * (1 - a));
        col = mk3(c0.x * a + c1.x * (1 - a), c0.y * a + c1.y * (1 - a), c0.z * a + c1.z * (1 - a));
    } else {
        const f3 p0 = mk3(L->f[0], L->f[1], L->f[2]), e01 = mk3(L->f[3], L->f[4], L->f[5]);
        const f3 e02 = mk3(L->f[6], L->f[7], L->f[8]);
        const f3 c0 = mk3(L->f[9], L->f[10], L->f[11]), c1 = mk3(L->f[12], L->f[13], L->f[14]);
        const f3 c2 = mk3(L->f[15], L->f[16], L->f[17]), c3 = mk3(L->f[18], L->f[19], L->f[20]);
        const uint32_t i = s / 10u, j = s - i * 10u;
        const float r1 = rand01(seed, path, depth, li, s, 0);
        const float r2 = rand01(seed, path, depth, li, s, 1);
        const float x = ((float)i + r1) / 10.0f;
        const float y = ((float)j + r2) / 10.0f;
        pos = (p0 + (x * e01)) + (y * e02);
        col = ((((c0 * (1 - x)) * (1 - y)) + ((c1 * x) * (1 - y))) + ((c2 * y) * (1 - x))) + ((c3 * x) * y);
    }
}

typedef uint32_t u32_unaligned __attribute__((aligned(1)));

/* The shade kernel's half: lights and samples in the reference's order; vis holds this hit's visibility bits
 * (one byte each) starting at index bit0. */
SRT_DI f3 light_contribution(const SceneDev& sc, uint32_t features, uint32_t seed, uint32_t path, uint32_t depth, f3 o, f3 d, float T,
                             const Surface& s, const unsigned char* __restrict__ vis, unsigned long long bit0)
{
    if (!(features & SRT_F_SHADING))
        return s.kd;
    f3 res = mk3(0, 0, 0);
    unsigned long long bit = bit0;
    /* With these, compute_shading is finite for every light sample at an ordinary distance: the normal and the view
     * vector can be normalised, colours and exponent cannot overflow, powf gets a base in [0, 1] and an exponent >= 0. */
    const f3 view = (o + d * T) - o;
    const float nn = dot3(s.normal, s.normal), vv = dot3(view, view);
    const float big = 1e18f;
    auto small3 = [&](f3 a) { return __builtin_fabsf(a.x) < big && __builtin_fabsf(a.y) < big && __builtin_fabsf(a.z) < big; };
    bool tame = nn >= 1e-30f && nn <= 1e30f && vv >= 1e-30f && vv <= 1e30f && small3(s.kd) && small3(s.ks) && s.shininess >= 0.0f && s.shininess < big
        && small3(o + d * T);
    for (uint32_t li = 0; li < sc.n_lights && tame; ++li) {
        /* light colours (every float of the record after the first three is a colour or an edge) and positions */
        for (int k = 0; k < 21; ++k)
            tame = tame && __builtin_fabsf(sc.lights[li].f[k]) < big;
    }
    for (uint32_t li = 0; li < sc.n_lights; ++li) {
        const srt_light* L = sc.lights + li;
        const uint32_t type = L->type;
        const uint32_t n = shadow_samples(type, features);
        /* The reference multiplies a sample's Phong term by its visibility (0 or 1).  The term of an occluded sample - a
         * powf and two normalisations - is not computed where the product is known to be an exact zero that leaves the
         * sum as it is: the hit's own inputs are tame (`tame`, below) and the shadow kernel found the direction to the
         * sample of ordinary length (visibility byte 0; 2 marks an occluded sample it could not vouch for).  Otherwise
         * the term is computed and multiplied, so that a non-finite term poisons the sum as it does in the reference.
         * Neighbouring hits mostly share a shadow, so whole waves skip the term. */
        if (type == 0) {
            const uint32_t v = n ? vis[bit] : 1u;
            bit += n;
            if (v != 0u || !tame) {
                f3 pos, col;
                light_sample(L, seed, path, depth, li, 0, pos, col);
                const f3 sh = compute_shading(pos, col, o, d, T, s.normal, s.kd, s.ks, s.shininess);
                res = n ? res + sh * (v == 1u ? 1.0f : 0.0f) : res + sh;
            }
        } else {
            /* segment: 100 samples, each weighted 1/100; parallelogram: 10 x 10 samples, each weighted 1/100 */
            /* the sample's visibility bytes four at a time (the buffer is padded beyond its last byte; a hit's bytes
             * start anywhere, the hardware reads unaligned words) */
            uint32_t word = 0;
            for (uint32_t k = 0; k < n; ++k) {
                if ((k & 3u) == 0u)
                    word = *(const u32_unaligned*)(vis + bit + k);
                const uint32_t v = (word >> ((k & 3u) * 8u)) & 0xFFu;
                if (v != 0u || !tame) {
                    f3 pos, col;
                    light_sample(L, seed, path, depth, li, k, pos, col);
                    const f3 sh = compute_shading(pos, col, o, d, T, s.normal, s.kd, s.ks, s.shininess) / 100.0f;
                    res = res + sh * (v == 1u ? 1.0f : 0.0f);
                }
            }
            bit += n;
        }
    }
    return res;
}

/* ------------------------------------------------------------------------------------------ frame kernels */

/* Sample numbering inside a rectangle.  Pixels are grouped in 8 x 8 blocks (blocks row-major over the rectangle,
 * padded at its right and top edges), pixels of a block follow a Z curve, and the S samples of a pixel are
 * adjacent.  Any aligned run of 64 consecutive samples - one wave - therefore covers a compact patch of the
 * image (8 x 8 pixels at 1 spp, 4 x 4 at 4 spp, 4 x 2 at 8 spp), so its rays enter the tree together. */
SRT_DI uint32_t blocks_x(const RenderDev& rp) { return ((uint32_t)rp.tw + 7u) >> 3; }
SRT_DI uint32_t compact_even_bits(uint32_t v)
{
    v &= 0x55u;
    v = (v | (v >> 1)) & 0x33u;
    v = (v | (v >> 2)) & 0x0Fu;
    return v;
}
SRT_DI uint32_t spread_bits(uint32_t v)
{
    v = (v | (v << 2)) & 0x33u;
    v = (v | (v << 1)) & 0x55u;
    return v;
}
/* sample index -> pixel and sub-sample; false for the padding beyond the rectangle's edges */
SRT_DI bool sample_decode(const RenderDev& rp, uint32_t s, int& px, int& py, int& sub)
{
    const uint32_t p = s / (uint32_t)rp.S;
    sub = (int)(s - p * (uint32_t)rp.S);
    const uint32_t blk = p >> 6, z = p & 63u;
    const uint32_t bx = blk % blocks_x(rp), by = blk / blocks_x(rp);
    const uint32_t lx = (bx << 3) + compact_even_bits(z), ly = (by << 3) + compact_even_bits(z >> 1);
    px = rp.x0 + (int)lx;
    py = rp.y0 + (int)ly;
    return lx < (uint32_t)rp.tw && ly < (uint32_t)rp.th;
}
/* pixel (rectangle-local) -> index of its first sample */
SRT_DI uint32_t pixel_first_sample(const RenderDev& rp, uint32_t lx, uint32_t ly)
{
    const uint32_t blk = (ly >> 3) * blocks_x(rp) + (lx >> 3);
    const uint32_t z = spread_bits(lx & 7u) | (spread_bits(ly & 7u) << 1);
    return ((blk << 6) + z) * (uint32_t)rp.S;
}

/* Normalized pixel position of a sample, reference src/render.cpp:190-193 (one ray per pixel) and
 * :200-209 (n x n grid; the reference's random offset is below 4.7e-10 and vanishes in the float sums). */
SRT_DI void sample_ndc(const RenderDev& rp, int px, int py, int sub, float& nx, float& ny)
{
    const float W = (float)rp.cam.width, H = (float)rp.cam.height;
    if (rp.features & SRT_F_MULTISAMPLE) {
        const int i = sub / rp.nsy, j = sub - i * rp.nsy;
        const float a = ((float)i + 0.0f) / (float)rp.nsx + (float)px;
        const float b = ((float)j + 0.0f) / (float)rp.nsy + (float)py;
        nx = a / W * 2.0f - 1.0f;
        ny = b / H * 2.0f - 1.0f;
    } else {
        nx = (float)px / W * 2.0f - 1.0f;
        ny = (float)py / H * 2.0f - 1.0f;
    }
}

SRT_DI long long out_index(const RenderDev& rp, int px, int py)
{
    return (long long)(rp.out_h - 1 - (py - rp.out_y0)) * rp.out_w + (px - rp.out_x0);
}

/* Primary ray of sample s: a camera ray, or the caller's own ray for srt_final_color_batch. */
SRT_DI bool primary_ray(const RenderDev& rp, uint32_t s, f3& o, f3& d, float& T, int& px, int& py, int& sub)
{
    if (rp.custom_rays) {
        const float* r = rp.custom_rays + 7ull * s;
        o = mk3(r[0], r[1], r[2]);
        d = mk3(r[3], r[4], r[5]);
        T = r[6];
        px = (int)s, py = 0, sub = 0;
        return true;
    }
    if (!sample_decode(rp, s, px, py, sub))
        return false;
    float nx, ny;
    sample_ndc(rp, px, py, sub, nx, ny);
    o = mk3(rp.cam.origin[0], rp.cam.origin[1], rp.cam.origin[2]);
    d = camera_dir(rp.cam, nx, ny);
    T = SRT_FLT_MAX;
    if (rp.lens_mode) {
        /* sampler key: the frame-global sample number, in a stream of its own (depth field all ones) */
        const uint32_t key = ((uint32_t)py * (uint32_t)rp.cam.width + (uint32_t)px) * (uint32_t)rp.S + (uint32_t)sub;
        if (rp.lens_mode == 1) {
            /* motionBlur, reference src/render.cpp:152-161 */
            const float random = rand01(rp.seed, key, 0xFFFFFFFFu, 0, 0, 0);
            const float t0 = rp.lens.time0, t1 = rp.lens.time1;
            const float k = (t0 + random * (t1 - t0)) - ((t1 - t0) / 2);
            o = o + normalize3(mk3(rp.lens.direction[0], rp.lens.direction[1], rp.lens.direction[2])) * k;
        } else {
            /* DOF, reference src/render.cpp:163-178: r in [-1, 1) */
            const f3 converge = o + d * rp.lens.focal_length;
            const float r1 = -1.0f + 2.0f * rand01(rp.seed, key, 0xFFFFFFFFu, 0, 0, 0);
            const float r2 = -1.0f + 2.0f * rand01(rp.seed, key, 0xFFFFFFFFu, 0, 0, 1);
            const float r3 = -1.0f + 2.0f * rand01(rp.seed, key, 0xFFFFFFFFu, 0, 0, 2);
            o = o + mk3(r1 * rp.lens.aperture, r2 * rp.lens.aperture, r3 * rp.lens.aperture);
            d = normalize3(converge - o);
        }
    }
    return true;
}

/* The miss branch of getFinalColor with extra.enableEnvironmentMapping (reference src/render.cpp:40-81,141-147):
 * where the ray leaves (or enters) the box [-64, 64]^3, mapped onto the cross-shaped cube-map image. */
SRT_DI f3 environment_color(const SceneDev& sc, f3 o, f3 d, float T)
{
    const float lo[3] = { -64.0f, -64.0f, -64.0f }, hi[3] = { 64.0f, 64.0f, 64.0f };
    float t;
    bool inside;
    if (sc.env_tex < 0 || !slab_test(lo, hi, o, d, T, t, inside))
        return mk3(0, 0, 0);
    const f3 p = (o + d * t) / 64.0f;
    const float ax = __builtin_fabsf(p.x), ay = __builtin_fabsf(p.y), az = __builtin_fabsf(p.z);
    f2 center = { -1.0f, -1.0f }, shift = { -1.0f, -1.0f };
    if (ax >= ay && ax >= az) {
        if (p.x > 0)
            center = { 2.5f / 4, 1.5f / 3 }, shift = { -p.z / 8, p.y / 6 };
        else
            center = { 0.5f / 4, 1.5f / 3 }, shift = { p.z / 8, p.y / 6 };
    }
    if (ay >= ax && ay >= az) {
        if (p.y > 0)
            center = { 1.5f / 4, 2.5f / 3 }, shift = { p.x / 8, -p.z / 6 };
        else
            center = { 1.5f / 4, 0.5f / 3 }, shift = { p.x / 8, p.z / 6 };
    }
    if (az >= ax && az >= ay) {
        if (p.z > 0)
            center = { 1.5f / 4, 1.5f / 3 }, shift = { p.x / 8, p.y / 6 };
        else
            center = { 3.5f / 4, 1.5f / 3 }, shift = { -p.x / 8, p.y / 6 };
    }
    f2 ans = { center.x + shift.x, center.y + shift.y };
    if (!(0 <= ans.x && ans.x <= 1 && 0 <= ans.y && ans.y <= 1))
        ans.x = 0, ans.y = 0;
    return fetch_texel(sc, sc.env_tex, ans);
}

/* Appends the lanes with want == true to a queue: one atomic per wave, slots in lane order. */
SRT_DI uint32_t queue_slot(bool want, uint32_t* count)
{
    const unsigned long long mask = ballot64(want);
    if (mask == 0)
        return 0;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
    uint32_t base = 0;
    const int leader = __ffsll((long long)mask) - 1;
    if ((int)lane == leader)
        base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = __shfl(base, leader, 64);
    return base + rank;
}

template <bool CULL, bool COUNT>
__global__ __launch_bounds__(SRT_BLOCK) void k_trace_primary(SceneDev sc, RenderDev rp, HitQ hq, uint32_t* hq_count, DevCounters* cnt, float* acc,
                                                               unsigned char* nlev)
{
    extern __shared__ uint2 lds_stack[];
    uint2* stk = lds_stack + threadIdx.x;
    Stats st = { 0, 0, 0, 0, 0 };
    uint32_t nPath = 0;
    const uint32_t stride = gridDim.x * SRT_BLOCK;
    for (uint32_t base = blockIdx.x * SRT_BLOCK + (threadIdx.x & ~63u); base < rp.n_samples; base += stride) {
        const uint32_t s = base + (threadIdx.x & 63u);
        const bool active = s < rp.n_samples;
        bool hit = false;
        float T = SRT_FLT_MAX;
        uint32_t tri = 0xFFFFFFFFu;
        int px = 0, py = 0, sub = 0;
        f3 o, d;
        if (active && primary_ray(rp, s, o, d, T, px, py, sub)) {
            nPath += 1;
            hit = bvh_query<CULL, COUNT>(sc, rp.features, o, d, T, tri, stk, st);
            /* the sample's slot starts at black, or at the environment's colour: what a ray that hits nothing returns
             * (the frame's launches need no cleared buffers: every slot that is read later is written here first) */
            f3 e = mk3(0, 0, 0);
            if (!hit && (rp.features & SRT_F_ENVIRONMENT))
                e = environment_color(sc, o, d, T);
            acc[3ull * s] = e.x, acc[3ull * s + 1] = e.y, acc[3ull * s + 2] = e.z;
            if (nlev)
                nlev[s] = 0;
            if (sub == 0 && (rp.out_mesh || rp.out_tri || rp.out_t)) {
                const long long oi = out_index(rp, px, py);
                int mesh = -1, ti = -1;
                if (hit) {
                    mesh = (int)__float_as_uint(sc.tris[6ull * tri + 1].w);
                    ti = (int)__float_as_uint(sc.tris[6ull * tri + 2].w);
                }
                if (rp.out_mesh) rp.out_mesh[oi] = mesh;
                if (rp.out_tri) rp.out_tri[oi] = ti;
                if (rp.out_t) rp.out_t[oi] = T;
            }
        }
        const uint32_t q = queue_slot(hit, hq_count);
        if (hit) {
            hq.slot[q] = s;
            hq.t[q] = T;
            hq.tri[q] = tri;
            const f3 ip = o + d * T;
            const uint32_t path = ((uint32_t)py * (uint32_t)rp.cam.width + (uint32_t)px) * (uint32_t)rp.S + (uint32_t)sub;
            hq.ip[q] = make_float4(ip.x, ip.y, ip.z, __uint_as_float(path));
        }
    }
    flush_counters(cnt, nPath, 0, 0, st);
}

template <bool CULL, bool COUNT>
__global__ __launch_bounds__(SRT_BLOCK) void k_trace_secondary(SceneDev sc, uint32_t features, const float4* paths, const uint32_t* n_paths,
                                                                 HitQ hq, uint32_t* hq_count, DevCounters* cnt, float* acc)
{
    extern __shared__ uint2 lds_stack[];
    uint2* stk = lds_stack + threadIdx.x;
    Stats st = { 0, 0, 0, 0, 0 };
    uint32_t nPath = 0;
    const uint32_t n = *n_paths;
    const uint32_t stride = gridDim.x * SRT_BLOCK;
    for (uint32_t base = blockIdx.x * SRT_BLOCK + (threadIdx.x & ~63u); base < n; base += stride) {
        const uint32_t i = base + (threadIdx.x & 63u);
        bool hit = false;
        float T = SRT_FLT_MAX;
        uint32_t tri = 0xFFFFFFFFu;
        f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1);
        if (i < n) {
            const float4 a = paths[3ull * i], b = paths[3ull * i + 1];
            o = mk3(a.x, a.y, a.z), d = mk3(a.w, b.x, b.y);
            nPath += 1;
            hit = bvh_query<CULL, COUNT>(sc, features, o, d, T, tri, stk, st);
            if (!hit && (features & SRT_F_ENVIRONMENT)) {
                /* the environment seen by a reflected / transmitted ray, weighted like a hit's light (k_shade) */
                const float4 c = paths[3ull * i + 2];
                const f3 thr = mk3(b.z, b.w, c.x);
                const f3 e = environment_color(sc, o, d, T);
                float* a3 = acc + 3ull * __float_as_uint(c.y);
                if (features & SRT_F_TRANSPARENCY) {
                    atomicAdd(a3 + 0, thr.x * e.x);
                    atomicAdd(a3 + 1, thr.y * e.y);
                    atomicAdd(a3 + 2, thr.z * e.z);
                } else {
                    /* the value of the sample's deepest level, folded into the levels above it by sample_value */
                    a3[0] = e.x, a3[1] = e.y, a3[2] = e.z;
                }
            }
        }
        const uint32_t q = queue_slot(hit, hq_count);
        if (hit) {
            hq.slot[q] = i;
            hq.t[q] = T;
            hq.tri[q] = tri;
            const f3 ip = o + d * T;
            hq.ip[q] = make_float4(ip.x, ip.y, ip.z, paths[3ull * i + 2].z); /* the sampler key travels with the ray */
        }
    }
    flush_counters(cnt, nPath, 0, 0, st);
}

/* k_trace_secondary with the shadow kernel's feeding: reflection rays differ in length as shadow queries do, so a lane
 * whose query ends hands in its hit at once and, as soon as SRT_REFILL_MIN lanes of the wave are idle, the idle lanes take
 * the wave's next rays, drawn in chunks from a counter all waves share (chunk size per launch as in k_trace_shadow).
 * A ray outside the ranges of the fast slab test is walked to its end by its lane when it is taken. */
template <bool CULL, bool COUNT>
__global__ __launch_bounds__(SRT_BLOCK) void k_trace_secondary_refill(SceneDev sc, uint32_t features, const float4* paths, const uint32_t* n_paths, HitQ hq,
                                                                       uint32_t* hq_count, DevCounters* cnt, float* acc, unsigned long long* next_chunk)
{
    extern __shared__ uint2 lds_stack[];
    uint2* stk = lds_stack + threadIdx.x;
    Stats st = { 0, 0, 0, 0, 0 };
    uint32_t nPath = 0;
    const unsigned long long n = *n_paths;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long waves = (unsigned long long)gridDim.x * (SRT_BLOCK / 64);
    uint32_t chunkItems = 64u;
    {
        const unsigned long long per = n / (waves * 16ull);
        while (chunkItems < 1024u && 2ull * chunkItems <= per)
            chunkItems *= 2u;
        chunkItems = __builtin_amdgcn_readfirstlane(chunkItems);
    }
    const unsigned long long first_shared = waves * chunkItems;
    auto uniform = [](unsigned long long v) {
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    };
    auto grab = [&]() {
        unsigned long long base = 0;
        if (lane == 0)
            base = atomicAdd(next_chunk, (unsigned long long)chunkItems);
        return first_shared + uniform(base);
    };
    unsigned long long chunk = uniform(((unsigned long long)blockIdx.x * (SRT_BLOCK / 64) + (threadIdx.x >> 6)) * chunkItems);
    uint32_t used = 0;
    bool dry = false;
    Walk w;
    w.alive = false;
    w.o = mk3(0, 0, 0), w.d = mk3(0, 0, 1);
    w.cur = SRT_WALK_DONE, w.sp = 0, w.T = SRT_FLT_MAX, w.C = SRT_FLT_MAX, w.hitTri = 0xFFFFFFFFu;
    w.rr.x = w.rr.y = w.rr.z = 0.0f;
    uint32_t item = 0;
    bool busy = false; /* the lane holds a ray: in flight (w.alive) or finished and waiting to be handed in */
    for (;;) {
        const unsigned long long spare = ballot64(!w.alive);
        if (__popcll(spare) >= SRT_REFILL_MIN) {
            /* hand in what has finished since the last time: one atomic for all of it */
            const bool fin = busy && !w.alive;
            const bool hit = fin && w.C != SRT_FLT_MAX;
            if (fin && !hit && (features & SRT_F_ENVIRONMENT)) {
                /* the environment seen by a reflected / transmitted ray, weighted like a hit's light (k_shade) */
                const float4 b = paths[3ull * item + 1], c = paths[3ull * item + 2];
                const f3 thr = mk3(b.z, b.w, c.x);
                const f3 e = environment_color(sc, w.o, w.d, w.T);
                float* a3 = acc + 3ull * __float_as_uint(c.y);
                if (features & SRT_F_TRANSPARENCY) {
                    atomicAdd(a3 + 0, thr.x * e.x);
                    atomicAdd(a3 + 1, thr.y * e.y);
                    atomicAdd(a3 + 2, thr.z * e.z);
                } else {
                    a3[0] = e.x, a3[1] = e.y, a3[2] = e.z;
                }
            }
            const uint32_t q = queue_slot(hit, hq_count);
            if (hit) {
                hq.slot[q] = item;
                hq.t[q] = w.T;
                hq.tri[q] = w.hitTri;
                const f3 ip = w.o + w.d * w.T;
                hq.ip[q] = make_float4(ip.x, ip.y, ip.z, paths[3ull * item + 2].z);
            }
            busy = busy && w.alive;
            if (!dry && used == chunkItems) {
                chunk = grab();
                used = 0;
            }
            dry = dry || chunk >= n;
            if (!dry) {
                const uint32_t cntSpare = (uint32_t)__popcll(spare);
                const uint32_t pos = used + (uint32_t)__popcll(spare & ((1ull << lane) - 1ull));
                unsigned long long mine = chunk + pos;
                used += cntSpare;
                if (used > chunkItems) {
                    chunk = grab();
                    used -= chunkItems;
                    if (pos >= chunkItems)
                        mine = chunk + (pos - chunkItems);
                }
                if (!busy && mine < n) {
                    item = (uint32_t)mine;
                    const float4 a = paths[3ull * item], b = paths[3ull * item + 1];
                    const f3 o = mk3(a.x, a.y, a.z), d = mk3(a.w, b.x, b.y);
                    nPath += 1;
                    busy = true;
                    if (ray_in_fast_range(sc, o, d)) {
                        walk_begin<CULL, COUNT, true>(sc, w, o, d, SRT_FLT_MAX, st);
                    } else {
                        float T = SRT_FLT_MAX;
                        uint32_t tri = 0xFFFFFFFFu;
                        const bool h = traverse_loop<CULL, COUNT, false>(sc, o, d, T, tri, stk, st);
                        w.o = o, w.d = d, w.T = T, w.hitTri = tri, w.C = h ? T : SRT_FLT_MAX;
                        w.alive = false, w.cur = SRT_WALK_DONE;
                    }
                }
            }
        }
        if (!walk_trip<CULL, COUNT, true>(sc, w, stk, st)) {
            /* nothing in flight: every lane is spare, so the next pass hands in and refills, or the wave is done */
            if (dry && ballot64(busy) == 0)
                break;
        }
    }
    flush_counters(cnt, nPath, 0, 0, st);
}

/* Path ray and hit point data of queued hit h, as the shade kernel sees them. */
SRT_DI void hit_ray(const RenderDev& rp, int depth, const HitQ& hq, const float4* cur_paths, uint32_t h, f3& o, f3& d, f3& thr, float& T,
                    uint32_t& tri, uint32_t& sample, uint32_t& path)
{
    const uint32_t slot = hq.slot[h];
    T = hq.t[h];
    tri = hq.tri[h];
    int px, py, sub;
    if (depth == 0) {
        sample = slot;
        float T0;
        primary_ray(rp, sample, o, d, T0, px, py, sub);
        thr = mk3(1, 1, 1);
    } else {
        const float4 a = cur_paths[3ull * slot], b = cur_paths[3ull * slot + 1], c = cur_paths[3ull * slot + 2];
        o = mk3(a.x, a.y, a.z);
        d = mk3(a.w, b.x, b.y);
        thr = mk3(b.z, b.w, c.x);
        sample = __float_as_uint(c.y);
        if (rp.custom_rays)
            px = (int)sample, py = 0, sub = 0;
        else
            sample_decode(rp, sample, px, py, sub);
    }
    /* sampler key: frame-global sample number, so a frame does not depend on how it is cut into rectangles */
    path = ((uint32_t)py * (uint32_t)rp.cam.width + (uint32_t)px) * (uint32_t)rp.S + (uint32_t)sub;
}

/* Launch shape of the shadow kernel.  (Keeping the head of the node array in LDS next to the stacks was measured and
 * dropped: a wave's lanes are rarely all inside the cached levels, so both fetch paths run in most steps.) */
#ifndef SRT_SHADOW_BLOCK
#define SRT_SHADOW_BLOCK 256
#endif
#define SRT_LDS_LIGHTS 16u
#ifndef SRT_WIDE_LDS_NODES
#define SRT_WIDE_LDS_NODES 85u /* wide records (64 bytes each) of the tree's top levels a block of the shadow kernel keeps in LDS: four whole levels */
#endif
#ifndef SRT_SVOTE_INNER_W
/* Vote weights of the shadow walk, box step : leaf step (dragon / teapot shadow launch: 1:2 141.4 / 0.752 ms, 3:4 138.7 /
 * 0.746, 1:1 139.2 / 0.759, 4:3 141.6 / 0.767) */
#define SRT_SVOTE_INNER_W 1
#define SRT_SVOTE_LEAF_W 3
#endif
#ifndef SRT_SHADOW_MORE_STEPS
#define SRT_SHADOW_MORE_STEPS 5  /* box steps a trip may add to its first one ... */
#define SRT_SHADOW_MORE_MIN 32u  /* ... while at least this many lanes can take one */
#endif
#ifndef SRT_SHADOW_CHUNK_MAX
#define SRT_SHADOW_CHUNK_MAX 1024u /* most shadow queries a wave draws from the shared counter at a time */
#endif /* lights a block keeps in LDS; further ones are read from global memory */

/* ---- shadow queries: a walk with four-byte stack entries.
 * Until a shadow query accepts its first hit, C is FLT_MAX: no pop test can skip a pending sibling, and the
 * value the general walk keeps beside each sibling is never looked at.  An accepted hit below 0.99 answers the
 * query.  So the common walk (swalk_*) stacks the sibling's reference alone, which halves the LDS a wave needs
 * and lets more waves share a compute unit.  Two kinds of query do not fit it and are answered by
 * shadow_walk_exact instead: one whose first accepted hit lies in [0.99, 1] (the reference walks on with C = t
 * and may skip siblings), and one whose operands are outside the ranges of the fast slab test. */

/* The reference's walk for one shadow ray (t = 1 at the start), start to end, on one lane: true when it ends with
 * a hit below 0.99.  Stacks the PARENT's index for a pending left sibling; at the pop the sibling's box is taken
 * from the parent's record and tested again.  That gives the value the general walk would have stored: the value
 * of a passing box does not depend on t, and a box can only stop passing because t_in > t, which is the case the
 * stored value would have been skipped for (t = C after any hit). */
template <bool CULL, bool COUNT, bool FAST>
SRT_DI bool shadow_walk_exact(const SceneDev& sc, f3 o, f3 d, uint32_t* stk, Stats& st)
{
    RayRcp rr;
    rr.x = rr.y = rr.z = 0.0f;
    if (FAST)
        rr.x = refined_rcp(d.x), rr.y = refined_rcp(d.y), rr.z = refined_rcp(d.z);
    float T = 1.0f, C = SRT_FLT_MAX, val;
    bool inside;
    const bool pass = FAST ? slab_test_fast(sc.root_lo, sc.root_hi, o, d, rr, T, val, inside) : slab_test(sc.root_lo, sc.root_hi, o, d, T, val, inside);
    if (COUNT)
        st.box += 1;
    if (!(val < C && (pass || !CULL)))
        return false;
    uint32_t cur = sc.root_ref;
    int sp = 0;
    for (;;) {
        if (cur & SRT_LEAF_FLAG) {
            const uint32_t first = cur & 0x00FFFFFFu;
            const uint32_t cnt = (cur >> 24) & 0x7Fu;
            for (uint32_t k = 0; k < cnt; ++k) {
                if (COUNT)
                    st.tri += 1;
                if (tri_test(tri_record(sc, first + k), o, d, T))
                    C = min_std(C, T);
            }
            if (T < 0.99f && C != SRT_FLT_MAX)
                return true;
        } else {
            if (COUNT) {
                st.inner += 1;
                st.box += 2;
            }
            const float4* nd = node_record(sc, cur);
            const float4 a = nd[0], b = nd[1], c = nd[2], e = nd[3];
            const float llo[3] = { a.x, a.y, a.z }, lhi[3] = { a.w, b.x, b.y };
            const float rlo[3] = { b.z, b.w, c.x }, rhi[3] = { c.y, c.z, c.w };
            float lval, rval;
            bool lin, rin;
            const bool lpass = FAST ? slab_test_fast(llo, lhi, o, d, rr, T, lval, lin) : slab_test(llo, lhi, o, d, T, lval, lin);
            const bool rpass = FAST ? slab_test_fast(rlo, rhi, o, d, rr, T, rval, rin) : slab_test(rlo, rhi, o, d, T, rval, rin);
            const bool pushL = lval < C && (lpass || !CULL);
            const bool pushR = rval < C && (rpass || !CULL);
            if (pushL && pushR) {
                stk[sp * SRT_SHADOW_BLOCK] = cur;
                ++sp;
            }
            if (pushR) {
                cur = __float_as_uint(e.y);
                continue;
            }
            if (pushL) {
                cur = __float_as_uint(e.x);
                continue;
            }
        }
        bool found = false;
        while (sp > 0 && !found) {
            --sp;
            const float4* nd = node_record(sc, stk[sp * SRT_SHADOW_BLOCK]);
            const float4 a = nd[0], b = nd[1];
            const float llo[3] = { a.x, a.y, a.z }, lhi[3] = { a.w, b.x, b.y };
            float lval;
            bool lin;
            const bool lpass = FAST ? slab_test_fast(llo, lhi, o, d, rr, T, lval, lin) : slab_test(llo, lhi, o, d, T, lval, lin);
            if (CULL && !lpass)
                continue; /* t_in > t = C by now: the stored t_in would have been skipped */
            const float v = CULL ? lval : ((lpass && lin) ? lval : -1.0f);
            if (!(v > C)) {
                cur = __float_as_uint(nd[3].x);
                found = true;
            }
        }
        if (!found)
            return false; /* a hit, if any, lies in [0.99, 1] */
    }
}

SRT_DI void swalk_pop(Walk& w, const uint32_t* stk)
{
    if (w.top == stk) {
        w.cur = SRT_WALK_DONE;
        return;
    }
    w.top -= SRT_SHADOW_BLOCK;
    w.cur = *w.top;
}

/* walk_inner with t = 1 and C = FLT_MAX: every value of a box in the fast ranges is finite, so the push test
 * `value < C` holds and a child is entered when its box passes (or always, without miss culling). */
template <bool CULL, bool COUNT>
SRT_DI void swalk_inner(const SceneDev& sc, Walk& w, uint32_t* stk, Stats& st)
{
    if (COUNT) {
        st.inner += 1;
        st.box += 2;
    }
    const float4* nd = node_record(sc, w.cur);
    const float4 a = nd[0], b = nd[1], c = nd[2], e = nd[3];
    const float llo[3] = { a.x, a.y, a.z }, lhi[3] = { a.w, b.x, b.y };
    const float rlo[3] = { b.z, b.w, c.x }, rhi[3] = { c.y, c.z, c.w };
    const uint32_t left = __float_as_uint(e.x), right = __float_as_uint(e.y);
    float lval, rval;
    bool lin, rin;
    const bool pushL = slab_test_fast(llo, lhi, w.o, w.d, w.rr, 1.0f, lval, lin) || !CULL;
    const bool pushR = slab_test_fast(rlo, rhi, w.o, w.d, w.rr, 1.0f, rval, rin) || !CULL;
    if (pushL && pushR) {
        *w.top = left;
        w.top += SRT_SHADOW_BLOCK;
        w.cur = right;
    } else if (pushR) {
        w.cur = right;
    } else if (pushL) {
        w.cur = left;
    } else {
        swalk_pop(w, stk);
    }
}

/* ---- coarse, four-wide nodes for shadow queries.
 * Before its first hit a shadow query asks of a box only whether the ray meets it.  Any box that contains the
 * child's box answers that question safely (a superset of the nodes is visited, the triangle tests decide), so the
 * shadow walk reads boxes on a 16-bit grid over the root box, rounded outwards by more than a grid step: 12 bytes per
 * box instead of 24, and the slab values by one fused multiply-add per coordinate in grid space, without the exact
 * division the closest-hit walks need.  The margin of one grid step on either side covers the rounding of everything
 * involved: the grid coordinate of a bound or of the origin is off by less than 0.02 steps, a slab value by less
 * than 0.03 steps of travel along its axis for an origin within one root-box size of the root box (other rays take
 * the exact walk).
 * Two levels of the tree are folded into one record.  A child's box lies inside its parent's, outward rounding and
 * the grid-space slab values are monotone, so a ray that meets a grandchild's coarse box meets the coarse box of the
 * child in between: testing the four grandchildren directly visits exactly the leaves the level-by-level walk on the
 * same grid visits.  The reference's walk pops the right child before the left one (src/bounding_volume_hierarchy.cpp:
 * 370-385), so the grandchildren are met in the order right-right, right-left, left-right, left-left; the record
 * lists them in that order (a child that is a leaf takes one slot), the walk goes on in the first one the ray meets
 * and stacks the others, last one first.  A query therefore takes half the dependent node visits, for the same 64
 * bytes per two levels and never more box tests. */
#define SRT_WIDE_EMPTY 0xFFFFFFFFu /* no leaf reference has this value: first + count would pass 2^24 triangles */
#define SRT_WIDE_NONE 0xFFFFFFFFu  /* wide_of[] of an inner node of odd depth */

SRT_DI uint32_t coarse_axis(float lo, float hi, float rlo, float s, bool& bad)
{
    const float ul = (lo - rlo) * s, uh = (hi - rlo) * s;
    if (!(ul >= 0.0f && uh <= 65534.0f && ul <= uh))
        bad = true; /* outside the root box, or NaN */
    const float ql = __builtin_fminf(__builtin_fmaxf(__builtin_floorf(ul), 0.0f), 65535.0f);
    const float qh = __builtin_fminf(__builtin_fmaxf(__builtin_ceilf(uh) + 2.0f, 0.0f), 65535.0f);
    return (uint32_t)ql | ((uint32_t)qh << 16);
}

struct GridMap {
    float lo[3], scale[3];
};

SRT_DI uint4 coarse_slot(f3 lo, f3 hi, uint32_t ref, const GridMap& g, bool& bad)
{
    uint4 s;
    s.x = coarse_axis(lo.x, hi.x, g.lo[0], g.scale[0], bad);
    s.y = coarse_axis(lo.y, hi.y, g.lo[1], g.scale[1], bad);
    s.z = coarse_axis(lo.z, hi.z, g.lo[2], g.scale[2], bad);
    s.w = ref;
    return s;
}

/* One thread per inner node; those of even depth (wide_of[i] is their position in the wide array) write a record.
 * When the triangles of the record's leaves fit the four slots one by one, every triangle gets a slot of its own, with
 * its own box, in the order the leaves hold them (the reference tests a leaf's triangles in stored order,
 * src/bounding_volume_hierarchy.cpp:359-368): the bottom records of a depth-capped tree then cull single triangles
 * before any triangle record is read. */
__global__ __launch_bounds__(SRT_BLOCK) void k_widen(const float4* __restrict__ nodes, uint32_t n, const uint32_t* __restrict__ wide_of,
                                                       const float4* __restrict__ tris, GridMap g, uint4* __restrict__ out, uint32_t* bad_flag)
{
    const uint32_t i = blockIdx.x * SRT_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t wid = wide_of[i];
    if (wid == SRT_WIDE_NONE)
        return;
    bool bad = false;
    f3 lo[4], hi[4];
    uint32_t ref[4];
    int used = 0;
    const float4 a = nodes[4ull * i], b = nodes[4ull * i + 1], c = nodes[4ull * i + 2], e = nodes[4ull * i + 3];
    /* right child first: the order the reference pops them in */
    const f3 clo[2] = { mk3(b.z, b.w, c.x), mk3(a.x, a.y, a.z) }, chi[2] = { mk3(c.y, c.z, c.w), mk3(a.w, b.x, b.y) };
    const uint32_t cref[2] = { __float_as_uint(e.y), __float_as_uint(e.x) };
    for (int side = 0; side < 2; ++side) {
        if (cref[side] & SRT_LEAF_FLAG) {
            lo[used] = clo[side], hi[used] = chi[side], ref[used] = cref[side];
            used += 1;
            continue;
        }
        const float4* cn = nodes + 4ull * cref[side];
        const float4 p = cn[0], q = cn[1], r = cn[2], t = cn[3];
        const uint32_t gr = __float_as_uint(t.y), gl = __float_as_uint(t.x);
        lo[used] = mk3(q.z, q.w, r.x), hi[used] = mk3(r.y, r.z, r.w), ref[used] = (gr & SRT_LEAF_FLAG) ? gr : wide_of[gr];
        used += 1;
        lo[used] = mk3(p.x, p.y, p.z), hi[used] = mk3(p.w, q.x, q.y), ref[used] = (gl & SRT_LEAF_FLAG) ? gl : wide_of[gl];
        used += 1;
    }
    uint32_t singles = 0;
    bool anyLeaf = false;
    for (int k = 0; k < used; ++k) {
        const bool leaf = (ref[k] & SRT_LEAF_FLAG) != 0;
        anyLeaf = anyLeaf || leaf;
        singles += leaf ? ((ref[k] >> 24) & 0x7Fu) : 1u;
    }
    uint4 slot[4];
    for (int k = 0; k < 4; ++k)
        slot[k] = make_uint4(0x0000FFFFu, 0x0000FFFFu, 0x0000FFFFu, SRT_WIDE_EMPTY); /* lower 65535, upper 0: never met */
    int w = 0;
    for (int k = 0; k < used; ++k) {
        const uint32_t cnt = (ref[k] >> 24) & 0x7Fu;
        if (anyLeaf && singles <= 4u && (ref[k] & SRT_LEAF_FLAG) && cnt != 1u) {
            /* one slot per triangle (an empty leaf leaves none) */
            const uint32_t first = ref[k] & 0x00FFFFFFu;
            for (uint32_t j = 0; j < cnt; ++j) {
                const float4* rec = tris + 6ull * (first + j);
                const f3 v0 = xyz(rec[1]), v1 = xyz(rec[2]), v2 = xyz(rec[3]);
                const f3 tlo = mk3(min_std(min_std(v0.x, v1.x), v2.x), min_std(min_std(v0.y, v1.y), v2.y), min_std(min_std(v0.z, v1.z), v2.z));
                const f3 thi = mk3(max_std(max_std(v0.x, v1.x), v2.x), max_std(max_std(v0.y, v1.y), v2.y), max_std(max_std(v0.z, v1.z), v2.z));
                slot[w++] = coarse_slot(tlo, thi, SRT_LEAF_FLAG | (1u << 24) | (first + j), g, bad);
            }
        } else {
            slot[w++] = coarse_slot(lo[k], hi[k], ref[k], g, bad);
        }
    }
    for (int k = 0; k < 4; ++k)
        out[4ull * wid + k] = slot[k];
    if (bad)
        atomicOr(bad_flag, 1u);
}

/* The shadow walk's 48-byte triangle records (tri_test_compact): unit normal and vertices of every traversal record, in
 * the same order.  A record only qualifies if what the compact test recomputes is what the full record stores: D equal
 * as a number, the three second operands of the inside tests bit for bit.  Records made by k_build_triangles or by the
 * host mirror's flatten() always do; a table uploaded with other values raises the flag and the scene keeps the full
 * records for its shadow walk. */
__global__ __launch_bounds__(SRT_BLOCK) void k_compact_triangles(const float4* __restrict__ tris, uint32_t n, float4* __restrict__ out, uint32_t* bad_flag)
{
    const uint32_t i = blockIdx.x * SRT_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    const float4* rec = tris + 6ull * i;
    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4], q5 = rec[5];
    const f3 nrm = xyz(q0), v0 = xyz(q1), v1 = xyz(q2), v2 = mk3(q3.x, q3.y, q3.z);
    const f3 e21 = v2 - v1, e20 = v2 - v0, e10 = v1 - v0;
    const f3 c0 = cross3(e21, v0 - v1), c1 = cross3(e20, e10), c2 = cross3(e10, e20);
    const float stored[9] = { q3.w, q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w };
    const float made[9] = { c0.x, c0.y, c0.z, c1.x, c1.y, c1.z, c2.x, c2.y, c2.z };
    bool same = dot3(v0, nrm) == q0.w;
    for (int k = 0; k < 9; ++k)
        same = same && __float_as_uint(stored[k]) == __float_as_uint(made[k]);
    out[3ull * i] = make_float4(nrm.x, nrm.y, nrm.z, v0.x);
    out[3ull * i + 1] = make_float4(v0.y, v0.z, v1.x, v1.y);
    out[3ull * i + 2] = make_float4(v1.z, v2.x, v2.y, v2.z);
    if (!same)
        atomicOr(bad_flag, 1u);
}

/* Grid-space constants of a query's ray; false when the origin is too far from the root box for the margin to hold. */
SRT_DI bool coarse_ray(const SceneDev& sc, Walk& w)
{
    const float ox = (w.o.x - sc.root_lo[0]) * sc.q_scale[0] + 1.0f, oy = (w.o.y - sc.root_lo[1]) * sc.q_scale[1] + 1.0f, oz = (w.o.z - sc.root_lo[2]) * sc.q_scale[2] + 1.0f;
    w.qi.x = 1.0f / (w.d.x * sc.q_scale[0]), w.qi.y = 1.0f / (w.d.y * sc.q_scale[1]), w.qi.z = 1.0f / (w.d.z * sc.q_scale[2]);
    w.qc.x = -(ox * w.qi.x), w.qc.y = -(oy * w.qi.y), w.qc.z = -(oz * w.qi.z);
    w.qsx = w.qi.x < 0.0f ? 16u : 0u, w.qsy = w.qi.y < 0.0f ? 16u : 0u, w.qsz = w.qi.z < 0.0f ? 16u : 0u;
    const float lim_lo = -65536.0f, lim_hi = 131072.0f;
    return ox >= lim_lo && ox <= lim_hi && oy >= lim_lo && oy <= lim_hi && oz >= lim_lo && oz <= lim_hi;
}

/* Does the ray meet the slot's box within t in [0, 1]?  Each axis word holds lower | upper << 16; rotating it by 16
 * where the ray runs towards smaller coordinates puts the NEAR bound into the low half, so t_in is the largest of the
 * three low-half slab values and t_out the smallest of the three high-half ones: what the per-axis min / max of the
 * slab test select (the slab value is monotone in the bound), without computing them.  An unused slot holds
 * lower = 65535, upper = 0 on every axis: its t_in exceeds its t_out whatever the ray, so it is never met. */
SRT_DI bool coarse_meets(const uint4& b, const Walk& w)
{
    const uint32_t bx = __builtin_amdgcn_alignbit(b.x, b.x, w.qsx), by = __builtin_amdgcn_alignbit(b.y, b.y, w.qsy), bz = __builtin_amdgcn_alignbit(b.z, b.z, w.qsz);
    const float xn = __builtin_fmaf((float)(bx & 0xFFFFu), w.qi.x, w.qc.x), xf = __builtin_fmaf((float)(bx >> 16), w.qi.x, w.qc.x);
    const float yn = __builtin_fmaf((float)(by & 0xFFFFu), w.qi.y, w.qc.y), yf = __builtin_fmaf((float)(by >> 16), w.qi.y, w.qc.y);
    const float zn = __builtin_fmaf((float)(bz & 0xFFFFu), w.qi.z, w.qc.z), zf = __builtin_fmaf((float)(bz >> 16), w.qi.z, w.qc.z);
    const float tin = __builtin_fmaxf(xn, __builtin_fmaxf(yn, zn));
    const float tout = __builtin_fminf(xf, __builtin_fminf(yf, zf));
    return !(tin > tout || tout < 0.0f || tin > 1.0f);
}

/* Visit of the wide node the lane stands on (miss culling only: without it no box is tested at all): the walk goes on
 * in the first slot the ray meets and stacks the other met ones, last first, so that they come off the stack in slot
 * order.  Returns true, with the walk stopped, when the stack is too full for the visit (fewer than four of its
 * wide_cap entries free and something to stack): the caller then answers the query with the exact walk. */
template <bool COUNT>
SRT_DI bool swalk_inner_wide(const SceneDev& sc, const uint4* top, Walk& w, uint32_t* stk, Stats& st)
{
    if (COUNT) {
        st.inner += 1;
        st.box += 4;
    }
    uint4 s0, s1, s2, s3;
    if (SRT_WIDE_LDS_NODES > 0u && w.cur < sc.n_top) {
        const uint4* nd = top + 4u * w.cur; /* the block's LDS copy of the top levels */
        s0 = nd[0], s1 = nd[1], s2 = nd[2], s3 = nd[3];
    } else {
        const uint4* nd = wide_record(sc, w.cur);
        s0 = nd[0], s1 = nd[1], s2 = nd[2], s3 = nd[3];
    }
    /* all four slots are tested, without branches: the record's four quarters are requested together (an unused slot
     * fails the test by its box) */
    const bool m0 = coarse_meets(s0, w), m1 = coarse_meets(s1, w), m2 = coarse_meets(s2, w), m3 = coarse_meets(s3, w);
    /* logical operators on the finished tests: the lane masks are then combined in scalar registers (`&` / `|` on bools
     * are integer operations: the masks would be turned into 0 / 1 values in vector registers, combined there and
     * compared again, a dozen vector instructions per visit; shadow launch 98.3 -> 95.5 ms) */
    const bool m01 = m0 || m1, m012 = m01 || m2;
    const bool p1 = m1 && m0, p2 = m2 && m01, p3 = m3 && m012;
    /* A visit that stacks anything needs room for three entries and one to spare, whatever it stacks: the three stores
     * below are then unconditional (an entry that is not wanted is overwritten by the next store or never read; a visit
     * that stacks nothing stores to the entry above the top, which the spare one guarantees to exist) and the only
     * branch left is this one. */
    uint32_t* t = w.top;
    if ((p1 || p2 || p3) && (t + 3 * SRT_SHADOW_BLOCK >= stk + sc.wide_cap * SRT_SHADOW_BLOCK)) {
        w.cur = SRT_WALK_DONE;
        return true;
    }
    /* the stack position is kept as the entry's address and moved by selects on the tests' lane masks: the masks are
     * combined in scalar registers and never become 0 / 1 values in vector ones */
    *t = s3.w;
    t = p3 ? t + SRT_SHADOW_BLOCK : t;
    *t = s2.w;
    t = p2 ? t + SRT_SHADOW_BLOCK : t;
    *t = s1.w;
    t = p1 ? t + SRT_SHADOW_BLOCK : t;
    w.top = t;
    if (m012 || m3)
        w.cur = m0 ? s0.w : (m1 ? s1.w : (m2 ? s2.w : s3.w));
    else
        swalk_pop(w, stk);
    return false;
}

/* Leaf visit of the common shadow walk.  Returns 0: no hit, walk on; 1: occluded; 2: first hit in [0.99, 1]. */
template <bool COUNT, bool COMPACT>
SRT_DI int swalk_leaf(const SceneDev& sc, Walk& w, const uint32_t* stk, Stats& st)
{
    const uint32_t first = w.cur & 0x00FFFFFFu;
    const uint32_t cnt = (w.cur >> 24) & 0x7Fu;
    float T = 1.0f;
    bool hit = false;
    /* One triangle after the other; most visits are of a single triangle (the wide records list the triangles of small
     * leaves one by one).  COMPACT: on the scene's 48-byte shadow records (normal and vertices; three loads per test, the
     * plane offset and the second operands of the inside test recomputed), else on the full 96-byte traversal records. */
    if (COMPACT) {
        for (uint32_t k = 0; k < cnt; ++k) {
            if (COUNT)
                st.tri += 1;
            if (tri_test_compact(shadow_tri_record(sc, first + k), w.o, w.d, T))
                hit = true;
        }
    } else {
        for (uint32_t k = 0; k < cnt; ++k) {
            if (COUNT)
                st.tri += 1;
            if (tri_test_flat(tri_record(sc, first + k), w.o, w.d, T))
                hit = true;
        }
    }
    if (hit) {
        w.cur = SRT_WALK_DONE;
        return T < 0.99f ? 1 : 2;
    }
    swalk_pop(w, stk);
    return 0;
}

/* Shadow queries of one recursion depth: item i = hit (i / shadows_per_hit), shadow sample (i % shadows_per_hit)
 * in the order the reference's light loop issues them; the answer is one byte per item (1 = visible, 0 = occluded,
 * 2 = occluded and the shade kernel must still evaluate the sample's term: see light_contribution).
 *
 * Shadow queries differ wildly in length (an occluded one stops at its first hit, a free one walks every box
 * around its origin), so a wave that started 64 of them together would soon run mostly empty.  Instead a
 * lane whose query ends hands in its answer and, as soon as 24 lanes of the wave are idle (SRT_REFILL_MIN; 16 to 32
 * measure within 2 %), the idle lanes are given the wave's next items.  A wave draws its items in chunks of
 * consecutive items, consecutive items going to consecutive idle lanes, so its lanes keep working on neighbouring
 * samples of the same few hits (same origin, nearby targets on the light). */
#ifndef SRT_SHADOW_WAVES
/* waves per SIMD the register allocation aims at.  Six: 85 registers, nothing spilled, and room in LDS for four levels of
 * top records; with the walk's instruction stream as short as it is now, 5, 6 and 7 waves measure within 0.5 % of each
 * other (7 waves spill nine registers in the refill path). */
#define SRT_SHADOW_WAVES (SRT_SHADOW_BLOCK >= 1024 ? 4 : 6)
#endif
template <bool CULL, bool COUNT, bool COARSE>
__global__ __launch_bounds__(SRT_SHADOW_BLOCK, SRT_SHADOW_WAVES) void k_trace_shadow(SceneDev sc, RenderDev rp, int depth, HitQ hq, const uint32_t* hq_count,
                                                              const float4* cur_paths, unsigned char* vis, uint32_t shadows_per_hit, DevCounters* cnt,
                                                              unsigned long long* next_chunk)
{
    /* LDS: the first lights of the scene (a refill reads its light from here instead of global memory), the wide
     * records of the tree's top levels, then the per-lane stacks */
    extern __shared__ uint32_t lds_shadow[];
    const srt_light* lds_lights = (const srt_light*)lds_shadow;
    const uint32_t nLdsLights = sc.n_lights < SRT_LDS_LIGHTS ? sc.n_lights : SRT_LDS_LIGHTS;
    for (uint32_t i = threadIdx.x; i < nLdsLights * (uint32_t)(sizeof(srt_light) / 4); i += SRT_SHADOW_BLOCK)
        lds_shadow[i] = ((const uint32_t*)sc.lights)[i];
    uint4* lds_top = (uint4*)(lds_shadow + SRT_LDS_LIGHTS * (uint32_t)(sizeof(srt_light) / 4));
    const uint32_t nTop = COARSE ? sc.n_top : 0u;
    for (uint32_t i = threadIdx.x; i < nTop * 4u; i += SRT_SHADOW_BLOCK)
        lds_top[i] = sc.wnodes[i];
    __syncthreads();
    uint32_t* stk = lds_shadow + SRT_LDS_LIGHTS * (uint32_t)(sizeof(srt_light) / 4) + SRT_WIDE_LDS_NODES * 16u + threadIdx.x;
    Stats st = { 0, 0, 0, 0, 0 };
    uint32_t nShadow = 0;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long n = (unsigned long long)(*hq_count) * shadows_per_hit;
    /* Items are handed out in chunks.  A wave's first chunk follows from its position in the grid; every
     * further one is drawn from a counter all waves share (one atomic per chunk), so waves that happen to get
     * long queries simply draw fewer chunks: with a fixed stride instead, the slowest wave's share decided the
     * launch time (measured 220 -> 200 ms on the dragon frame). */
    /* Chunk size: every draw is an atomic on one address, and all waves of the GPU drawing 64 items at a time kept that
     * address busier than the caches (14 M draws per dragon frame: the launch took 172 ms whatever else was changed,
     * 150 ms with 256 items per draw).  Large launches therefore draw up to 1024 items at a time, small ones keep 64 so
     * that their few items still spread over the waves: at least 16 draws per wave. */
    uint32_t chunkItems = 64u;
    {
        const unsigned long long per = n / ((unsigned long long)gridDim.x * (SRT_SHADOW_BLOCK / 64) * 16ull);
        while (chunkItems < SRT_SHADOW_CHUNK_MAX && 2ull * chunkItems <= per)
            chunkItems *= 2u;
        chunkItems = __builtin_amdgcn_readfirstlane(chunkItems);
    }
    const unsigned long long first_shared = (unsigned long long)gridDim.x * (SRT_SHADOW_BLOCK / 64) * chunkItems;
    /* chunk, used and dry are the same in all lanes of a wave; reading them through readfirstlane tells the compiler
     * so, and the refill conditions and the loop's exit become scalar branches instead of exec-mask bookkeeping
     * in every trip */
    auto uniform = [](unsigned long long v) {
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    };
    auto grab = [&]() {
        unsigned long long base = 0;
        if (lane == 0)
            base = atomicAdd(next_chunk, (unsigned long long)chunkItems);
        return first_shared + uniform(base); /* lane 0 is the first active lane: no lane leaves the loop before the wave does */
    };
    unsigned long long chunk = uniform(((unsigned long long)blockIdx.x * (SRT_SHADOW_BLOCK / 64) + (threadIdx.x >> 6)) * chunkItems);
    /* hit and sample number of a chunk's first item: one 64-bit division per chunk; the items of the chunk then
     * need 32-bit arithmetic only */
    uint32_t hitBase = 0, sampleBase = 0;
    auto locate = [&](unsigned long long c) {
        const unsigned long long q = c / shadows_per_hit;
        hitBase = __builtin_amdgcn_readfirstlane((uint32_t)q);
        sampleBase = __builtin_amdgcn_readfirstlane((uint32_t)(c - q * shadows_per_hit));
    };
    locate(chunk);
    bool dry = false;                                                                              /* the items have run out */
    uint32_t used = 0; /* items taken from the current chunk */
    Walk w;
    w.alive = false;
    w.cur = SRT_WALK_DONE, w.sp = 0, w.top = stk, w.T = 1.0f, w.C = SRT_FLT_MAX, w.hitTri = 0;
    unsigned long long item = 0;
    bool busy = false;
    bool redo = false; /* the lane's query cannot take the common walk */
    for (;;) {
        /* at this point a lane is idle exactly when its walk has ended (the answer of an ended walk is handed in at the
         * bottom of the same trip) */
        const unsigned long long idle = ballot64(w.cur == SRT_WALK_DONE);
        if (!dry && __popcll(idle) >= SRT_REFILL_MIN && used == chunkItems) {
            chunk = grab();
            locate(chunk);
            used = 0;
        }
        dry = dry || chunk >= n;
        if (!dry && __popcll(idle) >= SRT_REFILL_MIN) {
            const uint32_t cnt = (uint32_t)__popcll(idle);
            const uint32_t pos = used + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            unsigned long long mine = chunk + pos;
            uint32_t hb = hitBase, within = sampleBase + pos;
            used += cnt;
            if (used > chunkItems) {
                /* the idle lanes outnumber what is left of the chunk: the rest come from the next one */
                chunk = grab();
                locate(chunk);
                used -= chunkItems;
                if (pos >= chunkItems) {
                    mine = chunk + (pos - chunkItems);
                    hb = hitBase, within = sampleBase + (pos - chunkItems);
                }
            }
            bool traced = false;
            if (!busy && mine < n) {
                item = mine;
                const uint32_t over = within / shadows_per_hit; /* hits between the chunk's first item and this one */
                const uint32_t h = hb + over;
                uint32_t k = within - over * shadows_per_hit;
                const float4 hp = hq.ip[h]; /* hit point and sampler key, left by the trace kernel */
                /* which light the sample belongs to: the lights in LDS first, in a loop of their own (one loop over both
                 * kinds reads the type through a flat load and waits for every load in flight, the hit point's included) */
                uint32_t li = 0;
                bool found = false;
                for (; li < nLdsLights; ++li) {
                    const uint32_t c = shadow_samples(lds_lights[li].type, rp.features);
                    if (k < c) {
                        found = true;
                        break;
                    }
                    k -= c;
                }
                if (!found) {
                    for (; li < sc.n_lights; ++li) {
                        const uint32_t c = shadow_samples(sc.lights[li].type, rp.features);
                        if (k < c)
                            break;
                        k -= c;
                    }
                }
                srt_light L;
                if (li < nLdsLights)
                    L = lds_lights[li];
                else
                    L = sc.lights[li];
                const uint32_t path = __float_as_uint(hp.w);
                f3 pos3, col;
                light_sample(&L, rp.seed, path, (uint32_t)depth, li, k, pos3, col);
                /* testVisibilityLightSample, reference src/light.cpp:54-62 */
                const f3 ip = mk3(hp.x, hp.y, hp.z);
                if (ip.x == pos3.x && ip.y == pos3.y && ip.z == pos3.z) {
                    vis[item] = 1;
                } else {
                    const f3 dir = pos3 - ip;
                    const f3 org = ip + normalize3(dir) * 0.001f;
                    traced = true;
                    if (COARSE && (rp.features & SRT_F_ACCEL) && ray_in_fast_range(sc, org, dir)) {
                        /* The walk on the wide records starts without a test of the root box: a ray that misses it
                         * meets none of the root record's slots (or a rounded-out one, and then only finds what the
                         * triangle tests accept), so the test - six exact quotients per query - decides nothing. */
                        w.o = org, w.d = dir, w.T = 1.0f, w.C = SRT_FLT_MAX;
                        w.cur = sc.root_ref, w.top = stk;
                        busy = true;
                        if (!coarse_ray(sc, w))
                            w.cur = SRT_WALK_DONE, redo = true; /* answered by the exact walk below */
                    } else if ((rp.features & SRT_F_ACCEL) && ray_in_fast_range(sc, org, dir)) {
                        walk_begin<CULL, COUNT, true>(sc, w, org, dir, 1.0f, st);
                        w.top = stk;
                        busy = w.alive;
                        if (!busy) {
                            w.cur = SRT_WALK_DONE;
                            vis[item] = 1; /* the ray misses the root box */
                        }
                    } else {
                        /* An occluded sample is stored as 0 when the shade kernel may skip its Phong term: the direction
                         * to the light has an ordinary length (always so for rays in the ranges of the fast slab test),
                         * so the term's own normalisation of it is finite.  Otherwise as 2: occluded, term to be computed. */
                        const float dd = dot3(dir, dir);
                        const unsigned char dark = (dd >= 1e-30f && dd <= 1e30f) ? 0 : 2;
                        if (rp.features & SRT_F_ACCEL) {
                            vis[item] = shadow_walk_exact<CULL, COUNT, false>(sc, org, dir, stk, st) ? dark : 1;
                        } else {
                            float Ts = 1.0f;
                            uint32_t ht;
                            /* `newRay.t < 1 - 0.01` is a double comparison in the reference; for a float t it is t < 0.99f */
                            vis[item] = (traverse_linear<COUNT>(sc, org, dir, Ts, ht, st) && Ts < 0.99f) ? dark : 1;
                        }
                    }
                }
            }
            nShadow += (uint32_t)__builtin_popcountll(ballot64(traced)); /* counted per wave, in a scalar register */
        }
        /* one trip of the wave, as walk_trip: vote, then the kind of step more lanes are waiting for */
        const bool atLeaf = (int)w.cur < -1;
        const bool atInner = (int)w.cur >= 0;
        const unsigned long long mInner = ballot64(atInner), mLeaf = ballot64(atLeaf);
        if ((mInner | mLeaf) == 0 && !(COARSE && ballot64(redo) != 0)) {
            /* nothing in flight: either items remain (every lane is idle, so the next pass refills) or the wave is done */
            if (dry)
                break;
            continue;
        }
        int verdict = redo ? 2 : 0;
        redo = false;
        if (__popcll(mInner) * SRT_SVOTE_INNER_W >= __popcll(mLeaf) * SRT_SVOTE_LEAF_W) {
            if (COUNT && (int)lane == __ffsll((long long)mInner) - 1)
                st.wave_inner += 1;
            if (atInner) {
                if (COARSE) {
                    if (swalk_inner_wide<COUNT>(sc, lds_top, w, stk, st))
                        verdict = 2; /* more pending siblings than the stack holds */
                } else {
                    swalk_inner<CULL, COUNT>(sc, w, stk, st);
                }
            }
            /* Further box steps in the same trip while enough lanes stand on inner nodes: what a trip costs around its
             * step (idle count, refill conditions, the two votes, the hand-in of answers) is paid once for up to
             * six box steps (dragon shadow launch 138.8 -> 119.8 ms; 3 steps 121.5, 17 steps 123.0; thresholds 8 to 24
             * lanes measure the same, 36 lose the gain). */
            for (int rep = 0; rep < SRT_SHADOW_MORE_STEPS; ++rep) {
                const unsigned long long mAgain = ballot64((int)w.cur >= 0) & mInner; /* lanes of this trip's box step still on an inner node */
                const bool again = atInner && (int)w.cur >= 0;
                if ((uint32_t)__builtin_popcountll(mAgain) < SRT_SHADOW_MORE_MIN)
                    break;
                if (COUNT && (int)lane == __ffsll((long long)mAgain) - 1)
                    st.wave_inner += 1;
                if (again) {
                    if (COARSE) {
                        if (swalk_inner_wide<COUNT>(sc, lds_top, w, stk, st))
                            verdict = 2;
                    } else {
                        swalk_inner<CULL, COUNT>(sc, w, stk, st);
                    }
                }
            }
        } else {
            if (COUNT && (int)lane == __ffsll((long long)mLeaf) - 1)
                st.wave_leaf += 1;
            /* (one branch on a scalar per leaf step; the coarse walk's scenes all but always have the compact records) */
            if (atLeaf)
                verdict = (COARSE && sc.stris) ? swalk_leaf<COUNT, true>(sc, w, stk, st) : swalk_leaf<COUNT, false>(sc, w, stk, st);
        }
        if (busy && w.cur == SRT_WALK_DONE) {
            if (verdict == 2)
                vis[item] = shadow_walk_exact<CULL, COUNT, true>(sc, w.o, w.d, stk, st) ? 0 : 1;
            else
                vis[item] = verdict == 1 ? 0 : 1;
            busy = false;
        }
    }
    flush_counters(cnt, 0, lane == 0 ? nShadow : 0u, 0, st);
}

/* getFinalColor for the queued hits of one recursion depth (reference src/render.cpp:96-110).
 * Reflections: the reference returns Lo_0 + ks_0 * (Lo_1 + ks_1 * (...)), innermost product first.  A sample has one
 * ray per depth here, so acc[sample] holds the value of the deepest level reached so far, and a hit that sends on a
 * reflection ray parks its (Lo, ks) in `chain` (level-major: 6 floats at (level * n_samples + sample)), counts the
 * level in `nlev` and leaves zero in acc - the value of a reflection ray that hits nothing.  sample_value folds the
 * parked levels from the deepest up, with the reference's operations in the reference's order.
 * Transparency: a sample's rays form a binary tree instead; their shares are weighted by the product of the factors
 * on the way (thr) and added front to back (within rounding of the reference's nested sums). */
/* MODE 0: all of it.  A frame whose launches are few and short runs the two halves as kernels of their own, on two
 * streams (render_impl): MODE 1 only sends on the reflection rays (the hit's attributes, the ray, the level's ks), so
 * that the next depth's trace can start while this depth's shadow queries are still being answered; MODE 2 only
 * evaluates the light and puts it where MODE 0 would (never used with transparency). */
template <int MODE>
__global__ __launch_bounds__(SRT_BLOCK) void k_shade(SceneDev sc, RenderDev rp, int depth, HitQ hq, const uint32_t* hq_count, const float4* cur_paths,
                                                       float4* next_paths, uint32_t* next_count, float* acc, float* chain, unsigned char* nlev,
                                                       const unsigned char* vis, uint32_t shadows_per_hit, DevCounters* cnt)
{
    Stats st = { 0, 0, 0, 0, 0 };
    uint32_t nShaded = 0;
    const uint32_t n = *hq_count;
    const uint32_t stride = gridDim.x * SRT_BLOCK;
    for (uint32_t base = blockIdx.x * SRT_BLOCK + (threadIdx.x & ~63u); base < n; base += stride) {
        const uint32_t h = base + (threadIdx.x & 63u);
        bool spawn = false, through = false;
        f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 1), nthr = mk3(0, 0, 0);
        f3 to = mk3(0, 0, 0), td = mk3(0, 0, 1), tthr = mk3(0, 0, 0);
        uint32_t sample = 0, pathKey = 0;
        if (h < n) {
            f3 o, d, thr;
            float T;
            uint32_t tri, path;
            hit_ray(rp, depth, hq, cur_paths, h, o, d, thr, T, tri, sample, path);
            pathKey = path;
            Surface sf;
            surface_at(sc, rp.features, o, d, T, tri, sf);
            if (MODE != 1)
                nShaded += 1;
            f3 Lo = mk3(0, 0, 0);
            if (MODE != 1)
                Lo = light_contribution(sc, rp.features, rp.seed, path, (uint32_t)depth, o, d, T, sf, vis, (unsigned long long)h * shadows_per_hit);
            /* Share of this hit's own light in the sample.  With transparency (reference src/render.cpp:111-120) a
             * hit at the depth cap, or of a material with transparency above 1, returns black; a material with
             * transparency tr < 1 keeps tr of its light (and of its reflection) and passes 1 - tr on.  A hit that
             * returns black sends on no ray at all; the reference still follows its reflection and discards the result,
             * so the pixels are the same and the ray count is smaller. */
            const bool seeThrough = (rp.features & SRT_F_TRANSPARENCY) != 0;
            const float tr = sf.transparency;
            const bool black = seeThrough && !(depth < rp.max_depth && tr <= 1.0f);
            const float keep = (seeThrough && tr < 1.0f) ? tr : 1.0f;
            const f3 own = thr * keep;
            float* a3 = acc + 3ull * sample;
            const bool sendsOn = !black && (rp.features & SRT_F_RECURSIVE) && depth < rp.max_depth && !(sf.ks.x == 0.0f && sf.ks.y == 0.0f && sf.ks.z == 0.0f);
            if (MODE == 2) {
                /* the light of a hit that sends on a reflection ray is parked beside the ks the other half parked */
                float* c6 = chain + ((size_t)depth * rp.n_samples + sample) * 6;
                float* dst = sendsOn ? c6 : a3;
                dst[0] = Lo.x, dst[1] = Lo.y, dst[2] = Lo.z;
            } else if (MODE == 0 && !black) {
                if (depth == 0) {
                    a3[0] = keep * Lo.x, a3[1] = keep * Lo.y, a3[2] = keep * Lo.z;
                } else if (!seeThrough) {
                    a3[0] = Lo.x, a3[1] = Lo.y, a3[2] = Lo.z;
                } else {
                    /* several rays of one sample can be at the same depth: their shares are added atomically
                     * (the order of the additions, hence the last bits of the sum, may vary between runs) */
                    atomicAdd(a3 + 0, own.x * Lo.x);
                    atomicAdd(a3 + 1, own.y * Lo.y);
                    atomicAdd(a3 + 2, own.z * Lo.z);
                }
            }
            if (MODE != 2 && sendsOn) {
                /* computeReflectionRay, reference src/shading.cpp:33-51; offset = float(pow(10, -5)) */
                const float offset = 9.99999975e-06f;
                const f3 dn = normalize3(d);
                const f3 nn = normalize3(sf.normal);
                rd = dn - ((2.0f * dot3(dn, nn)) * nn);
                ro = (o + d * T) + (offset * rd);
                nthr = own * sf.ks;
                spawn = true;
                if (!seeThrough) {
                    float* c6 = chain + ((size_t)depth * rp.n_samples + sample) * 6;
                    if (MODE == 0)
                        c6[0] = Lo.x, c6[1] = Lo.y, c6[2] = Lo.z;
                    c6[3] = sf.ks.x, c6[4] = sf.ks.y, c6[5] = sf.ks.z;
                    nlev[sample] = (unsigned char)(depth + 1);
                    a3[0] = a3[1] = a3[2] = 0.0f;
                }
            }
            if (MODE == 0 && !black && seeThrough && tr < 1.0f) {
                /* the ray continued behind the surface, reference src/render.cpp:113 */
                to = o + d * (0.000001f + T);
                td = d;
                tthr = thr * (1 - tr);
                through = true;
            }
        }
        if (MODE == 2)
            continue; /* no successor rays from this half (wave-uniform: every lane takes it) */
        const uint32_t q = queue_slot(spawn, next_count);
        if (spawn) {
            next_paths[3ull * q] = make_float4(ro.x, ro.y, ro.z, rd.x);
            next_paths[3ull * q + 1] = make_float4(rd.y, rd.z, nthr.x, nthr.y);
            next_paths[3ull * q + 2] = make_float4(nthr.z, __uint_as_float(sample), __uint_as_float(pathKey), 0.0f);
        }
        if (rp.features & SRT_F_TRANSPARENCY) {
            const uint32_t q2 = queue_slot(through, next_count);
            if (through) {
                next_paths[3ull * q2] = make_float4(to.x, to.y, to.z, td.x);
                next_paths[3ull * q2 + 1] = make_float4(td.y, td.z, tthr.x, tthr.y);
                next_paths[3ull * q2 + 2] = make_float4(tthr.z, __uint_as_float(sample), __uint_as_float(pathKey), 0.0f);
            }
        }
    }
    flush_counters(cnt, 0, 0, nShaded, st);
}

/* getFinalColor's return value for one sample: the deepest level's value, folded into the levels that sent on a
 * reflection ray (`Lo += ks * getFinalColor(...)`, reference src/render.cpp:109), innermost first. */
SRT_DI f3 sample_value(const RenderDev& rp, const float* acc, const float* chain, const unsigned char* nlev, uint32_t s)
{
    f3 v = mk3(acc[3ull * s], acc[3ull * s + 1], acc[3ull * s + 2]);
    if (chain) {
        for (int level = (int)nlev[s] - 1; level >= 0; --level) {
            const float* c6 = chain + ((size_t)level * rp.n_samples + s) * 6;
            v = mk3(c6[0] + c6[3] * v.x, c6[1] + c6[4] * v.y, c6[2] + c6[5] * v.z);
        }
    }
    return v;
}

/* Pixel value from its samples (reference src/render.cpp:213-217) stored at Screen::setPixel's position. */
__global__ __launch_bounds__(SRT_BLOCK) void k_resolve(RenderDev rp, const float* acc, const float* chain, const unsigned char* nlev)
{
    if (!rp.custom_rays && !rp.lens_mode && !chain && (rp.features & SRT_F_MULTISAMPLE) && (rp.S & 3) == 0) {
        /* The common large frame (n x m samples, a multiple of four, no parked reflection levels): threads take the
         * pixels in the order their samples are stored, so a wave reads one contiguous stretch of the accumulators,
         * four samples (three 16-byte loads) at a time; same additions in the same order as below. */
        const uint32_t n_slots = blocks_x(rp) * (((uint32_t)rp.th + 7u) >> 3) * 64u;
        for (uint32_t q = blockIdx.x * SRT_BLOCK + threadIdx.x; q < n_slots; q += gridDim.x * SRT_BLOCK) {
            int px, py, sub;
            const uint32_t first = q * (uint32_t)rp.S;
            if (!sample_decode(rp, first, px, py, sub))
                continue; /* padding of a block at the rectangle's right or top edge */
            const float4* a4 = (const float4*)(acc + 3ull * first);
            f3 c = mk3(0, 0, 0);
            for (int s4 = 0; s4 < rp.S; s4 += 4) {
                const float4 u = a4[0], v = a4[1], w = a4[2];
                a4 += 3;
                c = c + mk3(u.x, u.y, u.z);
                c = c + mk3(u.w, v.x, v.y);
                c = c + mk3(v.z, v.w, w.x);
                c = c + mk3(w.y, w.z, w.w);
            }
            c = c / (float)rp.S;
            float* o = rp.out_rgb + 3 * out_index(rp, px, py);
            o[0] = c.x, o[1] = c.y, o[2] = c.z;
        }
        return;
    }
    const uint32_t n_px = (uint32_t)rp.tw * (uint32_t)rp.th;
    for (uint32_t p = blockIdx.x * SRT_BLOCK + threadIdx.x; p < n_px; p += gridDim.x * SRT_BLOCK) {
        const int py = rp.y0 + (int)(p / (uint32_t)rp.tw), px = rp.x0 + (int)(p % (uint32_t)rp.tw);
        f3 c = mk3(0, 0, 0);
        const uint32_t first = rp.custom_rays ? p : pixel_first_sample(rp, (uint32_t)(px - rp.x0), (uint32_t)(py - rp.y0));
        if (rp.lens_mode) {
            /* Lo += getFinalColor(...) / samples, reference src/render.cpp:158,175 (black with zero samples) */
            for (int s = 0; s < rp.lens.samples; ++s) {
                c = c + sample_value(rp, acc, chain, nlev, first + (uint32_t)s) / (float)rp.lens.samples;
            }
        } else if (rp.features & SRT_F_MULTISAMPLE) {
            for (int s = 0; s < rp.S; ++s) {
                c = c + sample_value(rp, acc, chain, nlev, first + (uint32_t)s);
            }
            c = c / (float)rp.S;
        } else {
            c = sample_value(rp, acc, chain, nlev, first);
        }
        float* o = rp.out_rgb + 3 * out_index(rp, px, py);
        o[0] = c.x, o[1] = c.y, o[2] = c.z;
    }
}

template <bool CULL>
__global__ __launch_bounds__(SRT_BLOCK) void k_intersect_batch(SceneDev sc, float* rays7, long long n, uint32_t features, srt_hit* hits,
                                                                 DevCounters* cnt)
{
    extern __shared__ uint2 lds_stack[];
    uint2* stk = lds_stack + threadIdx.x;
    Stats st = { 0, 0, 0, 0, 0 };
    uint32_t nPath = 0;
    const long long stride = (long long)gridDim.x * SRT_BLOCK;
    for (long long i = (long long)blockIdx.x * SRT_BLOCK + threadIdx.x; i < n; i += stride) {
        float* r = rays7 + 7 * i;
        const f3 o = mk3(r[0], r[1], r[2]), d = mk3(r[3], r[4], r[5]);
        float T = r[6];
        uint32_t tri;
        nPath += 1;
        const bool hit = bvh_query<CULL, true>(sc, features, o, d, T, tri, stk, st);
        srt_hit h;
        h.hit = hit;
        h.t = T;
        h.mesh = -1, h.tri = -1;
        h.normal[0] = h.normal[1] = h.normal[2] = 0;
        h.bary[0] = h.bary[1] = h.bary[2] = 0;
        h.uv[0] = h.uv[1] = 0;
        h.kd[0] = h.kd[1] = h.kd[2] = 0;
        if (hit) {
            Surface sf;
            surface_at(sc, features, o, d, T, tri, sf);
            h.mesh = (int)sf.mesh, h.tri = (int)sf.tri;
            h.normal[0] = sf.normal.x, h.normal[1] = sf.normal.y, h.normal[2] = sf.normal.z;
            h.bary[0] = sf.bary.x, h.bary[1] = sf.bary.y, h.bary[2] = sf.bary.z;
            h.uv[0] = sf.uv.x, h.uv[1] = sf.uv.y;
            h.kd[0] = sf.kd.x, h.kd[1] = sf.kd.y, h.kd[2] = sf.kd.z;
        }
        r[6] = T;
        hits[i] = h;
    }
    /* every lane of the wave reaches the reduction: the loop above has no early return */
    flush_counters(cnt, nPath, 0, 0, st);
}

__global__ __launch_bounds__(SRT_BLOCK) void k_generate(srt_camera cam, const float* ndc2, long long n, float* rays7)
{
    for (long long i = (long long)blockIdx.x * SRT_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * SRT_BLOCK) {
        const f3 d = camera_dir(cam, ndc2[2 * i], ndc2[2 * i + 1]);
        float* r = rays7 + 7 * i;
        r[0] = cam.origin[0], r[1] = cam.origin[1], r[2] = cam.origin[2];
        r[3] = d.x, r[4] = d.y, r[5] = d.z;
        r[6] = SRT_FLT_MAX;
    }
}

/* Self-test of div_by against plain division over the operand ranges slab_test_fast is used for: random
 * mantissas and exponents plus the awkward cases (powers of two, all-ones mantissas, numerator zero, the range
 * ends).  Counts results that differ other than in the sign of a zero. */
__global__ __launch_bounds__(SRT_BLOCK) void k_selftest_division(uint32_t seed, unsigned long long n, unsigned long long* mismatches)
{
    unsigned long long bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * SRT_BLOCK + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * SRT_BLOCK) {
        const uint32_t h0 = mix32(mix32(seed, (uint32_t)i), (uint32_t)(i >> 32));
        const uint32_t h1 = mix32(h0, 0x1234567u), h2 = mix32(h1, 0x89ABCDEu), h3 = mix32(h2, 0x5555AAAAu);
        /* denominator: 2^-20 <= |d| < 2^21 */
        uint32_t dm = h0 & 0x7FFFFFu;
        const uint32_t kind = h3 & 15u;
        if (kind == 0) dm = 0;
        if (kind == 1) dm = 0x7FFFFFu;
        if (kind == 2) dm = 1;
        const uint32_t de = 107u + (h1 % 41u);
        const float d = __uint_as_float((h1 & 0x80000000u) | (de << 23) | dm);
        /* two coordinates in {0} U [2^-40, 2^20), numerator = their difference (as box bound - origin) */
        auto coord = [](uint32_t h, uint32_t sel) {
            if ((sel & 7u) == 0)
                return 0.0f;
            uint32_t m = h & 0x7FFFFFu;
            if ((sel & 7u) == 1) m = 0;
            if ((sel & 7u) == 2) m = 0x7FFFFFu;
            const uint32_t e = 87u + ((h >> 23) % 60u);
            return __uint_as_float((h & 0x80000000u) ^ ((sel & 8u) << 28) | (e << 23) | m);
        };
        const float b = coord(h2, h3 >> 4), o = coord(mix32(h2, 77u), h3 >> 8);
        const float x = ((h3 >> 12) & 3u) == 0 ? (b - b * 1.00000012f) : (b - o); /* also nearly cancelling differences */
        const float want = x / d;
        const float got = div_by(x, d, refined_rcp(d));
        const bool same = __float_as_uint(want) == __float_as_uint(got) || (want == 0.0f && got == 0.0f);
        if (!same)
            bad += 1;
    }
    if (bad)
        atomicAdd(mismatches, bad);
}

/* Bandwidth probe: every lane reads `loads` pseudo-random 64-byte records (four 16-byte loads, as a node visit
 * does) from the first `n_nodes64` 64-byte units of the node array followed by the first `n_tris64` units of the
 * triangle stream, eight records in flight per lane, and does nothing else with them.  What comes out is the rate at
 * which the cache level that holds the chosen window can feed gathers of this shape. */
__global__ __launch_bounds__(SRT_BLOCK) void k_probe_gather(const float4* __restrict__ nodes, uint32_t n_nodes64, const float4* __restrict__ tris,
                                                              uint32_t n_tris64, uint32_t loads, uint32_t seed, float* sink)
{
#ifdef SRT_EXPERIMENT_PROBE_LANES
    if ((threadIdx.x & 63u) % SRT_EXPERIMENT_PROBE_LANES != 0)
        return; /* experiment: only every n-th lane of a wave fetches */
#endif
    const uint32_t tid = blockIdx.x * SRT_BLOCK + threadIdx.x;
    const uint32_t n = n_nodes64 + n_tris64;
    float acc = 0.0f;
    uint32_t h = mix32(seed, tid);
    for (uint32_t it = 0; it < loads; it += 8) {
        float4 q[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h = h * 1664525u + 1013904223u;
            const uint32_t idx = (uint32_t)(((unsigned long long)(h ^ (h >> 15)) * n) >> 32);
            const float4* rec = idx < n_nodes64 ? nodes + 4ull * idx : tris + 4ull * (idx - n_nodes64);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                q[u][k] = rec[k];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            acc += (q[u][0].x + q[u][1].y) + (q[u][2].z + q[u][3].w);
    }
    if (acc == 12345.678f)
        sink[0] = acc; /* keeps the loads alive; practically never taken */
}

/* The same records fetched the other way round: the four lanes of a quad read the four quarters of ONE record with
 * one instruction (64 contiguous bytes), four instructions cover the quad's four records. */
__global__ __launch_bounds__(SRT_BLOCK) void k_probe_gather_quad(const float4* __restrict__ nodes, uint32_t n_nodes64, const float4* __restrict__ tris,
                                                                   uint32_t n_tris64, uint32_t loads, uint32_t seed, float* sink)
{
    const uint32_t tid = blockIdx.x * SRT_BLOCK + threadIdx.x;
    const uint32_t n = n_nodes64 + n_tris64;
    const uint32_t q = threadIdx.x & 3u;
    float acc = 0.0f;
    uint32_t h = mix32(seed, tid);
    for (uint32_t it = 0; it < loads; it += 8) {
        float4 r[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h = h * 1664525u + 1013904223u;
            const uint32_t mine = (uint32_t)(((unsigned long long)(h ^ (h >> 15)) * n) >> 32);
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const uint32_t idx = (uint32_t)__shfl((int)mine, (int)((threadIdx.x & 60u) + sidx), 64);
                const float4* rec = idx < n_nodes64 ? nodes + 4ull * idx : tris + 4ull * (idx - n_nodes64);
                r[u][sidx] = rec[q];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            acc += (r[u][0].x + r[u][1].y) + (r[u][2].z + r[u][3].w);
    }
    if (acc == 12345.678f)
        sink[0] = acc;
}

/* ------------------------------------------------------------------------------------------ host side */

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return fail(SRT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap)
            return SRT_OK;
        if (p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
        HIPCHK(hipMalloc(&p, bytes));
        cap = bytes;
        return SRT_OK;
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

#define SRT_MAX_EVENTS 8192
#define SRT_MAX_SYNC 34 /* one per recursion depth (max_depth <= 30) and a few to spare */
#define SRT_FAMILIES SRT_KERNEL_FAMILIES /* 0 primary trace, 1 secondary trace, 2 shade, 3 resolve, 4 shadow trace */

/* ---- work memory of the entry point that allocates nothing (srt_render_noalloc).
 * A file-scope device array: every GPU the library is loaded on holds its own copy, at its own address, until the
 * process ends.  Host code never passes the array's name to a kernel or a memset (in host code the name is only a
 * handle for hipGetSymbolAddress); the address is looked up once per GPU in srt_device_create, with that GPU
 * current, and kept in g_scratch_ptr. */
static __device__ __align__(256) unsigned char scratch[SRT_NOALLOC_SCRATCH_BYTES];
#define SRT_MAX_GPUS 64
static unsigned char* g_scratch_ptr[SRT_MAX_GPUS];
static std::mutex g_scratch_mu;

struct srt_device {
    int gpu = 0;
    int n_cu = 0;
    DevCounters* counters = nullptr; /* SRT_FAMILIES structs, one per kernel family, plus one for srt_intersect_batch */
    uint32_t* qcounts = nullptr; /* 64 words: hit-queue counts [0..31], path-queue counts [32..63] */
    unsigned long long* chunk_counters = nullptr; /* 64: next unclaimed shadow item [0..31] and path ray [32..63] of each depth */
    DevBuf acc, chain, nlev, hq_slot, hq_t, hq_tri, hq_ip, vis, path[2], out_rgb, out_mesh, out_tri, out_t, io_a, io_b;
    hipEvent_t ev[SRT_MAX_EVENTS]; /* timing: events 2 i and 2 i + 1 bracket launch i */
    int ev_kind[SRT_MAX_EVENTS / 2]; /* kernel family of launch i */
    int n_ev = 0;
    bool ev_ready = false;
    /* the second stream a short frame's light stage runs on, and the events that order the two (render_impl) */
    hipStream_t aux = nullptr;
    hipEvent_t sync_hit[SRT_MAX_SYNC], sync_lit[SRT_MAX_SYNC];
    int sync_made = 0; /* event pairs created so far */
    bool sync_ready = false;
    DevBuf hq2_slot, hq2_t, hq2_tri, hq2_ip; /* the second hit queue such a frame alternates with the first */
};

/* Persistent-grid sizes (blocks that fit the GPU at once) of the four queue kernels. */
struct Grids {
    int primary = 0, secondary = 0, secondary_refill = 0, shadow = 0, shade = 0, shadow_coarse = 0;
};

struct srt_scene {
    srt_device* dev = nullptr;
    SceneDev view {};
    uint32_t num_levels = 0;
    uint32_t num_leaves = 0;
    Grids grids[2][2]; /* [miss culling][counting kernels]; fixed at upload: they depend on the stack depth only */
    std::vector<uint32_t> light_types; /* host copy: sizes the shadow-query stage */
    std::vector<uint32_t> host_children; /* (left, right) of every inner node, from upload / build until scene_coarsen has used it */
    std::vector<void*> allocs;
};

static int scene_grids(srt_device* d, srt_scene* sc);
static int scene_coarsen(srt_scene* sc);

template <typename T>
static int upload(srt_scene* s, const T* host, size_t count, const T** dev_out)
{
    *dev_out = nullptr;
    void* p = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    HIPCHK(hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    if (count)
        HIPCHK(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
    *dev_out = (const T*)p;
    return SRT_OK;
}

extern "C" {

const char* srt_last_error(void) { return g_err.c_str(); }

int srt_device_count(int* count)
{
    if (!count)
        return fail(SRT_ERR_INVALID, "srt_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(SRT_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return SRT_OK;
}

/* Paths of the HIP runtimes mapped into this process.  More than one means trouble: PyTorch's ROCm wheels carry a
 * runtime of their own whose file name differs from the system one's, so a process that loads this library first and
 * PyTorch afterwards ends up with two, and the second one's device initialisation fails with an unrelated message. */
static int collect_hip_runtimes(struct dl_phdr_info* info, size_t, void* data)
{
    auto* found = (std::vector<std::string>*)data;
    const char* name = info->dlpi_name ? info->dlpi_name : "";
    const char* base = strrchr(name, '/');
    base = base ? base + 1 : name;
    if (!strncmp(base, "libamdhip64.so", 14)) {
        bool seen = false;
        for (const std::string& f : *found)
            seen = seen || f == name;
        if (!seen)
            found->push_back(name);
    }
    return 0;
}

int srt_device_create(int gpu_id, srt_device** out)
{
    if (!out)
        return fail(SRT_ERR_INVALID, "srt_device_create: null argument");
    *out = nullptr;
    {
        std::vector<std::string> runtimes;
        dl_iterate_phdr(collect_hip_runtimes, &runtimes);
        if (runtimes.size() > 1)
            return fail(SRT_ERR_HIP, "two HIP runtimes are mapped into this process (%s and %s); only the first one loaded can use the GPU. "
                                     "Load PyTorch (`import torch`) before this library, which then binds to PyTorch's runtime",
                runtimes[0].c_str(), runtimes[1].c_str());
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(SRT_ERR_NO_DEVICE, "no HIP device is visible: the MI355X back end cannot run (there is no CPU fallback)");
    if (gpu_id < 0 || gpu_id >= n)
        return fail(SRT_ERR_INVALID, "gpu_id %d out of range [0,%d)", gpu_id, n);
    HIPCHK(hipSetDevice(gpu_id));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, gpu_id));
    srt_device* d = new srt_device();
    d->gpu = gpu_id;
    d->n_cu = prop.multiProcessorCount;
    if (gpu_id < SRT_MAX_GPUS) {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        if (!g_scratch_ptr[gpu_id]) {
            void* p = nullptr;
            hipError_t es = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
            if (es != hipSuccess || !p) {
                delete d;
                return fail(SRT_ERR_HIP, "hipGetSymbolAddress(scratch) on GPU %d: %s", gpu_id, hipGetErrorString(es));
            }
            g_scratch_ptr[gpu_id] = (unsigned char*)p;
        }
    }
    hipError_t e = hipMalloc((void**)&d->counters, (SRT_FAMILIES + 1) * sizeof(DevCounters));
    /* queue counts and chunk counters in one piece: a frame clears them with one memset */
    if (e == hipSuccess)
        e = hipMalloc((void**)&d->qcounts, 64 * sizeof(uint32_t) + 64 * sizeof(unsigned long long));
    if (e == hipSuccess)
        d->chunk_counters = (unsigned long long*)(d->qcounts + 64);
    if (e == hipSuccess)
        e = hipMemset(d->counters, 0, (SRT_FAMILIES + 1) * sizeof(DevCounters));
    if (e != hipSuccess) {
        srt_device_destroy(d); /* frees whatever was allocated before the failure */
        return fail(SRT_ERR_HIP, "device set-up failed: %s", hipGetErrorString(e));
    }
    *out = d;
    return SRT_OK;
}

void srt_device_destroy(srt_device* d)
{
    if (!d)
        return;
    (void)hipSetDevice(d->gpu);
    (void)hipDeviceSynchronize();
    DevBuf* bufs[] = { &d->acc, &d->chain, &d->nlev, &d->hq_slot, &d->hq_t, &d->hq_tri, &d->hq_ip, &d->vis, &d->path[0], &d->path[1], &d->out_rgb, &d->out_mesh, &d->out_tri,
        &d->out_t, &d->io_a, &d->io_b, &d->hq2_slot, &d->hq2_t, &d->hq2_tri, &d->hq2_ip };
    for (DevBuf* b : bufs)
        b->release();
    if (d->counters)
        (void)hipFree(d->counters);
    if (d->qcounts)
        (void)hipFree(d->qcounts); /* the chunk counters are part of it */
    if (d->ev_ready)
        for (int i = 0; i < SRT_MAX_EVENTS; ++i)
            (void)hipEventDestroy(d->ev[i]);
    for (int i = 0; i < d->sync_made; ++i) {
        (void)hipEventDestroy(d->sync_hit[i]);
        (void)hipEventDestroy(d->sync_lit[i]);
    }
    if (d->aux)
        (void)hipStreamDestroy(d->aux);
    delete d;
}

int srt_device_synchronize(srt_device* d, void* stream)
{
    if (!d)
        return fail(SRT_ERR_INVALID, "null device");
    HIPCHK(hipSetDevice(d->gpu));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return SRT_OK;
}

static void to_public(const DevCounters& c, srt_counters* out)
{
    out->path_rays = c.path_rays;
    out->shadow_rays = c.shadow_rays;
    out->shaded_hits = c.shaded_hits;
    out->inner_visits = c.inner_visits;
    out->tri_tests = c.tri_tests;
    out->box_tests = c.box_tests;
    out->wave_inner_steps = c.wave_inner_steps;
    out->wave_leaf_steps = c.wave_leaf_steps;
}

int srt_device_read_kernel_counters(srt_device* d, void* stream, srt_counters out[SRT_KERNEL_FAMILIES], int reset)
{
    if (!d || !out)
        return fail(SRT_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(d->gpu));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    DevCounters c[SRT_FAMILIES];
    HIPCHK(hipMemcpy(c, d->counters, sizeof c, hipMemcpyDeviceToHost));
    for (int k = 0; k < SRT_FAMILIES; ++k)
        to_public(c[k], out + k);
    if (reset)
        HIPCHK(hipMemset(d->counters, 0, sizeof c));
    return SRT_OK;
}

int srt_device_read_counters(srt_device* d, void* stream, srt_counters* out, int reset)
{
    srt_counters k[SRT_FAMILIES];
    int rc = srt_device_read_kernel_counters(d, stream, k, reset);
    if (rc != SRT_OK || !out)
        return rc != SRT_OK ? rc : fail(SRT_ERR_INVALID, "null argument");
    memset(out, 0, sizeof *out);
    for (int i = 0; i < SRT_FAMILIES; ++i) {
        out->path_rays += k[i].path_rays, out->shadow_rays += k[i].shadow_rays, out->shaded_hits += k[i].shaded_hits;
        out->inner_visits += k[i].inner_visits, out->tri_tests += k[i].tri_tests, out->box_tests += k[i].box_tests;
        out->wave_inner_steps += k[i].wave_inner_steps, out->wave_leaf_steps += k[i].wave_leaf_steps;
    }
    return SRT_OK;
}

int srt_device_kernel_times(srt_device* d, float ms[SRT_KERNEL_FAMILIES], int32_t launches[SRT_KERNEL_FAMILIES], int reset)
{
    if (!d || !ms || !launches)
        return fail(SRT_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(d->gpu));
    for (int k = 0; k < SRT_FAMILIES; ++k)
        ms[k] = 0.0f, launches[k] = 0;
    for (int i = 0; 2 * i + 1 < d->n_ev; ++i) {
        float t = 0;
        HIPCHK(hipEventSynchronize(d->ev[2 * i + 1]));
        HIPCHK(hipEventElapsedTime(&t, d->ev[2 * i], d->ev[2 * i + 1]));
        ms[d->ev_kind[i]] += t;
        launches[d->ev_kind[i]] += 1;
    }
    if (reset)
        d->n_ev = 0;
    return SRT_OK;
}

static int check_ref(const srt_scene_desc* d, uint32_t ref, const char* what)
{
    if (ref & SRT_LEAF_FLAG) {
        uint32_t first = ref & 0x00FFFFFFu, cnt = (ref >> 24) & 0x7Fu;
        if ((uint64_t)first + cnt > d->n_tris)
            return fail(SRT_ERR_INVALID, "%s: leaf reference [%u,+%u) outside %u triangles", what, first, cnt, d->n_tris);
    } else if (ref >= d->n_nodes) {
        return fail(SRT_ERR_INVALID, "%s: node reference %u outside %u nodes", what, ref, d->n_nodes);
    }
    return SRT_OK;
}

int srt_scene_upload(srt_device* dev, const srt_scene_desc* desc, srt_scene** out)
{
    if (!dev || !desc || !out)
        return fail(SRT_ERR_INVALID, "srt_scene_upload: null argument");
    *out = nullptr;
    static_assert(sizeof(srt_node) == 64 && sizeof(srt_tri) == 96 && sizeof(srt_tri_attr) == 64, "record sizes");
    static_assert(sizeof(srt_material) == 48 && sizeof(srt_light) == 96, "record sizes");
    if (desc->n_tris > SRT_MAX_TRIS)
        return fail(SRT_ERR_INVALID, "scene has %u triangles; the packed child reference addresses at most %u", desc->n_tris, SRT_MAX_TRIS);
    if (desc->num_levels > SRT_MAX_LEVELS)
        return fail(SRT_ERR_INVALID, "BVH has %u levels; the traversal stack supports %d", desc->num_levels, SRT_MAX_LEVELS);
    if (desc->n_nodes >= (1u << 26))
        return fail(SRT_ERR_INVALID, "node table has %u records; the kernels address it by 32-bit byte offsets (fewer than 2^26 records)", desc->n_nodes);
    if ((desc->n_nodes && !desc->nodes) || (desc->n_tris && (!desc->tris || !desc->attrs || !desc->scene_order))
        || (desc->n_materials && !desc->materials) || (desc->n_lights && !desc->lights) || (desc->n_textures && (!desc->textures || !desc->texels)))
        return fail(SRT_ERR_INVALID, "srt_scene_upload: a non-empty array is null");
    /* Every index a kernel will follow is validated here, on the host, before anything is launched. */
    int rc = check_ref(desc, desc->root_ref, "root");
    if (rc == SRT_OK && !(desc->root_ref & SRT_LEAF_FLAG) && desc->root_ref != 0)
        rc = fail(SRT_ERR_INVALID, "an inner root must be node 0 (root reference %u)", desc->root_ref);
    /* An inner child's index is larger than its parent's, so a traversal cannot cycle, and one pass in index
     * order gives every inner node's depth: the per-lane stacks hold one pending sibling per inner level and
     * are sized from num_levels, which therefore must cover the deepest leaf (inner depth + 2 levels). */
    std::vector<uint32_t> depth(desc->n_nodes, 0);
    std::vector<uint8_t> referenced(desc->n_nodes, 0);
    uint32_t deepest = 0;
    for (uint32_t i = 0; i < desc->n_nodes && rc == SRT_OK; ++i) {
        rc = check_ref(desc, desc->nodes[i].left, "left child");
        if (rc == SRT_OK)
            rc = check_ref(desc, desc->nodes[i].right, "right child");
        for (uint32_t child : { desc->nodes[i].left, desc->nodes[i].right }) {
            if (rc != SRT_OK || (child & SRT_LEAF_FLAG))
                continue;
            if (child <= i) {
                rc = fail(SRT_ERR_INVALID, "node %u: child index %u does not follow its parent", i, child);
            } else if (referenced[child]) {
                /* a node with two parents could sit deeper than the pass below records, and the stacks are sized by depth */
                rc = fail(SRT_ERR_INVALID, "node %u is the child of more than one node", child);
            } else {
                referenced[child] = 1;
                depth[child] = depth[i] + 1;
            }
        }
        if (depth[i] > deepest)
            deepest = depth[i];
    }
    uint32_t n_leaves = (desc->root_ref & SRT_LEAF_FLAG) ? 1u : 0u;
    for (uint32_t i = 0; i < desc->n_nodes; ++i)
        n_leaves += ((desc->nodes[i].left & SRT_LEAF_FLAG) ? 1u : 0u) + ((desc->nodes[i].right & SRT_LEAF_FLAG) ? 1u : 0u);
    if (rc == SRT_OK && desc->n_nodes && deepest + 2 > desc->num_levels)
        rc = fail(SRT_ERR_INVALID, "num_levels is %u but the node table is %u levels deep", desc->num_levels, deepest + 2);
    if (rc != SRT_OK)
        return rc;
    for (uint32_t i = 0; i < desc->n_tris; ++i) {
        if (desc->tris[i].mesh >= desc->n_materials || desc->attrs[i].mesh >= desc->n_materials)
            return fail(SRT_ERR_INVALID, "triangle %u: mesh index outside %u materials", i, desc->n_materials);
        if (desc->scene_order[i] >= desc->n_tris)
            return fail(SRT_ERR_INVALID, "scene_order[%u] outside %u triangles", i, desc->n_tris);
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

    HIPCHK(hipSetDevice(dev->gpu));
    srt_scene* s = new srt_scene();
    s->dev = dev;
    s->num_levels = desc->num_levels;
    s->num_leaves = n_leaves;
    for (uint32_t i = 0; i < desc->n_lights; ++i)
        s->light_types.push_back(desc->lights[i].type);
    SceneDev& v = s->view;
    const srt_node* dn;
    const srt_tri* dt;
    const srt_tri_attr* da;
    rc = upload(s, desc->nodes, desc->n_nodes, &dn);
    if (rc == SRT_OK) rc = upload(s, desc->tris, desc->n_tris, &dt);
    if (rc == SRT_OK) rc = upload(s, desc->attrs, desc->n_tris, &da);
    if (rc == SRT_OK) rc = upload(s, desc->scene_order, desc->n_tris, &v.scene_order);
    if (rc == SRT_OK) rc = upload(s, desc->materials, desc->n_materials, &v.materials);
    if (rc == SRT_OK) rc = upload(s, desc->textures, desc->n_textures, &v.textures);
    if (rc == SRT_OK) rc = upload(s, desc->texels, (size_t)desc->n_texels * 3, &v.texels);
    if (rc == SRT_OK) rc = upload(s, desc->lights, desc->n_lights, &v.lights);
    if (rc != SRT_OK) {
        srt_scene_destroy(s);
        return rc;
    }
    v.nodes = (const float4*)dn;
    v.tris = (const float4*)dt;
    v.attrs = (const float4*)da;
    v.n_nodes = desc->n_nodes, v.n_tris = desc->n_tris, v.n_materials = desc->n_materials;
    v.n_textures = desc->n_textures, v.n_lights = desc->n_lights;
    v.root_ref = desc->root_ref;
    for (int k = 0; k < 3; ++k)
        v.root_lo[k] = desc->root_box[k], v.root_hi[k] = desc->root_box[3 + k];
    v.n_texels = desc->n_texels;
    v.env_tex = desc->environment ? (int32_t)desc->environment - 1 : -1;
    auto coord_ok = [](float c) {
        uint32_t b;
        memcpy(&b, &c, 4);
        b &= 0x7FFFFFFFu;
        const uint32_t e = b >> 23;
        return b == 0u || (e >= 87u && e < 147u);
    };
    auto box_ok = [&](const float* b) {
        bool ok = true;
        for (int k = 0; k < 6; ++k)
            ok = ok && coord_ok(b[k]);
        return ok && b[0] <= b[3] && b[1] <= b[4] && b[2] <= b[5];
    };
    bool fast = box_ok(desc->root_box);
    for (uint32_t i = 0; i < desc->n_nodes && fast; ++i)
        fast = box_ok(desc->nodes[i].lbox) && box_ok(desc->nodes[i].rbox);
    v.fast_boxes = fast ? 1u : 0u;
    s->host_children.resize(2ull * desc->n_nodes);
    for (uint32_t i = 0; i < desc->n_nodes; ++i)
        s->host_children[2ull * i] = desc->nodes[i].left, s->host_children[2ull * i + 1] = desc->nodes[i].right;
    rc = scene_grids(dev, s);
    if (rc != SRT_OK) {
        srt_scene_destroy(s);
        return rc;
    }
    *out = s;
    return SRT_OK;
}

void srt_scene_destroy(srt_scene* s)
{
    if (!s)
        return;
    if (s->dev)
        (void)hipSetDevice(s->dev->gpu);
    for (void* p : s->allocs)
        (void)hipFree(p);
    delete s;
}

} // extern "C"

/* Stack capacity: at most one pending left sibling per inner level. */
static uint32_t stack_entries(const srt_scene* s) { return s->num_levels + SRT_STACK_SLACK; }
static size_t stack_bytes(const srt_scene* s) { return (size_t)stack_entries(s) * SRT_BLOCK * sizeof(uint2); }
/* Entries of the shadow kernel's per-lane stacks.  The walks on binary nodes need one per level.  A wide node can leave
 * three siblings pending and wants three free entries and a spare one whenever it stacks (swalk_inner_wide), so a tree
 * of L levels - (L - 1 + 1) / 2 wide levels - can never overflow 3 * wide levels + 1 entries; shallow trees get that,
 * deeper ones at least 19 (what an 18-level tree's L + 1 comes to, and what six blocks per compute unit leave room
 * for) and the rare query that wants more is answered by the exact walk. */
static uint32_t shadow_stack_entries(uint32_t num_levels)
{
    const uint32_t wide_levels = num_levels / 2, never = 3 * wide_levels + 1;
    const uint32_t capped = never < 19u ? never : 19u;
    return num_levels + SRT_STACK_SLACK > capped ? num_levels + SRT_STACK_SLACK : capped;
}
static size_t shadow_lds_bytes(const srt_scene* s)
{
    return SRT_LDS_LIGHTS * sizeof(srt_light) + SRT_WIDE_LDS_NODES * 64u + (size_t)shadow_stack_entries(s->num_levels) * SRT_SHADOW_BLOCK * sizeof(uint32_t);
}

template <typename K>
static int persistent_grid(srt_device* d, K kernel, size_t lds, int* grid, int block = SRT_BLOCK)
{
    int per_cu = 0;
    if (lds > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds));
    if (per_cu < 1)
        return fail(SRT_ERR_INVALID, "kernel does not fit a compute unit with %zu bytes of LDS", lds);
    *grid = per_cu * d->n_cu;
    return SRT_OK;
}

/* Timing of one launch: tick() before it and tock() after it record a pair of events on the launch's stream; the pair
 * belongs to kernel family `kind`.  When the ring is full, further launches are simply not timed. */
static bool tick(srt_device* d, hipStream_t st, int kind, bool timing)
{
    if (!timing)
        return false;
    if (!d->ev_ready) {
        for (int i = 0; i < SRT_MAX_EVENTS; ++i)
            if (hipEventCreate(&d->ev[i]) != hipSuccess)
                return false;
        d->ev_ready = true;
    }
    if (d->n_ev + 2 > SRT_MAX_EVENTS)
        return false;
    (void)hipEventRecord(d->ev[d->n_ev], st);
    d->ev_kind[d->n_ev / 2] = kind;
    return true;
}
static void tock(srt_device* d, hipStream_t st, bool ticked)
{
    if (!ticked)
        return;
    (void)hipEventRecord(d->ev[d->n_ev + 1], st);
    d->n_ev += 2;
}

/* Shadow queries per hit: the same count the kernels derive from the light list (shadow_samples). */
static uint32_t shadows_per_hit(const srt_scene* sc, uint32_t features)
{
    uint32_t n = 0;
    if (!(features & SRT_F_SHADING))
        return 0;
    for (uint32_t t : sc->light_types)
        n += t == 0 ? ((features & SRT_F_HARD_SHADOW) ? 1u : 0u) : ((features & SRT_F_SOFT_SHADOW) ? 100u : 0u);
    return n;
}

/* Device memory one rectangle's kernels work in, beyond the scene, the inputs and the outputs.  srt_render takes it
 * from buffers the device handle grows on demand, srt_render_noalloc from the static scratch array. */
struct Work {
    float* acc = nullptr;            /* 3 floats per sample */
    float* chain = nullptr;          /* reflections without transparency: 6 floats per (level < max_depth, sample), see k_shade */
    unsigned char* nlev = nullptr;   /* ... and the number of levels a sample has parked there */
    HitQ hq {};                      /* one entry per sample (more with transparency) */
    HitQ hq2 {};                     /* optional second hit queue: lets a short frame overlap its light stage with the next depth's trace */
    unsigned char* vis = nullptr;    /* one byte per (hit, shadow query) */
    float4* path[2] = { nullptr, nullptr }; /* 3 float4 per queued path ray; only with reflections / transparency */
    uint32_t* qcounts = nullptr;     /* 64 words: hit-queue counts [0..31], path-queue counts [32..63] */
    unsigned long long* chunk_counters = nullptr; /* 64: next unclaimed shadow item [0..31] and path ray [32..63] of each depth */
    bool growable = false;           /* the queues may be enlarged between depths (transparency) */
};

template <bool CULL, bool COUNT>
static int render_impl(srt_device* d, srt_scene* sc, const RenderDev& rp, uint32_t flags, hipStream_t st, Work wk)
{
    const size_t lds = stack_bytes(sc);
    const bool timing = (flags & SRT_RENDER_TIMING) != 0;
    const uint32_t n = rp.n_samples;
    const uint32_t sph = shadows_per_hit(sc, rp.features);
    const Grids& g = sc->grids[CULL ? 1 : 0][COUNT ? 1 : 0];
    const bool coarse = CULL && (rp.features & SRT_F_COARSE_SHADOW) && sc->view.coarse_ok;
    int g_primary = g.primary, g_secondary = g.secondary, g_shadow = coarse ? g.shadow_coarse : g.shadow, g_shade = g.shade;
    const int need = (int)((n + SRT_BLOCK - 1) / SRT_BLOCK);
#ifdef SRT_EXPERIMENT_PRIMARY_BLOCKS
    g_primary = need; /* one block per 256 samples: the hardware dispatcher balances the blocks */
#else
    if (need < g_primary) g_primary = need;
#endif
    if (need < g_secondary) g_secondary = need;
    if (need < g_shade) g_shade = need;
    const unsigned long long need_shadow = ((unsigned long long)n * sph + SRT_SHADOW_BLOCK - 1) / SRT_SHADOW_BLOCK;
    if (need_shadow < (unsigned long long)g_shadow) g_shadow = (int)need_shadow;

    HitQ hq = wk.hq;
    uint32_t* hq_count = wk.qcounts;
    uint32_t* path_count = wk.qcounts + 32;
    float* acc = wk.acc;
    unsigned char* vis = wk.vis;
    /* the only buffers a frame needs cleared: queue counts and chunk counters (contiguous where the handle owns them) */
    if ((void*)wk.chunk_counters == (void*)(wk.qcounts + 64)) {
        HIPCHK(hipMemsetAsync(wk.qcounts, 0, 64 * sizeof(uint32_t) + 64 * sizeof(unsigned long long), st));
    } else {
        HIPCHK(hipMemsetAsync(wk.qcounts, 0, 64 * sizeof(uint32_t), st));
        HIPCHK(hipMemsetAsync(wk.chunk_counters, 0, 64 * sizeof(unsigned long long), st));
    }
    const bool chained = (rp.features & SRT_F_RECURSIVE) && !(rp.features & SRT_F_TRANSPARENCY) && rp.max_depth > 0;
    float* chain = chained ? wk.chain : nullptr;
    unsigned char* nlev = chained ? wk.nlev : nullptr;
    if (chained && (!chain || !nlev))
        return fail(SRT_ERR_INVALID, "reflections need the per-level buffers");
    const uint32_t n_px = (uint32_t)rp.tw * (uint32_t)rp.th;
    int g_res = (int)((n_px + SRT_BLOCK - 1) / SRT_BLOCK);
    if (g_res > d->n_cu * 8) g_res = d->n_cu * 8;
    if (rp.lens_mode && rp.lens.samples == 0) {
        /* motion blur / depth of field with zero samples: the reference's loops do not run and the pixel stays black
         * (k_resolve reads no sample slot then) */
        hipLaunchKernelGGL(k_resolve, dim3(g_res), dim3(SRT_BLOCK), 0, st, rp, (const float*)acc, (const float*)chain, (const unsigned char*)nlev);
        HIPCHK(hipGetLastError());
        return SRT_OK;
    }
    auto launch_shadow = [&](hipStream_t s, int depth, const HitQ& q, const float4* cur) {
        if (!sph)
            return;
        const bool t = tick(d, s, 4, timing);
        if (coarse)
            hipLaunchKernelGGL((k_trace_shadow<CULL, COUNT, CULL>), dim3(g_shadow), dim3(SRT_SHADOW_BLOCK), shadow_lds_bytes(sc), s, sc->view, rp, depth, q,
                hq_count + depth, cur, vis, sph, d->counters + 4, wk.chunk_counters + depth);
        else
            hipLaunchKernelGGL((k_trace_shadow<CULL, COUNT, false>), dim3(g_shadow), dim3(SRT_SHADOW_BLOCK), shadow_lds_bytes(sc), s, sc->view, rp, depth, q,
                hq_count + depth, cur, vis, sph, d->counters + 4, wk.chunk_counters + depth);
        tock(d, s, t);
    };
    auto launch_secondary = [&](hipStream_t s, int depth, const HitQ& q, const float4* rays) {
        /* Rays fed like shadow queries (chunks + lane refill) once a rectangle is large: dragon stand-in 7680x4320, 64
         * samples per pixel in 960x540 rectangles 223.6 -> 195.9 ms per frame (box steps at 45 instead of 34 lanes), but monkey
         * 1920x1080, whose launches hold a few rays per lane, 0.674 -> 0.905 ms.  SRT_SECONDARY_REFILL=0 / 1 forces the choice
         * (tests run both kernels at every size). */
        static const char* refill_env = getenv("SRT_SECONDARY_REFILL");
        const bool refill = refill_env ? refill_env[0] == '1' : rp.n_samples >= (1u << 24);
        const bool t = tick(d, s, 1, timing);
        if (refill && (rp.features & SRT_F_ACCEL))
            hipLaunchKernelGGL((k_trace_secondary_refill<CULL, COUNT>), dim3(need < g.secondary_refill ? need : g.secondary_refill), dim3(SRT_BLOCK), lds, s,
                sc->view, rp.features, rays, path_count + depth + 1, q, hq_count + depth + 1, d->counters + 1, acc, wk.chunk_counters + 32 + depth);
        else
            hipLaunchKernelGGL((k_trace_secondary<CULL, COUNT>), dim3(g_secondary), dim3(SRT_BLOCK), lds, s, sc->view, rp.features, rays,
                path_count + depth + 1, q, hq_count + depth + 1, d->counters + 1, acc);
        tock(d, s, t);
    };
    bool t = tick(d, st, 0, timing);
    hipLaunchKernelGGL((k_trace_primary<CULL, COUNT>), dim3(g_primary), dim3(SRT_BLOCK), lds, st, sc->view, rp, hq, hq_count, d->counters + 0, acc, nlev);
    tock(d, st, t);
    const bool branching = (rp.features & SRT_F_TRANSPARENCY) != 0;
    const int last_depth = (rp.features & (SRT_F_RECURSIVE | SRT_F_TRANSPARENCY)) ? rp.max_depth : 0;

    /* ---- short frames with reflections: two streams.
     * A hit's reflection ray does not depend on its light, only the sample's value does.  The depth loop is therefore
     * cut in two chains: on the caller's stream  primary, [send on, trace]_1, [send on, trace]_2 ...  (k_shade<1>,
     * k_trace_secondary) and, one stage behind on a stream of the handle's own,  [shadow, light]_0, [shadow, light]_1 ...
     * (k_trace_shadow, k_shade<2>).  The launches of a 1920 x 1080 frame are short and leave most of the GPU idle, so the
     * two chains run side by side and the frame takes the longer one instead of their sum.  Depths alternate between two
     * hit queues; an event per depth keeps a queue (and the ray buffer of the same parity) from being overwritten before
     * its light stage has read it.  Large rectangles fill the GPU with either chain alone and keep the single stream. */
    /* (Measured at 1920 x 1080: monkey, 11 levels, 1.35 -> 1.11 ms per frame; the textured cube, 5 levels, whose launches
     * are all but empty, 0.33 -> 0.39 ms: there the cross-stream waits cost more than the overlap saves, hence the
     * condition on the tree's depth.  SRT_OVERLAP=0 / 1 forces the choice; the parity tests run both.) */
    static const char* overlap_env = getenv("SRT_OVERLAP");
    const bool small = overlap_env ? overlap_env[0] == '1' : (rp.n_samples < (1u << 23) && sc->num_levels >= 8);
    if (chained && small && wk.hq2.slot && last_depth > 0 && last_depth < SRT_MAX_SYNC - 1) {
        if (!d->sync_ready) {
            if (!d->aux)
                HIPCHK(hipStreamCreateWithFlags(&d->aux, hipStreamNonBlocking));
            for (; d->sync_made < SRT_MAX_SYNC; ++d->sync_made) {
                HIPCHK(hipEventCreateWithFlags(&d->sync_hit[d->sync_made], hipEventDisableTiming));
                if (hipEventCreateWithFlags(&d->sync_lit[d->sync_made], hipEventDisableTiming) != hipSuccess) {
                    (void)hipEventDestroy(d->sync_hit[d->sync_made]);
                    return fail(SRT_ERR_HIP, "hipEventCreateWithFlags failed (second stream of a short frame)");
                }
            }
            d->sync_ready = true;
        }
        hipStream_t sb = d->aux;
        const HitQ q2[2] = { wk.hq, wk.hq2 };
        HIPCHK(hipEventRecord(d->sync_hit[0], st)); /* hits of depth 0 queued (and everything the caller enqueued before) */
        for (int depth = 0; depth <= last_depth; ++depth) {
            const float4* cur = wk.path[depth & 1];
            float4* nxt = wk.path[(depth + 1) & 1];
            if (depth < last_depth) {
                /* the rays this stage writes replace those of depth - 1, and the next trace fills the hit queue of
                 * depth - 1: both are read by that depth's light stage */
                if (depth >= 1)
                    HIPCHK(hipStreamWaitEvent(st, d->sync_lit[depth - 1], 0));
                t = tick(d, st, 2, timing);
                hipLaunchKernelGGL(k_shade<1>, dim3(g_shade), dim3(SRT_BLOCK), 0, st, sc->view, rp, depth, q2[depth & 1], hq_count + depth, cur, nxt,
                    path_count + depth + 1, acc, chain, nlev, (const unsigned char*)vis, sph, d->counters + 2);
                tock(d, st, t);
                launch_secondary(st, depth, q2[(depth + 1) & 1], (const float4*)nxt);
                HIPCHK(hipEventRecord(d->sync_hit[depth + 1], st));
            }
            HIPCHK(hipStreamWaitEvent(sb, d->sync_hit[depth], 0));
            launch_shadow(sb, depth, q2[depth & 1], cur);
            t = tick(d, sb, 2, timing);
            hipLaunchKernelGGL(k_shade<2>, dim3(g_shade), dim3(SRT_BLOCK), 0, sb, sc->view, rp, depth, q2[depth & 1], hq_count + depth, cur, nxt,
                path_count + depth + 1, acc, chain, nlev, (const unsigned char*)vis, sph, d->counters + 2);
            tock(d, sb, t);
            HIPCHK(hipEventRecord(d->sync_lit[depth], sb));
        }
        HIPCHK(hipStreamWaitEvent(st, d->sync_lit[last_depth], 0)); /* the second stream runs in order: all its stages are done */
        t = tick(d, st, 3, timing);
        hipLaunchKernelGGL(k_resolve, dim3(g_res), dim3(SRT_BLOCK), 0, st, rp, (const float*)acc, (const float*)chain, (const unsigned char*)nlev);
        tock(d, st, t);
        HIPCHK(hipGetLastError());
        return SRT_OK;
    }

    for (int depth = 0; depth <= last_depth; ++depth) {
        const float4* cur = wk.path[depth & 1];
        float4* nxt = wk.path[(depth + 1) & 1];
        launch_shadow(st, depth, hq, cur);
        t = tick(d, st, 2, timing);
        hipLaunchKernelGGL(k_shade<0>, dim3(g_shade), dim3(SRT_BLOCK), 0, st, sc->view, rp, depth, hq, hq_count + depth, cur, nxt, path_count + depth + 1,
            acc, chain, nlev, (const unsigned char*)vis, sph, d->counters + 2);
        tock(d, st, t);
        if (depth == last_depth)
            break;
        if (branching) {
            /* A hit may send on two rays, so the queues can outgrow the sample count: read how many rays this
             * depth spawned and make room for their hits, their visibility bytes and twice as many successors.
             * Everything the buffers held is dead by now (hits, visibility and rays of this depth).  This is the
             * one place where a frame waits for the device between its launches. */
            if (!wk.growable)
                return fail(SRT_ERR_INVALID, "transparency needs queues that can grow");
            uint32_t spawned = 0;
            HIPCHK(hipMemcpyAsync(&spawned, path_count + depth + 1, sizeof spawned, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (spawned == 0)
                break;
            if (spawned >= (1u << 30))
                return fail(SRT_ERR_INVALID, "transparency: %u rays at depth %d; render a smaller rectangle", spawned, depth + 1);
            int rc = d->hq_slot.reserve((size_t)spawned * 4);
            if (rc == SRT_OK) rc = d->hq_t.reserve((size_t)spawned * 4);
            if (rc == SRT_OK) rc = d->hq_tri.reserve((size_t)spawned * 4);
            if (rc == SRT_OK) rc = d->hq_ip.reserve((size_t)spawned * 16);
            if (rc == SRT_OK) rc = d->vis.reserve((size_t)spawned * sph + 64);
            if (rc == SRT_OK) rc = d->path[depth & 1].reserve((size_t)spawned * 2 * 48);
            if (rc != SRT_OK)
                return rc;
            hq = { (uint32_t*)d->hq_slot.p, (float*)d->hq_t.p, (uint32_t*)d->hq_tri.p, (float4*)d->hq_ip.p };
            vis = (unsigned char*)d->vis.p;
            wk.path[depth & 1] = (float4*)d->path[depth & 1].p;
        }
        launch_secondary(st, depth, hq, (const float4*)nxt);
    }
    t = tick(d, st, 3, timing);
    hipLaunchKernelGGL(k_resolve, dim3(g_res), dim3(SRT_BLOCK), 0, st, rp, (const float*)acc, (const float*)chain, (const unsigned char*)nlev);
    tock(d, st, t);
    HIPCHK(hipGetLastError());
    return SRT_OK;
}

template <bool CULL, bool COUNT>
static int fill_grids(srt_device* d, srt_scene* sc)
{
    Grids& g = sc->grids[CULL ? 1 : 0][COUNT ? 1 : 0];
    int rc = persistent_grid(d, k_trace_primary<CULL, COUNT>, stack_bytes(sc), &g.primary);
    if (rc == SRT_OK) rc = persistent_grid(d, k_trace_secondary<CULL, COUNT>, stack_bytes(sc), &g.secondary);
    if (rc == SRT_OK) rc = persistent_grid(d, k_trace_secondary_refill<CULL, COUNT>, stack_bytes(sc), &g.secondary_refill);
    if (rc == SRT_OK) rc = persistent_grid(d, k_trace_shadow<CULL, COUNT, false>, shadow_lds_bytes(sc), &g.shadow, SRT_SHADOW_BLOCK);
    if (rc == SRT_OK && CULL) rc = persistent_grid(d, k_trace_shadow<CULL, COUNT, true>, shadow_lds_bytes(sc), &g.shadow_coarse, SRT_SHADOW_BLOCK);
    if (rc == SRT_OK) rc = persistent_grid(d, k_shade<0>, 0, &g.shade);
    return rc;
}

/* The wide coarse nodes of a freshly uploaded or built scene (see k_widen).  s->host_children holds the child
 * references of every inner node (left, right), as validated by the caller: an inner child's index is larger than its
 * parent's, so one pass in index order gives every node's depth. */
static int scene_coarsen(srt_scene* s)
{
    SceneDev& v = s->view;
    v.wnodes = nullptr, v.coarse_ok = 0, v.wide_cap = 0, v.n_wide = 0, v.n_top = 0, v.stris = nullptr;
    GridMap g;
    for (int k = 0; k < 3; ++k) {
        const float ext = v.root_hi[k] - v.root_lo[k];
        v.q_scale[k] = ext > 0.0f ? 65533.0f / ext : 1.0f;
        g.lo[k] = v.root_lo[k], g.scale[k] = v.q_scale[k];
    }
    std::vector<uint32_t> children;
    children.swap(s->host_children);
    if (!v.n_nodes || !v.fast_boxes)
        return SRT_OK;
    if (children.size() != 2ull * v.n_nodes)
        return fail(SRT_ERR_INVALID, "internal: child references of %u nodes expected, %zu given", v.n_nodes, children.size() / 2);
    std::vector<uint32_t> depth(v.n_nodes, 0), wide_of(v.n_nodes, SRT_WIDE_NONE);
    uint32_t deepest = 0;
    std::vector<uint32_t> per_depth;
    for (uint32_t i = 0; i < v.n_nodes; ++i) {
        if (depth[i] > deepest)
            deepest = depth[i];
        if (!(depth[i] & 1u)) {
            if (per_depth.size() <= depth[i] / 2)
                per_depth.resize(depth[i] / 2 + 1, 0);
            per_depth[depth[i] / 2] += 1;
        }
        for (int k = 0; k < 2; ++k) {
            const uint32_t c = children[2ull * i + k];
            if (!(c & SRT_LEAF_FLAG))
                depth[c] = depth[i] + 1;
        }
    }
    /* Numbering of the wide array: first the records of the tree's top levels, as many whole levels as the shadow
     * kernel's LDS area holds (every query starts there, and gathers from LDS do not queue at the vector memory
     * path), then all others in index order - depth first, right child first, like the node array. */
    uint32_t top_levels = 0, n_top = 0;
    while (top_levels < per_depth.size() && n_top + per_depth[top_levels] <= SRT_WIDE_LDS_NODES)
        n_top += per_depth[top_levels++];
    uint32_t next_top = 0, next_rest = n_top;
    for (uint32_t i = 0; i < v.n_nodes; ++i)
        if (!(depth[i] & 1u))
            wide_of[i] = depth[i] / 2 < top_levels ? next_top++ : next_rest++;
    const uint32_t n_wide = next_rest;
    /* Stack capacity of the wide walk, in entries of the shadow kernel's LDS stacks (shadow_stack_entries). */
    v.wide_cap = shadow_stack_entries(s->num_levels);
    if (const char* cap = getenv("SRT_WIDE_CAP")) {
        const long c = strtol(cap, nullptr, 10);
        if (c >= 0 && (uint32_t)c < v.wide_cap)
            v.wide_cap = (uint32_t)c;
    }
    void *p = nullptr, *map = nullptr, *flag = nullptr;
    HIPCHK(hipMalloc(&p, (size_t)n_wide * 64));
    s->allocs.push_back(p);
    HIPCHK(hipMalloc(&map, (size_t)v.n_nodes * 4));
    hipError_t e = hipMalloc(&flag, 4);
    uint32_t bad = 0;
    if (e == hipSuccess)
        e = hipMemcpy(map, wide_of.data(), (size_t)v.n_nodes * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemset(flag, 0, 4);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_widen, dim3((v.n_nodes + SRT_BLOCK - 1) / SRT_BLOCK), dim3(SRT_BLOCK), 0, 0, v.nodes, v.n_nodes, (const uint32_t*)map, v.tris, g, (uint4*)p,
            (uint32_t*)flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpy(&bad, flag, 4, hipMemcpyDeviceToHost);
    (void)hipFree(map);
    if (flag)
        (void)hipFree(flag);
    if (e != hipSuccess)
        return fail(SRT_ERR_HIP, "wide node array: %s", hipGetErrorString(e));
    v.wnodes = (const uint4*)p;
    v.n_wide = n_wide;
    v.n_top = n_top;
    v.coarse_ok = bad ? 0u : 1u;
    /* the coarse walk's compact triangle records */
    if (v.coarse_ok && v.n_tris) {
        void *ct = nullptr, *cflag = nullptr;
        uint32_t differs = 1;
        HIPCHK(hipMalloc(&ct, (size_t)v.n_tris * 48));
        s->allocs.push_back(ct);
        e = hipMalloc(&cflag, 4);
        if (e == hipSuccess)
            e = hipMemset(cflag, 0, 4);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_compact_triangles, dim3((v.n_tris + SRT_BLOCK - 1) / SRT_BLOCK), dim3(SRT_BLOCK), 0, 0, v.tris, v.n_tris, (float4*)ct, (uint32_t*)cflag);
            e = hipGetLastError();
        }
        if (e == hipSuccess)
            e = hipMemcpy(&differs, cflag, 4, hipMemcpyDeviceToHost);
        if (cflag)
            (void)hipFree(cflag);
        if (e != hipSuccess)
            return fail(SRT_ERR_HIP, "compact triangle records: %s", hipGetErrorString(e));
        if (!differs && !getenv("SRT_FULL_SHADOW_TRIANGLES"))
            v.stris = (const float4*)ct;
    }
    return SRT_OK;
}

/* Called once per scene upload, so that no render call has to query occupancies. */
static int scene_grids(srt_device* d, srt_scene* sc)
{
    int rc = scene_coarsen(sc);
    if (rc != SRT_OK)
        return rc;
    rc = fill_grids<false, false>(d, sc);
    if (rc == SRT_OK) rc = fill_grids<false, true>(d, sc);
    if (rc == SRT_OK) rc = fill_grids<true, false>(d, sc);
    if (rc == SRT_OK) rc = fill_grids<true, true>(d, sc);
    return rc;
}

static int dispatch_render(srt_device* d, srt_scene* sc, const RenderDev& rp, uint32_t features, uint32_t flags, hipStream_t st, const Work& wk)
{
    const bool cull = (features & SRT_F_MISS_CULL) != 0;
    const bool stats = (flags & SRT_RENDER_STATS) != 0;
    if (cull && stats) return render_impl<true, true>(d, sc, rp, flags, st, wk);
    if (cull) return render_impl<true, false>(d, sc, rp, flags, st, wk);
    if (stats) return render_impl<false, true>(d, sc, rp, flags, st, wk);
    return render_impl<false, false>(d, sc, rp, flags, st, wk);
}

/* Work memory out of the device handle's growable buffers, sized for n samples. */
static int handle_work(srt_device* d, srt_scene* sc, uint32_t n, uint32_t features, int32_t max_depth, Work* wk)
{
    const uint32_t sph = shadows_per_hit(sc, features);
    int rc = d->acc.reserve((size_t)n * 12);
    if (rc == SRT_OK) rc = d->hq_slot.reserve((size_t)n * 4);
    if (rc == SRT_OK) rc = d->hq_t.reserve((size_t)n * 4);
    if (rc == SRT_OK) rc = d->hq_tri.reserve((size_t)n * 4);
    if (rc == SRT_OK) rc = d->hq_ip.reserve((size_t)n * 16);
    /* one visibility byte per (hit, shadow sample); every sample of the rectangle may hit */
    if (rc == SRT_OK) rc = d->vis.reserve((size_t)n * sph + 64);
    if (rc == SRT_OK && (features & (SRT_F_RECURSIVE | SRT_F_TRANSPARENCY))) {
        /* with transparency a hit may send on two rays */
        const size_t per = (features & SRT_F_TRANSPARENCY) ? 96 : 48;
        rc = d->path[0].reserve((size_t)n * per);
        if (rc == SRT_OK) rc = d->path[1].reserve((size_t)n * per);
    }
    bool second = false;
    if (rc == SRT_OK && (features & SRT_F_RECURSIVE) && !(features & SRT_F_TRANSPARENCY) && max_depth > 0) {
        rc = d->chain.reserve((size_t)n * 24 * (size_t)max_depth);
        if (rc == SRT_OK) rc = d->nlev.reserve(n);
        /* frames short enough to overlap their two chains of launches (render_impl) get a second hit queue */
        if (rc == SRT_OK && n < (1u << 23)) {
            rc = d->hq2_slot.reserve((size_t)n * 4);
            if (rc == SRT_OK) rc = d->hq2_t.reserve((size_t)n * 4);
            if (rc == SRT_OK) rc = d->hq2_tri.reserve((size_t)n * 4);
            if (rc == SRT_OK) rc = d->hq2_ip.reserve((size_t)n * 16);
            second = rc == SRT_OK;
        }
    }
    if (rc != SRT_OK)
        return rc;
    if (second)
        wk->hq2 = { (uint32_t*)d->hq2_slot.p, (float*)d->hq2_t.p, (uint32_t*)d->hq2_tri.p, (float4*)d->hq2_ip.p };
    wk->acc = (float*)d->acc.p;
    wk->chain = (float*)d->chain.p, wk->nlev = (unsigned char*)d->nlev.p;
    wk->hq = { (uint32_t*)d->hq_slot.p, (float*)d->hq_t.p, (uint32_t*)d->hq_tri.p, (float4*)d->hq_ip.p };
    wk->vis = (unsigned char*)d->vis.p;
    wk->path[0] = (float4*)d->path[0].p, wk->path[1] = (float4*)d->path[1].p;
    wk->qcounts = d->qcounts;
    wk->chunk_counters = d->chunk_counters;
    wk->growable = true;
    return SRT_OK;
}

/* Checks a render call's arguments and turns them into the kernels' view of the rectangle (everything but the
 * output pointers). */
static int rect_setup(srt_device* d, srt_scene* sc, const srt_render_params* p, const srt_output* out, const char* who, RenderDev* rpo)
{
    if (!d || !sc || !p || !out || !out->rgb)
        return fail(SRT_ERR_INVALID, "%s: null argument", who);
    if (sc->dev != d)
        return fail(SRT_ERR_INVALID, "%s: scene belongs to another device handle", who);
    const srt_camera& cam = p->camera;
    if (cam.width <= 0 || cam.height <= 0)
        return fail(SRT_ERR_INVALID, "camera resolution %dx%d", cam.width, cam.height);
    const int x0 = p->rect[0], y0 = p->rect[1], x1 = p->rect[2], y1 = p->rect[3];
    if (x0 < 0 || y0 < 0 || x1 > cam.width || y1 > cam.height || x0 >= x1 || y0 >= y1)
        return fail(SRT_ERR_INVALID, "rectangle [%d,%d)x[%d,%d) is empty or outside the %dx%d frame", x0, x1, y0, y1, cam.width, cam.height);
    if (x0 < out->x0 || y0 < out->y0 || x1 > out->x0 + out->out_width || y1 > out->y0 + out->out_height)
        return fail(SRT_ERR_INVALID, "rectangle does not fit the output buffer (%dx%d at %d,%d)", out->out_width, out->out_height, out->x0, out->y0);
    int nsx = 1, nsy = 1;
    if (p->features & SRT_F_MULTISAMPLE) {
        nsx = p->samples_x, nsy = p->samples_y;
        if (nsx < 1 || nsy < 1 || nsx * nsy > 4096)
            return fail(SRT_ERR_INVALID, "sample grid %dx%d", nsx, nsy);
    }
    if (p->max_depth < 0 || p->max_depth > 30)
        return fail(SRT_ERR_INVALID, "max_depth %d outside [0,30]", p->max_depth);
    /* the else-if chain of reference src/render.cpp:196-227: multisampling, else motion blur, else depth of field */
    int lens_mode = 0;
    if (!(p->features & SRT_F_MULTISAMPLE))
        lens_mode = (p->features & SRT_F_MOTION_BLUR) ? 1 : ((p->features & SRT_F_DEPTH_OF_FIELD) ? 2 : 0);
    if (lens_mode) {
        if (p->lens.samples < 0 || p->lens.samples > 4096)
            return fail(SRT_ERR_INVALID, "lens.samples %d outside [0,4096]", p->lens.samples);
        nsx = p->lens.samples > 0 ? p->lens.samples : 1, nsy = 1; /* with zero samples nothing is traced, one unused slot per pixel */
    }
    if ((p->features & SRT_F_ENVIRONMENT) && sc->view.env_tex < 0)
        return fail(SRT_ERR_INVALID, "SRT_F_ENVIRONMENT: the scene was uploaded without an environment image");
    /* sample slots: the rectangle padded to whole 8 x 8-pixel blocks (sample_decode) */
    const uint64_t n64 = (uint64_t)((x1 - x0 + 7) / 8) * (uint64_t)((y1 - y0 + 7) / 8) * 64ull * (uint64_t)(nsx * nsy);
    if (n64 >= (1ull << 31))
        return fail(SRT_ERR_INVALID, "rectangle holds %llu samples; split it below 2^31", (unsigned long long)n64);
    if ((uint64_t)cam.width * cam.height * (uint64_t)(nsx * nsy) >= (1ull << 32))
        return fail(SRT_ERR_INVALID, "frame holds more than 2^32 samples");
    RenderDev rp {};
    rp.cam = cam;
    rp.features = p->features;
    rp.nsx = nsx, rp.nsy = nsy, rp.S = nsx * nsy;
    rp.max_depth = p->max_depth;
    rp.seed = p->seed;
    rp.x0 = x0, rp.y0 = y0, rp.tw = x1 - x0, rp.th = y1 - y0;
    rp.n_samples = (uint32_t)n64;
    rp.out_w = out->out_width, rp.out_h = out->out_height, rp.out_x0 = out->x0, rp.out_y0 = out->y0;
    rp.lens = p->lens;
    rp.lens_mode = lens_mode;
    *rpo = rp;
    return SRT_OK;
}

extern "C" {

int srt_render(srt_device* d, srt_scene* sc, const srt_render_params* p, const srt_output* out, void* stream)
{
    RenderDev rp {};
    int rc = rect_setup(d, sc, p, out, "srt_render", &rp);
    if (rc != SRT_OK)
        return rc;
    HIPCHK(hipSetDevice(d->gpu));
    hipStream_t st = (hipStream_t)stream;
    Work wk;
    rc = handle_work(d, sc, rp.n_samples, p->features, rp.max_depth, &wk);
    if (rc != SRT_OK)
        return rc;
    if (out->on_device) {
        rp.out_rgb = out->rgb, rp.out_mesh = out->prim_mesh, rp.out_tri = out->prim_tri, rp.out_t = out->prim_t;
    } else {
        /* host output: the rectangle is rendered into staging buffers of its own size and its rows are copied into the
         * caller's arrays; whatever those hold outside the rectangle is not touched, and a frame rendered rectangle by
         * rectangle moves every pixel once */
        const size_t px = (size_t)rp.tw * rp.th;
        rc = d->out_rgb.reserve(px * 12);
        if (rc == SRT_OK && out->prim_mesh) rc = d->out_mesh.reserve(px * 4);
        if (rc == SRT_OK && out->prim_tri) rc = d->out_tri.reserve(px * 4);
        if (rc == SRT_OK && out->prim_t) rc = d->out_t.reserve(px * 4);
        if (rc != SRT_OK)
            return rc;
        rp.out_rgb = (float*)d->out_rgb.p;
        rp.out_mesh = out->prim_mesh ? (int32_t*)d->out_mesh.p : nullptr;
        rp.out_tri = out->prim_tri ? (int32_t*)d->out_tri.p : nullptr;
        rp.out_t = out->prim_t ? (float*)d->out_t.p : nullptr;
        rp.out_w = rp.tw, rp.out_h = rp.th, rp.out_x0 = rp.x0, rp.out_y0 = rp.y0;
    }
    rc = dispatch_render(d, sc, rp, p->features, p->flags, st, wk);
    if (rc != SRT_OK)
        return rc;
    if (!out->on_device) {
        /* staging row 0 is the rectangle's top row (y = y1 - 1): row out_height - (y1 - out->y0) of the caller's array */
        const size_t row0 = (size_t)(out->out_height - (rp.y0 + rp.th - out->y0)), col0 = (size_t)(rp.x0 - out->x0);
        const size_t first = row0 * (size_t)out->out_width + col0;
        HIPCHK(hipMemcpy2DAsync(out->rgb + 3 * first, (size_t)out->out_width * 12, rp.out_rgb, (size_t)rp.tw * 12, (size_t)rp.tw * 12, (size_t)rp.th,
            hipMemcpyDeviceToHost, st));
        if (rp.out_mesh)
            HIPCHK(hipMemcpy2DAsync(out->prim_mesh + first, (size_t)out->out_width * 4, rp.out_mesh, (size_t)rp.tw * 4, (size_t)rp.tw * 4, (size_t)rp.th,
                hipMemcpyDeviceToHost, st));
        if (rp.out_tri)
            HIPCHK(hipMemcpy2DAsync(out->prim_tri + first, (size_t)out->out_width * 4, rp.out_tri, (size_t)rp.tw * 4, (size_t)rp.tw * 4, (size_t)rp.th,
                hipMemcpyDeviceToHost, st));
        if (rp.out_t)
            HIPCHK(hipMemcpy2DAsync(out->prim_t + first, (size_t)out->out_width * 4, rp.out_t, (size_t)rp.tw * 4, (size_t)rp.tw * 4, (size_t)rp.th,
                hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return SRT_OK;
}

int srt_render_noalloc(srt_device* d, srt_scene* sc, const srt_render_params* p, const srt_output* out, void* stream)
{
    RenderDev rp {};
    int rc = rect_setup(d, sc, p, out, "srt_render_noalloc", &rp);
    if (rc != SRT_OK)
        return rc;
    if (!out->on_device)
        return fail(SRT_ERR_INVALID, "srt_render_noalloc: the output must be device memory (host output needs staging buffers; use srt_render)");
    if (p->features & SRT_F_TRANSPARENCY)
        return fail(SRT_ERR_INVALID, "srt_render_noalloc: transparency needs queues that grow with the frame; use srt_render");
    if (p->flags & SRT_RENDER_TIMING)
        return fail(SRT_ERR_INVALID, "srt_render_noalloc: SRT_RENDER_TIMING creates events; use srt_render");
    if (d->gpu >= SRT_MAX_GPUS || !g_scratch_ptr[d->gpu])
        return fail(SRT_ERR_INVALID, "srt_render_noalloc: no scratch address for GPU %d", d->gpu);
    /* carve the rectangle's work memory out of the scratch array, every piece 256-byte aligned */
    const uint32_t n = rp.n_samples;
    const uint32_t sph = shadows_per_hit(sc, p->features);
    const bool paths = (p->features & SRT_F_RECURSIVE) != 0;
    unsigned char* const base = g_scratch_ptr[d->gpu];
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        unsigned char* q = base + off;
        off += (bytes + 255) & ~(size_t)255;
        return q;
    };
    Work wk;
    wk.qcounts = (uint32_t*)carve(64 * sizeof(uint32_t));
    wk.chunk_counters = (unsigned long long*)carve(64 * sizeof(unsigned long long));
    wk.acc = (float*)carve((size_t)n * 12);
    wk.hq.slot = (uint32_t*)carve((size_t)n * 4);
    wk.hq.t = (float*)carve((size_t)n * 4);
    wk.hq.tri = (uint32_t*)carve((size_t)n * 4);
    wk.hq.ip = (float4*)carve((size_t)n * 16);
    wk.vis = carve((size_t)n * sph + 64);
    if (paths) {
        wk.path[0] = (float4*)carve((size_t)n * 48);
        wk.path[1] = (float4*)carve((size_t)n * 48);
        wk.chain = (float*)carve((size_t)n * 24 * (size_t)(rp.max_depth > 0 ? rp.max_depth : 0));
        wk.nlev = carve(n);
    }
    if (off > (size_t)SRT_NOALLOC_SCRATCH_BYTES)
        return fail(SRT_ERR_INVALID, "srt_render_noalloc: the rectangle's %u sample slots need %zu bytes of work memory, the fixed scratch area holds %zu: "
                                     "render a smaller rectangle or use srt_render", n, off, (size_t)SRT_NOALLOC_SCRATCH_BYTES);
    HIPCHK(hipSetDevice(d->gpu));
    rp.out_rgb = out->rgb, rp.out_mesh = out->prim_mesh, rp.out_tri = out->prim_tri, rp.out_t = out->prim_t;
    /* render_impl clears the queue counts, the chunk counters and the accumulators of this call; everything else
     * in the area is written before it is read */
    return dispatch_render(d, sc, rp, p->features, p->flags, (hipStream_t)stream, wk);
}

int srt_intersect_batch(srt_device* d, srt_scene* sc, float* rays7, int64_t n, uint32_t features, srt_hit* hits, srt_counters* counters)
{
    if (!d || !sc || n < 0 || (n > 0 && (!rays7 || !hits)))
        return fail(SRT_ERR_INVALID, "srt_intersect_batch: bad argument");
    if (sc->dev != d)
        return fail(SRT_ERR_INVALID, "srt_intersect_batch: scene belongs to another device handle");
    if (counters)
        memset(counters, 0, sizeof *counters);
    if (n == 0)
        return SRT_OK;
    static_assert(sizeof(srt_hit) == 60, "srt_hit layout");
    HIPCHK(hipSetDevice(d->gpu));
    int rc = d->io_a.reserve((size_t)n * 28);
    if (rc == SRT_OK) rc = d->io_b.reserve((size_t)n * sizeof(srt_hit));
    if (rc != SRT_OK)
        return rc;
    HIPCHK(hipMemcpy(d->io_a.p, rays7, (size_t)n * 28, hipMemcpyHostToDevice));
    DevCounters* tmp = d->counters + SRT_FAMILIES; /* the handle's slot for this entry point */
    hipError_t e = hipMemset(tmp, 0, sizeof(DevCounters));
    const size_t lds = stack_bytes(sc);
    int grid = (int)((n + SRT_BLOCK - 1) / SRT_BLOCK);
    if (grid > d->n_cu * 16) grid = d->n_cu * 16;
    if (e == hipSuccess) {
        if (features & SRT_F_MISS_CULL)
            hipLaunchKernelGGL(k_intersect_batch<true>, dim3(grid), dim3(SRT_BLOCK), lds, 0, sc->view, (float*)d->io_a.p, (long long)n, features,
                (srt_hit*)d->io_b.p, tmp);
        else
            hipLaunchKernelGGL(k_intersect_batch<false>, dim3(grid), dim3(SRT_BLOCK), lds, 0, sc->view, (float*)d->io_a.p, (long long)n, features,
                (srt_hit*)d->io_b.p, tmp);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(rays7, d->io_a.p, (size_t)n * 28, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(hits, d->io_b.p, (size_t)n * sizeof(srt_hit), hipMemcpyDeviceToHost);
    DevCounters c {};
    if (e == hipSuccess) e = hipMemcpy(&c, tmp, sizeof c, hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return fail(SRT_ERR_HIP, "srt_intersect_batch: %s", hipGetErrorString(e));
    if (counters) {
        counters->path_rays = c.path_rays, counters->shadow_rays = c.shadow_rays, counters->shaded_hits = c.shaded_hits;
        counters->inner_visits = c.inner_visits, counters->tri_tests = c.tri_tests, counters->box_tests = c.box_tests;
        counters->wave_inner_steps = c.wave_inner_steps, counters->wave_leaf_steps = c.wave_leaf_steps;
    }
    return SRT_OK;
}

int srt_final_color_batch(srt_device* d, srt_scene* sc, const float* rays7, int64_t n, uint32_t features, int32_t max_depth, uint32_t seed, float* rgb)
{
    if (!d || !sc || n < 0 || (n > 0 && (!rays7 || !rgb)))
        return fail(SRT_ERR_INVALID, "srt_final_color_batch: bad argument");
    if (sc->dev != d)
        return fail(SRT_ERR_INVALID, "srt_final_color_batch: scene belongs to another device handle");
    if (n >= (1ll << 31) || max_depth < 0 || max_depth > 30)
        return fail(SRT_ERR_INVALID, "srt_final_color_batch: n or max_depth out of range");
    if (n == 0)
        return SRT_OK;
    HIPCHK(hipSetDevice(d->gpu));
    int rc = d->io_a.reserve((size_t)n * 28);
    if (rc == SRT_OK) rc = d->io_b.reserve((size_t)n * 12);
    Work wk;
    if (rc == SRT_OK) rc = handle_work(d, sc, (uint32_t)n, features, max_depth, &wk);
    if (rc != SRT_OK)
        return rc;
    HIPCHK(hipMemcpy(d->io_a.p, rays7, (size_t)n * 28, hipMemcpyHostToDevice));
    RenderDev rp {};
    rp.cam.width = (int32_t)n, rp.cam.height = 1;
    rp.features = features & ~SRT_F_MULTISAMPLE;
    rp.nsx = rp.nsy = rp.S = 1;
    rp.max_depth = max_depth;
    rp.seed = seed;
    rp.x0 = rp.y0 = 0, rp.tw = (int32_t)n, rp.th = 1;
    rp.n_samples = (uint32_t)n;
    rp.out_rgb = (float*)d->io_b.p;
    rp.out_w = (int32_t)n, rp.out_h = 1;
    rp.custom_rays = (const float*)d->io_a.p;
    rc = dispatch_render(d, sc, rp, features, 0, 0, wk);
    if (rc != SRT_OK)
        return rc;
    HIPCHK(hipMemcpy(rgb, d->io_b.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    return SRT_OK;
}

int srt_selftest_division(srt_device* d, uint32_t seed, uint64_t n, uint64_t* mismatches)
{
    if (!d || !mismatches)
        return fail(SRT_ERR_INVALID, "srt_selftest_division: null argument");
    HIPCHK(hipSetDevice(d->gpu));
    unsigned long long* dev = nullptr;
    HIPCHK(hipMalloc((void**)&dev, 8));
    hipError_t e = hipMemset(dev, 0, 8);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_selftest_division, dim3(d->n_cu * 8), dim3(SRT_BLOCK), 0, 0, seed, (unsigned long long)n, dev);
        e = hipGetLastError();
    }
    unsigned long long host = 0;
    if (e == hipSuccess) e = hipMemcpy(&host, dev, 8, hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e != hipSuccess)
        return fail(SRT_ERR_HIP, "srt_selftest_division: %s", hipGetErrorString(e));
    *mismatches = host;
    return SRT_OK;
}

int srt_probe_gather(srt_device* d, srt_scene* sc, uint64_t window_bytes, uint32_t loads_per_lane, uint32_t quad, double* gbytes_per_s,
                     uint64_t* window_used)
{
    if (!d || !sc || !gbytes_per_s || sc->dev != d)
        return fail(SRT_ERR_INVALID, "srt_probe_gather: bad argument");
    const uint64_t node_units = sc->view.n_nodes, tri_units = (uint64_t)sc->view.n_tris * 96 / 64;
    if (node_units + tri_units == 0)
        return fail(SRT_ERR_INVALID, "srt_probe_gather: the scene has no records");
    uint64_t units = window_bytes / 64;
    if (units == 0 || units > node_units + tri_units)
        units = node_units + tri_units;
    const uint32_t n_nodes64 = (uint32_t)(units < node_units ? units : node_units);
    const uint32_t n_tris64 = (uint32_t)(units - n_nodes64);
    loads_per_lane = (loads_per_lane + 7u) & ~7u;
    if (loads_per_lane == 0)
        loads_per_lane = 64;
    HIPCHK(hipSetDevice(d->gpu));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    const int grid = d->n_cu * 8;
    float* sink = (float*)(d->counters + SRT_FAMILIES);
    /* one untimed pass brings the window into the caches it fits */
    auto kern = quad ? k_probe_gather_quad : k_probe_gather;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SRT_BLOCK), 0, 0, sc->view.nodes, n_nodes64, sc->view.tris, n_tris64, loads_per_lane, 1u, sink);
    hipError_t e = hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SRT_BLOCK), 0, 0, sc->view.nodes, n_nodes64, sc->view.tris, n_tris64, loads_per_lane, 2u, sink);
    if (e == hipSuccess) e = hipEventRecord(e1, 0);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess || !(ms > 0))
        return fail(SRT_ERR_HIP, "srt_probe_gather: %s", hipGetErrorString(e));
    *gbytes_per_s = (double)grid * SRT_BLOCK * loads_per_lane * 64.0 / (ms * 1e-3) / 1e9;
    if (window_used)
        *window_used = units * 64;
    return SRT_OK;
}

int srt_generate_rays(srt_device* d, const srt_camera* cam, const float* ndc2, int64_t n, float* rays7)
{
    if (!d || !cam || n < 0 || (n > 0 && (!ndc2 || !rays7)))
        return fail(SRT_ERR_INVALID, "srt_generate_rays: bad argument");
    if (n == 0)
        return SRT_OK;
    HIPCHK(hipSetDevice(d->gpu));
    int rc = d->io_a.reserve((size_t)n * 8);
    if (rc == SRT_OK) rc = d->io_b.reserve((size_t)n * 28);
    if (rc != SRT_OK)
        return rc;
    HIPCHK(hipMemcpy(d->io_a.p, ndc2, (size_t)n * 8, hipMemcpyHostToDevice));
    int grid = (int)((n + SRT_BLOCK - 1) / SRT_BLOCK);
    if (grid > d->n_cu * 16) grid = d->n_cu * 16;
    hipLaunchKernelGGL(k_generate, dim3(grid), dim3(SRT_BLOCK), 0, 0, *cam, (const float*)d->io_a.p, (long long)n, (float*)d->io_b.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(rays7, d->io_b.p, (size_t)n * 28, hipMemcpyDeviceToHost));
    return SRT_OK;
}

} // extern "C"

#include "srt_build.h"
