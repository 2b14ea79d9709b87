/*
 * srt_device.h -- device-side arithmetic of the ray tracer's hot path for gfx950 (wave64).
 *
 * Every function performs the reference's float operations in the reference's order; the translation unit
 * must be compiled with -ffp-contract=off (hipcc contracts a*b+c into an FMA by default, the reference's
 * x86-64 build does not) and with hipcc's default correctly-rounded division and square root.
 * std::min / std::max / glm::min / glm::max are (b < a) ? b : a and (a < b) ? b : a: NaN handling differs
 * from fminf / fmaxf, so they are spelled out.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/srt_hip.h"

#define SRT_FLT_MAX 3.402823466e+38f
#ifndef SRT_BLOCK
#define SRT_BLOCK 256
#endif

struct f3 {
    float x, y, z;
};
struct f2 {
    float x, y;
};

#define SRT_DI __device__ __forceinline__

SRT_DI f3 mk3(float x, float y, float z)
{
    f3 r;
    r.x = x, r.y = y, r.z = z;
    return r;
}
SRT_DI f3 xyz(float4 q) { return mk3(q.x, q.y, q.z); }
SRT_DI f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
SRT_DI f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
SRT_DI f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
SRT_DI f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
SRT_DI f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
SRT_DI f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
SRT_DI f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
/* glm dot / cross / length / normalize, scalar paths (glm/detail/func_geometric.inl:8-14,48-55,68-90) */
SRT_DI float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
SRT_DI f3 cross3(f3 x, f3 y) { return mk3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
SRT_DI float length3(f3 v) { return __builtin_sqrtf(dot3(v, v)); }
SRT_DI f3 normalize3(f3 v) { return v * (1.0f / __builtin_sqrtf(dot3(v, v))); }
SRT_DI float min_std(float a, float b) { return (b < a) ? b : a; }
SRT_DI float max_std(float a, float b) { return (a < b) ? b : a; }
SRT_DI float clamp01(float x) { return min_std(max_std(x, 0.0f), 1.0f); }

/* Ray/box slab test, reference src/intersect.cpp:134-160, in the form the traversal needs.
 * Returns whether the test passes; val is what getClosestIntersectionWithBox
 * (src/bounding_volume_hierarchy.cpp:305-310) returns: T on a failure, t_out when the origin is inside or
 * on the box (t_in <= 0), else t_in.  inside tells the two passing cases apart. */
SRT_DI bool slab_test(const float lo[3], const float hi[3], f3 o, f3 d, float T, float& val, bool& inside)
{
    float txmin = (lo[0] - o.x) / d.x;
    float tymin = (lo[1] - o.y) / d.y;
    float tzmin = (lo[2] - o.z) / d.z;
    float txmax = (hi[0] - o.x) / d.x;
    float tymax = (hi[1] - o.y) / d.y;
    float tzmax = (hi[2] - o.z) / d.z;
    float tinx = min_std(txmin, txmax), toutx = max_std(txmin, txmax);
    float tiny = min_std(tymin, tymax), touty = max_std(tymin, tymax);
    float tinz = min_std(tzmin, tzmax), toutz = max_std(tzmin, tzmax);
    float tin = max_std(tinx, max_std(tiny, tinz));
    float tout = min_std(toutx, min_std(touty, toutz));
    inside = false;
    if (tin > tout || tout < 0 || tin > T) {
        val = T;
        return false;
    }
    val = tin;
    if (tin <= 0) {
        val = tout;
        inside = true;
    }
    return true;
}

/* ---- division by a ray-constant denominator.
 * The slab test divides twelve differences by the same three direction components at every node.  hipcc expands
 * a correctly rounded x / d into: scale operands if their exponents are extreme (v_div_scale), r0 = v_rcp(d),
 * e = fma(-d, r0, 1), r1 = fma(e, r0, r0), q = x * r1, s = fma(-d, q, x), q = fma(s, r1, q), s = fma(-d, q, x),
 * q = fma(s, r1, q) (v_div_fmas), then a fix-up for zero / infinite / NaN operands (v_div_fixup).  When no scaling
 * and no fix-up can trigger, the quotient is exactly the five operations after r1, and r1 depends on d alone, so
 * it is computed once per ray.  "No scaling, no fix-up" is guaranteed by keeping operands in these ranges:
 *   direction component   2^-20 <= |d| <= 2^20
 *   origin / box coordinate   0, or 2^-40 <= |c| < 2^20      (so x = box - origin is 0 or |x| >= 2^-63)
 *   box                       lower <= upper on every axis   (false only for the empty scene's root)
 * which puts every quotient and every intermediate in the normal range.  A zero numerator gives a zero of
 * possibly the other sign than x / d; slab values only feed comparisons, for which the sign of zero is irrelevant.
 * Rays or scenes outside the ranges take the plain-division path. */
struct RayRcp {
    float x, y, z;
};
SRT_DI float refined_rcp(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
SRT_DI float div_by(float x, float d, float r1)
{
#ifdef SRT_EXPERIMENT_INEXACT_DIV
    return x * r1; /* timing experiment only: not the reference's quotient */
#endif
    float q = x * r1;
    float s = __builtin_fmaf(-d, q, x);
    q = __builtin_fmaf(s, r1, q);
    s = __builtin_fmaf(-d, q, x);
    return __builtin_fmaf(s, r1, q);
}
SRT_DI bool dir_in_fast_range(float d)
{
    const uint32_t e = (__float_as_uint(d) >> 23) & 0xFFu;
    return e >= 107u && e <= 147u; /* 2^-20 <= |d| < 2^21; 2^20 itself has e == 147 */
}
SRT_DI bool coord_in_fast_range(float c)
{
    const uint32_t b = __float_as_uint(c) & 0x7FFFFFFFu;
    const uint32_t e = b >> 23;
    return b == 0u || (e >= 87u && e < 147u);
}

/* slab_test for operands in the fast ranges and a box with lower <= upper on every axis; same val / inside /
 * result.  Correctly rounded subtraction and division are monotone, so with d > 0 the lower bound gives the
 * smaller quotient and std::min(t_min, t_max) IS t_min (t_max for d < 0): the per-axis min / max of the reference
 * become a selection of the near and far bound by the sign of d, made before the division.  No operand can be
 * NaN here, so the three-way max / min of the reference equal v_max3_f32 / v_min3_f32. */
SRT_DI bool slab_test_fast(const float lo[3], const float hi[3], f3 o, f3 d, RayRcp r, float T, float& val, bool& inside)
{
    const bool nx = d.x < 0, ny = d.y < 0, nz = d.z < 0;
    const float tinx = div_by((nx ? hi[0] : lo[0]) - o.x, d.x, r.x);
    const float tiny = div_by((ny ? hi[1] : lo[1]) - o.y, d.y, r.y);
    const float tinz = div_by((nz ? hi[2] : lo[2]) - o.z, d.z, r.z);
    const float toutx = div_by((nx ? lo[0] : hi[0]) - o.x, d.x, r.x);
    const float touty = div_by((ny ? lo[1] : hi[1]) - o.y, d.y, r.y);
    const float toutz = div_by((nz ? lo[2] : hi[2]) - o.z, d.z, r.z);
    const float tin = __builtin_fmaxf(tinx, __builtin_fmaxf(tiny, tinz));
    const float tout = __builtin_fminf(toutx, __builtin_fminf(touty, toutz));
    /* slab_test's two ifs as selects (nested exec-mask regions otherwise, twice per visit) */
    const bool pass = !(tin > tout || tout < 0 || tin > T);
    inside = pass && tin <= 0;
    val = pass ? (inside ? tout : tin) : T;
    return pass;
}

/* Ray/triangle test, reference src/intersect.cpp:13-77, on a precomputed srt_tri record (six float4).
 * On acceptance T becomes the new distance. */
SRT_DI bool tri_test(const float4* __restrict__ rec, f3 o, f3 d, float& T)
{
    const float4 q0 = rec[0];
    const f3 n = xyz(q0);
    const float D = q0.w;
    const float dn = dot3(d, n);
    const float on = dot3(o, n);
    if (dn == 0.0f || on == D)
        return false;
    const float nt = (D - on) / dn;
    if (nt < 0)
        return false;
    if (!(nt <= T))
        return false;
    const f3 p = o + d * nt;
    const float4 q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4], q5 = rec[5];
    const f3 v0 = xyz(q1), v1 = xyz(q2), v2 = mk3(q3.x, q3.y, q3.z);
    const f3 c0 = mk3(q3.w, q4.x, q4.y), c1 = mk3(q4.z, q4.w, q5.x), c2 = mk3(q5.y, q5.z, q5.w);
    /* sameSide(p, v0, v1, v2) && sameSide(p, v1, v0, v2) && sameSide(p, v2, v0, v1).  All three are evaluated
     * (no short circuit): the record's five remaining quarters are then fetched together and waited for once,
     * instead of in three dependent rounds; the lanes of a wave rarely agree on an early exit anyway. */
    const bool s0 = dot3(cross3(v2 - v1, p - v1), c0) >= 0;
    const bool s1 = dot3(cross3(v2 - v0, p - v0), c1) >= 0;
    const bool s2 = dot3(cross3(v1 - v0, p - v0), c2) >= 0;
    if (!(s0 & s1 & s2))
        return false;
    T = nt;
    return true;
}

/* tri_test without branches, for the shadow walk: all six quarters of the record are requested at once and every
 * lane runs the whole test; the early exits of tri_test become terms of the final conjunction.  Same operations on the
 * same operands in the same order, so the same decision and the same distance bit for bit (a lane that tri_test would
 * have sent away early computes values nobody looks at: a division by zero raises nothing here).  One round trip to
 * memory per triangle instead of two (plane first, the rest after the plane test), which is what a leaf step of the
 * shadow walk waits for. */
SRT_DI bool tri_test_flat(const float4* __restrict__ rec, f3 o, f3 d, float& T)
{
    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4], q5 = rec[5];
    const f3 n = xyz(q0);
    const float D = q0.w;
    const float dn = dot3(d, n);
    const float on = dot3(o, n);
    const float nt = (D - on) / dn;
    const bool plane = !(dn == 0.0f || on == D) && !(nt < 0) && (nt <= T);
    const f3 p = o + d * nt;
    const f3 v0 = xyz(q1), v1 = xyz(q2), v2 = mk3(q3.x, q3.y, q3.z);
    const f3 c0 = mk3(q3.w, q4.x, q4.y), c1 = mk3(q4.z, q4.w, q5.x), c2 = mk3(q5.y, q5.z, q5.w);
    const bool s0 = dot3(cross3(v2 - v1, p - v1), c0) >= 0;
    const bool s1 = dot3(cross3(v2 - v0, p - v0), c1) >= 0;
    const bool s2 = dot3(cross3(v1 - v0, p - v0), c2) >= 0;
    const bool hit = plane && s0 && s1 && s2;
    T = hit ? nt : T;
    return hit;
}

/* tri_test_flat on the 48-byte record of the shadow walk: unit normal and the three vertices, nothing else.  The plane
 * offset D and the second operands of the three inside tests are recomputed from those with the operations that made the
 * stored ones (k_build_triangles; reference src/intersect.cpp:13-26,45-59), so decision and distance are the same bit for
 * bit; the scene only gets such records when that has been checked for every triangle (k_compact_triangles).  Three
 * 16-byte loads per lane and test instead of six, for 35 more vector instructions: the shadow walk is short of the
 * former. */
SRT_DI bool tri_test_compact(const float4* __restrict__ rec, f3 o, f3 d, float& T)
{
    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
    const f3 n = xyz(q0);
    const f3 v0 = mk3(q0.w, q1.x, q1.y), v1 = mk3(q1.z, q1.w, q2.x), v2 = mk3(q2.y, q2.z, q2.w);
    const float D = dot3(v0, n);
    const float dn = dot3(d, n);
    const float on = dot3(o, n);
    const float nt = (D - on) / dn;
    const bool plane = !(dn == 0.0f || on == D) && !(nt < 0) && (nt <= T);
    const f3 p = o + d * nt;
    const f3 e21 = v2 - v1, e20 = v2 - v0, e10 = v1 - v0;
    const f3 c0 = cross3(e21, v0 - v1), c1 = cross3(e20, e10), c2 = cross3(e10, e20);
    const bool s0 = dot3(cross3(e21, p - v1), c0) >= 0;
    const bool s1 = dot3(cross3(e20, p - v0), c1) >= 0;
    const bool s2 = dot3(cross3(e10, p - v0), c2) >= 0;
    const bool hit = plane && s0 && s1 && s2;
    T = hit ? nt : T;
    return hit;
}

/* The two halves of tri_test, for callers that test several triangles of a leaf side by side.
 * tri_plane: intersectRayWithPlane up to, not including, the comparison with the ray's t (reference
 * src/intersect.cpp:28-43): false when the ray is parallel to the plane, starts on it or meets it behind its origin. */
SRT_DI bool tri_plane(const float4 q0, f3 o, f3 d, float& nt)
{
    const f3 n = xyz(q0);
    const float dn = dot3(d, n);
    const float on = dot3(o, n);
    nt = (q0.w - on) / dn;
    return !(dn == 0.0f || on == q0.w) && !(nt < 0);
}
/* pointInTriangle for p = o + d * nt (reference src/intersect.cpp:13-26) on the record's three vertex quarters, for
 * the shadow walk.  The second operands of the three sameSide tests are ray independent and the record stores them
 * (tri_test reads them), but here they are recomputed from the vertices, with the operations that made the stored
 * ones: 30 more vector instructions per test against two fewer 16-byte loads per lane, and it is per-lane loads
 * the shadow kernel runs out of (measured: shadow launch of the dragon frame 192.4 -> 183.6 ms; the closest-hit
 * kernels, which are not, lose 2 % that way and keep the loads). */
SRT_DI bool tri_inside(const float4* __restrict__ rec, f3 o, f3 d, float nt)
{
    const f3 p = o + d * nt;
    const float4 q1 = rec[1], q2 = rec[2], q3 = rec[3];
    const f3 v0 = xyz(q1), v1 = xyz(q2), v2 = mk3(q3.x, q3.y, q3.z);
    const f3 e21 = v2 - v1, e20 = v2 - v0, e10 = v1 - v0;
    const f3 c0 = cross3(e21, v0 - v1), c1 = cross3(e20, e10), c2 = cross3(e10, e20);
    const bool s0 = dot3(cross3(e21, p - v1), c0) >= 0;
    const bool s1 = dot3(cross3(e20, p - v0), c1) >= 0;
    const bool s2 = dot3(cross3(e10, p - v0), c2) >= 0;
    return s0 & s1 & s2;
}

/* computeBarycentricCoord, reference src/interpolate.cpp:4-15 */
SRT_DI f3 barycentric3(f3 v0, f3 v1, f3 v2, f3 p)
{
    f3 d0 = v1 - v0, d1 = v2 - v1, d2 = v0 - v2;
    float a0 = length3(cross3(d1, v1 - p));
    float a1 = length3(cross3(d2, v2 - p));
    float a2 = length3(cross3(d0, v0 - p));
    float sum = a0 + a1 + a2;
    if (sum == 0)
        return mk3(1, 0, 0);
    return mk3(a0 / sum, a1 / sum, a2 / sum);
}

/* computeShading, reference src/shading.cpp:6-30 (powf is the only operation that is not correctly
 * rounded on either side; it only reaches colours) */
SRT_DI f3 compute_shading(f3 lightPos, f3 lightCol, f3 ro, f3 rd, float rt, f3 hitNormal, f3 kd, f3 ks, float shininess)
{
    f3 rayPosition = ro + rd * rt;
    f3 L = normalize3(lightPos - rayPosition);
    f3 N = normalize3(hitNormal);
    f3 camV = rayPosition - ro;
    f3 R = ((2.0f * dot3(L, N)) * N) - L;
    f3 lambert = (lightCol * kd) * clamp01(dot3(N, L));
    f3 spec = (lightCol * ks) * powf(clamp01(dot3(R, normalize3(camV))), shininess);
    return lambert + spec;
}

/* Trackball::generateRay, reference framework/src/trackball.cpp:101-110; quaternion * vector as
 * glm/detail/type_quat.inl:347-354 */
SRT_DI f3 camera_dir(const srt_camera& cam, float nx, float ny)
{
    f3 d = normalize3(mk3(-nx * cam.half_w, ny * cam.half_h, 1.0f));
    f3 q = mk3(cam.quat[1], cam.quat[2], cam.quat[3]);
    f3 uv = cross3(q, d);
    f3 uuv = cross3(q, uv);
    return d + ((uv * cam.quat[0]) + uuv) * 2.0f;
}

/* Counter-based uniform sample in [0,1) for the soft-shadow samplers (the reference uses libc rand()). */
SRT_DI uint32_t mix32(uint32_t h, uint32_t k)
{
    h ^= k;
    h *= 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    h *= 0xC2B2AE3Du;
    h ^= h >> 16;
    return h;
}
/* The arguments are folded into one word (each times an odd constant of its own) and that word goes through one
 * two-multiplication finaliser: eight multiplications for the two coordinates of an area-light sample, where a chain of
 * one full mixing round per argument took eighteen (the sampler was 2.5 % of the dragon frame). */
SRT_DI uint32_t final32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
SRT_DI float rand01(uint32_t seed, uint32_t path, uint32_t depth, uint32_t light, uint32_t sample, uint32_t dim)
{
    const uint32_t x = seed ^ (path * 0x9E3779B1u) ^ (depth * 0x85EBCA77u) ^ (light * 0xC2B2AE3Du) ^ (sample * 0x27D4EB2Fu) ^ (dim * 0x165667B1u);
    return (float)(final32(x) >> 8) * (1.0f / 16777216.0f);
}
