/*
 * sim_beam.cpp -- CPU simulation (development tool, not product code): how many hits of a frame could answer all the
 * samples of an area light at once, by walking the tree with the whole bundle of shadow rays (the pyramid from the hit
 * point to the light) and finding no triangle that any ray of the bundle could meet?
 *
 * Compiled together with the oracle's source (it needs the oracle's tree and its per-query work counters); built and
 * driven by tools/sim_beam.py.
 */
#include "../oracle/srt_oracle.cpp"

namespace {

struct Beam {
    V3 P;          /* apex: the hit point */
    V3 S[4];       /* light corners */
    V3 n[5];       /* inward unit normals: four sides through P, the light's plane */
    float off[5];  /* plane offsets: inside means dot(n, x) - off >= 0 */
    V3 lo, hi;     /* bounding box of apex and corners */
    float r;       /* dilation: origin shift of the rays + slack */
    float dmax;    /* longest direction */
    bool ok;
};

Beam make_beam(V3 P, const float* L)
{
    Beam b;
    b.P = P;
    V3 p0 = v3(L + 1), e1 = v3(L + 4), e2 = v3(L + 7);
    b.S[0] = p0, b.S[1] = p0 + e1, b.S[2] = (p0 + e1) + e2, b.S[3] = p0 + e2;
    b.r = 0.001f + 1e-4f;
    b.ok = true;
    b.lo = b.hi = P;
    b.dmax = 0;
    V3 c = { 0, 0, 0 };
    for (int k = 0; k < 4; ++k) {
        b.lo = { fmin_std(b.lo.x, b.S[k].x), fmin_std(b.lo.y, b.S[k].y), fmin_std(b.lo.z, b.S[k].z) };
        b.hi = { fmax_std(b.hi.x, b.S[k].x), fmax_std(b.hi.y, b.S[k].y), fmax_std(b.hi.z, b.S[k].z) };
        b.dmax = fmax_std(b.dmax, length(b.S[k] - P));
        c = c + b.S[k] * 0.25f;
    }
    for (int k = 0; k < 4; ++k) {
        V3 a = b.S[k] - P, d = b.S[(k + 1) & 3] - P;
        V3 n = cross(a, d);
        float l = length(n);
        if (!(l > 1e-12f)) {
            b.ok = false;
            return b;
        }
        n = n * (1.0f / l);
        if (dot(n, c - P) < 0)
            n = -n;
        if (!(dot(n, c - P) > 1e-6f))
            b.ok = false;
        b.n[k] = n;
        b.off[k] = dot(n, P);
    }
    V3 nl = normalize(cross(e1, e2));
    if (dot(nl, P - b.S[0]) < 0)
        nl = -nl;
    if (!(dot(nl, P - b.S[0]) > 1e-4f))
        b.ok = false;
    b.n[4] = nl;
    b.off[4] = dot(nl, b.S[0]);
    return b;
}

/* may the bundle meet the box?  (conservative: false only when surely not) */
bool beam_box(const Beam& b, const Box& x, float grow)
{
    const float r = b.r + grow;
    if (x.lo.x - r > b.hi.x || x.hi.x + r < b.lo.x || x.lo.y - r > b.hi.y || x.hi.y + r < b.lo.y || x.lo.z - r > b.hi.z || x.hi.z + r < b.lo.z)
        return false;
    for (int k = 0; k < 5; ++k) {
        const V3 n = b.n[k];
        const V3 pv = { n.x >= 0 ? x.hi.x : x.lo.x, n.y >= 0 ? x.hi.y : x.lo.y, n.z >= 0 ? x.hi.z : x.lo.z };
        if (dot(n, pv) - b.off[k] < -r)
            return false;
    }
    return true;
}

/* may any ray of the bundle be accepted by the triangle with t < 0.99?  (conservative) */
bool beam_triangle(const Beam& b, V3 v0, V3 v1, V3 v2, int* why)
{
    const float r = b.r;
    *why = 0;
    V3 lo = { fmin_std(v0.x, fmin_std(v1.x, v2.x)), fmin_std(v0.y, fmin_std(v1.y, v2.y)), fmin_std(v0.z, fmin_std(v1.z, v2.z)) };
    V3 hi = { fmax_std(v0.x, fmax_std(v1.x, v2.x)), fmax_std(v0.y, fmax_std(v1.y, v2.y)), fmax_std(v0.z, fmax_std(v1.z, v2.z)) };
    if (lo.x - r > b.hi.x || hi.x + r < b.lo.x || lo.y - r > b.hi.y || hi.y + r < b.lo.y || lo.z - r > b.hi.z || hi.z + r < b.lo.z)
        return false;
    for (int k = 0; k < 5; ++k) {
        const V3 n = b.n[k];
        if (dot(n, v0) - b.off[k] < -r && dot(n, v1) - b.off[k] < -r && dot(n, v2) - b.off[k] < -r)
            return false;
    }
    /* the triangle's own plane */
    V3 c = cross(v1 - v0, v2 - v0);
    if (length(c) == 0.0f)
        return false; /* the reference's test refuses a triangle without a normal */
    V3 n = normalize(c);
    const float D = dot(v0, n);
    const float gP = dot(b.P, n) - D;
    float mmin = FLT_MAX, mmax = -FLT_MAX;
    for (int k = 0; k < 4; ++k) {
        const float m = dot(n, b.S[k] - b.P);
        mmin = fmin_std(mmin, m), mmax = fmax_std(mmax, m);
    }
    const float eps = 1e-5f;
    if (mmin > 0) {
        if (gP + 0.001f * mmin / b.dmax > eps)
            return false;
    } else if (mmax < 0) {
        if (gP + 0.001f * mmax / b.dmax < -eps)
            return false;
    }
    if (gP - 0.001f > eps && gP + mmin - 0.001f > eps)
        return false;
    if (gP + 0.001f < -eps && gP + mmax + 0.001f < -eps)
        return false;
    *why = 1;
    return true;
}


/* A shadow walk over the four-wide coarse slots that is free to choose its order (exact whenever the ray cannot have a
 * hit in [0.99, 1]).  mode 0: the reference's pop order; 1: nearest entry first; 2: farthest first; 3: the slot whose span
 * along the ray lies nearest to t_guide first; 4: longest span first.  Returns occluded; work in *visits, *tris. */
bool walk_order(Ctx& c, const Ray& ray, const HitInfo& like, int mode, float t_guide, double* visits, double* tris, float* t_found, int start = -1)
{
    const orc_bvh& bvh = c.bvh;
    const float o[3] = { ray.o.x, ray.o.y, ray.o.z }, d[3] = { ray.d.x, ray.d.y, ray.d.z };
    const Box& root = bvh.nodes.back().box;
    const float rlo[3] = { root.lo.x, root.lo.y, root.lo.z };
    CoarseRay cr;
    for (int k = 0; k < 3; ++k) {
        const float og = (o[k] - rlo[k]) * bvh.q_scale[k] + 1.0f;
        cr.qi[k] = 1.0f / (d[k] * bvh.q_scale[k]);
        cr.qc[k] = -(og * cr.qi[k]);
    }
    std::vector<int> stack;
    int cur = start >= 0 ? start : (int)bvh.nodes.size() - 1;
    for (;;) {
        bool descended = false;
        if (cur < 0 || bvh.nodes[cur].leaf) {
            const int first = cur < 0 ? -(cur + 1) : bvh.nodes[cur].a, count = cur < 0 ? 1 : bvh.nodes[cur].b;
            Ray r = ray;
            HitInfo h = like;
            bool hit = false;
            for (int k = 0; k < count; ++k) {
                TriRef t = bvh.order[first + k];
                *tris += 1;
                if (leaf_triangle(c, r, h, t.mesh, t.tri))
                    hit = true;
            }
            if (hit && r.t < 0.99f) {
                *t_found = r.t;
                return true;
            }
        } else {
            const Node& n = bvh.nodes[cur];
            *visits += 1;
            int slot[4], ns = 0;
            for (int child : { n.b, n.a }) {
                const Node& cn = bvh.nodes[child];
                if (cn.leaf)
                    slot[ns++] = child;
                else
                    slot[ns++] = cn.b, slot[ns++] = cn.a;
            }
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
                    if (sn.leaf && sn.b != 1)
                        for (int j = 0; j < sn.b; ++j)
                            wide[nw++] = -(sn.a + j + 1);
                    else
                        wide[nw++] = slot[k];
                }
                ns = nw;
                for (int k = 0; k < ns; ++k)
                    slot[k] = wide[k];
            }
            int met[4], nm = 0;
            float key[4];
            for (int k = 0; k < ns; ++k) {
                const std::array<uint32_t, 3>& q = slot[k] < 0 ? bvh.tq[-(slot[k] + 1)] : bvh.cq[slot[k]];
                float tl[3], th[3];
                for (int a = 0; a < 3; ++a) {
                    const float x = std::fmaf((float)(q[a] & 0xFFFFu), cr.qi[a], cr.qc[a]), y = std::fmaf((float)(q[a] >> 16), cr.qi[a], cr.qc[a]);
                    tl[a] = fmin_std(x, y), th[a] = fmax_std(x, y);
                }
                const float tin = fmax_std(tl[0], fmax_std(tl[1], tl[2]));
                const float tout = fmin_std(th[0], fmin_std(th[1], th[2]));
                if (tin > tout || tout < 0.0f || tin > 1.0f)
                    continue;
                const float a = fmax_std(tin, 0.0f), b = fmin_std(tout, 1.0f);
                float ky = (float)nm; /* mode 0: keep the slot order */
                if (mode == 1)
                    ky = a;
                else if (mode == 2)
                    ky = -b;
                else if (mode == 3)
                    ky = t_guide < a ? a - t_guide : (t_guide > b ? t_guide - b : 0.0f);
                else if (mode == 4)
                    ky = -(b - a);
                else if (mode == 5)
                    ky = std::fabs(0.5f * (a + b) - t_guide);
                met[nm] = slot[k], key[nm] = ky, nm += 1;
            }
            /* stable insertion sort by key */
            for (int i = 1; i < nm; ++i)
                for (int j = i; j > 0 && key[j] < key[j - 1]; --j) {
                    std::swap(key[j], key[j - 1]);
                    std::swap(met[j], met[j - 1]);
                }
            for (int k = nm - 1; k > 0; --k)
                stack.push_back(met[k]);
            if (nm > 0) {
                cur = met[0];
                descended = true;
            }
        }
        if (descended)
            continue;
        if (stack.empty())
            return false;
        cur = stack.back();
        stack.pop_back();
    }
}


/* ---- a surface-area-heuristic tree over the same triangles, for shadow queries only (any accepted hit below 0.99
 * answers; order and shape of the tree do not matter to the answer when no hit in [0.99, 1] is possible) */
struct SNode {
    Box box;
    int left, right; /* inner: children; leaf: left = -1, right = triangle position in order[] ... */
    int first, count;
};
struct SahTree {
    std::vector<SNode> nodes;
    std::vector<int> tri; /* positions in bvh.order */
};
static inline float half_area(const Box& b)
{
    const float dx = b.hi.x - b.lo.x, dy = b.hi.y - b.lo.y, dz = b.hi.z - b.lo.z;
    return dx * dy + dy * dz + dz * dx;
}
static inline void grow(Box& b, const Box& o)
{
    b.lo = { fmin_std(b.lo.x, o.lo.x), fmin_std(b.lo.y, o.lo.y), fmin_std(b.lo.z, o.lo.z) };
    b.hi = { fmax_std(b.hi.x, o.hi.x), fmax_std(b.hi.y, o.hi.y), fmax_std(b.hi.z, o.hi.z) };
}
static int sah_build(SahTree& t, const std::vector<Box>& tb, const std::vector<V3>& tc, int first, int count, int leaf_max)
{
    SNode n;
    n.box = { { FLT_MAX, FLT_MAX, FLT_MAX }, { -FLT_MAX, -FLT_MAX, -FLT_MAX } };
    Box cb = n.box;
    for (int i = first; i < first + count; ++i) {
        grow(n.box, tb[t.tri[i]]);
        const V3 c = tc[t.tri[i]];
        grow(cb, Box { c, c });
    }
    n.left = n.right = -1, n.first = first, n.count = count;
    const int me = (int)t.nodes.size();
    t.nodes.push_back(n);
    if (count <= leaf_max)
        return me;
    const int NB = 16;
    float best = FLT_MAX;
    int best_axis = -1, best_bin = -1;
    for (int a = 0; a < 3; ++a) {
        const float lo = axis_of(cb.lo, a), hi = axis_of(cb.hi, a);
        if (!(hi > lo))
            continue;
        Box bb[NB];
        int bc[NB];
        for (int k = 0; k < NB; ++k)
            bb[k] = { { FLT_MAX, FLT_MAX, FLT_MAX }, { -FLT_MAX, -FLT_MAX, -FLT_MAX } }, bc[k] = 0;
        const float sc = NB / (hi - lo);
        for (int i = first; i < first + count; ++i) {
            int k = (int)((axis_of(tc[t.tri[i]], a) - lo) * sc);
            k = k < 0 ? 0 : (k >= NB ? NB - 1 : k);
            grow(bb[k], tb[t.tri[i]]);
            bc[k] += 1;
        }
        float ra[NB];
        int rc[NB];
        Box acc = { { FLT_MAX, FLT_MAX, FLT_MAX }, { -FLT_MAX, -FLT_MAX, -FLT_MAX } };
        int cnt = 0;
        for (int k = NB - 1; k > 0; --k) {
            if (bc[k]) grow(acc, bb[k]);
            cnt += bc[k];
            ra[k] = cnt ? half_area(acc) : 0.0f, rc[k] = cnt;
        }
        acc = { { FLT_MAX, FLT_MAX, FLT_MAX }, { -FLT_MAX, -FLT_MAX, -FLT_MAX } };
        cnt = 0;
        for (int k = 0; k < NB - 1; ++k) {
            if (bc[k]) grow(acc, bb[k]);
            cnt += bc[k];
            if (cnt == 0 || rc[k + 1] == 0)
                continue;
            const float cost = half_area(acc) * cnt + ra[k + 1] * rc[k + 1];
            if (cost < best)
                best = cost, best_axis = a, best_bin = k;
        }
    }
    int mid;
    if (best_axis < 0) {
        mid = first + count / 2;
    } else {
        const float lo = axis_of(cb.lo, best_axis), hi = axis_of(cb.hi, best_axis);
        const float sc = NB / (hi - lo);
        auto it = std::partition(t.tri.begin() + first, t.tri.begin() + first + count, [&](int ti) {
            int k = (int)((axis_of(tc[ti], best_axis) - lo) * sc);
            k = k < 0 ? 0 : (k >= NB ? NB - 1 : k);
            return k <= best_bin;
        });
        mid = (int)(it - t.tri.begin());
        if (mid == first || mid == first + count)
            mid = first + count / 2;
    }
    const int l = sah_build(t, tb, tc, first, mid - first, leaf_max);
    const int r = sah_build(t, tb, tc, mid, first + count - mid, leaf_max);
    t.nodes[me].left = l, t.nodes[me].right = r;
    return me;
}
/* any-hit walk of the binary SAH tree on exact boxes; mode 0: left child first, 1: nearer child first, 2: farther first */
static bool sah_walk(Ctx& c, const SahTree& t, const Ray& ray, const HitInfo& like, int mode, double* visits, double* tris, int* maxsp)
{
    const orc_bvh& bvh = c.bvh;
    int stack[128], sp = 0;
    stack[sp++] = 0;
    float val;
    { Ray r0 = ray; if (!slab(t.nodes[0].box, r0, val)) return false; }
    while (sp > 0) {
        const SNode& n = t.nodes[stack[--sp]];
        if (n.left < 0) {
            Ray r = ray;
            HitInfo h = like;
            bool hit = false;
            for (int k = 0; k < n.count; ++k) {
                TriRef tr = bvh.order[t.tri[n.first + k]];
                *tris += 1;
                if (leaf_triangle(c, r, h, tr.mesh, tr.tri))
                    hit = true;
            }
            if (hit && r.t < 0.99f)
                return true;
            continue;
        }
        *visits += 1;
        float vl, vr;
        Ray r1 = ray;
        const bool pl = slab(t.nodes[n.left].box, r1, vl), pr = slab(t.nodes[n.right].box, r1, vr);
        int a = n.left, b = n.right;
        bool pa = pl, pb = pr;
        if ((mode == 1 && vr < vl) || (mode == 2 && vr > vl))
            std::swap(a, b), std::swap(pa, pb);
        if (pb) stack[sp++] = b;
        if (pa) stack[sp++] = a;
        if (sp > *maxsp) *maxsp = sp;
    }
    return false;
}

} // namespace

extern "C" {

/* out[0..]: see sim_beam.py */
void sim_beam(const orc_bvh* bvh, const orc_camera* cam, uint32_t features, int light_index, int step, double* out)
{
    const orc_scene& s = bvh->s;
    const float* L = s.lights + (size_t)ORC_LIGHT_STRIDE * light_index;
    const Box& root = bvh->nodes.back().box;
    const float grid = fmax_std(fmax_std(root.hi.x - root.lo.x, root.hi.y - root.lo.y), root.hi.z - root.lo.z) / 65533.0f;
    std::vector<int> parent(bvh->nodes.size(), -1), leaf_of(bvh->order.size(), -1);
    for (size_t i = 0; i < bvh->nodes.size(); ++i) {
        const Node& n = bvh->nodes[i];
        if (n.leaf)
            for (int k = 0; k < n.b; ++k)
                leaf_of[n.a + k] = (int)i;
        else
            parent[n.a] = parent[n.b] = (int)i;
    }
    double hits = 0, clean = 0, allvis = 0, allocc = 0, mixed = 0;
    double cost_total = 0, cost_clean = 0, cost_allvis = 0, cost_allocc = 0, cost_mixed = 0;
    double beam_visits = 0, beam_tris = 0, beam_visits_clean = 0, beam_tris_clean = 0, wrong = 0;
    double fail_own = 0, fail_far = 0, fail_overflow = 0;
    double fallbacks = 0;
    SahTree sah[2];
    {
        std::vector<Box> tb(bvh->order.size());
        std::vector<V3> tc(bvh->order.size());
        for (size_t i = 0; i < bvh->order.size(); ++i) {
            tb[i] = get_box(s, &bvh->order[i], &bvh->order[i] + 1);
            tc[i] = (tb[i].lo + tb[i].hi) * 0.5f;
        }
        for (int v = 0; v < 2; ++v) {
            sah[v].tri.resize(bvh->order.size());
            for (size_t i = 0; i < bvh->order.size(); ++i)
                sah[v].tri[i] = (int)i;
            sah_build(sah[v], tb, tc, 0, (int)bvh->order.size(), v == 0 ? 1 : 2);
        }
    }
    double sah_visits[6] = {0,0,0,0,0,0}, sah_tris[6] = {0,0,0,0,0,0}, sah_occ[6] = {0,0,0,0,0,0};
    int sah_maxsp = 0;
    double order_work[6] = { 0, 0, 0, 0, 0, 0 }, order_occ[6] = { 0, 0, 0, 0, 0, 0 };
    double top_tests[5] = {0,0,0,0,0}, top_roots[5] = {0,0,0,0,0}, top_visits[5] = {0,0,0,0,0}, top_tris[5] = {0,0,0,0,0}, top_occ[5] = {0,0,0,0,0};
    double scout_work[4] = { 0, 0, 0, 0 }, scout_tried[4] = { 0, 0, 0, 0 }, scout_found[4] = { 0, 0, 0, 0 };
    double vis_in_unclean = 0, q_total = 0, zero_q = 0, zero_cost = 0, zero_vis = 0, vis_cost = 0, occ_cost = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : hits, clean, allvis, allocc, mixed, cost_total, cost_clean, cost_allvis, cost_allocc, cost_mixed, beam_visits, beam_tris, beam_visits_clean, beam_tris_clean, wrong, fail_own, fail_far, fail_overflow, vis_in_unclean, q_total, zero_q, zero_cost, zero_vis, vis_cost, occ_cost, scout_work[:4], scout_tried[:4], scout_found[:4], order_work[:6], order_occ[:6], top_tests[:5], top_roots[:5], top_visits[:5], top_tris[:5], top_occ[:5], fallbacks, sah_visits[:6], sah_tris[:6], sah_occ[:6]) reduction(max : sah_maxsp)
    for (int y = 0; y < cam->height; y += step) {
        for (int x = 0; x < cam->width; x += step) {
            Ctx c { *bvh, features, 5, 0u };
            float nx = (float)x / (float)cam->width * 2.0f - 1.0f, ny = (float)y / (float)cam->height * 2.0f - 1.0f;
            Ray ray = camera_ray(*cam, nx, ny);
            HitInfo hit;
            if (!intersect(c, ray, hit))
                continue;
            hits += 1;
            V3 P = ray.o + ray.d * ray.t;
            /* the 100 queries, with their work */
            V3 p0 = v3(L + 1), e01 = v3(L + 4), e02 = v3(L + 7);
            int nvis = 0;
            orc_counters before = c.cnt;
            for (int i = 0; i < 10; ++i)
                for (int j = 0; j < 10; ++j) {
                    float r1 = rand01(0, (uint32_t)(y * cam->width + x), 0, (uint32_t)light_index, (uint32_t)(i * 10 + j), 0);
                    float r2 = rand01(0, (uint32_t)(y * cam->width + x), 0, (uint32_t)light_index, (uint32_t)(i * 10 + j), 1);
                    float u = ((float)i + r1) / 10.0f, v = ((float)j + r2) / 10.0f;
                    V3 pos = (p0 + (u * e01)) + (v * e02);
                    {
                        Ray nr = { P, pos - P, 1.0f };
                        nr.o = nr.o + normalize(nr.d) * 0.001f;
                        bool occ2 = false;
                        Ctx c2 { *bvh, features, 5, 0u };
                        if (!shadow_coarse(c2, nr, hit, occ2))
                            fallbacks += 1;
                    }
                    {
                        Ray nr = { P, pos - P, 1.0f };
                        nr.o = nr.o + normalize(nr.d) * 0.001f;
                        for (int v = 0; v < 2; ++v)
                            for (int m = 0; m < 3; ++m) {
                                double vv = 0, tt = 0;
                                int msp = 0;
                                const bool occ = sah_walk(c, sah[v], nr, hit, m, &vv, &tt, &msp);
                                sah_visits[v * 3 + m] += vv, sah_tris[v * 3 + m] += tt, sah_occ[v * 3 + m] += occ;
                                if (msp > sah_maxsp) sah_maxsp = msp;
                            }
                    }
                    orc_counters b4 = c.cnt;
                    const bool vis = visibility(c, pos, ray, hit) != 0.0f;
                    nvis += vis;
                    V3 col = { 1, 1, 1 };
                    V3 term = compute_shading(pos, col, ray, hit.normal, hit.material.kd, hit.material.ks, hit.material.shininess);
                    const double qc = (double)(c.cnt.inner_visits - b4.inner_visits) + (double)(c.cnt.tri_tests - b4.tri_tests);
                    if (term.x == 0 && term.y == 0 && term.z == 0)
                        zero_q += 1, zero_cost += qc, zero_vis += vis;
                    if (vis) vis_cost += qc; else occ_cost += qc;
                }
            const double cost = (double)(c.cnt.inner_visits - before.inner_visits) + (double)(c.cnt.tri_tests - before.tri_tests);
            cost_total += cost;
            q_total += 100;
            if (nvis == 100)
                allvis += 1, cost_allvis += cost;
            else if (nvis == 0)
                allocc += 1, cost_allocc += cost;
            else
                mixed += 1, cost_mixed += cost;
            /* free traversal orders */
            {
                float tg = 0.5f;
                bool have = false;
                for (int i = 0; i < 10; ++i)
                    for (int j = 0; j < 10; ++j) {
                        float r1 = rand01(0, (uint32_t)(y * cam->width + x), 0, (uint32_t)light_index, (uint32_t)(i * 10 + j), 0);
                        float r2 = rand01(0, (uint32_t)(y * cam->width + x), 0, (uint32_t)light_index, (uint32_t)(i * 10 + j), 1);
                        float u = ((float)i + r1) / 10.0f, v = ((float)j + r2) / 10.0f;
                        V3 pos = (p0 + (u * e01)) + (v * e02);
                        Ray nr = { P, pos - P, 1.0f };
                        nr.o = nr.o + normalize(nr.d) * 0.001f;
                        for (int m = 0; m < 6; ++m) {
                            double vv = 0, tt = 0;
                            float tf = -1;
                            /* the guide: t of the previous occluded sample of this hit (sequential stand-in for a scout) */
                            bool occ = walk_order(c, nr, hit, m, have ? tg : 0.5f, &vv, &tt, &tf);
                            order_work[m] += vv + tt;
                            order_occ[m] += occ;
                            if (m == 5 && occ)
                                tg = tf, have = true;
                        }
                    }
            }
            /* top levels walked once per hit by the bundle: the samples start from the listed sub-trees */
            {
                Beam b = make_beam(P, L);
                for (int D = 2; D <= 4; ++D) { /* wide levels folded into the list */
                    std::vector<int> roots, frontier;
                    frontier.push_back((int)bvh->nodes.size() - 1);
                    double tests = 0;
                    bool usable = b.ok;
                    for (int lev = 0; lev < D && usable; ++lev) {
                        std::vector<int> next;
                        for (int cur : frontier) {
                            const Node& n = bvh->nodes[cur];
                            if (n.leaf) { usable = false; break; }
                            for (int child : { n.b, n.a }) {
                                const Node& cn = bvh->nodes[child];
                                if (cn.leaf) { usable = false; break; }
                                for (int g : { cn.b, cn.a }) {
                                    tests += 1;
                                    if (beam_box(b, bvh->nodes[g].box, 3.0f * grid))
                                        next.push_back(g);
                                }
                            }
                        }
                        frontier.swap(next);
                    }
                    if (!usable) continue;
                    top_tests[D] += tests;
                    top_roots[D] += (double)frontier.size();
                    for (int i = 0; i < 10; ++i)
                        for (int j = 0; j < 10; ++j) {
                            float r1 = rand01(0, (uint32_t)(y * cam->width + x), 0, (uint32_t)light_index, (uint32_t)(i * 10 + j), 0);
                            float r2 = rand01(0, (uint32_t)(y * cam->width + x), 0, (uint32_t)light_index, (uint32_t)(i * 10 + j), 1);
                            float u = ((float)i + r1) / 10.0f, v = ((float)j + r2) / 10.0f;
                            V3 pos = (p0 + (u * e01)) + (v * e02);
                            Ray nr = { P, pos - P, 1.0f };
                            nr.o = nr.o + normalize(nr.d) * 0.001f;
                            double vv = 0, tt = 0;
                            float tf;
                            bool occ = false;
                            for (int root : frontier) {
                                if (bvh->nodes[root].leaf) { usable = false; break; }
                                if (walk_order(c, nr, hit, 0, 0.5f, &vv, &tt, &tf, root)) { occ = true; break; }
                            }
                            top_visits[D] += vv, top_tris[D] += tt, top_occ[D] += occ;
                        }
                }
            }
            /* the bundle */
            Beam b = make_beam(P, L);
            bool ok = b.ok;
            double bv = 0, bt = 0;
            int failed_near = 0;
            if (ok) {
                std::vector<int> stack;
                stack.push_back((int)bvh->nodes.size() - 1);
                while (!stack.empty() && ok) {
                    const int cur = stack.back();
                    stack.pop_back();
                    const Node& n = bvh->nodes[cur];
                    if (n.leaf) {
                        for (int k = 0; k < n.b && ok; ++k) {
                            TriRef t = bvh->order[n.a + k];
                            const uint32_t* tr = tri(s, t.mesh, t.tri);
                            V3 v0 = v3(vert(s, t.mesh, tr[0])), v1 = v3(vert(s, t.mesh, tr[1])), v2 = v3(vert(s, t.mesh, tr[2]));
                            bt += 1;
                            int why;
                            if (beam_triangle(b, v0, v1, v2, &why)) {
                                ok = false;
                                V3 cc = ((v0 + v1) + v2) / 3.0f;
                                failed_near = length(cc - P) < 0.05f;
                            }
                        }
                        continue;
                    }
                    bv += 1; /* one wide visit: up to four grandchildren */
                    int slot[4], ns = 0;
                    for (int child : { n.b, n.a }) {
                        const Node& cn = bvh->nodes[child];
                        if (cn.leaf)
                            slot[ns++] = child;
                        else
                            slot[ns++] = cn.b, slot[ns++] = cn.a;
                    }
                    for (int k = 0; k < ns; ++k)
                        if (beam_box(b, bvh->nodes[slot[k]].box, 3.0f * grid))
                            stack.push_back(slot[k]);
                    if (stack.size() > 24) {
                        ok = false;
                        fail_overflow += 1;
                    }
                }
            }
            beam_visits += bv, beam_tris += bt;
            if (ok) {
                clean += 1;
                cost_clean += cost;
                beam_visits_clean += bv, beam_tris_clean += bt;
                if (nvis != 100)
                    wrong += 1;
            } else {
                vis_in_unclean += nvis;
                if (failed_near)
                    fail_own += 1;
                else
                    fail_far += 1;
            }
        }
    }
    double* o = out;
    *o++ = hits, *o++ = clean, *o++ = allvis, *o++ = allocc, *o++ = mixed;
    *o++ = cost_total, *o++ = cost_clean, *o++ = cost_allvis, *o++ = cost_allocc, *o++ = cost_mixed;
    *o++ = beam_visits, *o++ = beam_tris, *o++ = beam_visits_clean, *o++ = beam_tris_clean, *o++ = wrong;
    *o++ = fail_own, *o++ = fail_far, *o++ = fail_overflow, *o++ = vis_in_unclean, *o++ = q_total; for (int K = 0; K < 4; ++K) *o++ = scout_work[K], *o++ = scout_tried[K], *o++ = scout_found[K];
    for (int m = 0; m < 6; ++m) *o++ = order_work[m], *o++ = order_occ[m];
    for (int D = 2; D <= 4; ++D) *o++ = top_tests[D], *o++ = top_roots[D], *o++ = top_visits[D], *o++ = top_tris[D], *o++ = top_occ[D];
    *o++ = fallbacks;
    for (int k = 0; k < 6; ++k) *o++ = sah_visits[k], *o++ = sah_tris[k], *o++ = sah_occ[k];
    *o++ = (double)sah_maxsp; *o++ = (double)sah[0].nodes.size();
    *o++ = zero_q, *o++ = zero_cost, *o++ = zero_vis, *o++ = vis_cost, *o++ = occ_cost;
}
}
