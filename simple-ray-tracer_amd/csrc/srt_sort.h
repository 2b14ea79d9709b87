/*
 * srt_sort.h -- the reference builder's sorting on the device; textually part of srt_kernels.hip (through srt_build.h).
 *
 * The reference sorts every node's triangles with std::sort on the centroid coordinate (reference
 * src/bounding_volume_hierarchy.cpp:81-85).  std::sort is not stable, and meshes hold many equal coordinates, so the
 * order of equal keys - which decides what lands in which child and in which order a leaf lists its triangles - is
 * whatever libstdc++'s introsort leaves behind.  This file reproduces that order exactly, in parallel, for all nodes
 * of one tree level at a time.
 *
 * libstdc++'s std::sort (bits/stl_algo.h: __sort, __introsort_loop, __unguarded_partition_pivot,
 * __move_median_to_first, __unguarded_partition, __final_insertion_sort) is
 *   1. while a range holds more than 16 elements: swap the median of its second, middle and last element into its
 *      first place, Hoare-partition the rest around that pivot, continue on both parts;
 *   2. one insertion sort over everything.
 * Step 2 is a stable sort, and step 1 leaves pieces of at most 16 elements none of which holds an element smaller
 * than one of an earlier piece, so step 2 amounts to a stable insertion sort of every piece by itself.  Step 1 is
 * sequential as written, but what one partition does can be stated without its loop: with a_1 < a_2 < ... the
 * positions (after the pivot) whose key is not smaller than the pivot and b_1 > b_2 > ... those whose key is not larger,
 * the loop swaps a_k with b_k for every k with a_k < b_k (a monotone condition), and returns
 * min(a_{S+1}, b_S) for S swaps (b_0 = the range's end).  Ranks come from a prefix sum, so all ranges of all nodes are
 * partitioned side by side, one recursion depth per round:
 *   - ranges of more than SRT_SORT_SMALL elements: rounds of kernels over the whole array with a device-wide prefix sum;
 *   - ranges up to that size: one workgroup per tile of positions works on the ranges that start in its tile, in
 *     LDS, round after round, and finishes with the insertion sorts.
 * A range that would exhaust introsort's depth limit (2 * floor(log2 n) partitions; libstdc++ then switches to
 * heapsort) raises a flag and the caller falls back to std::sort itself on the host.
 */
#include <hipcub/hipcub.hpp>

#define SRT_SORT_SMALL 1024u /* largest range the tile kernel takes; also its tile size */
#define SRT_SORT_EXT 2048u   /* a tile's ranges span fewer than 2 * SRT_SORT_SMALL positions */
#define SRT_SORT_THREADS 256u
#define SRT_SORT_PER_THREAD (SRT_SORT_EXT / SRT_SORT_THREADS)

struct SortState {
    float* key;        /* centroid coordinate of the triangle at each position, current axis */
    uint32_t* idx;     /* triangle at each position */
    uint32_t *rs, *re; /* the range each position belongs to: [rs, re) */
    uint32_t* rnd;     /* partitions on the way to that range */
    uint32_t* segn;    /* size of the node's whole range (sets the depth limit) */
    unsigned long long *flags, *prefix; /* stopper flags (left | right << 32) and their exclusive prefix sums, N + 1 entries */
    uint32_t *posL, *posR;              /* posL[f + k]: position of the range's k-th left stopper; posR likewise from the right */
    float* piv;
    uint32_t *cut, *swaps;              /* per range, stored at its first position */
    uint32_t *heads[2];                 /* first positions of the ranges above SRT_SORT_SMALL, this round / next round */
    uint32_t* counters;                 /* [0], [1]: lengths of the two head lists; [2]: depth limit exhausted somewhere */
};

SRT_DI uint32_t floor_log2(uint32_t n) { return 31u - (uint32_t)__clz((int)n); }

/* std::__move_median_to_first(result = f, a = f + 1, b = mid, c = l - 1) on (key, idx) pairs */
template <typename K, typename I>
SRT_DI void median_to_first(K* key, I* idx, uint32_t f, uint32_t l)
{
    const uint32_t a = f + 1, b = f + (l - f) / 2, c = l - 1;
    const float ka = key[a], kb = key[b], kc = key[c];
    uint32_t pick;
    if (ka < kb)
        pick = (kb < kc) ? b : ((ka < kc) ? c : a);
    else
        pick = (ka < kc) ? a : ((kb < kc) ? c : b);
    const K tk = key[f];
    key[f] = key[pick], key[pick] = tk;
    const I ti = idx[f];
    idx[f] = idx[pick], idx[pick] = ti;
}

/* Number of swaps S and the returned cut of std::__unguarded_partition over (f, l), from the stopper position lists. */
template <typename P>
SRT_DI void partition_outcome(const P* posL, const P* posR, uint32_t f, uint32_t l, uint32_t nL, uint32_t nR, uint32_t& S, uint32_t& cut)
{
    uint32_t lo = 0, hi = nL < nR ? nL : nR; /* largest k in [0, hi] with posL[f + k] < posR[f + k] (k = 0 counts as true) */
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if ((uint32_t)posL[f + mid] < (uint32_t)posR[f + mid])
            lo = mid;
        else
            hi = mid - 1;
    }
    S = lo;
    const uint32_t nextLeft = (S + 1 <= nL) ? (uint32_t)posL[f + S + 1] : 0xFFFFFFFFu;
    const uint32_t lastRight = S >= 1 ? (uint32_t)posR[f + S] : l;
    cut = nextLeft < lastRight ? nextLeft : lastRight;
}

/* ---- all positions of one level: the node's range, the key, the first head list */
__global__ __launch_bounds__(SRT_BLOCK) void k_sort_level_init(SortState st, const float* __restrict__ centroid, uint32_t n, uint32_t level)
{
    const uint32_t i = blockIdx.x * SRT_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    uint32_t l = 0, r = n;
    for (uint32_t lv = 0; lv < level && r - l > 1; ++lv) {
        const uint32_t m = (uint32_t)(((unsigned long long)l + r + 1) / 2);
        if (i < m)
            r = m;
        else
            l = m;
    }
    st.rs[i] = l, st.re[i] = r, st.rnd[i] = 0, st.segn[i] = r - l;
    st.key[i] = centroid[(size_t)(level % 3u) * n + st.idx[i]];
    if (i == l && r - l > SRT_SORT_SMALL)
        st.heads[0][atomicAdd(&st.counters[0], 1u)] = l;
}

__global__ __launch_bounds__(SRT_BLOCK) void k_sort_pivot(SortState st, uint32_t which, uint32_t n_heads)
{
    const uint32_t h = blockIdx.x * SRT_BLOCK + threadIdx.x;
    if (h >= n_heads)
        return;
    const uint32_t f = st.heads[which][h], l = st.re[f];
    if (st.rnd[f] >= 2u * floor_log2(st.segn[f]))
        st.counters[2] = 1u;
    median_to_first(st.key, st.idx, f, l);
    st.piv[f] = st.key[f];
}

__global__ __launch_bounds__(SRT_BLOCK) void k_sort_flags(SortState st, uint32_t n)
{
    const uint32_t i = blockIdx.x * SRT_BLOCK + threadIdx.x;
    if (i > n)
        return;
    unsigned long long fl = 0;
    if (i < n) {
        const uint32_t f = st.rs[i], l = st.re[i];
        if (l - f > SRT_SORT_SMALL && i > f) {
            const float p = st.piv[f], k = st.key[i];
            fl = (!(k < p) ? 1ull : 0ull) | (!(p < k) ? (1ull << 32) : 0ull);
        }
    }
    st.flags[i] = fl;
}

__global__ __launch_bounds__(SRT_BLOCK) void k_sort_scatter(SortState st, uint32_t n)
{
    const uint32_t i = blockIdx.x * SRT_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    const unsigned long long fl = st.flags[i];
    if (fl == 0)
        return;
    const uint32_t f = st.rs[i], l = st.re[i];
    const unsigned long long base = st.prefix[f + 1], mine = st.prefix[i] - base, total = st.prefix[l] - base;
    if (fl & 1ull)
        st.posL[f + (uint32_t)(mine & 0xFFFFFFFFull) + 1u] = i;
    if (fl >> 32)
        st.posR[f + ((uint32_t)(total >> 32) - (uint32_t)(mine >> 32))] = i; /* stoppers at or right of i */
}

__global__ __launch_bounds__(SRT_BLOCK) void k_sort_outcome(SortState st, uint32_t which, uint32_t n_heads)
{
    const uint32_t h = blockIdx.x * SRT_BLOCK + threadIdx.x;
    if (h >= n_heads)
        return;
    const uint32_t f = st.heads[which][h], l = st.re[f];
    const unsigned long long total = st.prefix[l] - st.prefix[f + 1];
    uint32_t S, cut;
    partition_outcome(st.posL, st.posR, f, l, (uint32_t)(total & 0xFFFFFFFFull), (uint32_t)(total >> 32), S, cut);
    st.cut[f] = cut, st.swaps[f] = S;
    if (cut - f > SRT_SORT_SMALL)
        st.heads[which ^ 1u][atomicAdd(&st.counters[which ^ 1u], 1u)] = f;
    if (l - cut > SRT_SORT_SMALL)
        st.heads[which ^ 1u][atomicAdd(&st.counters[which ^ 1u], 1u)] = cut;
}

__global__ __launch_bounds__(SRT_BLOCK) void k_sort_apply(SortState st, uint32_t n)
{
    const uint32_t i = blockIdx.x * SRT_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t f = st.rs[i], l = st.re[i];
    if (l - f <= SRT_SORT_SMALL)
        return;
    const uint32_t j = i - f;
    if (j >= 1 && j <= st.swaps[f]) {
        const uint32_t a = st.posL[f + j], b = st.posR[f + j];
        const float ka = st.key[a];
        st.key[a] = st.key[b], st.key[b] = ka;
        const uint32_t ia = st.idx[a];
        st.idx[a] = st.idx[b], st.idx[b] = ia;
    }
    const uint32_t cut = st.cut[f];
    if (i < cut)
        st.re[i] = cut;
    else
        st.rs[i] = cut;
    st.rnd[i] += 1;
}

/* ---- ranges of at most SRT_SORT_SMALL elements: the rest of their introsort in LDS, one workgroup per tile */
__global__ __launch_bounds__(SRT_SORT_THREADS) void k_sort_tiles(SortState st, uint32_t n)
{
    __shared__ float skey[SRT_SORT_EXT];
    __shared__ uint32_t sidx[SRT_SORT_EXT];
    __shared__ uint16_t srs[SRT_SORT_EXT], sre[SRT_SORT_EXT];
    __shared__ uint16_t sposL[SRT_SORT_EXT + 2], sposR[SRT_SORT_EXT + 2];
    __shared__ uint32_t sprefix[SRT_SORT_EXT + 1]; /* inclusive prefix of (left | right << 16), entry i + 1 for position i */
    __shared__ uint8_t sflag[SRT_SORT_EXT]; /* bit 0: left stopper, bit 1: right stopper */
    __shared__ float spiv[SRT_SORT_EXT];
    __shared__ uint16_t scut[SRT_SORT_EXT], sswaps[SRT_SORT_EXT];
    __shared__ uint8_t srnd[SRT_SORT_EXT];
    __shared__ uint32_t swave[SRT_SORT_THREADS / 64];
    __shared__ uint32_t sactive;

    const uint32_t ts = blockIdx.x * SRT_SORT_SMALL, te = ts + SRT_SORT_SMALL < n ? ts + SRT_SORT_SMALL : n;
    /* the ranges that START in this tile: from the end of the range the previous tile's last position lies in ... */
    uint32_t a = ts;
    if (ts > 0) {
        const uint32_t prevEnd = st.re[ts - 1];
        if (prevEnd > ts)
            a = prevEnd;
    }
    if (a >= te)
        return; /* one range covers the tile and began before it */
    const uint32_t b = st.re[te - 1]; /* ... to the end of the range its last position lies in */
    const uint32_t ext = b - a;
    const uint32_t t = threadIdx.x;
    for (uint32_t i = t; i < ext; i += SRT_SORT_THREADS) {
        skey[i] = st.key[a + i], sidx[i] = st.idx[a + i];
        srs[i] = (uint16_t)(st.rs[a + i] - a), sre[i] = (uint16_t)(st.re[a + i] - a);
        srnd[i] = (uint8_t)st.rnd[a + i];
    }
    __syncthreads();
    for (;;) {
        if (t == 0)
            sactive = 0;
        __syncthreads();
        /* pivots */
        for (uint32_t i = t; i < ext; i += SRT_SORT_THREADS) {
            const uint32_t f = srs[i], l = sre[i];
            if (i == f && l - f > 16u) {
                if (srnd[f] >= 2u * floor_log2(st.segn[a + f]))
                    st.counters[2] = 1u;
                median_to_first(skey, sidx, f, l);
                spiv[f] = skey[f];
                sactive = 1;
            }
        }
        __syncthreads();
        if (!sactive)
            break;
        /* stopper flags and their prefix sums (thread t owns SRT_SORT_PER_THREAD consecutive positions) */
        uint32_t local[SRT_SORT_PER_THREAD], sum = 0;
        for (uint32_t k = 0; k < SRT_SORT_PER_THREAD; ++k) {
            const uint32_t i = t * SRT_SORT_PER_THREAD + k;
            uint32_t fl = 0;
            if (i < ext) {
                const uint32_t f = srs[i], l = sre[i];
                if (l - f > 16u && i > f) {
                    const float p = spiv[f], key = skey[i];
                    fl = (!(key < p) ? 1u : 0u) | (!(p < key) ? (1u << 16) : 0u);
                }
                sflag[i] = (uint8_t)((fl & 1u) | (fl >> 15));
            }
            sum += fl;
            local[k] = sum;
        }
        /* exclusive scan of the threads' sums: inside each wave, then over the waves */
        uint32_t incl = sum;
        for (uint32_t off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if ((t & 63u) >= off)
                incl += up;
        }
        if ((t & 63u) == 63u)
            swave[t >> 6] = incl;
        __syncthreads();
        uint32_t before = incl - sum;
        for (uint32_t w = 0; w < (t >> 6); ++w)
            before += swave[w];
        if (t == 0)
            sprefix[0] = 0;
        for (uint32_t k = 0; k < SRT_SORT_PER_THREAD; ++k) {
            const uint32_t i = t * SRT_SORT_PER_THREAD + k;
            if (i < ext)
                sprefix[i + 1] = before + local[k];
        }
        __syncthreads();
        /* stopper position lists */
        for (uint32_t i = t; i < ext; i += SRT_SORT_THREADS) {
            const uint32_t fl = sflag[i];
            if (!fl)
                continue;
            const uint32_t f = srs[i], l = sre[i];
            const uint32_t base = sprefix[f + 1], mine = sprefix[i] - base, total = sprefix[l] - base;
            if (fl & 1u)
                sposL[f + (mine & 0xFFFFu) + 1u] = (uint16_t)i;
            if (fl & 2u)
                sposR[f + ((total >> 16) - (mine >> 16))] = (uint16_t)i;
        }
        __syncthreads();
        for (uint32_t i = t; i < ext; i += SRT_SORT_THREADS) {
            const uint32_t f = srs[i], l = sre[i];
            if (i == f && l - f > 16u) {
                const uint32_t total = sprefix[l] - sprefix[f + 1];
                uint32_t S, cut;
                partition_outcome(sposL, sposR, f, l, total & 0xFFFFu, total >> 16, S, cut);
                scut[f] = (uint16_t)cut, sswaps[f] = (uint16_t)S;
            }
        }
        __syncthreads();
        /* swaps; every position reads its range's outcome before any range bound changes */
        uint32_t newRs[SRT_SORT_PER_THREAD], newRe[SRT_SORT_PER_THREAD];
        for (uint32_t k = 0; k < SRT_SORT_PER_THREAD; ++k) {
            const uint32_t i = t + k * SRT_SORT_THREADS;
            newRs[k] = newRe[k] = 0xFFFFFFFFu;
            if (i >= ext)
                continue;
            const uint32_t f = srs[i], l = sre[i];
            if (l - f <= 16u)
                continue;
            const uint32_t j = i - f;
            if (j >= 1 && j <= sswaps[f]) {
                const uint32_t pa = sposL[f + j], pb = sposR[f + j];
                const float ka = skey[pa];
                skey[pa] = skey[pb], skey[pb] = ka;
                const uint32_t ia = sidx[pa];
                sidx[pa] = sidx[pb], sidx[pb] = ia;
            }
            const uint32_t cut = scut[f];
            newRs[k] = i < cut ? f : cut;
            newRe[k] = i < cut ? cut : l;
        }
        __syncthreads();
        for (uint32_t k = 0; k < SRT_SORT_PER_THREAD; ++k) {
            const uint32_t i = t + k * SRT_SORT_THREADS;
            if (i < ext && newRs[k] != 0xFFFFFFFFu) {
                srs[i] = (uint16_t)newRs[k], sre[i] = (uint16_t)newRe[k];
                srnd[i] = (uint8_t)(srnd[i] + 1u);
            }
        }
        __syncthreads();
    }
    /* __final_insertion_sort: a stable insertion sort of every piece */
    for (uint32_t i = t; i < ext; i += SRT_SORT_THREADS) {
        const uint32_t f = srs[i], l = sre[i];
        if (i != f)
            continue;
        for (uint32_t p = f + 1; p < l; ++p) {
            const float kv = skey[p];
            const uint32_t iv = sidx[p];
            uint32_t q = p;
            while (q > f && kv < skey[q - 1]) {
                skey[q] = skey[q - 1], sidx[q] = sidx[q - 1];
                --q;
            }
            skey[q] = kv, sidx[q] = iv;
        }
    }
    __syncthreads();
    for (uint32_t i = t; i < ext; i += SRT_SORT_THREADS)
        st.idx[a + i] = sidx[i];
}

/* centroid = ((p1 + p2) + p3) / 3.0f per axis (getMedian, reference :37-45); plane k of the table holds axis k */
__global__ __launch_bounds__(SRT_BLOCK) void k_sort_centroids(const float* __restrict__ verts8, const uint32_t* __restrict__ tris3, uint32_t n, float* centroid,
                                                                uint32_t* idx, uint32_t* bad)
{
    const uint32_t t = blockIdx.x * SRT_BLOCK + threadIdx.x;
    if (t >= n)
        return;
    const float *a = verts8 + 8ull * tris3[3ull * t], *b = verts8 + 8ull * tris3[3ull * t + 1], *c = verts8 + 8ull * tris3[3ull * t + 2];
    for (uint32_t k = 0; k < 3; ++k) {
        const float v = ((a[k] + b[k]) + c[k]) / 3.0f;
        centroid[(size_t)k * n + t] = v;
        if (!(v == v))
            *bad = 1u; /* a NaN key: std::sort's behaviour is undefined; left to the host path */
    }
    idx[t] = t;
}

namespace srt_build {

/* The triangle order of the median-split tree over n triangles, on the device: *d_order (owned by `temp`) holds the
 * triangle at every position.  SRT_OK with *fell_back = true means the device gave up (depth limit, NaN key). */
static int device_order(const float* d_verts, const uint32_t* d_tris3, uint32_t n, uint32_t n_levels_to_sort, std::vector<void*>& temp,
                        uint32_t** d_order, bool* fell_back)
{
    *fell_back = false;
    auto alloc = [&](size_t bytes, void** p) -> int {
        *p = nullptr;
        HIPCHK(hipMalloc(p, bytes ? bytes : 4));
        temp.push_back(*p);
        return SRT_OK;
    };
    SortState st {};
    float* centroid = nullptr;
    const size_t N = n, n_heads_cap = N / SRT_SORT_SMALL + 16;
    void* scan_temp = nullptr;
    size_t scan_bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (int)(N + 1), (hipStream_t)0);
    int rc = alloc(3 * N * 4, (void**)&centroid);
    if (rc == SRT_OK) rc = alloc(N * 4, (void**)&st.key);
    if (rc == SRT_OK) rc = alloc(N * 4, (void**)&st.idx);
    if (rc == SRT_OK) rc = alloc(N * 4, (void**)&st.rs);
    if (rc == SRT_OK) rc = alloc(N * 4, (void**)&st.re);
    if (rc == SRT_OK) rc = alloc(N * 4, (void**)&st.rnd);
    if (rc == SRT_OK) rc = alloc(N * 4, (void**)&st.segn);
    if (rc == SRT_OK) rc = alloc((N + 1) * 8, (void**)&st.flags);
    if (rc == SRT_OK) rc = alloc((N + 1) * 8, (void**)&st.prefix);
    if (rc == SRT_OK) rc = alloc((N + 2) * 4, (void**)&st.posL);
    if (rc == SRT_OK) rc = alloc((N + 2) * 4, (void**)&st.posR);
    if (rc == SRT_OK) rc = alloc(N * 4, (void**)&st.piv);
    if (rc == SRT_OK) rc = alloc(N * 4, (void**)&st.cut);
    if (rc == SRT_OK) rc = alloc(N * 4, (void**)&st.swaps);
    if (rc == SRT_OK) rc = alloc(n_heads_cap * 4, (void**)&st.heads[0]);
    if (rc == SRT_OK) rc = alloc(n_heads_cap * 4, (void**)&st.heads[1]);
    if (rc == SRT_OK) rc = alloc(16, (void**)&st.counters);
    if (rc == SRT_OK) rc = alloc(scan_bytes, &scan_temp);
    if (rc != SRT_OK)
        return rc;
    const dim3 block(SRT_BLOCK), grid_n((n + SRT_BLOCK - 1) / SRT_BLOCK), grid_n1((n + 1 + SRT_BLOCK - 1) / SRT_BLOCK);
    HIPCHK(hipMemsetAsync(st.counters, 0, 16, 0));
    hipLaunchKernelGGL(k_sort_centroids, grid_n, block, 0, 0, d_verts, d_tris3, n, centroid, st.idx, st.counters + 3);
    for (uint32_t level = 0; level < n_levels_to_sort; ++level) {
        HIPCHK(hipMemsetAsync(st.counters, 0, 8, 0));
        hipLaunchKernelGGL(k_sort_level_init, grid_n, block, 0, 0, st, (const float*)centroid, n, level);
        uint32_t which = 0, n_heads = 0;
        HIPCHK(hipMemcpy(&n_heads, st.counters, 4, hipMemcpyDeviceToHost));
        while (n_heads) {
            if (n_heads > n_heads_cap)
                return fail(SRT_ERR_INVALID, "srt_scene_build: sort bookkeeping overflow");
            const dim3 grid_h((n_heads + SRT_BLOCK - 1) / SRT_BLOCK);
            HIPCHK(hipMemsetAsync(st.counters + (which ^ 1u), 0, 4, 0));
            hipLaunchKernelGGL(k_sort_pivot, grid_h, block, 0, 0, st, which, n_heads);
            hipLaunchKernelGGL(k_sort_flags, grid_n1, block, 0, 0, st, n);
            if (hipcub::DeviceScan::ExclusiveSum(scan_temp, scan_bytes, st.flags, st.prefix, (int)(N + 1), (hipStream_t)0) != hipSuccess)
                return fail(SRT_ERR_HIP, "srt_scene_build: prefix sum failed");
            hipLaunchKernelGGL(k_sort_scatter, grid_n, block, 0, 0, st, n);
            hipLaunchKernelGGL(k_sort_outcome, grid_h, block, 0, 0, st, which, n_heads);
            hipLaunchKernelGGL(k_sort_apply, grid_n, block, 0, 0, st, n);
            which ^= 1u;
            HIPCHK(hipMemcpy(&n_heads, st.counters + which, 4, hipMemcpyDeviceToHost));
        }
        hipLaunchKernelGGL(k_sort_tiles, dim3((n + SRT_SORT_SMALL - 1) / SRT_SORT_SMALL), dim3(SRT_SORT_THREADS), 0, 0, st, n);
    }
    uint32_t flags[4] = {};
    HIPCHK(hipMemcpy(flags, st.counters, 16, hipMemcpyDeviceToHost));
    HIPCHK(hipGetLastError());
    *fell_back = flags[2] != 0 || flags[3] != 0;
    *d_order = st.idx;
    return SRT_OK;
}

} // namespace srt_build
