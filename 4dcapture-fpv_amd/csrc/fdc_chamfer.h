// Body -> scene nearest neighbour (squared L2) over the whole scene.
// Replaces ChamferDistancePytorch's NmDistanceKernel as called at
// /root/reference/global_optimization.py:292-294 (only dist1 = query -> scene is consumed).
//
// Layout decisions:
//   * the scene is stored ONCE as float4 {x, y, z, bits(index)} and shared by every frame; all
//     frames' contact vertices form one flat query array, so a scene tile staged in LDS is
//     reused by every query of a workgroup, whatever frame it belongs to (the reference re-reads a
//     per-frame copy of the scene);
//   * the scene is cut into `nsplit` contiguous ranges; the blockIdx -> (query block, split) map is
//     XCD-aware (nn_block_map): every XCD gets an equal share of every split's work (speed only;
//     any placement is correct);
//   * every kernel reports (d, index) with d from nn_exact_d2 and the LOWEST index among exact
//     ties -- what the ascending strict-`<` scan of the reference kernel gives -- independent of the
//     order in which points are visited; per-split minima are merged by the same (d, index) order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <atomic>

#include "fdc_math.h"

namespace fdc {

constexpr int NN_TILE = 1024;          // scene points per LDS tile of the plain scan
#ifndef FDC_MF_CH
#define FDC_MF_CH 512
#endif
constexpr int MF_CH = FDC_MF_CH;       // scene points per chunk of the MFMA scans = one k-d cell = culling granularity

// scene as the NN kernels see it
struct NNTarget {
    const float4* pts;        // [n] {x, y, z, bits(original index)}
    int n;
    const float4* bounds;     // optional [2 * ceil(n / MF_CH)] axis-aligned box {lo xyz, -}, {hi xyz, -} of each chunk (pts spatially sorted)
    const float4* sbounds;    // with bounds: [2 * ceil(nchunk / 16)] boxes of 16 consecutive chunks (two-level survivor test)
    const float4* qbounds;    // with frags: [2 * 4 * nchunk] boxes of the quarter chunks (the unit the streaming kernel lists and scans)
    const int* inv_perm;      // optional [n] original index -> position in pts (null: identity)
    // optional, static per scene: MFMA A fragments precomputed per chunk RELATIVE TO THE CHUNK'S OWN CENTRE
    // ([chunk][tile 0..15][k-half][point 0..31] uint4) + the centres {cx, cy, cz, radius}: lets a wave
    // stream a chunk straight from global memory into registers (no LDS staging, no workgroup barrier)
    const uint4* frags;
    const float4* centers;
};

// The one definition of the squared distance every kernel reports (direct-difference form as
// the reference CUDA kernel computes it; fixed operation order so all kernels agree bitwise).
__device__ __forceinline__ float nn_exact_d2(float qx, float qy, float qz, float px, float py, float pz) {
    float dx = qx - px, dy = qy - py, dz = qz - pz;
    return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}
__device__ __forceinline__ bool nn_better(float d, int i, float bd, int bi) {
    return d < bd || (d == bd && i < bi);
}
__host__ __device__ __forceinline__ int nn_split_len(int nt, int nsplit) {
    int per = (nt + nsplit - 1) / nsplit;
    return (per + MF_CH - 1) / MF_CH * MF_CH;             // chunk-aligned so bounds[] indexes uniformly
}

// Workgroup -> (query block, scene split).  Blocks are dealt to the 8 XCDs round-robin (b % 8), and
// with chunk culling nearly all the work of a query block sits in the one or two splits that hold
// its neighbourhood, so the split must NOT be a function of b % 8 (half the XCDs would idle --
// measured 2x).  Within each XCD consecutive slots walk the splits of one query block.
__device__ __forceinline__ void nn_block_map(int b, int nsplit, int* qb, int* split) {
    const int xcd = b & 7, slot = b >> 3;
    *split = slot % nsplit;
    *qb = (slot / nsplit) * 8 + xcd;
}
static inline int nn_grid_blocks(int qblocks, int nsplit) { return (qblocks + 7) / 8 * 8 * nsplit; }

template <int QPT>
__global__ __launch_bounds__(256) void nn_direct_kernel(const float* __restrict__ q, int nq, NNTarget T, int nsplit,
                                                        float* __restrict__ pd, int* __restrict__ pi) {
    __shared__ float4 tile[NN_TILE];
    const int tid = threadIdx.x;
    int split, qb;
    nn_block_map(blockIdx.x, nsplit, &qb, &split);
    if (qb * (256 * QPT) >= nq) return;
    const int per = nn_split_len(T.n, nsplit);
    const int t_begin = min(T.n, split * per);
    const int t_end = min(T.n, t_begin + per);
    float qx[QPT], qy[QPT], qz[QPT], best[QPT];
    int bi[QPT];
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        int qi = qb * (256 * QPT) + u * 256 + tid;
        bool ok = qi < nq;
        qx[u] = ok ? q[3 * (size_t)qi] : 0.f;
        qy[u] = ok ? q[3 * (size_t)qi + 1] : 0.f;
        qz[u] = ok ? q[3 * (size_t)qi + 2] : 0.f;
        best[u] = INFINITY;
        bi[u] = -1;
    }
    for (int base = t_begin; base < t_end; base += NN_TILE) {
        const int cnt = min(NN_TILE, t_end - base);
        for (int j = tid; j < cnt; j += 256) tile[j] = T.pts[base + j];
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < cnt; ++j) {
            const float4 p = tile[j];
            const int gi = __float_as_int(p.w);
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                float d = nn_exact_d2(qx[u], qy[u], qz[u], p.x, p.y, p.z);
                if (nn_better(d, gi, best[u], bi[u])) { best[u] = d; bi[u] = gi; }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        int qi = qb * (256 * QPT) + u * 256 + tid;
        if (qi < nq) {
            pd[(size_t)split * nq + qi] = best[u];
            pi[(size_t)split * nq + qi] = bi[u];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA-filtered exact nearest neighbour.
//
// Every (query, scene point) pair that is visited is scored on the matrix cores: one
// v_mfma_f32_32x32x16_bf16 evaluates 32 scene points x 32 queries of
//     s_ij = |y'_j|^2 - 2 x'_i . y'_j        (x' = x - c, y' = y - c, c = workgroup's query centroid)
// with both coordinate vectors split into bf16 hi+lo parts (16 significant bits each) and
// |y'|^2 into three parts, laid out along K = 16:
//     A (scene) : [yh_x yh_x yl_x yl_x | yh_y yh_y yl_y yl_y | yh_z yh_z yl_z yl_z | nh nm nl 0]
//     B (query) : [-2xh_x -2xl_x -2xh_x -2xl_x | ... y ... | ... z ... | 1 1 1 0]
// The score only FILTERS: |s_ij + |x'_i|^2 - d_ij| <= eps_i (bound below), so a pair is skipped
// only when s_ij >= thr_i = best_i - |x'_i|^2 + eps_i proves d_ij > best_i.  Every surviving pair
// is re-evaluated with nn_exact_d2 in fp32 and compared by (d, index), so the result is
// bit-identical to nn_direct_kernel (checked on the GPU in tests/test_gpu_parity.py).
// The VALU's share per MFMA is a 16-way v_min3 tree and one compare.
//
// eps_i = K1 * X * Y + K2 * (X^2 + Y^2),  X = |x'_i|,  Y = X + sqrt(best_i) >= |y'_j| for any j
// that could beat best_i.  Dominant term: each coordinate keeps 16 bits, so the cross term errs
// by <= 4 * 2^-16 * X * Y = 6.1e-5 X Y; fp32 accumulation, the 3-part norm and the fp32
// centring add < 2e-6 (X Y + Y^2).  K1 = 1e-4, K2 = 8e-6 leave >= 50 % margin.
//
// Two exact accelerators sit on top (both only prune; results do not depend on them):
//   seed   : an initial neighbour per query (the optimiser passes last iteration's result) gives a
//            tight bound from the first tile on;
//   bounds : if the scene is spatially sorted and comes with an axis-aligned box per MF_CH-point
//            chunk, a chunk is skipped unless some query i has dist(x_i, box) <= sqrt(best_i) -- no
//            point of a skipped chunk can beat or tie any query's bound.  (Boxes, not spheres: scanned
//            surfaces give flat chunks, and a query 1 m above a floor patch must not pull in every
//            patch within 1 m.)  Without seeds the bound is infinite and the scan is plain brute force.
#ifndef FDC_MF_K1
#define FDC_MF_K1 1e-4f
#endif
#ifndef FDC_MF_K2
#define FDC_MF_K2 8e-6f
#endif
constexpr float MF_K1 = FDC_MF_K1, MF_K2 = FDC_MF_K2;
constexpr int MF_MAXCHUNK = 2048;      // chunks per split the survivor list can hold (host keeps splits below it)

#ifdef FDC_NN_TIMELINE
__device__ unsigned long long g_nn_timeline[16384 * 8];            // instrumentation build only: per workgroup {start, end, xcc, after set-up, after list, after filter, after main loop, work items}
#endif
#ifdef FDC_NN_STATS
// instrumentation build only (never shipped): [0] MFMA results reduced, [1] results that entered
// the exact path (wave level), [2] rows re-evaluated exactly (lane level), [3] chunks staged
__device__ unsigned long long g_nn_hist[96];    // waves by log2(work items)
__device__ unsigned long long g_nn_stats[8];   // [4] waves on a kept list, [5] waves that built a list, [6] raw list items, [7] items after the filter
#define FDC_STAT(i, v) st_cnt[i] += (v)
#else
#define FDC_STAT(i, v)
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned f2bf(float f) {
    __bf16 h = (__bf16)f;                                  // v_cvt_pk_bf16_f32, round-to-nearest-even
    return (unsigned)__builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf2f(unsigned h) { return __uint_as_float(h << 16); }
// hi/lo bf16 split of v; returns (hi | hi<<16) in .x and (lo | lo<<16) in .y
__device__ __forceinline__ uint2 bf_split_dup(float v) {
    unsigned h = f2bf(v);
    unsigned l = f2bf(v - bf2f(h));
    return make_uint2(h | (h << 16), l | (l << 16));
}
__device__ __forceinline__ float mf_thr(float best, float X, float X2) {
    float Y = X + sqrtf(best);
    return best - X2 + (MF_K1 * X * Y + MF_K2 * (X2 + Y * Y));
}

// squared distance from a point to an axis-aligned box
__device__ __forceinline__ float box_d2(float4 lo, float4 hi, float x, float y, float z) {
    // |x - clamp(x, lo, hi)| per axis: v_med3_f32 + v_sub_f32 (the same magnitudes as max(lo - x, x - hi, 0), one instruction
    // less per axis; an empty box is (+inf, +inf): infinitely far)
    const float dx = x - __builtin_amdgcn_fmed3f(x, lo.x, hi.x), dy = y - __builtin_amdgcn_fmed3f(y, lo.y, hi.y);
    const float dz = z - __builtin_amdgcn_fmed3f(z, lo.z, hi.z);
    return dx * dx + dy * dy + dz * dz;
}

template <int NQ>
__global__ __launch_bounds__(256) void nn_mfma_kernel(const float* __restrict__ q, int nq, NNTarget T, int nsplit,
                                                      const int* __restrict__ seed, float* __restrict__ pd,
                                                      int* __restrict__ pi) {
    __shared__ uint4 sA[2][MF_CH / 32][2][32];     // [buffer][tile][k-half][point] bf16 x 8
    __shared__ float4 sP[2][MF_CH];                // fp32 coordinates (+ index) for the exact re-evaluation
    __shared__ float sred[4][4];
    __shared__ unsigned short slist[MF_MAXCHUNK];   // surviving chunks of this split, ascending
    __shared__ int swcnt[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
    int split, qb;
    nn_block_map(blockIdx.x, nsplit, &qb, &split);
    if (qb * (128 * NQ) >= nq) return;
    const int per = nn_split_len(T.n, nsplit);
    const int t_begin = min(T.n, split * per);
    const int t_end = min(T.n, t_begin + per);

    float qx[NQ], qy[NQ], qz[NQ], own_d[NQ], thr[NQ], X[NQ], X2[NQ];
    int own_i[NQ], qidx[NQ];
    float sx = 0.f, sy = 0.f, sz = 0.f, sc = 0.f;
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        qidx[n] = qb * (128 * NQ) + (wave * NQ + n) * 32 + col;
        bool ok = qidx[n] < nq;
        qx[n] = ok ? q[3 * (size_t)qidx[n]] : 0.f;
        qy[n] = ok ? q[3 * (size_t)qidx[n] + 1] : 0.f;
        qz[n] = ok ? q[3 * (size_t)qidx[n] + 2] : 0.f;
        if (ok && half == 0) { sx += qx[n]; sy += qy[n]; sz += qz[n]; sc += 1.f; }
    }
    // workgroup centroid of the queries
    {
        float v[4] = {sx, sy, sz, sc};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_xor(v[i], off, 64);
            if (lane == 0) sred[wave][i] = v[i];
        }
    }
    __syncthreads();
    const float cnt = fmaxf(sred[0][3] + sred[1][3] + sred[2][3] + sred[3][3], 1.f);
    const float cx = (sred[0][0] + sred[1][0] + sred[2][0] + sred[3][0]) / cnt;
    const float cy = (sred[0][1] + sred[1][1] + sred[2][1] + sred[3][1]) / cnt;
    const float cz = (sred[0][2] + sred[1][2] + sred[2][2] + sred[3][2]) / cnt;
    __syncthreads();                                       // sred is reused below

    bf16x8 bfrag[NQ];
    float reach = 0.f;                                     // max_i (X_i + sqrt(best_i)) over this lane's queries
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        float xx = qx[n] - cx, xy = qy[n] - cy, xz = qz[n] - cz;
        X2[n] = __fmaf_rn(xz, xz, __fmaf_rn(xy, xy, xx * xx));
        X[n] = sqrtf(X2[n]);
        // -2 * (bf16 value) is exact in bf16; pairs are (hi, lo)
        unsigned hx = f2bf(xx), hy = f2bf(xy), hz = f2bf(xz);
        unsigned lx = f2bf(xx - bf2f(hx)), ly = f2bf(xy - bf2f(hy)), lz = f2bf(xz - bf2f(hz));
        unsigned px = f2bf(-2.f * bf2f(hx)) | (f2bf(-2.f * bf2f(lx)) << 16);
        unsigned py = f2bf(-2.f * bf2f(hy)) | (f2bf(-2.f * bf2f(ly)) << 16);
        unsigned pz = f2bf(-2.f * bf2f(hz)) | (f2bf(-2.f * bf2f(lz)) << 16);
        const unsigned one = 0x3F80u;                       // bf16 1.0
        uint4 u = half == 0 ? make_uint4(px, px, py, py) : make_uint4(pz, pz, one | (one << 16), one);
        bfrag[n] = __builtin_bit_cast(bf16x8, u);
        own_d[n] = INFINITY;
        own_i[n] = -1;
        thr[n] = (qidx[n] < nq) ? INFINITY : -INFINITY;     // padding queries are never flagged
        if (seed != nullptr && qidx[n] < nq) {
            const int sj = seed[qidx[n]];
            if (sj >= 0 && sj < T.n) {
                const float4 p = T.pts[T.inv_perm ? T.inv_perm[sj] : sj];
                own_d[n] = nn_exact_d2(qx[n], qy[n], qz[n], p.x, p.y, p.z);
                own_i[n] = sj;
                thr[n] = mf_thr(own_d[n], X[n], X2[n]);
            }
        }
        if (qidx[n] < nq) reach = fmaxf(reach, X[n] + sqrtf(own_d[n]));
    }
    // workgroup reach for chunk culling (infinite unless every query is seeded)
    float Rw = INFINITY;
    if (T.bounds != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) reach = fmaxf(reach, __shfl_xor(reach, off, 64));
        if (lane == 0) sred[wave][0] = reach;
        __syncthreads();
        Rw = fmaxf(fmaxf(sred[0][0], sred[1][0]), fmaxf(sred[2][0], sred[3][0]));
        Rw = Rw * 1.00001f + 1e-6f;                        // rounding of the bound test itself
    }
    const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#ifdef FDC_NN_STATS
    unsigned st_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    // Survivor list.  A chunk (axis-aligned box) matters to query i only if dist(x_i, box) <= sqrt(best_i).  Pass 1 (one chunk per thread, parallel loads -- a serial scan
    // would pay a dependent global-load latency per chunk) keeps the chunks within the workgroup's
    // reach Rw of the centroid; pass 2 tests those against every query individually (much tighter for
    // queries far from the centroid or with a distant neighbour); survivors are compacted in ascending
    // order with ballots.
    const int nchunk = (t_end - t_begin + MF_CH - 1) / MF_CH;
    const bool cull = T.bounds != nullptr && Rw < INFINITY && nchunk <= MF_MAXCHUNK;
    int nsurv = nchunk;
    if (cull) {
        float sb[NQ];                                         // best_i with the rounding slack of the box test
#pragma unroll
        for (int n = 0; n < NQ; ++n) sb[n] = (qidx[n] < nq) ? own_d[n] * 1.00002f + 1e-9f : -INFINITY;
        const float Rw2 = Rw * Rw;
        for (int c0 = 0; c0 < nchunk; c0 += 256) {            // pass 1: centroid reach
            const int ci = c0 + tid;
            if (ci < nchunk) {
                const float4 lo = T.bounds[2 * (t_begin / MF_CH + ci)], hi = T.bounds[2 * (t_begin / MF_CH + ci) + 1];
                slist[ci] = (box_d2(lo, hi, cx, cy, cz) > Rw2) ? 0 : 1;
            }
        }
        __syncthreads();
        for (int ci = half; ci < nchunk; ci += 2) {           // pass 2: per query (each half-wave takes every other chunk)
            if (slist[ci] == 0) continue;                     // workgroup-uniform
            const float4 lo = T.bounds[2 * (t_begin / MF_CH + ci)], hi = T.bounds[2 * (t_begin / MF_CH + ci) + 1];
            bool hit = false;
#pragma unroll
            for (int n = 0; n < NQ; ++n) hit |= box_d2(lo, hi, qx[n], qy[n], qz[n]) <= sb[n];
            if (hit) slist[ci] = 2;                           // same-value stores from several lanes: benign
        }
        __syncthreads();
        nsurv = 0;
        for (int c0 = 0; c0 < nchunk; c0 += 256) {            // ordered compaction (in place: writes trail reads)
            const int ci = c0 + tid;
            const bool keep = ci < nchunk && slist[ci] == 2;
            const unsigned long long mask = __ballot(keep);
            if (lane == 0) swcnt[wave] = __popcll(mask);
            __syncthreads();
            int off = nsurv;
            for (int w = 0; w < wave; ++w) off += swcnt[w];
            if (keep) slist[off + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)ci;
            nsurv += swcnt[0] + swcnt[1] + swcnt[2] + swcnt[3];
            __syncthreads();
        }
    }
    auto chunk_base = [&](int s) -> int { return t_begin + (cull ? (int)slist[s] : s) * MF_CH; };

    // staging (issue-early / write-late): the global loads of the next chunk are issued before the
    // current chunk's MFMA loop, the centre / bf16 split / LDS write happens after it
    float4 pre[MF_CH / 256];
    auto stage_load = [&](int base) {
#pragma unroll
        for (int it = 0; it < MF_CH / 256; ++it) {
            int g = base + tid + it * 256;
            pre[it] = (g < t_end) ? T.pts[g] : make_float4(0.f, 0.f, 0.f, __int_as_float(0x7fffffff));
        }
    };
    auto stage_write = [&](int buf, int base) {
#pragma unroll
        for (int it = 0; it < MF_CH / 256; ++it) {
            int j = tid + it * 256;
            int g = base + j;
            float4 p = pre[it];
            float yx = p.x - cx, yy = p.y - cy, yz = p.z - cz;
            float n2 = __fmaf_rn(yz, yz, __fmaf_rn(yy, yy, yx * yx));
            if (g >= t_end) { yx = yy = yz = 0.f; n2 = 1e30f; }   // padding rows: score 1e30, never below a finite thr
            uint2 sxp = bf_split_dup(yx), syp = bf_split_dup(yy), szp = bf_split_dup(yz);
            unsigned nh = f2bf(n2);
            float r1 = n2 - bf2f(nh);
            unsigned nm = f2bf(r1);
            unsigned nl = f2bf(r1 - bf2f(nm));
            int tile = j >> 5, pt = j & 31;
            sA[buf][tile][0][pt] = make_uint4(sxp.x, sxp.y, syp.x, syp.y);
            sA[buf][tile][1][pt] = make_uint4(szp.x, szp.y, nh | (nm << 16), nl);
            sP[buf][j] = p;
        }
    };

    int buf = 0;
    if (nsurv > 0) { stage_load(chunk_base(0)); stage_write(0, chunk_base(0)); }
    __syncthreads();
    for (int s = 0; s < nsurv; ++s) {
        const int base = chunk_base(s);
        const bool more = s + 1 < nsurv;
        const int nxt = more ? chunk_base(s + 1) : 0;
        if (more) stage_load(nxt);
        FDC_STAT(3, tid == 0);
        const int ntile = (min(MF_CH, t_end - base) + 31) >> 5;
        // software pipeline: the next tile's A fragment is fetched from LDS and two MFMAs are in
        // flight before a result is reduced, so matrix pipe, LDS read and v_min3 tree overlap
        bf16x8 afrag_next = __builtin_bit_cast(bf16x8, sA[buf][0][half][col]);
        for (int tile = 0; tile < ntile; ++tile) {
            const bf16x8 afrag = afrag_next;
            if (tile + 1 < ntile) afrag_next = __builtin_bit_cast(bf16x8, sA[buf][tile + 1][half][col]);
            f32x16_t acc_q[NQ];
            acc_q[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag[0], zero, 0, 0, 0);
            if (NQ > 1) acc_q[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag[1], zero, 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NQ; ++n) {
                const f32x16_t acc = acc_q[n];
                if (n + 2 < NQ) acc_q[n + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag[n + 2], zero, 0, 0, 0);
                // 16-way minimum as a depth-3 tree of v_min3
                const float t0 = fminf(fminf(acc[0], acc[1]), acc[2]), t1 = fminf(fminf(acc[3], acc[4]), acc[5]);
                const float t2 = fminf(fminf(acc[6], acc[7]), acc[8]), t3 = fminf(fminf(acc[9], acc[10]), acc[11]);
                const float t4 = fminf(fminf(acc[12], acc[13]), acc[14]);
                const float m = fminf(fminf(fminf(t0, t1), t2), fminf(fminf(t3, t4), acc[15]));
                FDC_STAT(0, lane == 0);
                if (__any(m < thr[n])) {
                    FDC_STAT(1, lane == 0);
                    // rare path: exact fp32 re-evaluation of the surviving rows
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (acc[r] < thr[n]) {
                            FDC_STAT(2, 1);
                            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                            const float4 p = sP[buf][tile * 32 + row];
                            const int gi = __float_as_int(p.w);
                            const float d = nn_exact_d2(qx[n], qy[n], qz[n], p.x, p.y, p.z);
                            if (base + tile * 32 + row < t_end && nn_better(d, gi, own_d[n], own_i[n])) {
                                own_d[n] = d;
                                own_i[n] = gi;
                                thr[n] = mf_thr(d, X[n], X2[n]);
                            }
                        }
                    }
                    // both half-waves hold the same 32 queries: share the tighter bound
                    const float sb = fminf(own_d[n], __shfl_xor(own_d[n], 32, 64));
                    if (qidx[n] < nq) thr[n] = mf_thr(sb, X[n], X2[n]);
                }
            }
        }
        if (more) stage_write(buf ^ 1, nxt);
        __syncthreads();
        buf ^= 1;
    }
#ifdef FDC_NN_STATS
    for (int i = 0; i < 8; ++i) atomicAdd(&g_nn_stats[i], (unsigned long long)st_cnt[i]);
#endif
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        float od = __shfl_xor(own_d[n], 32, 64);
        int oi = __shfl_xor(own_i[n], 32, 64);
        if (oi >= 0 && (own_i[n] < 0 || nn_better(od, oi, own_d[n], own_i[n]))) { own_d[n] = od; own_i[n] = oi; }
        if (half == 0 && qidx[n] < nq) {
            pd[(size_t)split * nq + qidx[n]] = own_d[n];
            pi[(size_t)split * nq + qidx[n]] = own_i[n];
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Streaming variant for seeded + culled launches (the optimiser's steady state).
// With ~1 % of the cells surviving, the staged kernel above is bound by the per-chunk global-load -> convert ->
// LDS -> barrier latency, not by the matrix pipe.  Here a group of 32*NQ queries (spatially compact: the contact
// slots are Morton-ordered) is served by WPG = 1, 2 or 4 waves of a workgroup.  Each wave builds its own survivor
// list over the cells dealt to it round-robin (cell c belongs to wave c % WPG: neighbouring cells -- which tend to
// survive together -- spread evenly) in three stages: boxes of 16-cell k-d subtrees and then cell boxes against the box
// around the queries' balls (lane-parallel), the near cells against every query (two per round, one per half-wave, boxes
// through LDS), and the QUARTERS of the surviving cells (128 points = four MFMA tiles = one k-d node, NNTarget::qbounds)
// against every query again.  The work list holds quarter ids; the wave re-centres the queries on a cell's centre when
// the cell changes and streams each listed quarter's PRECOMPUTED A fragments (NNTarget::frags, static per scene) from
// global memory straight into registers through a short ring (ST4_PF ahead, running across work items; ids in SGPRs, so
// the loads take a scalar base).  No LDS staging of points, no barrier in the main loop, no partial minima / combine pass;
// a filter hit that is only a query meeting its own seed is recognised from wave ballots, the rare real candidates take a
// compact bit-mask loop; the waves of a group meet in LDS once at the end, merged by the same (d, index) order =>
// bit-identical results to every other kernel here.
// eps per cell: |y'| <= cell radius rc, so eps = K1 * X * rc + K2 * (X^2 + rc^2), X = |x - centre|.
// More waves per group = shorter serial chains and a fuller machine at shard sizes, but the group's set-up is
// repeated by each of them: the host picks WPG by launch size.
// blockIdx -> query group: r1-r2 gave each XCD a contiguous range of groups (an XCD's L2 then holds 1/8 of the clip's
// neighbourhoods); r3 interleaves (see below) -- load balance between the XCDs is worth more than that locality.
#ifndef FDC_ST4_MAXLIST
#define FDC_ST4_MAXLIST 768
#endif
#ifndef FDC_ST4_MAXCELL
#define FDC_ST4_MAXCELL 256
#endif
constexpr int ST4_MAXLIST = FDC_ST4_MAXLIST;   // quarter chunks one wave can list out of its share of the chunks
constexpr int ST4_MAXCELL = FDC_ST4_MAXCELL;   // chunks one wave can list before their quarters are tested
// (tiny values, e.g. -DFDC_ST4_MAXCELL=4 -DFDC_ST4_MAXLIST=8, force the overflow fall-back everywhere: the GPU suite passes with them)
// Ring depth vs occupancy (measured at 512 k queries, NQ = 1, one wave per group): 8 fragments / 4 waves per SIMD
// (128 VGPR) 0.148 ms, 4 / 5 (92 VGPR) 0.139, 2 / 6 (80 VGPR) 0.136: the launch is latency-bound on its set-up
// chain, so resident waves hide more than a deeper ring does.
#ifndef FDC_ST4_PF
#define FDC_ST4_PF 2
#endif
#ifndef FDC_ST4_ADAPTIVE_SLACK
#define FDC_ST4_ADAPTIVE_SLACK 1
#endif
#ifndef FDC_AS_MULT
#define FDC_AS_MULT 4.f
#endif
#ifndef FDC_AS_MAX
#define FDC_AS_MAX 3.f
#endif
#ifndef FDC_AS_NEAR
#define FDC_AS_NEAR 0.8f
#endif
#ifndef FDC_ST4_OCC
#define FDC_ST4_OCC 8
#endif
#ifndef FDC_ST4_OCC2
#define FDC_ST4_OCC2 3
#endif
constexpr int ST4_PF = FDC_ST4_PF;     // A fragments in flight per wave
constexpr int ST4_SUPER = 16;          // chunks per super-cell of the two-level survivor test (consecutive chunks = one k-d subtree)
constexpr int ST4_QCAP = 8;            // filter survivors a lane can queue for the batched exact evaluation after the main loop

// Work-list cache of the streaming kernel (optional; pruning only -- results never depend on it).
// The queries move a few millimetres per optimiser iteration, so the set of quarter chunks a group can possibly need
// changes slowly.  When a wave has to build its list it builds it with every query's bound radius INFLATED by `slack`
// and keeps (a) the list and (b) per query the anchor position a_i and the radius R_i the list was built for.  A later
// launch whose queries satisfy  sqrt(bound_i) + |x_i - a_i| <= R_i  for all i may take the kept list: a quarter that is
// not on it lies further than R_i from a_i, hence further than R_i - |x_i - a_i| >= sqrt(bound_i) from x_i, for every i --
// exactly the condition under which the full three-level test would have dropped it.  Either way the (inflated) list is
// then filtered by the per-query box test with the CURRENT bounds, so the scan visits the same quarters as without a cache.
// Measured on the bench clip (tools/motion_probe.py): with 4 cm of slack 14 % of the groups rebuild per iteration on
// average (33 % during the first 100 iterations, 1.4 % during the last 100).
#ifndef FDC_NN_CACHE_CAP
#define FDC_NN_CACHE_CAP 128
#endif
constexpr int NN_CACHE_CAP = FDC_NN_CACHE_CAP;   // quarter ids kept per (group, wave share); longer lists are not cached.  64 or 128 (r5: a lane
                                        // carries TWO 16-bit ids in its one set-up register: BASELINE config 5's far queries list 60-120 quarters)
static_assert(NN_CACHE_CAP == 64 || NN_CACHE_CAP == 128, "one or two ids per lane");
struct NNCache {
    unsigned short* ids;      // [ngroups * WPG][NN_CACHE_CAP]
    int* hdr;                 // [ngroups * WPG] list length | launches since the anchors were set << 8; -1: no list (anchors one launch old)
    float4* anchor;           // [WPG][nq] {a_i, R_i (validity margin already taken off)}
    float slack;              // metres
    // launch order (one-wave workgroups only; scheduling, never results): the workgroups are dispatched in blockIdx order, and a
    // launch ends with whatever started last -- so the groups with the most work go first (nn_order_kernel, NNOrder below).
    // The tables sit behind hdr in the same allocation (G = ceil(nq / 32) ints each): hdr + 4 G: work items scanned by the
    // workgroup at launch position b in this launch; hdr + 5 G, hdr + 6 G: launch position -> group, two tables written in turn.
    // Indexed by POSITION because blockIdx lives in a scalar register for free: the one-wave kernel sits at its 64-register
    // limit, and keeping the group index (or two more pointer arguments) alive for this store cost 8-10 spilled registers.
    int order_mode = 0;       // 0: off, 1: record the work items, 2 / 3: record + launch in the order of table 0 / 1
};

template <int NQ, int WPG, int WPB = 4>
__global__ __launch_bounds__(64 * WPB, NQ == 1 ? FDC_ST4_OCC : FDC_ST4_OCC2) void nn_stream4_kernel(const float* __restrict__ q, int nq, NNTarget T,
                                                         const int* __restrict__ seed, float4* __restrict__ seedpt,
                                                         float* __restrict__ dist, int* __restrict__ idx, NNCache cache) {
    // WPB waves per workgroup.  Waves of different groups never talk to each other, and a workgroup's slot on the CU is only
    // handed on when its slowest wave is done: with one wave per group the launch runs one-wave workgroups (WPB = 1).
    __shared__ unsigned short slist[WPB][ST4_MAXLIST];           // a wave's work list: quarter chunks 4 k + quarter (chunk WPG k + sub)
    __shared__ unsigned short clist[WPB][ST4_MAXCELL];           // ... before that, the chunks that passed the per-query test
    __shared__ float4 sbox[WPB][64][2];                          // a wave's near chunk boxes of the current batch {lo, bits(chunk)}, {hi, -}
    __shared__ float s_d[WPB][32 * NQ];
    __shared__ int s_i[WPB][32 * NQ];
    __shared__ float4 s_p[WPB][32 * NQ];
    static_assert(WPG == 1 || WPG == 2 || WPG == 4, "waves per query group");
    static_assert(WPB % WPG == 0 && WPB <= 4, "a group's waves share a workgroup");
    constexpr int GPW = WPB / WPG;                               // query groups per workgroup
    constexpr int CPS = ST4_SUPER / WPG;                         // chunks of a super-cell that belong to one wave
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int sub = wave % WPG, gslot = wave / WPG;             // this wave's share of its group's chunks: WPG k + sub
    const int ngroups = (nq + 32 * NQ - 1) / (32 * NQ);
    const int nwg = (ngroups + GPW - 1) / GPW;
#ifdef FDC_NN_TIMELINE
    const unsigned long long tl_t0 = wall_clock64();             // instrumentation build only: 100 MHz device-wide clock
    unsigned long long tl_p[4] = {0, 0, 0, 0};
#define TL_STAMP(k) tl_p[k] = wall_clock64()
#else
#define TL_STAMP(k)
#endif
    // r3: plain round-robin over the XCDs (workgroup b runs on XCD b % 8).  Contiguous frame ranges per XCD leave the XCDs with
    // 13 % different amounts of work (frames near the floor cost more) and the launch ends with its slowest XCD; interleaved,
    // every XCD sees every part of the clip -- consecutive frames share their scene cells anyway.  72.57 -> 72.31 ms per step.
    const int wg = (cache.order_mode >= 2 && (int)blockIdx.x < nwg) ? cache.hdr[(3 + cache.order_mode) * ((nq + 31) / 32) + (int)blockIdx.x] : (int)blockIdx.x;
    const int group = wg * GPW + gslot;
    const bool idle = wg >= nwg || group >= ngroups;            // idle waves still meet the barrier below
    const int wq0 = idle ? nq : group * (32 * NQ);
    const int nchunk = (T.n + MF_CH - 1) / MF_CH;
    const int myn = idle ? 0 : (nchunk - sub + WPG - 1) / WPG;  // this wave's chunks: WPG k + sub, k < myn
#ifdef FDC_NN_STATS
    unsigned st_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // Set-up loads, one batch: the queries, their seeds and seed points, and everything a kept work list needs (header, anchors,
    // ids -- addressed by the group alone).  All unconditional (indices clamped) and consumed together by the empty asm below:
    // left to itself hipcc sinks each load into the branch that uses it -- seed point behind the seed's validity test, the
    // list behind the anchors' test -- and the wave pays four dependent round trips instead of one (a wave alone on its SIMD
    // during the launch's drain is nothing but such round trips).
    const bool caching = cache.hdr != nullptr && !idle;            // (idle waves have no group: nothing kept for them)
    const int cidx = group * WPG + sub;
    float lqx[NQ], lqy[NQ], lqz[NQ];
    int lsj[NQ];
    float4 lsp[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const size_t qc = (size_t)min(wq0 + n * 32 + col, nq - 1);
        lqx[n] = q[3 * qc]; lqy[n] = q[3 * qc + 1]; lqz[n] = q[3 * qc + 2];
        lsj[n] = seed[qc];
        lsp[n] = seedpt[qc];                                      // the seed's coordinates, kept from the launch that found it
    }
    int hv_pre = -1;
    unsigned id_pre = 0;
    float4 anc_pre[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) anc_pre[n] = make_float4(0.f, 0.f, 0.f, -1.f);
    if (caching) {                                                 // wave-uniform
        hv_pre = cache.hdr[cidx];
        id_pre = NN_CACHE_CAP == 64 ? (unsigned)cache.ids[(size_t)cidx * NN_CACHE_CAP + lane]
                                    : ((const unsigned*)cache.ids)[(size_t)cidx * (NN_CACHE_CAP / 2) + lane];
#pragma unroll
        for (int n = 0; n < NQ; ++n) anc_pre[n] = cache.anchor[(size_t)sub * nq + min(wq0 + n * 32 + col, nq - 1)];
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int n = 0; n < NQ; ++n)
        asm volatile("" : "+v"(lqx[n]), "+v"(lqy[n]), "+v"(lqz[n]), "+v"(lsj[n]), "+v"(lsp[n].x), "+v"(lsp[n].y), "+v"(lsp[n].z), "+v"(lsp[n].w),
                          "+v"(anc_pre[n].x), "+v"(anc_pre[n].y), "+v"(anc_pre[n].z), "+v"(anc_pre[n].w), "+v"(hv_pre), "+v"(id_pre));
#endif
    float qx[NQ], qy[NQ], qz[NQ], own_d[NQ], sb[NQ];
    int own_i[NQ], qidx[NQ];
    float4 own_p[NQ];          // the current best {x, y, z, bits(position in T.pts)}: it passes the filter by construction and
                               // must not cost a global load every time it is met; next iteration's seed point
    bool all_seeded = true;
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        qidx[n] = wq0 + n * 32 + col;
        const bool ok = qidx[n] < nq;
        qx[n] = ok ? lqx[n] : 0.f;
        qy[n] = ok ? lqy[n] : 0.f;
        qz[n] = ok ? lqz[n] : 0.f;
        own_d[n] = INFINITY;
        own_i[n] = -1;
        own_p[n] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        const int sj = ok ? lsj[n] : -1;
        const float4 p = ok ? lsp[n] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            if (sj >= 0 && sj < T.n) {
                own_p[n] = p;
                own_d[n] = nn_exact_d2(qx[n], qy[n], qz[n], p.x, p.y, p.z);
                own_i[n] = sj;
            } else {
                all_seeded = false;
            }
        }
        sb[n] = ok ? own_d[n] * 1.00002f + 1e-9f : -INFINITY;   // bound with the rounding slack of the box test
    }
    float rq[NQ];                                                // radius of each query's bound (rounded up)
    bool finite = true;
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        rq[n] = 0.f;
        if (qidx[n] < nq) {
            rq[n] = __builtin_amdgcn_sqrtf(sb[n]) * 1.00001f + 1e-6f;   // 1-ulp v_sqrt_f32 inside the 1e-5 slack
            finite &= rq[n] < INFINITY;
        }
    }
    const bool cull = __all(all_seeded && finite);
    TL_STAMP(0);
    // DPP reductions (fdc_math.h), no LDS traffic; lanes 32-63 repeat lanes 0-31 here, so two of the four row results suffice
    auto min_rows01 = [](float v) {
        v = fminf(v, dpp_move<0xB1>(v)); v = fminf(v, dpp_move<0x4E>(v)); v = fminf(v, dpp_move<0x141>(v)); v = fminf(v, dpp_move<0x140>(v));
        return fminf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16)));
    };
    auto max_rows01 = [](float v) {
        v = fmaxf(v, dpp_move<0xB1>(v)); v = fmaxf(v, dpp_move<0x4E>(v)); v = fmaxf(v, dpp_move<0x141>(v)); v = fmaxf(v, dpp_move<0x140>(v));
        return fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16)));
    };

    // Work-list cache: may this wave take the list it kept?
    int n_kept = -1;
    bool inflate = false;                                        // build the list with slack and keep it?
    int slk = __builtin_amdgcn_readfirstlane(__float_as_int(cache.slack));   // ... with how much (wave-uniform; bits of a float)
    if (cull && caching) {
        const int hv = __builtin_amdgcn_readfirstlane(hv_pre);
        bool ok = true;
        float dmax = 0.f;
#pragma unroll
        for (int n = 0; n < NQ; ++n)
            if (qidx[n] < nq) {
                const float4 a = anc_pre[n];
                const float ex = qx[n] - a.x, ey = qy[n] - a.y, ez = qz[n] - a.z;
                const float dl = __builtin_amdgcn_sqrtf(ez * ez + (ey * ey + ex * ex)) * 1.00001f + 1e-7f;
                ok &= rq[n] + dl <= a.w;
                dmax = fmaxf(dmax, dl);
            }
        if (hv >= 0 && __all(ok)) {
            n_kept = hv & 255;
            if (lane == 0) cache.hdr[cidx] = hv + 256;            // one launch older
        } else {
            // A kept list pays for its slack (a wider build, a longer list to filter) only if it survives a few launches:
            // the queries' speed since the anchors were set (age launches ago) must leave it about three.
            dmax = max_rows01(dmax);
            dmax = fmaxf(dmax, __shfl_xor(dmax, 32, 64));
            const float age = hv >= 0 ? (float)max(hv >> 8, 1) : 1.f;
#if FDC_ST4_ADAPTIVE_SLACK
            // r5: the slack follows the queries' speed -- 4 launches' worth of motion, between the configured slack and 3 x that --
            // so that lists are also kept while the bodies move centimetres per launch (the first ~150 iterations of a fit; all
            // of phase 2 when its contact term is logged: camera_ext turns the bodies ~1 cm per iteration and no list survived,
            // every wave rebuilt its list in every launch, 14 us of a 26 us lifetime: tools/phase_switch_timeline.py)
            const float speed = dmax / age;
            slk = __builtin_amdgcn_readfirstlane(__float_as_int(fminf(fmaxf(FDC_AS_MULT * speed, cache.slack), FDC_AS_MAX * cache.slack)));
            // ... measured against the SMALLEST slack a query of the group gets (far queries' is scaled down, slack_of below): a list
            // is only as durable as its most constrained anchor (late r5: with every vertex a contact and the bodies moving centimetres
            // per launch, config 5's groups built inflated lists that the far queries voided one launch later)
            float smin = __int_as_float(slk);
#pragma unroll
            for (int n = 0; n < NQ; ++n)
                if (qidx[n] < nq && rq[n] > FDC_AS_NEAR) smin = fminf(smin, fmaxf(cache.slack, __int_as_float(slk) * (FDC_AS_NEAR / rq[n])));
            smin = -max_rows01(-smin);
            smin = fminf(smin, __shfl_xor(smin, 32, 64));
            inflate = 3.f * speed <= smin;
#else
            inflate = dmax * 6.f <= cache.slack * age;
#endif
        }
    }
    // The slack of ONE query: the wave's (speed-adaptive) slack for a query near its neighbour; for a far one scaled down by
    // FDC_AS_NEAR / radius, never below the configured slack.  A ball of radius r inflated by s reaches sqrt(2 r s) sideways over a
    // surface it hovers above: with 9 cm on a query 1.5 m from the floor that is half a metre of floor in every direction -- BASELINE
    // config 5's lists (every vertex a query, most of them far) grew past what is kept and its mean launch 10 % (1154 -> 1272 us).
    // (FDC_AS_NEAR 0.8 m: config 5 789 -> 770 ms per fit, config 3 unchanged; 0.4: 764, config 3 within noise; 0.15: config 3 +1.3 %)
    auto slack_of = [&](float r) -> float {
        const float sw = __int_as_float(slk);
        return r <= FDC_AS_NEAR ? sw : fmaxf(cache.slack, sw * (FDC_AS_NEAR / r));
    };
    // the radii / squared bounds the list is BUILT for: inflated by the slack when the list is going to be kept
    float sbT[NQ];
    float glx = INFINITY, gly = INFINITY, glz = INFINITY, ghx = -INFINITY, ghy = -INFINITY, ghz = -INFINITY;
    float scx = 0.f, scy = 0.f, scz = 0.f, sR2 = 0.f;
    if (cull && n_kept < 0) {
        float rT[NQ];
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            rT[n] = inflate ? rq[n] + slack_of(rq[n]) : rq[n];
            sbT[n] = (qidx[n] < nq) ? (inflate ? rT[n] * rT[n] * 1.00002f : sb[n]) : -INFINITY;
        }
        // Group bound for the box tests: the axis-aligned box around the queries' balls (centre x_i, radius r_i).
        // A cell some query needs intersects that query's ball, hence this box -- and for elongated groups (a shin above a
        // floor) the box is far tighter than a sphere around the centroid with the largest reach.
#pragma unroll
        for (int n = 0; n < NQ; ++n)
            if (qidx[n] < nq) {
                const float r = rT[n];
                glx = fminf(glx, qx[n] - r); gly = fminf(gly, qy[n] - r); glz = fminf(glz, qz[n] - r);
                ghx = fmaxf(ghx, qx[n] + r); ghy = fmaxf(ghy, qy[n] + r); ghz = fmaxf(ghz, qz[n] + r);
            }
        glx = min_rows01(glx); gly = min_rows01(gly); glz = min_rows01(glz);
        ghx = max_rows01(ghx); ghy = max_rows01(ghy); ghz = max_rows01(ghz);
        // ... and the sphere around the box centre that contains every ball (radius max_i |x_i - c| + r_i): for a far group the
        // ball box's corners reach much further than any ball does, the sphere cuts them off; a cell has to touch both.
        scx = 0.5f * (glx + ghx); scy = 0.5f * (gly + ghy); scz = 0.5f * (glz + ghz);
        float sR = 0.f;
#pragma unroll
        for (int n = 0; n < NQ; ++n)
            if (qidx[n] < nq) {
                const float ex = qx[n] - scx, ey = qy[n] - scy, ez = qz[n] - scz;
                sR = fmaxf(sR, (__builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez) + rT[n]) * 1.00001f + 1e-6f);
            }
        sR = max_rows01(sR);
        sR2 = sR * sR * 1.00001f;
    } else {
#pragma unroll
        for (int n = 0; n < NQ; ++n) sbT[n] = sb[n];
    }
    // box-box overlap (closed): false only if the boxes are strictly apart along some axis
    // (bitwise |: with short-circuit || the compiler sinks the component loads into a chain of dependent branches)
    auto overlaps = [&](float4 lo, float4 hi) -> bool {
        return !((lo.x > ghx) | (hi.x < glx) | (lo.y > ghy) | (hi.y < gly) | (lo.z > ghz) | (hi.z < glz)) & (box_d2(lo, hi, scx, scy, scz) <= sR2);
    };

    // Survivor list of this wave, two levels: the boxes of 16-chunk super-cells (k-d subtrees) are tested against
    // the group's box 64 per round by every wave; inside the near super-cells each wave tests ITS chunks
    // (4 j + wave, j < 4) -- 16 super-cells x 4 chunks per round of lanes -- against the reach and then per query.
    // (Testing all chunk boxes directly costs every workgroup the whole box array through L1/L2: 31 KB x 16000
    // workgroups per launch at 500k points, more than everything else the kernel reads.)
    // The per-query box tests of the list stages are lane-parallel (r3): the group's queries {x, y, z, squared bound} sit in
    // LDS, a PAIR of lanes holds one box in registers and each lane of the pair walks half of the queries.
    float4* const sq = &s_p[wave][0];                            // [32 NQ] queries (s_p is only needed for the final merge)
    float4* const sbx = &sbox[wave][0][0];                       // [64][2] compacted boxes of a chunk-stage batch
    auto put_queries = [&](const float* bound) {
        if (half == 0) {
#pragma unroll
            for (int n = 0; n < NQ; ++n) sq[n * 32 + col] = make_float4(qx[n], qy[n], qz[n], bound[n]);
        }
        __builtin_amdgcn_wave_barrier();
    };
    auto pair_hits = [&](const float4 blo, const float4 bhi) -> bool {
        const float4* const qs = sq + (lane & 1) * (16 * NQ);
        bool hit = false;
#pragma unroll 8
        for (int i = 0; i < 16 * NQ; ++i) {
            const float4 v = qs[i];
            hit |= box_d2(blo, bhi, v.x, v.y, v.z) <= v.w;
        }
        return hit;
    };
    int nsurv = 4 * myn;                                         // work items are quarter chunks: 4 k + quarter
    bool listed = false;
    if (cull && n_kept >= 0) {                                    // the kept list is still a superset of what this launch can need
        nsurv = n_kept;
        listed = true;
        FDC_STAT(4, lane == 0);
        if (NN_CACHE_CAP == 64) {
            if (lane < n_kept) slist[wave][lane] = (unsigned short)id_pre;
        } else {
            if (2 * lane < n_kept) slist[wave][2 * lane] = (unsigned short)(id_pre & 0xFFFFu);
            if (2 * lane + 1 < n_kept) slist[wave][2 * lane + 1] = (unsigned short)(id_pre >> 16);
        }
        __builtin_amdgcn_wave_barrier();
    } else if (cull) {
        nsurv = 0;
        listed = true;
        int ncell = 0;
        put_queries(sbT);
        const int nsuper = (nchunk + ST4_SUPER - 1) / ST4_SUPER;
        for (int s0 = 0; s0 < nsuper && listed; s0 += 64) {
            const int si = s0 + lane;
            bool nearS = false;
            if (si < nsuper) nearS = overlaps(T.sbounds[2 * si], T.sbounds[2 * si + 1]);
            unsigned long long ms = __ballot(nearS);
            while (ms && listed) {                               // batches of 64 / CPS near super-cells
                int mysuper = -1, e = 0;
                while (ms && e < 64 / CPS) {                     // lane l serves the (l / CPS)-th near super-cell of the batch
                    const int bsup = __ffsll((long long)ms) - 1;
                    ms &= ms - 1;
                    if ((lane / CPS) == e) mysuper = s0 + bsup;
                    ++e;
                }
                const int ci = mysuper * ST4_SUPER + WPG * (lane % CPS) + sub;
                float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
                bool near = false;
                if (mysuper >= 0 && ci < nchunk) {
                    lo = T.bounds[2 * ci];
                    hi = T.bounds[2 * ci + 1];
                    near = overlaps(lo, hi);
                }
                // Per-query test of the near chunks (17 per wave at 512 k queries, 5 survive), lane-parallel: their boxes are
                // compacted in chunk order through LDS, a PAIR of lanes takes one chunk and each of the two walks half of the
                // group's queries (pair_hits) -- up to 32 chunks per pass, no scalar hand-over inside.
                const unsigned long long m = __ballot(near);
                const int nnear = __builtin_amdgcn_readfirstlane(__popcll(m));
                const int k = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                if (near) {
                    sbx[2 * k] = make_float4(lo.x, lo.y, lo.z, __int_as_float(ci));
                    sbx[2 * k + 1] = hi;
                }
                __builtin_amdgcn_wave_barrier();
                for (int p0 = 0; p0 < nnear && listed; p0 += 32) {
                    const int j = min(p0 + (lane >> 1), nnear - 1);
                    const float4 blo = sbx[2 * j], bhi = sbx[2 * j + 1];
                    const bool hit = (p0 + (lane >> 1) < nnear) && pair_hits(blo, bhi);
                    unsigned long long hm = __ballot(hit);
                    hm = (hm | (hm >> 1)) & 0x5555555555555555ull;           // bit 2 j: chunk j of the pass is needed by some query
                    const int cnt = __popcll(hm);
                    if (ncell + cnt > ST4_MAXCELL) { listed = false; break; }
                    if ((lane & 1) == 0 && ((hm >> lane) & 1ull))
                        clist[wave][ncell + __popcll(hm & ((1ull << lane) - 1ull))] = (unsigned short)(__float_as_int(blo.w) / WPG);   // k of chunk WPG k + sub
                    ncell += cnt;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // Second stage: the QUARTERS (128 points = four MFMA tiles = one k-d node) of the surviving chunks against every
        // query.  Only ~11 of the 80 tiles a wave used to scan hold a point within any query's bound; a chunk is a 30 cm
        // patch, the part of it some query's ball reaches usually one or two of its quarters.  A pair of lanes per quarter
        // (32 quarters = 8 chunks per pass), boxes straight from global memory into the pair's registers.
        ncell = __builtin_amdgcn_readfirstlane(ncell);
        for (int p0 = 0; p0 < 4 * ncell && listed; p0 += 32) {
            const int jq = p0 + (lane >> 1);                      // cell jq >> 2 of the list, its quarter jq & 3
            const bool valid = jq < 4 * ncell;
            const int k4 = 4 * (int)clist[wave][valid ? (jq >> 2) : 0];
            const size_t qb = ((size_t)(WPG * (k4 >> 2) + sub) * 4 + (jq & 3)) * 2;
            const float4 blo = T.qbounds[qb], bhi = T.qbounds[qb + 1];
            const bool hit = valid && pair_hits(blo, bhi);
            unsigned long long hm = __ballot(hit);
            hm = (hm | (hm >> 1)) & 0x5555555555555555ull;
            const int cnt = __popcll(hm);
            if (nsurv + cnt > ST4_MAXLIST) { listed = false; break; }
            if ((lane & 1) == 0 && ((hm >> lane) & 1ull))
                slist[wave][nsurv + __popcll(hm & ((1ull << lane) - 1ull))] = (unsigned short)(k4 + (jq & 3));
            nsurv += cnt;
            __builtin_amdgcn_wave_barrier();
        }
        FDC_STAT(5, lane == 0 && !idle);
        if (caching) {                                            // keep the (inflated) list and what it was built for
            const bool keep = inflate && listed && nsurv <= NN_CACHE_CAP;
            if (NN_CACHE_CAP == 64) {
                if (keep && lane < nsurv) cache.ids[(size_t)cidx * NN_CACHE_CAP + lane] = slist[wave][lane];
            } else if (keep && 2 * lane < nsurv)
                ((unsigned*)cache.ids)[(size_t)cidx * (NN_CACHE_CAP / 2) + lane] =
                    (unsigned)slist[wave][2 * lane] | ((2 * lane + 1 < nsurv ? (unsigned)slist[wave][2 * lane + 1] : 0u) << 16);
            if (lane == 0) cache.hdr[cidx] = keep ? (nsurv | 256) : -1;
            if (half == 0) {
#pragma unroll
                for (int n = 0; n < NQ; ++n)
                    if (qidx[n] < nq)                              // R_i with the validity test's rounding margin taken off; no list: never valid
                        cache.anchor[(size_t)sub * nq + qidx[n]] =
                            make_float4(qx[n], qy[n], qz[n], keep ? (rq[n] + slack_of(rq[n])) * 0.99998f - 2e-6f : -1.f);
            }
        }
        if (!listed) nsurv = 4 * myn;                           // list overflow: scan this wave's whole share (still exact)
    }
    TL_STAMP(1);
    // With a cache the list (kept or just built) was made for inflated radii: filter it by the per-query box test with the
    // CURRENT bounds, in place.  r3: lane-parallel -- a pair of lanes per listed quarter (32 quarters per pass), each lane walks
    // half of the group's queries; ONE ballot and one ordered compaction per pass.  The r2 form tested two quarters per round
    // with every lane on its own query: 11 rounds of LDS read -> 13 VALU -> compare -> scalar branch -> lane-0 store, ~1150
    // cycles per round even for a wave alone on its SIMD (timeline stamps: 5.2 of a lone wave's 19.7 us) -- a chain of
    // VALU / SALU / LDS hand-overs, not work.
    if (cull && caching && listed && (inflate || n_kept >= 0)) {
        const int n_raw = __builtin_amdgcn_readfirstlane(nsurv);
        int nout = 0;
        FDC_STAT(6, lane == 0 ? n_raw : 0);
        put_queries(sb);
        for (int k0 = 0; k0 < n_raw; k0 += 32) {
            const int kk = k0 + (lane >> 1);
            const bool valid = kk < n_raw;
            const int id = (int)slist[wave][valid ? kk : 0];
            const size_t qb = ((size_t)(WPG * (id >> 2) + sub) * 4 + (id & 3)) * 2;
            const float4 blo = T.qbounds[qb], bhi = T.qbounds[qb + 1];
            const bool hit = valid && pair_hits(blo, bhi);
            unsigned long long hm = __ballot(hit);
            hm = (hm | (hm >> 1)) & 0x5555555555555555ull;
            if ((lane & 1) == 0 && ((hm >> lane) & 1ull))         // (in place: a pass reads its ids before it writes, and writes trail reads)
                slist[wave][nout + __popcll(hm & ((1ull << lane) - 1ull))] = (unsigned short)id;
            nout += __popcll(hm);
            __builtin_amdgcn_wave_barrier();
        }
        nsurv = nout;
        FDC_STAT(7, lane == 0 ? nout : 0);
    }
    TL_STAMP(2);
#ifdef FDC_NN_TIMELINE
    const int tl_nsurv = nsurv;
#endif
    const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int NT = MF_CH / 32;                               // 16 tiles per chunk (padding rows score 1e30)
    constexpr int QT = NT / 4;                                   // tiles per work item (quarter chunk)
    static_assert(QT % ST4_PF == 0, "the prefetch ring turns a whole number of times per work item");

#ifdef FDC_NN_STATS
    if (lane == 0 && !idle) atomicAdd(&g_nn_hist[(listed ? 0 : 16) + (nsurv > 0 ? 32 - __clz(nsurv) : 0)], 1ull);
#endif
    // queue of filter survivors awaiting their exact evaluation (NQ == 1; see the main loop): per lane up to ST4_QCAP positions
    // in the LDS the list stages used (sbox: 2 KB per wave, free from here on)
    unsigned* const cq = (unsigned*)&sbox[wave][0][0];
    int cq_n = 0;
    if (nsurv > 0) {
        // work items (quarter chunks) are wave-uniform: ids kept in SGPRs (readfirstlane), so fragment addresses are scalar
        // base + lane offset + immediate and the centres come through the scalar cache, one item ahead
        int id = __builtin_amdgcn_readfirstlane(listed ? (int)slist[wave][0] : 0);
        int ch = __builtin_amdgcn_readfirstlane(WPG * (id >> 2) + sub), qd = id & 3;
        // The fragment stream goes through BUFFER loads (r3): one resource descriptor for the whole fragment array in SGPRs, the
        // item's byte offset as the scalar offset, lane * 16 as the vector offset, the tile within the item as the immediate --
        // `buffer_load_dwordx4 v, v_lane, s[rsrc], s_item offen offset:imm`.  The r2 form (global loads from a scalar base
        // forced by an empty asm) cost a v_mov per tile and 64-bit scalar address arithmetic per item; left to itself the
        // compiler turned every tile's address into 64-bit VALU adds.
        const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)T.frags, 0, (int)min((size_t)nchunk * NT * 1024, (size_t)0x7fffffff), 0x00020000);
        unsigned fo = (unsigned)(ch * NT + qd * QT) * 1024u;               // [tile][half][col] == [tile][lane]: 1 KiB per tile
        const unsigned lofs = (unsigned)lane * 16u;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 f[ST4_PF];
#pragma unroll
        for (int j = 0; j < ST4_PF; ++j) f[j] = __builtin_amdgcn_raw_buffer_load_b128(frs, lofs, fo + j * 1024, 0);
        // The cell centres are wave-uniform and static per scene: read through the CONSTANT address space they become s_load_dwordx4
        // into four SGPRs.  (r5: as a plain global load the compiler kept the next centre in four VGPRs across the whole item --
        // with one more live value in this 64-register kernel it spilled exactly those, behind an s_waitcnt vmcnt(0): a scratch
        // round trip and a drained fragment ring per work item, +20 us per launch.)
        typedef float cf4_t __attribute__((ext_vector_type(4)));
        typedef const cf4_t __attribute__((address_space(4)))* cf4_ptr;
        const cf4_ptr centers_c = (cf4_ptr)(const void*)T.centers;
        cf4_t cc_next = centers_c[ch];
        nsurv = __builtin_amdgcn_readfirstlane(nsurv);
        int cur = -1;                                                      // chunk the queries are centred on
        bf16x8 bfrag[NQ];
        // Per query and current cell: thr (rows with a score below it pass), e2 = 2.01 eps and c2 = eps - |x'|^2, with
        // eps = K1 |x'| rc + K2 (|x'|^2 + rc^2) the filter's error bound for this cell (|score + |x'|^2 - d| <= eps / 1.5).
        // r5: the SCORES tighten the threshold, without an exact evaluation.  The error bound cuts both ways: a row with score s lies
        // at d <= s + |x'|^2 + eps, so once a tile shows a row with score m the neighbour is no further than that -- and every row
        // with s >= m + 2.01 eps is out, whatever the seed promised.  With a tight seed (steady state) nothing changes; with a loose
        // one -- a seed that moved centimetres since it was found: the first iterations of a fit, the iterations after the phase
        // switch, when camera_ext turns the bodies about the world origin -- the rows inside the seed's ball used to be evaluated
        // exactly one by one (23 per query and launch against 0.1, most of them as dependent round trips once a lane's queue was
        // full: launches of 150-290 us, tools/phase_switch_probe.py / phase_switch_timeline.py); now only the rows within 2 eps
        // of the tile's best are.  Pruning only: every row that can win or tie still passes.  (What a cell's scores proved is not
        // carried to the next cell: one more live register in this 64-register kernel spills a fragment of the ring.)
        float thr[NQ], e2[NQ], c2[NQ];
#pragma unroll
        for (int n = 0; n < NQ; ++n) { thr[n] = -INFINITY; e2[n] = c2[n] = 0.f; bfrag[n] = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u)); }
        // NQ == 1: the tile of this lane's current best (-1: none in this half) and the mask that clears its row's bit --
        // own_p only changes in the queue-full path below, which refreshes them
        int seed_tile = -1;
        unsigned seed_keep = ~0u;
        auto seed_refresh = [&]() {
            const int spos = __float_as_int(own_p[0].w);
            seed_tile = (spos >= 0 && ((spos >> 2) & 1) == half) ? (spos >> 5) : -1;
            seed_keep = ~(1u << ((spos & 3) + 4 * ((spos >> 3) & 3)));
        };
        seed_refresh();
        for (int s = 0; s < nsurv; ++s) {
            const int s1 = min(s + 1, nsurv - 1);                          // last item: harmless re-fetch of itself
            const int id_next = __builtin_amdgcn_readfirstlane(listed ? (int)slist[wave][s1] : s1);
            const int ch_next = __builtin_amdgcn_readfirstlane(WPG * (id_next >> 2) + sub);
            const unsigned fo_next = (unsigned)(ch_next * NT + (id_next & 3) * QT) * 1024u;
            const cf4_t cc = cc_next;
            cc_next = centers_c[ch_next];
            FDC_STAT(3, lane == 0);
            if (ch != cur) {                                               // wave-uniform: quarters of one chunk follow each other
                cur = ch;
                // re-centre the queries on the chunk centre
                const float rc = cc.w;
#pragma unroll
                for (int n = 0; n < NQ; ++n) {
                    const float xx = qx[n] - cc.x, xy = qy[n] - cc.y, xz = qz[n] - cc.z;
                    const float X2 = __fmaf_rn(xz, xz, __fmaf_rn(xy, xy, xx * xx));
                    const float X = __builtin_amdgcn_sqrtf(X2) * 1.000001f;  // only feeds eps: 1-ulp v_sqrt_f32, rounded up
                    const unsigned hx = f2bf(xx), hy = f2bf(xy), hz = f2bf(xz);
                    const unsigned lx = f2bf(xx - bf2f(hx)), ly = f2bf(xy - bf2f(hy)), lz = f2bf(xz - bf2f(hz));
                    const unsigned px = hx | (lx << 16), py = hy | (ly << 16), pz = hz | (lz << 16);   // (the factor -2 is in the A fragments)
                    const unsigned one = 0x3F80u;
                    const uint4 u = half == 0 ? make_uint4(px, px, py, py) : make_uint4(pz, pz, one | (one << 16), one);
                    bfrag[n] = __builtin_bit_cast(bf16x8, u);
                    const float eps = MF_K1 * X * rc + MF_K2 * (X2 + rc * rc);
                    e2[n] = 2.01f * eps;
                    c2[n] = eps - X2;
                    // this half's own bound (the other half's may be tighter after an exact hit; it is folded in at the next
                    // hit -- a looser threshold only lets more pairs through, and saves a cross-half exchange per cell)
                    thr[n] = (qidx[n] < nq) ? own_d[n] + c2[n] : -INFINITY;
                }
            }
            const int base = ch * MF_CH + qd * (QT * 32);
#pragma unroll
            for (int tile = 0; tile < QT; ++tile) {
                const bf16x8 afrag = __builtin_bit_cast(bf16x8, f[tile % ST4_PF]);
                const int tn = tile + ST4_PF;                               // a quarter's four tiles are within reach of the immediates
                // (the tile's offset rides in the SCALAR offset: added to the lane offset it becomes a v_or per tile)
                f[tile % ST4_PF] = __builtin_amdgcn_raw_buffer_load_b128(frs, lofs, (tn < QT ? fo : fo_next) + (unsigned)(tn < QT ? tn : tn - QT) * 1024u, 0);
                f32x16_t acc_q[NQ];
#pragma unroll
                for (int n = 0; n < NQ; ++n) acc_q[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag[n], zero, 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NQ; ++n) {
                    const f32x16_t acc = acc_q[n];
                    const float t0 = fminf(fminf(acc[0], acc[1]), acc[2]), t1 = fminf(fminf(acc[3], acc[4]), acc[5]);
                    const float t2 = fminf(fminf(acc[6], acc[7]), acc[8]), t3 = fminf(fminf(acc[9], acc[10]), acc[11]);
                    const float t4 = fminf(fminf(acc[12], acc[13]), acc[14]);
                    const float g0 = fminf(fminf(t0, t1), t2), g1 = fminf(fminf(t3, t4), acc[15]);      // rows 0-8, rows 9-15
                    const float m = fminf(g0, g1);
                    FDC_STAT(0, lane == 0);
                    if (__any(m < thr[n])) {
                        FDC_STAT(1, lane == 0);
                        // This lane's rows that passed, as a bit mask: sign(acc[r] - thr) shifted in row by row (v_sub_f32 +
                        // v_alignbit_b32 per row; a difference of two distinct finite floats is never rounded to zero, an
                        // invalid lane's thr is -inf, and a row passes iff its triple's minimum does).  r3: 13.5 entries per wave, and FDC_NN_STATS shows most of them carry
                        // a real candidate next to the seeds (about one other point per query lies within the filter's eps
                        // of the bound), so the r2 form -- 16 wave ballots + ~50 SALU to recognise seed-only entries, then
                        // 16 64-bit shifts to recover the per-lane bits -- paid both halves nearly every time: half of the
                        // kernel's VALU cycles (ablation: 79 -> 44 us steady state without the body).
                        // Only the HALF of the min tree (rows 0-8 / rows 9-15) in which some lane has a passing row is expanded: an
                        // entry is usually one candidate or one seed, i.e. one half -- 2 compares + 18 or 14 instead of 32
                        // instructions, 13.5 entries per wave.  (Finer -- the five row triples of the tree's first level, 6 + 7
                        // instructions per entry -- costs 15-26 spilled registers in this 64-register kernel; nested under the
                        // halves it fits, and the extra wave votes and branches make the launch 3 % slower: 67.8 vs 65.9 us.)
                        unsigned mask = 0;
                        if (qidx[n] < nq) thr[n] = fminf(thr[n], m + e2[n]);               // (a lane without a passing row: m >= thr, no change)
                        const float th = thr[n];
                        if (__any(g0 < th)) {
#pragma unroll
                            for (int r = 8; r >= 0; --r) mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(acc[r] - th), 31);
                        }
                        if (__any(g1 < th)) {
                            unsigned hi = 0;
#pragma unroll
                            for (int r = 15; r >= 9; --r) hi = __builtin_amdgcn_alignbit(hi, __float_as_uint(acc[r] - th), 31);
                            mask |= hi << 9;
                        }
                        // the lane's current best passes by construction (it sits in this tile's rows of this half when
                        // its tile comes up) and is never re-evaluated: its row's bit is cleared
                        if constexpr (NQ == 1) {
                            mask &= (seed_tile == (base >> 5) + tile) ? seed_keep : ~0u;
                        } else {
                            const int spos = __float_as_int(own_p[n].w);
                            const bool mine = (spos >> 5) == (base >> 5) + tile && ((spos >> 2) & 1) == half;
                            if (mine) mask &= ~(1u << ((spos & 3) + 4 * ((spos >> 3) & 3)));
                        }
                        if (!__any(mask != 0)) continue;
                        // Real candidates (FDC_NN_STATS over a fit: 84 % of the waves meet at least one per launch, ~13 entries
                        // per wave -- with bodies hovering above a densely sampled floor about one other point per query lies
                        // within the filter's eps of the bound).  Evaluated on the spot each one is a dependent L2 round trip
                        // in the middle of the main loop (nothing else of the wave proceeds meanwhile: ~10 of them in a row are
                        // most of a lone wave's 11-13 us main loop).  So they are QUEUED -- positions per lane, in
                        // the LDS the list stages no longer need -- and evaluated together after the loop, all loads in flight at
                        // once.  The bound is not tightened in between: the filter then lets through a superset of what it would
                        // have, every member of which is evaluated exactly and merged in (d, index) order -- same result, bit for bit.
                        if constexpr (NQ == 1) {
                            while (mask) {
                                const int r = __ffs(mask) - 1;
                                mask &= mask - 1;
                                const int pos = base + tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                                if (pos < T.n) {   // (a padding row scores 1e30, which still passes an INFINITE bound -- unseeded scans; the
                                                   // lane's own best was cleared from the mask above)
                                    FDC_STAT(2, 1);
                                    if (cq_n < ST4_QCAP) { cq[cq_n * 64 + lane] = (unsigned)pos; ++cq_n; }
                                    else {                                 // queue full (rare): evaluated on the spot
                                        const float4 p = T.pts[pos];
                                        const int gi = __float_as_int(p.w);
                                        const float d = nn_exact_d2(qx[n], qy[n], qz[n], p.x, p.y, p.z);
                                        if (nn_better(d, gi, own_d[n], own_i[n])) {
                                            own_d[n] = d; own_i[n] = gi;
                                            own_p[n] = make_float4(p.x, p.y, p.z, __int_as_float(pos));
                                            seed_refresh();
                                        }
                                    }
                                }
                            }
                            continue;
                        }
                        while (mask) {                                     // rows of this lane that passed the filter
                            const int r = __ffs(mask) - 1;
                            mask &= mask - 1;
                            const int pos = base + tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            if (pos < T.n && pos != __float_as_int(own_p[n].w)) {
                                FDC_STAT(2, 1);
                                const float4 p = T.pts[pos];
                                const int gi = __float_as_int(p.w);
                                const float d = nn_exact_d2(qx[n], qy[n], qz[n], p.x, p.y, p.z);
                                if (nn_better(d, gi, own_d[n], own_i[n])) {
                                    own_d[n] = d; own_i[n] = gi;
                                    own_p[n] = make_float4(p.x, p.y, p.z, __int_as_float(pos));
                                }
                            }
                        }
                        const float sbest = fminf(own_d[n], __shfl_xor(own_d[n], 32, 64));
                        if (qidx[n] < nq) thr[n] = fminf(thr[n], sbest + c2[n]);
                    }
                }
            }
            ch = ch_next;
            qd = id_next & 3;
            fo = fo_next;
        }
    }
    if constexpr (NQ == 1) {
        // the queued candidates: up to four loads in flight per lane and round (a wave's LDS traffic is in order: no barrier)
        for (int k0 = 0; __any(k0 < cq_n); k0 += 4) {
            float4 p[4];
            int ps[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ps[k] = (k0 + k < cq_n) ? (int)cq[(k0 + k) * 64 + lane] : 0;
                p[k] = T.pts[ps[k]];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k0 + k < cq_n) {
                    const int gi = __float_as_int(p[k].w);
                    const float d = nn_exact_d2(qx[0], qy[0], qz[0], p[k].x, p[k].y, p[k].z);
                    if (nn_better(d, gi, own_d[0], own_i[0])) {
                        own_d[0] = d; own_i[0] = gi;
                        own_p[0] = make_float4(p[k].x, p[k].y, p[k].z, __int_as_float(ps[k]));
                    }
                }
        }
    }
    TL_STAMP(3);
    // (the launch order's key: the work items scanned, filed under the launch position.  Also tried as keys: + a bonus when the list was built this launch -- no
    // better, such a wave does not build again next launch --, and the wave's measured main-loop time -- worse, 70.4 vs 68.6 us:
    // it follows contention)
    if (cache.order_mode != 0 && lane == 0 && !idle) cache.hdr[4 * ((nq + 31) / 32) + (int)blockIdx.x] = nsurv;
    // the two halves of a wave hold different scene rows of the same queries; then the four waves meet in LDS
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const float od = __shfl_xor(own_d[n], 32, 64);
        const int oi = __shfl_xor(own_i[n], 32, 64);
        float4 op;
        op.x = __shfl_xor(own_p[n].x, 32, 64); op.y = __shfl_xor(own_p[n].y, 32, 64);
        op.z = __shfl_xor(own_p[n].z, 32, 64); op.w = __shfl_xor(own_p[n].w, 32, 64);
        if (oi >= 0 && (own_i[n] < 0 || nn_better(od, oi, own_d[n], own_i[n]))) { own_d[n] = od; own_i[n] = oi; own_p[n] = op; }
        if (half == 0) { s_d[wave][n * 32 + col] = own_d[n]; s_i[wave][n * 32 + col] = own_i[n]; s_p[wave][n * 32 + col] = own_p[n]; }
    }
    __syncthreads();
#ifdef FDC_NN_STATS
    {   // per wave: slow-path entries (wave-level count sits in lane 0) and the longest per-lane chain of exact re-evaluations
        unsigned mx = st_cnt[2];
        for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));
        if (lane == 0 && !idle) {
            atomicAdd(&g_nn_hist[32 + (st_cnt[1] > 0 ? 32 - __clz(st_cnt[1]) : 0)], 1ull);
            atomicAdd(&g_nn_hist[64 + (mx > 0 ? 32 - __clz(mx) : 0)], 1ull);
        }
    }
    for (int i = 0; i < 8; ++i) atomicAdd(&g_nn_stats[i], (unsigned long long)st_cnt[i]);
#endif
#ifdef FDC_NN_TIMELINE
    if (tid == 0 && blockIdx.x < 16384) {
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* tl = g_nn_timeline + (size_t)blockIdx.x * 8;
        tl[0] = tl_t0; tl[1] = wall_clock64(); tl[2] = xcc & 15; tl[3] = tl_p[0]; tl[4] = tl_p[1]; tl[5] = tl_p[2]; tl[6] = tl_p[3]; tl[7] = (unsigned long long)tl_nsurv;
    }
#endif
    if (tid < GPW * 32 * NQ) {
        const int g = tid / (32 * NQ), e = tid % (32 * NQ);      // group slot of this workgroup, query of the group
        const int qo = (wg * GPW + g) * (32 * NQ) + e;
        if (wg < nwg && qo < nq) {
            int bw = g * WPG;
            float bd = s_d[bw][e];
            int bi = s_i[bw][e];
#pragma unroll
            for (int w = 1; w < WPG; ++w) {
                const float d = s_d[g * WPG + w][e];
                const int i = s_i[g * WPG + w][e];
                if (i >= 0 && (bi < 0 || nn_better(d, i, bd, bi))) { bd = d; bi = i; bw = g * WPG + w; }
            }
            dist[qo] = bd;
            idx[qo] = bi;
            seedpt[qo] = s_p[bw][e];
        }
    }
}

// Seeds for queries that have none (the first iteration of a fit): any scene point gives a valid upper
// bound, a good one makes the culled scan cheap.  Thread per query: nearest chunk box (all boxes, read
// wave-uniformly), then the nearest point of that one chunk.  Pruning aid only -- results do not depend on it.
__global__ __launch_bounds__(256) void nn_seed_kernel(const float* __restrict__ q, int nq, NNTarget T, int* __restrict__ seed,
                                                      float4* __restrict__ seedpt) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    const int old = seed[qi];
    if (old >= 0 && old < T.n) {                              // a seed without its coordinates (kept from another kernel): fetch them
        const int pos = T.inv_perm ? T.inv_perm[old] : old;
        const float4 pt = T.pts[pos];
        seedpt[qi] = make_float4(pt.x, pt.y, pt.z, __int_as_float(pos));
        return;
    }
    const float x = q[3 * (size_t)qi], y = q[3 * (size_t)qi + 1], z = q[3 * (size_t)qi + 2];
    const int nchunk = (T.n + MF_CH - 1) / MF_CH;
    float bb = INFINITY;
    int bc = -1;
    int p0, p1;
    if (T.sbounds) {
        // the nearest chunk box, by branch and bound over the 16-chunk subtrees: a subtree whose box is not nearer than the best
        // chunk so far cannot hold a nearer one (~62 + a few x 16 box tests instead of 977; the same chunk as the plain loop
        // finds, ties included: chunks are visited in ascending order and must be strictly nearer to replace)
        const int nsuper = (nchunk + ST4_SUPER - 1) / ST4_SUPER;
        for (int s = 0; s < nsuper; ++s) {
            const float ds = box_d2(T.sbounds[2 * s], T.sbounds[2 * s + 1], x, y, z);
            if (!(ds < bb)) continue;
            for (int c = s * ST4_SUPER; c < min(nchunk, (s + 1) * ST4_SUPER); ++c) {
                const float d = box_d2(T.bounds[2 * c], T.bounds[2 * c + 1], x, y, z);
                if (d < bb) { bb = d; bc = c; }
            }
        }
    } else {
        for (int c = 0; c < nchunk; ++c) {
            const float d = box_d2(T.bounds[2 * c], T.bounds[2 * c + 1], x, y, z);
            if (d < bb) { bb = d; bc = c; }
        }
    }
    if (bc < 0) return;                                       // NaN query: stays unseeded (the scan handles it)
    float bd = INFINITY;
    int bi = -1;
    float4 bp = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
    auto scan = [&](int c) {
        p0 = c * MF_CH;
        p1 = min(T.n, p0 + MF_CH);
        for (int p = p0; p < p1; ++p) {
            const float4 pt = T.pts[p];
            const float d = nn_exact_d2(x, y, z, pt.x, pt.y, pt.z);
            if (d < bd) { bd = d; bi = __float_as_int(pt.w); bp = make_float4(pt.x, pt.y, pt.z, __int_as_float(p)); }
        }
    };
    scan(bc);
    // r5: the nearest BOX can be a poor choice far from every surface -- a cell at the scene's rim is a large box with its points at
    // the far end -- and the first launch of a fit pays for a loose seed with a ball that reaches thousands of quarters (BASELINE
    // config 5, 5.4 M queries most of them metres from anything: the first search launch took 83 ms, every later one ~1).  A cell's
    // centre and radius bound its farthest point: the cell with the smallest |x - c| + r is scanned as well when that bound beats the
    // first cell's best.
    if (T.centers) {
        float bm = INFINITY;
        int bc2 = -1;
        auto centre = [&](int c) {
            const float4 cc = T.centers[c];
            const float dx = x - cc.x, dy = y - cc.y, dz = z - cc.z;
            const float m = __builtin_amdgcn_sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx))) + cc.w;
            if (m < bm) { bm = m; bc2 = c; }
        };
        if (T.sbounds) {                                      // (a subtree's box distance is a lower bound of every |x - c| + r in it)
            const int nsuper = (nchunk + ST4_SUPER - 1) / ST4_SUPER;
            for (int sb = 0; sb < nsuper; ++sb) {
                if (!(box_d2(T.sbounds[2 * sb], T.sbounds[2 * sb + 1], x, y, z) < bm * bm)) continue;
                for (int c = sb * ST4_SUPER; c < min(nchunk, (sb + 1) * ST4_SUPER); ++c) centre(c);
            }
        } else
            for (int c = 0; c < nchunk; ++c) centre(c);
        if (bc2 >= 0 && bc2 != bc && bm * bm < bd) scan(bc2);
    }
    seed[qi] = bi;
    seedpt[qi] = bp;
}

__global__ void nn_combine_kernel(const float* __restrict__ pd, const int* __restrict__ pi, int nsplit, int nq,
                                  float* __restrict__ dist, int* __restrict__ idx) {
    int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    float best = INFINITY;
    int bi = -1;
    for (int s = 0; s < nsplit; ++s) {
        float d = pd[(size_t)s * nq + qi];
        int j = pi[(size_t)s * nq + qi];
        if (j >= 0 && (bi < 0 || nn_better(d, j, best, bi))) { best = d; bi = j; }
    }
    dist[qi] = best;
    idx[qi] = bi;
}

// xyz [n,3] -> float4 {x, y, z, bits(i)}
__global__ void pack_points_kernel(const float* __restrict__ xyz, int n, float4* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __int_as_float(i));
}

// d dist / d query: NmDistanceGradKernel restricted to the query side (the scene has no grad);
// tgt is indexed by original index
__global__ void nn_grad_kernel(const float* __restrict__ q, const float4* __restrict__ tgt, const float* __restrict__ g,
                               const int* __restrict__ idx, int nq, float* __restrict__ gq) {
    int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    int j = idx[qi];
    float gg = 2.f * g[qi];
    float4 p = j >= 0 ? tgt[j] : make_float4(0, 0, 0, 0);
    gq[3 * (size_t)qi] = j >= 0 ? gg * (q[3 * (size_t)qi] - p.x) : 0.f;
    gq[3 * (size_t)qi + 1] = j >= 0 ? gg * (q[3 * (size_t)qi + 1] - p.y) : 0.f;
    gq[3 * (size_t)qi + 2] = j >= 0 ? gg * (q[3 * (size_t)qi + 2] - p.z) : 0.f;
}

// kernel choice: 0 = by size (MFMA-filtered for non-trivial sizes), 1 = plain VALU scan,
// 2 = MFMA-filtered.  Env FDCAP_NN_KERNEL=direct|mfma or fdcap_set_nn_kernel() override (A/B).
inline std::atomic<int>& nn_mode_ref() {
    static std::atomic<int> mode{-1};
    if (mode < 0) {
        const char* e = getenv("FDCAP_NN_KERNEL");
        mode = (e && e[0] == 'd') ? 1 : (e && e[0] == 'm') ? 2 : 0;
    }
    return mode;
}
static inline bool nn_use_mfma(int nq, int nt) {
    int mode = nn_mode_ref();
    if (mode == 1) return false;
    if (mode == 2) return true;
    return (long long)nq * nt >= (1LL << 22);
}

static inline int nn_pick_nsplit(int nq, int nt, bool culled = false) {
    // Each split re-reads the queries / seeds and writes its own partial minima, so fewer, longer
    // splits win once there are enough query blocks to occupy 256 CUs x 3 resident workgroups.
    // Measured on 1024 frames x 500 contacts vs 500k points: brute-force scan nsplit 2 (9.7 ms) beats
    // 1 (11.1) and 8 (9.9); seeded + chunk-culled scan nsplit 1 (1.09 ms) beats 2 (1.23) and 4 (1.43).
    int qblocks = culled ? (nq + 255) / 256 : (nq + 511) / 512;
    int ns = 1;
    const int target = culled ? 1536 : 1024;
    while (qblocks * ns < target && ns < 64 && nt / (ns * 2) >= 4 * MF_CH) ns *= 2;
    if (!culled && qblocks >= 512 && nt >= 8 * MF_CH) ns = max(ns, 2);
    while (nn_split_len(nt, ns) / MF_CH > MF_MAXCHUNK) ns *= 2;
    return ns;
}

// Launch order of nn_stream4_kernel's one-wave workgroups: position -> group, most work items first (counting sort on the
// counts the last launch wrote; ties in no particular order -- the order decides when a group runs, never what it returns).
__global__ __launch_bounds__(1024) void nn_order_kernel(const int* __restrict__ work, int n, const int* __restrict__ prev,
                                                        int* __restrict__ order) {
    // work[b]: items scanned by the workgroup at launch position b, which served group prev[b] (prev == nullptr: group b)
    // a histogram per wave: most groups fall into a dozen bins, and LDS atomics on one address serialise
    __shared__ int cnt[16][256], base[256];
    const int tid = threadIdx.x, w = tid >> 6;
    for (int e = tid; e < 16 * 256; e += 1024) (&cnt[0][0])[e] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) atomicAdd(&cnt[w][255 - min(max(work[i], 0), 255)], 1);
    __syncthreads();
    if (tid < 256) {                                           // bin tid: the waves' shares become offsets inside the bin
        int run = 0;
        for (int k = 0; k < 16; ++k) { const int t = cnt[k][tid]; cnt[k][tid] = run; run += t; }
        base[tid] = run;
    }
    __syncthreads();
    if (tid < 64) {                                            // exclusive prefix over the 256 bins: four per lane + a wave scan
        const int a = base[4 * tid], b = base[4 * tid + 1], c = base[4 * tid + 2], d = base[4 * tid + 3];
        int s = a + b + c + d, incl = s;
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (tid >= off) incl += t; }
        const int ex = incl - s;
        base[4 * tid] = ex; base[4 * tid + 1] = ex + a; base[4 * tid + 2] = ex + a + b; base[4 * tid + 3] = ex + a + b + c;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const int k = 255 - min(max(work[i], 0), 255);
        order[base[k] + atomicAdd(&cnt[w][k], 1)] = prev ? prev[i] : i;
    }
}
// host-side bookkeeping of the launch order (owned by the optimiser state)
struct NNOrder {
    bool on = false;            // the cache's hdr allocation has room for the tables (7 ints per 32-query group)
    int sorted_groups = 0;      // number of groups the current table was sorted for (0: none yet)
    int cur = 0;                // which of the two order tables is current
    int age = 0;                // launches since it was written
    int every = 32;             // re-sort period (work per group drifts over tens of iterations)
};

// workspace: pd/pi [nsplit*nq]; seed: optional [nq] original indices (may alias idx: read before idx is rewritten)
static inline hipError_t nn_search(const float* q, int nq, const NNTarget& T, float* dist, int* idx, float* pd, int* pi,
                                   int nsplit, hipStream_t st, const int* seed = nullptr, bool seed_missing = false,
                                   float4* seedpt = nullptr, bool* seedpt_written = nullptr, const NNCache* cache = nullptr,
                                   NNOrder* ord = nullptr) {
    if (seedpt_written) *seedpt_written = false;             // true: seedpt[q] = coordinates of the neighbour idx[q] after this launch
    if (nq <= 0) return hipSuccess;
    // Query blocks per workgroup: 4 waves x NQ x 32.  A brute-force scan wants NQ = 4 (most MFMAs per
    // staged chunk: 9.7 ms vs 10.9 at NQ = 2); a seeded + chunk-culled scan wants NQ = 2 (the union of
    // the chunks 256 queries need is smaller than what 512 need, twice the workgroups: 0.92 ms vs
    // 1.10 ms at NQ = 4, 1.09 ms at NQ = 1).  FDCAP_NN_NQ overrides.
    static std::atomic<int> forced_nq{-1};
    if (forced_nq < 0) { const char* e = getenv("FDCAP_NN_NQ"); forced_nq = e ? atoi(e) : 0; }
    const bool culled = seed != nullptr && T.bounds != nullptr;
    // FDCAP_NN_STREAM (A/B): 0 staged kernel, WQ = nn_stream4_kernel with W waves per group of 32 Q queries
    // (41, 42, 21, 22, 11, 12); default: 32-query groups, waves per group by launch size --
    // measured (1024 / 512 / 256 / 128 frames x 500 queries): 11: 0.139 / 0.095 / 0.054 / 0.072 ms, 21: 0.140 / 0.087 /
    // 0.049 / 0.047, 41: 0.153 / 0.086 / 0.047 / 0.036
    static std::atomic<int> use_stream{-1};
    if (use_stream < 0) { const char* e = getenv("FDCAP_NN_STREAM"); use_stream = e ? atoi(e) : -2; }
    // (the streaming kernel's work list holds 16-bit ids 4 k + quarter: scenes up to 16384 chunks = 8.4 M points; beyond, the staged kernel)
    if (culled && T.frags != nullptr && use_stream && nn_use_mfma(nq, T.n) && seedpt != nullptr && seed == idx &&
        (T.n + MF_CH - 1) / MF_CH <= 16384) {
        if (seed_missing)                                     // first launch of a fit: cheap seeds (+ their coordinates) instead of a full scan
            hipLaunchKernelGGL(nn_seed_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, q, nq, T, idx, seedpt);
        // seed aliases idx: every workgroup reads its seeds before it writes its own results, and no other workgroup touches them
        if (seedpt_written) *seedpt_written = true;
        {
            // (queries per group / 32, waves per group): 42 / 41 four waves, 22 / 21 two, 12 / 11 one
            int nqv = 1, wpg;
            if (use_stream < 0) {                             // enough waves to fill 1024 SIMDs x 4 twice over, no more (the
                const int g32 = (nq + 31) / 32;               // per-group setup is repeated by every wave of the group)
                // r6 sweep (tools/launch_times.py at 64 .. 224 frames x 500 queries, us per launch; waves per group 1 / 2 / 4):
                //   2000 groups 25.8 / 23.7 / 22.7, 2500: 25.1 / 24.2 / 25.5, 3000: 25.8 / 26.8 / 28.7, 3500: 24.1 / 26.9 / 30.1
                wpg = g32 >= 2816 ? 1 : g32 >= 2304 ? 2 : 4;  // (3072 / 4 until r6) re-measured with quarter work items: 128 / 256 / 512 / 768 frames: 11: 0.036 / 0.033 / 0.057 / 0.068 ms, 21: 0.028 / 0.034 / 0.059 / 0.074, 41: 0.024 / 0.036 / 0.063 / 0.084
            } else {
                nqv = (use_stream % 10 == 2) ? 2 : 1;
                wpg = (use_stream / 10 == 4) ? 4 : (use_stream / 10 == 2) ? 2 : 1;
            }
            static std::atomic<int> wpb1{-1};                             // FDCAP_NN_WPB=4 (A/B): four-wave workgroups for one-wave groups too
            if (wpb1 < 0) { const char* e = getenv("FDCAP_NN_WPB"); wpb1 = (e && atoi(e) == 4) ? 0 : 1; }
            const int groups = (nq + 32 * nqv - 1) / (32 * nqv);
            const int wpb = (wpg == 1 && nqv == 1 && wpb1) ? 1 : 4;
            const int nwg = (groups * wpg + wpb - 1) / wpb;
            const dim3 grid((nwg + 7) / 8 * 8);
            NNCache nc = cache ? *cache : NNCache{nullptr, nullptr, nullptr, 0.f};
            const bool ordered = ord != nullptr && ord->on && nc.hdr != nullptr && ord->every > 0 && nqv == 1 && wpg == 1 && wpb == 1;
            nc.order_mode = !ordered ? 0 : (ord->sorted_groups == groups ? 2 + ord->cur : 1);
#define FDC_ST4(NQV, WPGV) hipLaunchKernelGGL((nn_stream4_kernel<NQV, WPGV>), grid, dim3(256), 0, st, q, nq, T, seed, seedpt, dist, idx, nc)
            note_form(wpg == 4 ? "nn_stream4_kernel(4 waves per group)" : wpg == 2 ? "nn_stream4_kernel(2 waves per group)" :
                      wpb == 4 ? "nn_stream4_kernel(1 wave per group, 4-wave workgroups)" : "nn_stream4_kernel<1,1,1>");
            if (nqv == 2 && wpg == 4) FDC_ST4(2, 4); else if (nqv == 2 && wpg == 2) FDC_ST4(2, 2); else if (nqv == 2) FDC_ST4(2, 1);
            else if (wpg == 4) FDC_ST4(1, 4); else if (wpg == 2) FDC_ST4(1, 2); else if (wpb == 4) FDC_ST4(1, 1);
            else hipLaunchKernelGGL((nn_stream4_kernel<1, 1, 1>), grid, dim3(64), 0, st, q, nq, T, seed, seedpt, dist, idx, nc);
#undef FDC_ST4
            if (ordered && (ord->sorted_groups != groups || ++ord->age >= ord->every)) {
                const bool had = nc.order_mode >= 2;
                const int nxt = had ? 1 - ord->cur : 0;
                hipLaunchKernelGGL(nn_order_kernel, dim3(1), dim3(1024), 0, st, (const int*)(nc.hdr + 4 * groups), groups,
                                   had ? (const int*)(nc.hdr + (5 + ord->cur) * groups) : (const int*)nullptr, nc.hdr + (5 + nxt) * groups);
                ord->cur = nxt;
                ord->sorted_groups = groups;
                ord->age = 0;
            }
        }
        return hipGetLastError();
    }
    const int fq_ = forced_nq;
    const int NQsel = fq_ ? fq_ : (culled ? 2 : 4);
    note_form(nn_use_mfma(nq, T.n) ? "nn_mfma_kernel" : "nn_direct_kernel");
    if (nn_use_mfma(nq, T.n) && NQsel == 1)
        hipLaunchKernelGGL((nn_mfma_kernel<1>), dim3(nn_grid_blocks((nq + 127) / 128, nsplit)), dim3(256), 0, st, q, nq, T, nsplit, seed, pd, pi);
    else if (nn_use_mfma(nq, T.n) && NQsel == 2)
        hipLaunchKernelGGL((nn_mfma_kernel<2>), dim3(nn_grid_blocks((nq + 255) / 256, nsplit)), dim3(256), 0, st, q, nq, T, nsplit, seed, pd, pi);
    else if (nn_use_mfma(nq, T.n))
        hipLaunchKernelGGL((nn_mfma_kernel<4>), dim3(nn_grid_blocks((nq + 511) / 512, nsplit)), dim3(256), 0, st, q, nq, T, nsplit, seed, pd, pi);
    else
        hipLaunchKernelGGL((nn_direct_kernel<2>), dim3(nn_grid_blocks((nq + 511) / 512, nsplit)), dim3(256), 0, st, q, nq, T, nsplit, pd, pi);
    hipLaunchKernelGGL(nn_combine_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, pd, pi, nsplit, nq, dist, idx);
    return hipGetLastError();
}

}  // namespace fdc
