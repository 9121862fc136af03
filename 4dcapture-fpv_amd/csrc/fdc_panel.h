// Skinny-M fp32 GEMMs of the optimiser loop on v_mfma_f32_16x16x4_f32 (exact fp32: bitwise a k-ordered fmaf chain).
//
// Every dense contraction of an iteration multiplies a few hundred to a thousand frame rows by a STATIC matrix (VPoser
// weights, the contact set's pose/shape blend directions): M ~ 130..1028, N and K 32..1500.  At these sizes a classic
// LDS-tiled GEMM spends its time in per-launch latency (operand staging, barriers, split-K reductions), not on the matrix
// pipe.  Here the static operand is re-laid out ONCE, on the host, in MFMA fragment order ("panel"): the fragment of tile
// t / super-step s is 1 KiB that lane l reads as one float4 -- so a wave streams its B operand straight from L2 into
// registers with fully coalesced 16-byte loads, no LDS, no barrier; only the 16-row A block lives in LDS (k-blocked, so the
// A fragment of a super-step is one conflict-free ds_read_b128).  One super-step = 16 k = 4 MFMAs per (A read, B load).
//
//   panel_gemm_kernel          C[M,N] = A[M,K] x B            pose/shape blend offsets of the contact vertices (:280-283, K8)
//                                                             and their data gradient (B = the transposed panel)
//   vposer_fwd_fused_kernel    the three decoder layers (:270) in ONE launch: a workgroup owns 16 rows and one QUARTER of the
//                              hidden columns; layer 1 (K = 32) is recomputed by the four quarter-workgroups, layer 2 is
//                              split by output column, the output layer by K -- its four partial sums are added, in a fixed
//                              order, by the consumer (pose_fwd_kernel), so no workgroup ever waits for another
//   vposer_bwd_fused_kernel    the data-gradient chain dO -> dH2 -> dH1 -> d latent in one launch, split the other way round
//                              (every step is linear in its input once the LeakyReLU masks are applied, so K-slices of a
//                              step can run independently down to four partial latent gradients, summed by the Adam kernel)
// Replaces six gemm_f32 launches per iteration (55 us at 1028 rows) by two (~8 us each) and the two blend products'
// 25 + 24 us by ~12 + 12.
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <vector>

#include "fdc_frame.h"
#include "fdc_loss.h"

namespace fdc {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a property of ONE device's code object: one bit per device ordinal, set after
// the attribute took (two threads racing set it twice: harmless)
inline bool fdc_attr_needed(const std::atomic<uint64_t>& mask) { int d = 0; (void)hipGetDevice(&d); return !(mask.load() & (1ull << (d & 63))); }
inline void fdc_attr_done(std::atomic<uint64_t>& mask) { int d = 0; (void)hipGetDevice(&d); mask.fetch_or(1ull << (d & 63)); }

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// Row count from which the clip-sized kernel forms are selected (two row blocks per fragment stream in the blend products, the
// fused contact forward): measured break-even 336 (r6 sweep: contact forward as two launches / fused 11.7 / 12.6 us at 320 rows,
// 13.7 / 12.8 at 352; 384 until then).  FDCAP_CLIP_FORMS_MIN_ROWS overrides it -- tests run the reference's own
// 300-frame fixtures (the reference hard-codes 300, :41-42) through the forms BASELINE configs 2 / 3 / 5 select.
inline int clip_forms_min_rows() {
    static std::atomic<int> v{-1};
    if (v < 0) { const char* e = getenv("FDCAP_CLIP_FORMS_MIN_ROWS"); v = e ? std::max(32, atoi(e)) : 336; }
    return v;
}

// ... and for the K-split data gradient (panel_gemm3_rb2k): from 257 rows.  r6 (tools/launch_times.py): the one-tile-stream form
// (panel_gemm3_ksw) keeps one workgroup per CU, so from 17 row blocks x 16 column groups = 272 workgroups on it runs in two rounds --
// 16.6 us at 272-352 rows against 9.7 at 256 -- while the two-row-block K-split form takes 10.7-11.0 us there: a 300-frame clip
// (the reference's real clip length) 42.4 -> 40.0 ms per fit.  At 256 rows and below the one-round form wins (9.7 vs 11.2 us).
inline int clip_kgrad_min_rows() {
    static std::atomic<int> v{-1};
    if (v < 0) { const char* e = getenv("FDCAP_CLIP_FORMS_MIN_ROWS"); v = e ? std::max(32, atoi(e)) : 257; }
    return v;
}

// B operand in fragment order: f[(tile * nss + s) * 64 + lane] = { B(16 s + 4 (lane >> 4) + m, 16 tile + (lane & 15)) }, m = 0..3
struct PanelB {
    const float4* f = nullptr;
    int ntile = 0;      // ceil(N / 16)
    int nss = 0;        // ceil(K / 16) super-steps
};

// host: B(k, n) = src[k * sk + n * sn], zero padded to whole tiles / super-steps
static inline void panel_pack(const float* src, long sk, long sn, int K, int N, std::vector<float>& out, int* ntile, int* nss) {
    const int nt = (N + 15) / 16, ns = (K + 15) / 16;
    out.assign((size_t)nt * ns * 256, 0.f);
    for (int t = 0; t < nt; ++t)
        for (int s = 0; s < ns; ++s)
            for (int l = 0; l < 64; ++l) {
                const int n = 16 * t + (l & 15);
                if (n >= N) continue;
                for (int m = 0; m < 4; ++m) {
                    const int k = 16 * s + 4 * (l >> 4) + m;
                    if (k < K) out[(((size_t)t * ns + s) * 64 + l) * 4 + m] = src[(long)k * sk + (long)n * sn];
                }
            }
    *ntile = nt;
    *nss = ns;
}

// device form of panel_pack (r6: a vertex set's panels are built where they are used -- the host loops cost 0.3 s per full-mesh set);
// one thread per fragment lane: out[(tile * nss + s) * 64 + lane]
__global__ __launch_bounds__(256) void panel_pack_kernel(const float* __restrict__ src, long sk, long sn, int K, int N, int nt, int ns,
                                                         float4* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)nt * ns * 64) return;
    const int l = (int)(idx & 63), s = (int)((idx >> 6) % ns), t = (int)((idx >> 6) / ns);
    const int n = 16 * t + (l & 15);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < N)
        for (int m = 0; m < 4; ++m) {
            const int k = 16 * s + 4 * (l >> 4) + m;
            if (k < K) v[m] = src[(long)k * sk + (long)n * sn];
        }
    out[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

// LDS image of a 16-row A block, k-blocked: element (row i, column k) -- a super-step's fragment is lane-linear
__host__ __device__ __forceinline__ int pn_lds_index(int i, int k) { return (((k >> 2) << 4) + i) * 4 + (k & 3); }

// stage rows [r0, r0 + 16) x columns [k0, k0 + kn) of A (row-major, lda) as a k-blocked image of `kpad` columns
// (kpad % 16 == 0; rows >= rmax and columns >= kn are zero).  NT threads.
template <int NT>
__device__ __forceinline__ void panel_stage(float* __restrict__ sA, const float* __restrict__ A, int lda, int r0, int rmax,
                                            int k0, int kn, int kpad, int tid) {
    const bool vec = ((lda & 3) == 0) && ((k0 & 3) == 0) && ((((size_t)A) & 15) == 0);
    if (vec) {
        for (int e = tid; e < (kpad >> 2) * 16; e += NT) {          // e = kblock * 16 + row
            const int kb = e >> 4, i = e & 15, row = r0 + i, k = 4 * kb;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < rmax && k < kn) {
                const float* p = A + (size_t)row * lda + k0 + k;
                if (k + 3 < kn) v = *(const float4*)p;
                else { v.x = p[0]; if (k + 1 < kn) v.y = p[1]; if (k + 2 < kn) v.z = p[2]; }
            }
            *(float4*)(sA + (size_t)e * 4) = v;
        }
    } else {
        for (int e = tid; e < kpad * 16; e += NT) {                  // e = (kblock * 16 + row) * 4 + m
            const int kb = e >> 6, i = (e >> 2) & 15, k = 4 * kb + (e & 3), row = r0 + i;
            sA[e] = (row < rmax && k < kn) ? A[(size_t)row * lda + k0 + k] : 0.f;
        }
    }
}

// acc[t] += A[16 x 16 nss] * B_t for T tiles that share the A block.  sA: image at the first super-step; bf[t]: tile t's
// fragments at the first super-step (without the lane offset).  PF fragments per tile in flight.
//
// The fragment ring: loads in the main loop are unconditional and pinned by a scheduling barrier right after their issue.
// Loads under wave-uniform branches make hipcc wait vmcnt(0) in every super-step; unpinned, its scheduler sinks each load
// to just before its use -- either way one exposed L2 round trip per 4 MFMAs (the blend product ran 27 us instead of ~12).
// (Inline-asm loads with hand-counted s_waitcnt were tried and are wrong here: the ring is loop-carried, and the register
// copies hipcc inserts for it read a destination before the wait.)  -DFDC_PN_PLAIN: no pins (A/B reference; same bits).
// (-DFDC_PN_ABL=1 / 2 / 3: timing ablations -- no fragment loads / no MFMAs / no LDS reads; results are wrong)
#ifndef FDC_PN_ABL
#define FDC_PN_ABL 0
#endif
// (addresses are per-lane VGPR pairs: a scalar base would be written by SALU / v_readfirstlane right before the load reads
// it -- a 5-wait-state hazard hipcc does not pad inside an asm statement; measured: wrong fragments)
struct PnStream { const char* p; };                                   // this lane's byte address in a tile's fragment stream
__device__ __forceinline__ PnStream pn_stream(const float4* tile_base, int lane) { return PnStream{(const char*)(tile_base + lane)}; }
__device__ __forceinline__ f32x4_t pn_load_b(PnStream st, int step) {
#if FDC_PN_ABL == 1
    return f32x4_t{(float)step, 1.f, 2.f, (float)(size_t)st.p};
#else
    return *(const f32x4_t*)(st.p + (size_t)step * 1024);
#endif
}
// nothing may be scheduled across this point: the fragment load issued before it stays PF super-steps ahead of its use
__device__ __forceinline__ void pn_pin() {
#ifndef FDC_PN_PLAIN
    __builtin_amdgcn_sched_barrier(0);
#endif
}
__device__ __forceinline__ float4 pn_load_a(const float* __restrict__ p) {
#if FDC_PN_ABL == 3
    return make_float4(__builtin_amdgcn_readfirstlane((int)(size_t)p) * 1.f, 1.f, 2.f, 3.f);
#else
    return *(const float4*)p;
#endif
}
__device__ __forceinline__ f32x4_t pn_mfma(float a, float b, f32x4_t c) {
#if FDC_PN_ABL == 2
    c[0] += a * b; return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}
// The static fragment is the MFMA's A operand (its 16 rows = output columns n), the LDS block its B operand (its 16 columns =
// frame rows), so a lane's four results are FOUR CONSECUTIVE OUTPUT COLUMNS of one frame: lane (j = lane & 15, g = lane >> 4),
// register r  <->  out[frame j][16 tile + 4 g + r].  Epilogues are one 16-byte store per tile -- to global memory and into
// the next layer's k-blocked LDS image alike (the other way round each lane holds four frames of one column: four
// scattered 4-byte LDS writes with an 8-way bank conflict).
__device__ __forceinline__ f32x4_t pn_step(float4 a, f32x4_t b, f32x4_t acc) {
    acc = pn_mfma(b[0], a.x, acc);
    acc = pn_mfma(b[1], a.y, acc);
    acc = pn_mfma(b[2], a.z, acc);
    return pn_mfma(b[3], a.w, acc);
}
__device__ __forceinline__ float4 pn_f4(f32x4_t v) { return make_float4(v[0], v[1], v[2], v[3]); }

// Two register sets, each PF super-steps deep: set A is consumed while set B is being filled and vice versa, so no fragment
// register is redefined while its previous value is live and the ring needs no copies at the loop's back edge (with one
// set hipcc rotates the registers there and has to drain the queue once per turn).  The first set is loaded by
// panel_prefetch -- issued BEFORE the barrier / staging that precedes the product, so the first L2 round trip is hidden.
template <int T, int PF>
struct PnRing { f32x4_t bA[T][PF]; PnStream st[T]; };

template <int T, int PF>
__device__ __forceinline__ void panel_prefetch(PnRing<T, PF>& rg, const float4* const* bf, int nss, int lane) {
    const int last = nss - 1;
#pragma unroll
    for (int t = 0; t < T; ++t) rg.st[t] = pn_stream(bf[t], lane);
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int t = 0; t < T; ++t) rg.bA[t][p] = pn_load_b(rg.st[t], min(p, last));
}

template <int T, int PF>
__device__ __forceinline__ void panel_mma(const float* __restrict__ sA, PnRing<T, PF>& rg, int nss, f32x4_t* acc, int lane) {
    f32x4_t (&bA)[T][PF] = rg.bA;
    f32x4_t bB[T][PF];
    PnStream (&st)[T] = rg.st;
    const int last = nss - 1;
    float4 a = pn_load_a(sA + lane * 4);                             // the A fragment runs one super-step ahead of its MFMAs
    int s = 0;
    for (; s + 2 * PF <= nss; s += 2 * PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const float4 a_next = pn_load_a(sA + (size_t)(s + p + 1) * 256 + lane * 4);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                bB[t][p] = pn_load_b(st[t], s + PF + p);
                pn_pin();
                acc[t] = pn_step(a, bA[t][p], acc[t]);
            }
            a = a_next;
        }
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const float4 a_next = pn_load_a(sA + (size_t)min(s + PF + p + 1, last) * 256 + lane * 4);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                bA[t][p] = pn_load_b(st[t], min(s + 2 * PF + p, last));
                pn_pin();
                acc[t] = pn_step(a, bB[t][p], acc[t]);
            }
            a = a_next;
        }
    }
    // tail: fewer than 2 PF super-steps; set A holds the first PF of them
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        if (s + p < nss) {
            const float4 a_next = pn_load_a(sA + (size_t)min(s + p + 1, last) * 256 + lane * 4);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                bB[t][p] = pn_load_b(st[t], min(s + PF + p, last));
                acc[t] = pn_step(a, bA[t][p], acc[t]);
            }
            a = a_next;
        }
    }
#pragma unroll
    for (int p = 0; p < PF - 1; ++p) {
        if (s + PF + p < nss) {
            const float4 a_next = pn_load_a(sA + (size_t)min(s + PF + p + 1, last) * 256 + lane * 4);
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] = pn_step(a, bB[t][p], acc[t]);
            a = a_next;
        }
    }
}

// RB row blocks of 16 share one B fragment stream (wide outputs: B traffic / RB).  sA: RB images, `img` floats apart.
template <int RB, int PF>
__device__ __forceinline__ void panel_mma_rows(const float* __restrict__ sA, int img, PnRing<1, PF>& rg, int nss,
                                               f32x4_t* acc, int lane) {
    f32x4_t (&bA)[PF] = rg.bA[0];
    f32x4_t bB[PF];
    const PnStream st = rg.st[0];
    const int last = nss - 1;
    float4 a[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) a[rb] = pn_load_a(sA + (size_t)rb * img + lane * 4);
    auto step = [&](int cur, f32x4_t bb) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const float4 a_next = pn_load_a(sA + (size_t)rb * img + (size_t)min(cur + 1, last) * 256 + lane * 4);
            acc[rb] = pn_step(a[rb], bb, acc[rb]);
            a[rb] = a_next;
        }
    };
    int s = 0;
    for (; s + 2 * PF <= nss; s += 2 * PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            bB[p] = pn_load_b(st, s + PF + p);
            pn_pin();
            step(s + p, bA[p]);
        }
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            bA[p] = pn_load_b(st, min(s + 2 * PF + p, last));
            pn_pin();
            step(s + PF + p, bB[p]);
        }
    }
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        if (s + p < nss) {
            bB[p] = pn_load_b(st, min(s + PF + p, last));
            step(s + p, bA[p]);
        }
    }
#pragma unroll
    for (int p = 0; p < PF - 1; ++p)
        if (s + PF + p < nss) step(s + PF + p, bB[p]);
}

#ifdef FDC_PN_TIMING
// instrumentation build only (never shipped): per-workgroup s_memtime stamps [block][8]
__device__ unsigned long long g_pn_times[8192 * 8];
#define PN_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_pn_times[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PN_STAMP(i)
#endif

// Workgroup -> (row block, column block), XCD-aware.  The dispatcher deals consecutive workgroups to the 8 XCDs round-robin
// (b % 8), and each XCD has its own 4 MiB L2: with a plain 2-D grid every XCD pulls nearly all of A AND all of B through the
// fabric in every launch (measured: 5.6 k cycles just to stage a 31 KB A block, fragment loads at Infinity-Cache latency).
// Here XCD x owns the rectangle (row group x / xc, column group x % xc): its slice of the STATIC operand B stays resident in
// its L2 from one optimiser iteration to the next, and A crosses the fabric xc times instead of 8.
struct PnMap { int nrb, ncb, xc, rpg, cpg, rfast; };   // row / column blocks; column groups; blocks per group; slot order
static inline PnMap panel_map(int nrb, int ncb, size_t a_bytes, size_t b_bytes) {
    PnMap best{nrb, ncb, 1, (nrb + 7) / 8, ncb, 0};
    double best_cost = 1e300;
    for (int xc = 1; xc <= 8; xc *= 2) {
        const int xr = 8 / xc;
        const int rpg = (nrb + xr - 1) / xr, cpg = (ncb + xc - 1) / xc;
        // fabric bytes per launch: A once per column group; B once per row group unless an XCD's slice is small enough to
        // stay in its L2 between launches; + a penalty for idle slots of ragged groups
        double cost = (double)xc * a_bytes + ((b_bytes / xc <= (size_t)(3u << 19)) ? 0.0 : (double)xr * b_bytes);
        cost *= (double)(8 * rpg * cpg) / (double)(nrb * ncb);
        if (cost < best_cost) { best_cost = cost; best = PnMap{nrb, ncb, xc, rpg, cpg, 0}; }
    }
    // an XCD's slice of B does not fit its L2 (wide outputs): walk the row blocks of one column block first, so the
    // workgroups resident at any time share a few column blocks of B and each slice crosses the fabric once
    best.rfast = b_bytes / best.xc > (size_t)(3u << 19);
    return best;
}

// C[M, N] = A[M, K] x B.  Workgroup = 8 waves = 16 RB rows x 128 columns (a 16-column tile per wave); K in slabs of `kslab`
// columns (multiple of 16) so any K fits the LDS.  1-D grid of 8 * rpg * cpg workgroups (PnMap).
// Dynamic LDS: RB * kslab * 16 floats.
template <int RB>
__global__ __launch_bounds__(512) void panel_gemm_kernel(const float* __restrict__ A, int lda, int M, int K, PanelB B, int kslab,
                                                         float* __restrict__ C, int ldc, int N, PnMap mp) {
    extern __shared__ __attribute__((aligned(16))) float pn_lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int r_in = mp.rfast ? slot % mp.rpg : slot / mp.cpg, c_in = mp.rfast ? slot / mp.rpg : slot % mp.cpg;
    const int rbk = (xcd / mp.xc) * mp.rpg + r_in, cbk = (xcd % mp.xc) * mp.cpg + c_in;
    if (r_in >= mp.rpg || c_in >= mp.cpg || rbk >= mp.nrb || cbk >= mp.ncb) return;  // ragged groups (whole workgroup)
    const int tile = cbk * 8 + wave;
    const int m0 = rbk * (16 * RB);
    const int img = kslab * 16;
    PN_STAMP(0);
    f32x4_t acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool active = tile < B.ntile;
    const float4* const tf = B.f + (size_t)(active ? tile : 0) * B.nss * 64;
    for (int k0 = 0; k0 < K; k0 += kslab) {
        const int kn = min(kslab, K - k0), kpad = (kn + 15) & ~15;
        PnRing<1, 4> rg;
        const float4* bfp = tf + (size_t)(k0 >> 4) * 64;
        panel_prefetch<1, 4>(rg, &bfp, kpad >> 4, lane);            // in flight while the A block is staged
        if (k0) __syncthreads();
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) panel_stage<512>(pn_lds + (size_t)rb * img, A, lda, m0 + 16 * rb, M, k0, kn, kpad, tid);
        __syncthreads();
        PN_STAMP(1);
        if (active) panel_mma_rows<RB, 4>(pn_lds, img, rg, kpad >> 4, acc, lane);
    }
    PN_STAMP(2);
    const int n4 = tile * 16 + 4 * g;
    if (active && n4 < N) {
        const bool vec = ((ldc & 3) == 0) && ((((size_t)C) & 15) == 0) && n4 + 3 < N;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int m = m0 + 16 * rb + j;
            if (m < M) {
                float* dst = C + (size_t)m * ldc + n4;
                if (vec) *(float4*)dst = pn_f4(acc[rb]);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n4 + r < N) dst[r] = acc[rb][r];
                }
            }
        }
    }
    PN_STAMP(3);
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 products on the bf16 matrix cores: three-way split.
// An fp32 number is EXACTLY the sum of three bf16 numbers, a = h + m + l (8 + 8 + 8 mantissa bits: h = bf16(a),
// m = bf16(a - h), l = bf16(a - h - m); both subtractions are exact).  A product a b is then nine bf16 x bf16 partial
// products (each exact in fp32); the three smallest (m l, l m, l l) are below 2^-23 |a b| together -- one fp32 rounding
// of the product itself -- and are dropped; the other six go through v_mfma_f32_16x16x32_bf16 with fp32 accumulation,
// smallest first.  Six bf16 MFMAs cover 32 columns of K in ~100 cycles per SIMD where the fp32 form needs eight
// v_mfma_f32_16x16x4_f32 = 256 cycles, and the loop's products are bound by exactly that pipe (93 % of its issue slots).
// Error vs the exact product: <= 2^-22 relative per term, the class of the fp32 fmaf chain's own rounding
// (tests/test_gpu_panel.py holds both forms to the same bar against fp64).
typedef __bf16 pn_bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4u_t __attribute__((ext_vector_type(4), aligned(4)));      // 16-byte store at 4-byte alignment (3 V is odd)
struct PanelB3 {
    const uint4* f = nullptr;       // f[((tile * nst + s) * 3 + plane) * 64 + lane] = 8 bf16: B(32 s + 8 (lane >> 4) + e, 16 tile + (lane & 15))
    int ntile = 0, nst = 0;         // ceil(N / 16), ceil(K / 32)
    const float* isc = nullptr;     // PnH2 panels: isc[16 tile + c] = 1 / (the power-of-two scale of output column c of the tile); two planes there
};
static inline unsigned pn3_bf_host(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return u >> 16; }   // RNE
static inline float pn3_bff_host(unsigned h) { unsigned u = h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline void panel_pack3(const float* src, long sk, long sn, int K, int N, std::vector<unsigned>& out, int* ntile, int* nst) {
    const int nt = (N + 15) / 16, ns = (K + 31) / 32;
    out.assign((size_t)nt * ns * 3 * 64 * 4, 0u);
    for (int t = 0; t < nt; ++t)
        for (int s = 0; s < ns; ++s)
            for (int l = 0; l < 64; ++l) {
                const int n = 16 * t + (l & 15);
                if (n >= N) continue;
                for (int e = 0; e < 8; ++e) {
                    const int k = 32 * s + 8 * (l >> 4) + e;
                    if (k >= K) continue;
                    const float a = src[(long)k * sk + (long)n * sn];
                    const unsigned h = pn3_bf_host(a);
                    const float r1 = a - pn3_bff_host(h);
                    const unsigned m = pn3_bf_host(r1);
                    const unsigned lo = pn3_bf_host(r1 - pn3_bff_host(m));
                    const unsigned part[3] = {h, m, lo};
                    for (int pl = 0; pl < 3; ++pl) {
                        unsigned& w = out[((((size_t)t * ns + s) * 3 + pl) * 64 + l) * 4 + (e >> 1)];
                        w |= part[pl] << (16 * (e & 1));
                    }
                }
            }
    *ntile = nt;
    *nst = ns;
}

__device__ __forceinline__ unsigned pn3_bf(float f) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)f); }   // RNE
__device__ __forceinline__ float pn3_bff(unsigned h) { return __uint_as_float(h << 16); }

// The three-way split of TWO values at once (late r4): v_cvt_pk_bf16_f32 converts a pair and packs it -- low half the first value --
// so a plane's dword needs no shift / or to assemble; the parts are the same round-to-nearest-even conversions of the same
// residuals as pn3_bf / pn3_bff element by element (46 instead of ~68 VALU instructions per eight values).
typedef __bf16 pn_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pn_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pn3_pk2(float a, float b) {
    const pn_f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, pn_bf16x2));
}
__device__ __forceinline__ void pn3_split2(float a, float b, unsigned& H, unsigned& M, unsigned& L) {
    H = pn3_pk2(a, b);
    const float ra = a - __uint_as_float(H << 16), rb = b - __uint_as_float(H & 0xffff0000u);
    M = pn3_pk2(ra, rb);
    L = pn3_pk2(ra - __uint_as_float(M << 16), rb - __uint_as_float(M & 0xffff0000u));
}

// stage rows [r0, r0 + 16) x columns [k0, k0 + kn) of A as three bf16 planes, each [kpad / 8 chunks][16 rows] x 16 bytes
// (kpad % 32 == 0; rows >= rmax and columns >= kn are zero).  NT threads; one (chunk, row) item = 8 columns.
template <int NT>
__device__ __forceinline__ void panel_stage3(uint4* __restrict__ sA3, const float* __restrict__ A, int lda, int r0, int rmax, int k0,
                                             int kn, int kpad, int tid) {
    const int nch = kpad >> 3;
    const bool vec = ((lda & 3) == 0) && ((k0 & 3) == 0) && ((((size_t)A) & 15) == 0);
    for (int it = tid; it < nch * 16; it += NT) {
        const int ch = it >> 4, i = it & 15, row = r0 + i, k = 8 * ch;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (row < rmax && k < kn) {
            const float* p = A + (size_t)row * lda + k0 + k;
            if (vec && k + 7 < kn) {
                const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (k + e < kn) v[e] = p[e];
            }
        }
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) pn3_split2(v[2 * e], v[2 * e + 1], h[e], m[e], l[e]);
        sA3[(size_t)0 * nch * 16 + it] = make_uint4(h[0], h[1], h[2], h[3]);
        sA3[(size_t)1 * nch * 16 + it] = make_uint4(m[0], m[1], m[2], m[3]);
        sA3[(size_t)2 * nch * 16 + it] = make_uint4(l[0], l[1], l[2], l[3]);
    }
}

__device__ __forceinline__ f32x4_t pn3_step(const uint4* a /*[3] planes h m l of the LDS block*/, const uint4* b /*[3] of the fragment*/, f32x4_t acc) {
    // static fragment = MFMA A operand (rows = output columns), LDS block = B operand (columns = frames): see pn_step
#define PN3_MMA(wp, ap) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pn_bf16x8, b[wp]), __builtin_bit_cast(pn_bf16x8, a[ap]), acc, 0, 0, 0)
#ifdef FDC_PN_EXP2      /* timing only: what two planes / three products per step would cost */
    PN3_MMA(1, 0); PN3_MMA(0, 1); PN3_MMA(0, 0);
#else
    PN3_MMA(2, 0); PN3_MMA(0, 2); PN3_MMA(1, 1); PN3_MMA(1, 0); PN3_MMA(0, 1); PN3_MMA(0, 0);
#endif
#undef PN3_MMA
    return acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// Operand formats of the split products.  PnB3: the three bf16 planes above (six products per step; -DFDC_PN_H2=0 builds everything on it).
// PnH2 (late r5; the blend products): TWO fp16 planes and THREE products per step.
//   a s = h + 2^-11 l,  h = fp16(a s),  l = fp16((a s - h) 2^11)   (both conversions round to nearest even; a s - h is exact)
// with s a power of two per frame row of the dynamic operand (its largest |a| lands in [2^13, 2^14): found while the block is
// staged, one extra barrier) and per output column of the static one (host).  |a - (h + 2^-11 l) / s| <= 2^-22 |a| for every element
// within 2^-27 of its row's largest (below that: 2^-50 of the largest, absolutely).  Products h h -> one accumulator, l h + h l ->
// a second one, joined as (acc_h + 2^-11 acc_l) / (s_a s_b) in the epilogue (exact scalings, one rounding); l l (<= 2^-22 |a b|)
// is dropped.  Error <= 3 x 2^-22 |a||b| per term: above PnB3's 2^-22, far below the K 2^-24 bound of an fp32 fmaf chain, and
// tests/test_gpu_panel.py holds it to the same 1e-6 sum|a||b| bar against fp64.  What it buys: half the MFMA issue, two thirds
// of the fragment bytes and of the LDS reads per 32 columns of K -- all three of the resources these kernels sit on at once.
#ifndef FDC_PN_H2
#define FDC_PN_H2 1
#endif
typedef _Float16 pn_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pn_f16x2 __attribute__((ext_vector_type(2)));
static inline unsigned pn2_f16_host(float f) {                  // fp32 -> fp16 bits, round to nearest even (no libgcc soft-float call)
    unsigned u; memcpy(&u, &f, 4);
    const unsigned sign = (u >> 16) & 0x8000u, ex = (u >> 23) & 255u;
    unsigned man = u & 0x7FFFFFu;
    if (ex == 255u) return sign | 0x7C00u | (man ? 0x200u : 0u);
    const int e = (int)ex - 127 + 15;
    if (e >= 31) return sign | 0x7C00u;
    if (e <= 0) {                                               // subnormal or zero
        if (e < -10) return sign;
        man |= 0x800000u;
        const int sh = 14 - e;                                  // 14 .. 24
        const unsigned q = man >> sh, rem = man & ((1u << sh) - 1u), half = 1u << (sh - 1);
        return sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u));
    }
    const unsigned q = ((unsigned)e << 10) | (man >> 13), rem = man & 0x1FFFu;
    return sign | (q + ((rem > 0x1000u || (rem == 0x1000u && (q & 1u))) ? 1u : 0u));   // a carry into the exponent is the right answer
}
static inline float pn2_f16f_host(unsigned h) {
    const unsigned sign = (h & 0x8000u) << 16, ex = (h >> 10) & 31u, man = h & 0x3FFu;
    float f;
    if (ex == 0) { f = ldexpf((float)man, -24); unsigned u; memcpy(&u, &f, 4); u |= sign; memcpy(&f, &u, 4); return f; }
    const unsigned u = sign | (ex == 31u ? 0x7F800000u : ((ex + 112u) << 23)) | (man << 13);
    memcpy(&f, &u, 4);
    return f;
}
struct PnB3 {
    static constexpr int NP = 3, SC_U4 = 0;                     // planes; uint4 of per-row scales (+ scratch) behind an image's planes
    typedef f32x4_t Acc;
    static __device__ __forceinline__ Acc zero() { return f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ Acc step(const uint4* a, const uint4* b, Acc acc) { return pn3_step(a, b, acc); }
    static __device__ __forceinline__ f32x4_t value(const Acc& a, float, const f32x4_t&) { return a; }
    static __device__ __forceinline__ float row_isc(const uint4*, int, int) { return 1.f; }
    static __device__ __forceinline__ f32x4_t tile_isc(const PanelB3&, int, int) { return f32x4_t{1.f, 1.f, 1.f, 1.f}; }
    template <int NT, int RB, int MAXIT>
    static __device__ __forceinline__ void stage(uint4* __restrict__ lds, int img, const float* __restrict__ A, int lda, int m0, int M, int k0,
                                                 int kn, int kpad, int tid) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) panel_stage3<NT>(lds + (size_t)rb * img, A, lda, m0 + 16 * rb, M, k0, kn, kpad, tid);
    }
    static __device__ __forceinline__ void pow2(float, float& sc, float& isc) { sc = 1.f; isc = 1.f; }
    static inline void pack(const float* src, long sk, long sn, int K, int N, std::vector<unsigned>& out, std::vector<float>& isc, int* ntile,
                            int* nst) {
        panel_pack3(src, sk, sn, K, N, out, ntile, nst);
        isc.assign((size_t)*ntile * 16, 1.f);
    }
};
__device__ __forceinline__ unsigned pn2_pk2(float a, float b) {
    const pn_f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, pn_f16x2));
}
struct PnH2 {
    static constexpr int NP = 2, SC_U4 = 4 + 4 * 12;            // 16 inverse row scales + 16 partial maxima per wave (<= 12 waves)
    struct Acc { f32x4_t h, l; };
    static __device__ __forceinline__ Acc zero() { return Acc{f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}}; }
    static __device__ __forceinline__ Acc step(const uint4* a /*[2] planes h l of the LDS block*/, const uint4* b /*[2] of the fragment*/, Acc acc) {
#define PN2_MMA(dst, wp, ap) dst = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pn_f16x8, b[wp]), __builtin_bit_cast(pn_f16x8, a[ap]), dst, 0, 0, 0)
        PN2_MMA(acc.l, 1, 0); PN2_MMA(acc.h, 0, 0); PN2_MMA(acc.l, 0, 1);
#undef PN2_MMA
        return acc;
    }
    static __device__ __forceinline__ f32x4_t value(const Acc& a, float row_isc, const f32x4_t& col_isc) {
        f32x4_t v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaf(a.l[r], 0x1p-11f, a.h[r]) * row_isc * col_isc[r];   // (powers of two: exact; one after the other: their product alone may leave fp32's range)
        return v;
    }
    static __device__ __forceinline__ float row_isc(const uint4* image, int pstride, int j) { return ((const float*)(image + (size_t)NP * pstride))[j]; }
    // the power of two that takes `bound` (>= every |a| of a row; 0: an all-zero row) into [2^13, 2^14), and its inverse
    static __device__ __forceinline__ void pow2(float bound, float& sc, float& isc) {
        const int se = bound == 0.f ? 127 : min(max(267 - (int)((__float_as_uint(bound) >> 23) & 255u), 4), 250);
        sc = __uint_as_float((unsigned)se << 23);
        isc = __uint_as_float((unsigned)(254 - se) << 23);
    }
    static __device__ __forceinline__ f32x4_t tile_isc(const PanelB3& B, int tile, int g);    // the lane's four columns 16 tile + 4 g ..
    // rows [m0, m0 + 16 RB) x columns [k0, k0 + kn) of A as RB images of two fp16 planes, each [kpad / 8 chunks][16 rows] x 16 bytes,
    // + the 16 inverse row scales behind each image's planes.  NT threads (a multiple of 64); contains ONE __syncthreads (every thread
    // of the workgroup must call); the caller's barrier after it publishes the images.  A thread's items all belong to row (tid & 15):
    // up to MAXIT items per image stay in registers between the maximum and the conversion, longer blocks are read twice.
    template <int NT, int RB, int MAXIT>
    static __device__ __forceinline__ void stage(uint4* __restrict__ lds, int img, const float* __restrict__ A, int lda, int m0, int M, int k0,
                                                 int kn, int kpad, int tid) {
        const int nch = kpad >> 3, nit = nch * 16, pstride = nit, wave = tid >> 6, lane = tid & 63, i = tid & 15;
        const bool vec = ((lda & 3) == 0) && ((k0 & 3) == 0) && ((((size_t)A) & 15) == 0);
        auto load8 = [&](int rb, int it, float* v) {
            const int ch = it >> 4, row = m0 + 16 * rb + i, k = 8 * ch;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
            if (row < M && k < kn) {
                const float* p = A + (size_t)row * lda + k0 + k;
                if (vec && k + 7 < kn) {
                    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (k + e < kn) v[e] = p[e];
                }
            }
        };
        auto amax8 = [&](const float* v, float m) {
#pragma unroll
            for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
            return m;
        };
        auto put = [&](int rb, int it, const float* v, float sc) {
            unsigned h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = v[2 * e] * sc, x1 = v[2 * e + 1] * sc;
                h[e] = pn2_pk2(x0, x1);
                const pn_f16x2 hh = __builtin_bit_cast(pn_f16x2, h[e]);
                l[e] = pn2_pk2((x0 - (float)hh[0]) * 2048.f, (x1 - (float)hh[1]) * 2048.f);
            }
            uint4* const im = lds + (size_t)rb * img;
            im[(size_t)0 * pstride + it] = make_uint4(h[0], h[1], h[2], h[3]);
            im[(size_t)1 * pstride + it] = make_uint4(l[0], l[1], l[2], l[3]);
        };
        const bool keep = nit <= MAXIT * NT;
        float kv[RB][MAXIT][8], mx[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) mx[rb] = 0.f;
        if (keep) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int n = 0; n < MAXIT; ++n) {
                    const int it = tid + n * NT;
                    if (it < nit) { load8(rb, it, kv[rb][n]); mx[rb] = amax8(kv[rb][n], mx[rb]); }
                }
        } else {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                for (int it = tid; it < nit; it += NT) { float v[8]; load8(rb, it, v); mx[rb] = amax8(v, mx[rb]); }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            float m = mx[rb];
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            float* const sc = (float*)(lds + (size_t)rb * img + (size_t)NP * pstride);
            if (lane < 16) sc[16 + wave * 16 + lane] = m;
        }
        __syncthreads();
        float scl[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            float* const sc = (float*)(lds + (size_t)rb * img + (size_t)NP * pstride);
            float m = 0.f;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) m = fmaxf(m, sc[16 + w * 16 + i]);
            float isc;                                          // the row's largest |a| -> [2^13, 2^14); an all-zero row keeps 1, the ends are clamped
            pow2(m, scl[rb], isc);
            if (tid < 16) sc[tid] = isc;
        }
        if (keep) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int n = 0; n < MAXIT; ++n) {
                    const int it = tid + n * NT;
                    if (it < nit) put(rb, it, kv[rb][n], scl[rb]);
                }
        } else {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                for (int it = tid; it < nit; it += NT) { float v[8]; load8(rb, it, v); put(rb, it, v, scl[rb]); }
        }
    }
    static inline void pack(const float* src, long sk, long sn, int K, int N, std::vector<unsigned>& out, std::vector<float>& isc, int* ntile,
                            int* nst) {
        const int nt = (N + 15) / 16, ns = (K + 31) / 32;
        out.assign((size_t)nt * ns * NP * 64 * 4, 0u);
        isc.assign((size_t)nt * 16, 1.f);
        std::vector<float> scl((size_t)nt * 16, 1.f);
        for (int n = 0; n < N; ++n) {
            float mx = 0.f;
            for (int k = 0; k < K; ++k) mx = std::max(mx, fabsf(src[(long)k * sk + (long)n * sn]));
            int ex = 0;
            if (mx > 0.f && std::isfinite(mx)) { frexpf(mx, &ex); ex = std::min(std::max(14 - ex, -100), 100); }   // mx in [2^(ex-1), 2^ex) -> [2^13, 2^14)
            scl[(size_t)n] = ldexpf(1.f, ex);
            isc[(size_t)n] = ldexpf(1.f, -ex);
        }
        for (int t = 0; t < nt; ++t) {
            for (int s = 0; s < ns; ++s)
                for (int l = 0; l < 64; ++l) {
                    const int n = 16 * t + (l & 15);
                    if (n >= N) continue;
                    for (int e = 0; e < 8; ++e) {
                        const int k = 32 * s + 8 * (l >> 4) + e;
                        if (k >= K) continue;
                        const float x = src[(long)k * sk + (long)n * sn] * scl[(size_t)n];
                        const unsigned h = pn2_f16_host(x);
                        const unsigned lo = pn2_f16_host((x - pn2_f16f_host(h)) * 2048.f);
                        const unsigned part[2] = {h, lo};
                        for (int pl = 0; pl < NP; ++pl) {
                            unsigned& w = out[((((size_t)t * ns + s) * NP + pl) * 64 + l) * 4 + (e >> 1)];
                            w |= part[pl] << (16 * (e & 1));
                        }
                    }
                }
        }
        *ntile = nt;
        *nst = ns;
    }
};
// device form of PnH2::pack (r6), same arithmetic: the column scales (largest |B(., n)| -> [2^13, 2^14), all-zero / non-finite
// columns keep 1) by one of two kernels -- B row-major (sn == 1): a thread per column; B column-major (sk == 1): a workgroup per
// column -- then one thread per fragment lane writes its 16 bytes of each plane.  scl / isc: [16 ntile], padding columns 1.
__device__ __forceinline__ void pnh2_col_scale(float mx, float* scl, float* isc) {
    int ex = 0;
    if (mx > 0.f && mx < INFINITY) { (void)frexpf(mx, &ex); ex = min(max(14 - ex, -100), 100); }
    *scl = ldexpf(1.f, ex);
    *isc = ldexpf(1.f, -ex);
}
__global__ __launch_bounds__(256) void pnh2_colscale_rowmajor_kernel(const float* __restrict__ src, long sk, int K, int N, int npad,
                                                                     float* __restrict__ scl, float* __restrict__ isc) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= npad) return;
    float mx = 0.f;
    if (n < N) for (int k = 0; k < K; ++k) mx = fmaxf(mx, fabsf(src[(long)k * sk + n]));
    pnh2_col_scale(mx, scl + n, isc + n);
}
__global__ __launch_bounds__(256) void pnh2_colscale_colmajor_kernel(const float* __restrict__ src, long sn, int K, int N, int npad,
                                                                     float* __restrict__ scl, float* __restrict__ isc) {
    __shared__ float part[4];
    const int n = blockIdx.x;
    float mx = 0.f;
    if (n < N) for (int k = threadIdx.x; k < K; k += 256) mx = fmaxf(mx, fabsf(src[(long)n * sn + k]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) pnh2_col_scale(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3])), scl + n, isc + n);
}
__global__ __launch_bounds__(256) void pnh2_pack_kernel(const float* __restrict__ src, long sk, long sn, int K, int N, int nt, int ns,
                                                        const float* __restrict__ scl, uint4* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)nt * ns * 64) return;
    const int l = (int)(idx & 63), s = (int)((idx >> 6) % ns), t = (int)((idx >> 6) / ns);
    const int n = 16 * t + (l & 15);
    unsigned h[4] = {0u, 0u, 0u, 0u}, lo[4] = {0u, 0u, 0u, 0u};
    if (n < N) {
        const float sc = scl[n];
        for (int e = 0; e < 8; ++e) {
            const int k = 32 * s + 8 * (l >> 4) + e;
            if (k >= K) continue;
            const float x = src[(long)k * sk + (long)n * sn] * sc;
            const _Float16 hh = (_Float16)x;
            const _Float16 ll = (_Float16)((x - (float)hh) * 2048.f);
            h[e >> 1] |= (unsigned)__builtin_bit_cast(unsigned short, hh) << (16 * (e & 1));
            lo[e >> 1] |= (unsigned)__builtin_bit_cast(unsigned short, ll) << (16 * (e & 1));
        }
    }
    uint4* const o = out + (((size_t)t * ns + s) * 2) * 64 + l;
    o[0] = make_uint4(h[0], h[1], h[2], h[3]);
    o[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

#if FDC_PN_H2
typedef PnH2 PnF;                                               // the format of the panel_gemm3_* family
#else
typedef PnB3 PnF;
#endif
__device__ __forceinline__ f32x4_t PnH2::tile_isc(const PanelB3& B, int tile, int g) { return ((const f32x4_t*)B.isc)[(size_t)min(tile, B.ntile - 1) * 4 + g]; }
constexpr int PNF = PnF::NP;
// uint4 per 16-row image of kpad columns; bytes of RB of them
constexpr __host__ __device__ int pnf_img_u4(int kpad) { return PnF::NP * (kpad >> 3) * 16 + PnF::SC_U4; }
constexpr __host__ __device__ size_t pnf_lds_bytes(int kpad, int rb) { return (size_t)rb * pnf_img_u4(kpad) * 16; }

// acc += A[16 x 32 nst] * B (split operands, PnF::step per 32-column step).  sA3: image at the first step (plane stride
// `pstride` uint4); bf: the tile's fragments at the first step (no lane offset).  Two register sets of PF steps x planes, loads
// pinned (see panel_mma).
template <int PF>
struct PnRing3 { uint4 bA[PF][PNF]; const uint4* st; };
template <int PF>
__device__ __forceinline__ void panel3_prefetch(PnRing3<PF>& rg, const uint4* bf, int nst, int lane) {
    rg.st = bf + lane;
    const int last = nst - 1;
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int pl = 0; pl < PNF; ++pl) rg.bA[p][pl] = rg.st[((size_t)min(p, last) * PNF + pl) * 64];
}
// RB row blocks (images `img` uint4 apart) share one fragment stream
template <int RB, int PF>
__device__ __forceinline__ void panel3_mma(const uint4* __restrict__ sA3, int pstride, int img, PnRing3<PF>& rg, int nst, PnF::Acc* acc, int lane) {
    uint4 (&bA)[PF][PNF] = rg.bA;
    uint4 bB[PF][PNF];
    const uint4* const st = rg.st;
    const int last = nst - 1;
    auto load_a = [&](uint4 (*a)[PNF], int step) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int pl = 0; pl < PNF; ++pl) a[rb][pl] = sA3[(size_t)rb * img + (size_t)pl * pstride + (size_t)step * 64 + lane];
    };
    auto load_b = [&](uint4* b, int step) {
#pragma unroll
        for (int pl = 0; pl < PNF; ++pl) b[pl] = st[((size_t)step * PNF + pl) * 64];
    };
    uint4 a[RB][PNF];
    load_a(a, 0);
    auto step = [&](const uint4* b, int next) {
        uint4 an[RB][PNF];
        load_a(an, next);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            acc[rb] = PnF::step(a[rb], b, acc[rb]);
#pragma unroll
            for (int pl = 0; pl < PNF; ++pl) a[rb][pl] = an[rb][pl];
        }
    };
    int s = 0;
    for (; s + 2 * PF <= nst; s += 2 * PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            load_b(bB[p], s + PF + p);
            pn_pin();
            step(bA[p], s + p + 1);
        }
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            load_b(bA[p], min(s + 2 * PF + p, last));
            pn_pin();
            step(bB[p], min(s + PF + p + 1, last));
        }
    }
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        if (s + p < nst) {
            load_b(bB[p], min(s + PF + p, last));
            step(bA[p], min(s + p + 1, last));
        }
    }
#pragma unroll
    for (int p = 0; p < PF - 1; ++p)
        if (s + PF + p < nst) step(bB[p], min(s + PF + p + 1, last));
}
// a lane's four results of one tile: columns n4 .. n4 + 3 of row m
__device__ __forceinline__ void pnf_store4(float* __restrict__ C, int ldc, int N, int m, int n4, const f32x4_t v) {
    float* dst = C + (size_t)m * ldc + n4;
    if (n4 + 3 < N) *(f32x4u_t*)dst = f32x4u_t{v[0], v[1], v[2], v[3]};
    else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n4 + r < N) dst[r] = v[r];
    }
}

// C[M, N] = A[M, K] x B on the split (same workgroup shape and XCD-aware map as panel_gemm_kernel<1>; K in one slab).
// Dynamic LDS: pnf_lds_bytes(kpad, 1) per row block.
// NW waves per workgroup = NW column tiles (r5; 8 until then).  A 128-frame shard's products ran on 96 (forward) and 32 (data
// gradient) workgroups of eight waves -- the gradient's 32 each pulled 1.1 MB of fragments through one CU's return path, 11.5 us
// for a product of 0.2 GFLOP; with fewer waves per workgroup there are enough workgroups for the chip (launcher: panel_gemm3).
template <int NW>
__global__ __launch_bounds__(64 * NW) void panel_gemm3_kernel(const float* __restrict__ A, int lda, int M, int K, PanelB3 B,
                                                          float* __restrict__ C, int ldc, int N, PnMap mp) {
    extern __shared__ __attribute__((aligned(16))) uint4 pn3_lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int r_in = mp.rfast ? slot % mp.rpg : slot / mp.cpg, c_in = mp.rfast ? slot / mp.rpg : slot % mp.cpg;
    const int rbk = (xcd / mp.xc) * mp.rpg + r_in, cbk = (xcd % mp.xc) * mp.cpg + c_in;
    if (r_in >= mp.rpg || c_in >= mp.cpg || rbk >= mp.nrb || cbk >= mp.ncb) return;  // ragged groups (whole workgroup)
    const int tile = cbk * NW + wave, m0 = rbk * 16;
    const bool active = tile < B.ntile;
    const int kpad = (K + 31) & ~31, nst = kpad >> 5, pstride = (kpad >> 3) * 16;
    PnRing3<2> rg;
    panel3_prefetch<2>(rg, B.f + (size_t)(active ? tile : 0) * B.nst * PNF * 64, nst, lane);      // in flight while the A block is staged
    const f32x4_t ts = PnF::tile_isc(B, tile, g);                // (with the first fragments: at the epilogue it would be a cold round trip of its own)
    PnF::stage<64 * NW, 1, (NW == 8 ? 6 : 4)>(pn3_lds, 0, A, lda, m0, M, 0, K, kpad, tid);
    __syncthreads();
    PnF::Acc acc = PnF::zero();
    if (active) panel3_mma<1, 2>(pn3_lds, pstride, 0, rg, nst, &acc, lane);
    const int n4 = tile * 16 + 4 * g, m = m0 + j;
    if (active && n4 < N && m < M) pnf_store4(C, ldc, N, m, n4, PnF::value(acc, PnF::row_isc(pn3_lds, pstride, j), ts));
}
// Two row blocks per fragment stream, for clip-sized M.  s_memtime in panel_gemm3_kernel (1024 rows): the MFMA phase of the
// K = 1500 data gradient takes 20.8 k cycles for 282 MFMAs per wave -- 74 cycles each where the pipe needs 16 -- and its eight
// waves receive 8 x 141 KB of fragments in that time: 54 bytes per clock, against the 64 a CU's vector-memory path returns.
// The bound is that return path (L1 hit or not: streaming one tile to all eight waves changed nothing), so what helps is
// fewer fragment bytes per MFMA: every fragment multiplies TWO 16-row blocks.  Twelve waves = twelve column tiles per
// workgroup, 32 rows, both images in LDS (K <= 768); column group = blockIdx & 7 = XCD, so each
// XCD streams its own eighth of the static operand.  Grid: 8 x ceil(M / 32) workgroups (256 at 1024 rows: one per CU).
__global__ __launch_bounds__(768) void panel_gemm3_rb2_kernel(const float* __restrict__ A, int lda, int M, int K, PanelB3 B,
                                                              float* __restrict__ C, int ldc, int N) {
    extern __shared__ __attribute__((aligned(16))) uint4 pn3_lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7, m0 = (int)(blockIdx.x >> 3) * 32;
    const int ncb = (B.ntile + 11) / 12, cpg = (ncb + 7) / 8;            // column blocks of 12 tiles; per XCD
    const int kpad = (K + 31) & ~31, nst = kpad >> 5, pstride = (kpad >> 3) * 16, img = pnf_img_u4(kpad);
    if (xcd * cpg >= ncb) return;
    PnRing3<2> rg;
    panel3_prefetch<2>(rg, B.f + (size_t)min((xcd * cpg) * 12 + wave, B.ntile - 1) * B.nst * PNF * 64, nst, lane);
    f32x4_t ts = PnF::tile_isc(B, (xcd * cpg) * 12 + wave, g);   // (requested with the fragments: see panel_gemm3_kernel)
    PnF::stage<768, 2, 2>(pn3_lds, img, A, lda, m0, M, 0, K, kpad, tid);
    __syncthreads();
    const float rs[2] = {PnF::row_isc(pn3_lds, pstride, j), PnF::row_isc(pn3_lds + img, pstride, j)};
    for (int cb = xcd * cpg; cb < min(ncb, (xcd + 1) * cpg); ++cb) {
        const int tile = cb * 12 + wave;
        PnF::Acc acc[2] = {PnF::zero(), PnF::zero()};
        if (tile < B.ntile) panel3_mma<2, 2>(pn3_lds, pstride, img, rg, nst, acc, lane);
        const f32x4_t tsc = ts;
        if (cb + 1 < min(ncb, (xcd + 1) * cpg)) {
            panel3_prefetch<2>(rg, B.f + (size_t)min((cb + 1) * 12 + wave, B.ntile - 1) * B.nst * PNF * 64, nst, lane);
            ts = PnF::tile_isc(B, (cb + 1) * 12 + wave, g);
        }
        const int n4 = tile * 16 + 4 * g;
        if (tile < B.ntile && n4 < N) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int m = m0 + 16 * rb + j;
                if (m < M) pnf_store4(C, ldc, N, m, n4, PnF::value(acc[rb], rs[rb], tsc));
            }
        }
    }
}

// ... and for long K (the data gradient, K = 1500: two images of all of K do not fit) each workgroup takes HALF of K for two
// row blocks and leaves a partial product: C = Cpart[0] + Cpart[1], added by the consumer.  XCD = blockIdx & 7 =
// (K half, column block of 8 tiles): each XCD streams its own eighth of the static operand; slot = blockIdx >> 3 = row pair.
// N <= 32 tiles (four column blocks), kpad / 2 rounded up to a step <= 768 columns per half.  Grid 8 x ceil(M / 32).
__global__ __launch_bounds__(512) void panel_gemm3_rb2k_kernel(const float* __restrict__ A, int lda, int M, int K, PanelB3 B,
                                                               float* __restrict__ Cpart, size_t part_stride, int ldc, int N) {
    extern __shared__ __attribute__((aligned(16))) uint4 pn3_lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7, half = xcd >> 2, cb = xcd & 3, m0 = (int)(blockIdx.x >> 3) * 32;
    const int nst_all = (K + 31) >> 5, s0 = half ? (nst_all + 1) / 2 : 0, s1 = half ? nst_all : (nst_all + 1) / 2;
    const int nst = s1 - s0, kpad = 32 * nst, k0 = 32 * s0, kn = min(K, 32 * s1) - k0;
    const int pstride = (kpad >> 3) * 16, img = pnf_img_u4(kpad);
    const int tile = cb * 8 + wave;
    const bool active = tile < B.ntile && nst > 0;
    PnRing3<2> rg;
    panel3_prefetch<2>(rg, B.f + ((size_t)(active ? tile : 0) * B.nst + s0) * PNF * 64, max(nst, 1), lane);
    const f32x4_t ts = PnF::tile_isc(B, tile, g);               // (requested with the fragments: see panel_gemm3_kernel)
    PnF::stage<512, 2, 3>(pn3_lds, img, A, lda, m0, M, k0, kn, kpad, tid);
    __syncthreads();
    PnF::Acc acc[2] = {PnF::zero(), PnF::zero()};
    if (active) panel3_mma<2, 2>(pn3_lds, pstride, img, rg, nst, acc, lane);
    const int n4 = tile * 16 + 4 * g;
    if (tile < B.ntile && n4 < N) {
        float* const C = Cpart + (size_t)half * part_stride;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int m = m0 + 16 * rb + j;
            if (m < M) pnf_store4(C, ldc, N, m, n4, PnF::value(acc[rb], PnF::row_isc(pn3_lds + (size_t)rb * img, pstride, j), ts));
        }
    }
}
// may the product take that form?  (the caller then provides the second partial buffer and adds the two)
static inline bool panel_gemm3_rb2k_ok(int M, int K, const PanelB3& B) {
    static std::atomic<int> rb2{-1};
    if (rb2 < 0) { const char* e = getenv("FDCAP_PN_RB2"); rb2 = e ? atoi(e) : 1; }     // 0 off, 1 both forms, 2 forward only, 3 K-split only
    const int nst_all = (K + 31) >> 5;
    return (rb2 == 1 || rb2 == 3) && M >= clip_kgrad_min_rows() && B.ntile <= 32 && nst_all >= 2 && 32 * ((nst_all + 1) / 2) <= 768;
}
static inline hipError_t panel_gemm3_rb2k(const float* A, int lda, int M, int K, const PanelB3& B, float* Cpart, size_t part_stride,
                                          int ldc, int N, hipStream_t st) {
    const int kh = 32 * ((((K + 31) >> 5) + 1) / 2);
    note_form("panel_gemm3_rb2k_kernel"); hipLaunchKernelGGL(panel_gemm3_rb2k_kernel, dim3(8 * ((M + 31) / 32)), dim3(512), pnf_lds_bytes(kh, 2), st, A, lda, M, K, B, Cpart,
                       part_stride, ldc, N);
    return hipGetLastError();
}

// wide outputs on the split: the column-walking form of panel_gemm_wide_kernel (A staged once per workgroup, all the
// column blocks of the XCD's share walked; next block's fragments requested before the stores)
template <int RB>
__global__ __launch_bounds__(512) void panel_gemm3_wide_kernel(const float* __restrict__ A, int lda, int M, int K, PanelB3 B,
                                                               float* __restrict__ C, int ldc, int N, int cs) {
    // cs (r5): an XCD's share of the column blocks is cut into cs parts, one workgroup each per row block -- grid 8 x row blocks x cs.
    // With cs = 1 a 512-row product (BASELINE config 5) ran on 128 workgroups and a 128-row shard of it on 32: every workgroup
    // walks an eighth of the 93 MB operand through ONE CU's 64 B / clock, 205-214 us whatever M is (profiles/r5_c5_*).
    extern __shared__ __attribute__((aligned(16))) uint4 pn3_lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int nrb = (M + 16 * RB - 1) / (16 * RB);
    const int xcd = blockIdx.x & 7, slot = (int)(blockIdx.x >> 3), m0 = (slot % nrb) * (16 * RB), part = slot / nrb;
    const int ncb = (B.ntile + 7) / 8, cpg = (ncb + 7) / 8;
    const int xb0 = xcd * cpg, xb1 = min(ncb, xb0 + cpg), per = (max(xb1 - xb0, 0) + cs - 1) / cs;
    const int cb0 = xb0 + part * per, cb1 = min(xb1, cb0 + per);
    const int kpad = (K + 31) & ~31, nst = kpad >> 5, pstride = (kpad >> 3) * 16, img = pnf_img_u4(kpad);
    if (cb0 >= cb1) return;
    PnRing3<2> rg;
    panel3_prefetch<2>(rg, B.f + (size_t)min(cb0 * 8 + wave, B.ntile - 1) * B.nst * PNF * 64, nst, lane);
    f32x4_t ts = PnF::tile_isc(B, cb0 * 8 + wave, g);           // (requested with the fragments: see panel_gemm3_kernel)
    PnF::stage<512, RB, 3>(pn3_lds, img, A, lda, m0, M, 0, K, kpad, tid);
    __syncthreads();
    float rs[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) rs[rb] = PnF::row_isc(pn3_lds + (size_t)rb * img, pstride, j);
    for (int cb = cb0; cb < cb1; ++cb) {
        const int tile = cb * 8 + wave;
        PnF::Acc acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = PnF::zero();
        if (tile < B.ntile) panel3_mma<RB, 2>(pn3_lds, pstride, img, rg, nst, acc, lane);
        const f32x4_t tsc = ts;
        if (cb + 1 < cb1) {
            panel3_prefetch<2>(rg, B.f + (size_t)min((cb + 1) * 8 + wave, B.ntile - 1) * B.nst * PNF * 64, nst, lane);
            ts = PnF::tile_isc(B, (cb + 1) * 8 + wave, g);
        }
        const int n4 = tile * 16 + 4 * g;
        if (tile < B.ntile && n4 < N) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int m = m0 + 16 * rb + j;
                if (m < M) pnf_store4(C, ldc, N, m, n4, PnF::value(acc[rb], rs[rb], tsc));
            }
        }
    }
}

// longest K whose single-slab LDS image (pnf_lds_bytes(kpad, 1)) fits the 160 KB of a gfx950 CU
constexpr int PN3_MAX_K = FDC_PN_H2 ? 2528 : 1696;
static_assert(pnf_lds_bytes(PN3_MAX_K, 1) <= 160 * 1024, "one image per CU");
static inline bool panel_gemm3_fits(int K) { return ((K + 31) & ~31) <= PN3_MAX_K; }

static inline hipError_t panel_gemm3(const float* A, int lda, int M, int K, const PanelB3& B, float* C, int ldc, int N, hipStream_t st) {
    if (M <= 0 || N <= 0) return hipSuccess;
    const int kpad = (K + 31) & ~31;
    if (kpad > PN3_MAX_K) return hipErrorInvalidValue;       // callers fall back to panel_gemm (K slabs) above this
    if ((size_t)B.ntile * B.nst * PNF * 1024 > (size_t)(16u << 20) && M >= 32 && kpad <= 768) {
        // one workgroup per CU (the 98 KB image leaves room for one): as many column parts as it takes to reach 256 workgroups
        const int nrb = (M + 31) / 32, ncb = (B.ntile + 7) / 8, cpg = (ncb + 7) / 8;
        static std::atomic<int> cs_env{-1};                          // FDCAP_PN_WIDE_CS=1 (A/B): the r2-r4 form, one part
        if (cs_env < 0) { const char* e = getenv("FDCAP_PN_WIDE_CS"); cs_env = e ? atoi(e) : 0; }
        // (r5: two column tiles per wave, as in the K-loop product, measured no faster here: 0.186 vs 0.179 ms at 1024 rows, equal at 512)
        static std::atomic<int> rb_env{-1};                          // FDCAP_PN_WIDE_RB=4 (A/B, r6): 64 rows per workgroup -- half the fragment bytes per MFMA
        if (rb_env < 0) { const char* e = getenv("FDCAP_PN_WIDE_RB"); rb_env = e ? atoi(e) : 2; }
        if (rb_env == 4 && M >= 64) {
            const int nrb4 = (M + 63) / 64;
            const int cs4 = cs_env > 0 ? (int)cs_env : std::max(1, std::min(cpg, (256 + 8 * nrb4 - 1) / (8 * nrb4)));
            static std::atomic<uint64_t> attr4{0};
            if (fdc_attr_needed(attr4)) {
                hipError_t e = hipFuncSetAttribute((const void*)panel_gemm3_wide_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pnf_lds_bytes(kpad, 4));
                if (e != hipSuccess) return e;
                fdc_attr_done(attr4);
            }
            note_form("panel_gemm3_wide_kernel<4>");
            hipLaunchKernelGGL(panel_gemm3_wide_kernel<4>, dim3(8 * nrb4 * cs4), dim3(512), pnf_lds_bytes(kpad, 4), st, A, lda, M, K, B, C, ldc, N, cs4);
            return hipGetLastError();
        }
        const int cs = cs_env > 0 ? (int)cs_env : std::max(1, std::min(cpg, (256 + 8 * nrb - 1) / (8 * nrb)));
        note_form("panel_gemm3_wide_kernel"); hipLaunchKernelGGL(panel_gemm3_wide_kernel<2>, dim3(8 * nrb * cs), dim3(512), pnf_lds_bytes(kpad, 2), st, A, lda, M, K, B, C, ldc, N, cs);
        return hipGetLastError();
    }
    static std::atomic<int> rb2{-1};                                  // FDCAP_PN_RB2=0 (A/B): one row block per fragment stream everywhere
    if (rb2 < 0) { const char* e = getenv("FDCAP_PN_RB2"); rb2 = e ? atoi(e) : 1; }
    if ((rb2 == 1 || rb2 == 2) && M >= clip_forms_min_rows() && kpad <= 768 && B.ntile >= 48) {   // (measured: 256 rows 50.2 vs 49.8 ms per step, 384 rows 56.7 vs 57.8)
        // (r5: six waves x two tiles over the same 32 x 192 block -- half the LDS bytes per MFMA, the lever that took the K-loop
        //  product from 192 to 139 us -- is SLOWER here: 14.4 vs 12.8 us; with K = 512 the twelve waves' latency hiding is worth more)
        note_form("panel_gemm3_rb2_kernel"); hipLaunchKernelGGL(panel_gemm3_rb2_kernel, dim3(8 * ((M + 31) / 32)), dim3(768), pnf_lds_bytes(kpad, 2), st, A, lda, M, K, B, C, ldc, N);
        return hipGetLastError();
    }
    // waves per workgroup: eight while that gives >= 192 workgroups, else four, else two (FDCAP_PN_NW pins it: A/B)
    static std::atomic<int> nw_env{-1};
    if (nw_env < 0) { const char* e = getenv("FDCAP_PN_NW"); nw_env = e ? atoi(e) : 0; }
    const int nrb = (M + 15) / 16;
    int nw = 8;                                            // (measured at 128 rows: forward 7.8 -> 6.6 us with four waves; two waves stage too slowly)
    if (nrb * ((B.ntile + 7) / 8) < 128 && nrb * ((B.ntile + 3) / 4) >= 128) nw = 4;
    if (nw_env == 8 || nw_env == 4 || nw_env == 2) nw = nw_env;
    const PnMap mp = panel_map(nrb, (B.ntile + nw - 1) / nw, (size_t)M * K * 4, (size_t)B.ntile * B.nst * PNF * 1024);
    const dim3 grid(8 * mp.rpg * mp.cpg);
    const size_t lds = pnf_lds_bytes(kpad, 1);
    note_form("panel_gemm3_kernel");
    if (nw == 8) hipLaunchKernelGGL(panel_gemm3_kernel<8>, grid, dim3(512), lds, st, A, lda, M, K, B, C, ldc, N, mp);
    else if (nw == 4) hipLaunchKernelGGL(panel_gemm3_kernel<4>, grid, dim3(256), lds, st, A, lda, M, K, B, C, ldc, N, mp);
    else hipLaunchKernelGGL(panel_gemm3_kernel<2>, grid, dim3(128), lds, st, A, lda, M, K, B, C, ldc, N, mp);
    return hipGetLastError();
}

// Wide outputs (the full-mesh blend: N = 3 V = 31 425 columns, B = 62 MB): a workgroup keeps its 16 RB rows of A in LDS and
// walks ALL the column blocks of its XCD's share of B, so A is staged once per workgroup instead of once per (row, column)
// block (measured with one column block per workgroup: staging 14.7 k of a 45 k-cycle lifetime, exposed because the 127 KB
// image leaves room for one workgroup per CU) and the workgroups of an XCD stream the same column blocks at the same
// time: each slice of B crosses the fabric once.  The next column block's first fragments are requested before the
// current block's stores.  K must fit one slab.  Grid: 8 * ceil(M / (16 RB)) workgroups, blockIdx & 7 = XCD = column share.
template <int RB>
__global__ __launch_bounds__(512) void panel_gemm_wide_kernel(const float* __restrict__ A, int lda, int M, int K, PanelB B,
                                                              float* __restrict__ C, int ldc, int N) {
    extern __shared__ __attribute__((aligned(16))) float pn_lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7, m0 = (int)(blockIdx.x >> 3) * (16 * RB);
    const int ncb = (B.ntile + 7) / 8, cpg = (ncb + 7) / 8;
    const int cb0 = xcd * cpg, cb1 = min(ncb, cb0 + cpg);
    const int kpad = (K + 15) & ~15, nss = kpad >> 4, img = kpad * 16;
    if (cb0 >= cb1) return;
    PN_STAMP(0);
    PnRing<1, 4> rg;
    {
        const int t0 = min(cb0 * 8 + wave, B.ntile - 1);
        const float4* bfp = B.f + (size_t)t0 * B.nss * 64;
        panel_prefetch<1, 4>(rg, &bfp, nss, lane);
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) panel_stage<512>(pn_lds + (size_t)rb * img, A, lda, m0 + 16 * rb, M, 0, K, kpad, tid);
    __syncthreads();
    PN_STAMP(1);
    for (int cb = cb0; cb < cb1; ++cb) {
        if (cb == cb0 + 1) PN_STAMP(2);
        const int tile = cb * 8 + wave;
        f32x4_t acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (tile < B.ntile) panel_mma_rows<RB, 4>(pn_lds, img, rg, nss, acc, lane);
        if (cb + 1 < cb1) {                                          // next block's first fragments: in flight during the stores
            const int tn = min((cb + 1) * 8 + wave, B.ntile - 1);
            const float4* bfp = B.f + (size_t)tn * B.nss * 64;
            panel_prefetch<1, 4>(rg, &bfp, nss, lane);
        }
        const int n4 = tile * 16 + 4 * g;
        if (tile < B.ntile && n4 < N) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int m = m0 + 16 * rb + j;
                if (m < M) {
                    float* dst = C + (size_t)m * ldc + n4;
                    if (n4 + 3 < N) *(f32x4u_t*)dst = f32x4u_t{acc[rb][0], acc[rb][1], acc[rb][2], acc[rb][3]};
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n4 + r < N) dst[r] = acc[rb][r];
                    }
                }
            }
        }
    }
    PN_STAMP(3);
}

static inline hipError_t panel_gemm(const float* A, int lda, int M, int K, const PanelB& B, float* C, int ldc, int N, hipStream_t st) {
    if (M <= 0 || N <= 0) return hipSuccess;
    // one row block per workgroup while an XCD's slice of the operand stays inside its L2 (the loop's products: a 3 MB panel);
    // wide outputs (full-mesh blend, N = 31 425) take the column-walking form with two row blocks per fragment stream
    const size_t b_bytes = (size_t)B.ntile * B.nss * 1024;
    const int kpad = (K + 15) & ~15;
    const int ncb = (B.ntile + 7) / 8;
    if (b_bytes > (size_t)(24u << 20) && M >= 32 && kpad <= 1024) {
        hipLaunchKernelGGL(panel_gemm_wide_kernel<2>, dim3(8 * ((M + 31) / 32)), dim3(512), (size_t)2 * kpad * 16 * sizeof(float), st, A, lda, M, K,
                           B, C, ldc, N);
    } else {
        const int kslab = kpad <= 1536 ? kpad : 1024;               // <= 96 KiB
        const PnMap mp = panel_map((M + 15) / 16, ncb, (size_t)M * K * 4, b_bytes);
        hipLaunchKernelGGL(panel_gemm_kernel<1>, dim3(8 * mp.rpg * mp.cpg), dim3(512), (size_t)kslab * 16 * sizeof(float), st, A, lda, M, K, B,
                           kslab, C, ldc, N, mp);
    }
    return hipGetLastError();
}

// T tiles share one 16-row LDS block (the fused VPoser kernels' layers)
template <int T, int PF, class F = PnB3>
struct PnRing3T { uint4 bA[T][PF][F::NP]; const uint4* st[T]; };
template <int T, int PF, class F = PnB3>
__device__ __forceinline__ void panel3_prefetch_t(PnRing3T<T, PF, F>& rg, const uint4* const* bf, int nst, int lane) {
    const int last = nst - 1;
#pragma unroll
    for (int t = 0; t < T; ++t) rg.st[t] = bf[t] + lane;
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int pl = 0; pl < F::NP; ++pl) rg.bA[t][p][pl] = rg.st[t][((size_t)min(p, last) * F::NP + pl) * 64];
}
template <int T, int PF, class F = PnB3>
__device__ __forceinline__ void panel3_mma_t(const uint4* __restrict__ sA3, int pstride, PnRing3T<T, PF, F>& rg, int nst, typename F::Acc* acc, int lane) {
    constexpr int NP = F::NP;
    uint4 (&bA)[T][PF][NP] = rg.bA;
    uint4 bB[T][PF][NP];
    const int last = nst - 1;
    auto load_a = [&](uint4* a, int step) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) a[pl] = sA3[(size_t)pl * pstride + (size_t)step * 64 + lane];
    };
    auto keep = [&](uint4* a, const uint4* an) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) a[pl] = an[pl];
    };
    uint4 a[NP];
    load_a(a, 0);
    int s = 0;
    for (; s + 2 * PF <= nst; s += 2 * PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            uint4 an[NP];
            load_a(an, s + p + 1);
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) bB[t][p][pl] = rg.st[t][((size_t)(s + PF + p) * NP + pl) * 64];
                pn_pin();
                acc[t] = F::step(a, bA[t][p], acc[t]);
            }
            keep(a, an);
        }
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            uint4 an[NP];
            load_a(an, min(s + PF + p + 1, last));
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) bA[t][p][pl] = rg.st[t][((size_t)min(s + 2 * PF + p, last) * NP + pl) * 64];
                pn_pin();
                acc[t] = F::step(a, bB[t][p], acc[t]);
            }
            keep(a, an);
        }
    }
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        if (s + p < nst) {
            uint4 an[NP];
            load_a(an, min(s + p + 1, last));
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) bB[t][p][pl] = rg.st[t][((size_t)min(s + PF + p, last) * NP + pl) * 64];
                acc[t] = F::step(a, bA[t][p], acc[t]);
            }
            keep(a, an);
        }
    }
#pragma unroll
    for (int p = 0; p < PF - 1; ++p) {
        if (s + PF + p < nst) {
            uint4 an[NP];
            load_a(an, min(s + PF + p + 1, last));
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] = F::step(a, bB[t][p], acc[t]);
            keep(a, an);
        }
    }
}
// RB row blocks x T tiles per wave (r5, the K-loop product): a step's RB LDS fragments are read ONCE for T tiles and its T static
// fragments once for RB row blocks -- RB T products per (RB LDS reads + T fragment loads).  acc[rb * T + t].  Format PnF.
template <int RB, int T, int PF>
__device__ __forceinline__ void panel3_mma_rt(const uint4* __restrict__ sA3, int pstride, int img, PnRing3T<T, PF, PnF>& rg, int nst, PnF::Acc* acc, int lane) {
    uint4 (&bA)[T][PF][PNF] = rg.bA;
    uint4 bB[T][PF][PNF];
    const int last = nst - 1;
    auto load_a = [&](uint4 (*a)[PNF], int step) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int pl = 0; pl < PNF; ++pl) a[rb][pl] = sA3[(size_t)rb * img + (size_t)pl * pstride + (size_t)step * 64 + lane];
    };
    auto load_b = [&](uint4 (*b)[PF][PNF], int p, int step) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int pl = 0; pl < PNF; ++pl) b[t][p][pl] = rg.st[t][((size_t)step * PNF + pl) * 64];
    };
    auto keep = [&](uint4 (*a)[PNF], uint4 (*an)[PNF]) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int pl = 0; pl < PNF; ++pl) a[rb][pl] = an[rb][pl];
    };
    uint4 a[RB][PNF];
    load_a(a, 0);
    auto mma = [&](uint4 (*b)[PF][PNF], int p) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb * T + t] = PnF::step(a[rb], b[t][p], acc[rb * T + t]);
    };
    int s = 0;
    for (; s + 2 * PF <= nst; s += 2 * PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            uint4 an[RB][PNF];
            load_a(an, s + p + 1);
            load_b(bB, p, s + PF + p);
            pn_pin();
            mma(bA, p);
            keep(a, an);
        }
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            uint4 an[RB][PNF];
            load_a(an, min(s + PF + p + 1, last));
            load_b(bA, p, min(s + 2 * PF + p, last));
            pn_pin();
            mma(bB, p);
            keep(a, an);
        }
    }
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        if (s + p < nst) {
            uint4 an[RB][PNF];
            load_a(an, min(s + p + 1, last));
            load_b(bB, p, min(s + PF + p, last));
            mma(bA, p);
            keep(a, an);
        }
    }
#pragma unroll
    for (int p = 0; p < PF - 1; ++p) {
        if (s + PF + p < nst) {
            uint4 an[RB][PNF];
            load_a(an, min(s + PF + p + 1, last));
            mma(bB, p);
            keep(a, an);
        }
    }
}
// Long K, few rows (r5: the data gradient [M,1500] x [1500,496] of a SHARD, M = 128..383 -- too few rows for the two-K-halves form,
// and as a plain one-row-block product 32 workgroups that each pull 1.1 MB of fragments through one CU: 15 us for 0.2 GFLOP).  K is
// split over the EIGHT WAVES of a workgroup: every wave streams its eighth of K for the workgroup's T column tiles, the eight partial
// tiles meet in LDS and are added in wave order -- one output, no partial products in global memory, M / 16 x tiles / T workgroups.
template <int T>
__global__ __launch_bounds__(512) void panel_gemm3_ksw_kernel(const float* __restrict__ A, int lda, int M, int K, PanelB3 B,
                                                              float* __restrict__ C, int ldc, int N, PnMap mp) {
    extern __shared__ __attribute__((aligned(16))) uint4 pn3_lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int r_in = mp.rfast ? slot % mp.rpg : slot / mp.cpg, c_in = mp.rfast ? slot / mp.rpg : slot % mp.cpg;
    const int rbk = (xcd / mp.xc) * mp.rpg + r_in, cbk = (xcd % mp.xc) * mp.cpg + c_in;
    if (r_in >= mp.rpg || c_in >= mp.cpg || rbk >= mp.nrb || cbk >= mp.ncb) return;  // ragged groups (whole workgroup)
    const int tile0 = cbk * T, m0 = rbk * 16;
    const int kpad = (K + 31) & ~31, nst_all = kpad >> 5, pstride = (kpad >> 3) * 16;
    const int per = (nst_all + 7) >> 3, s_lo = min(nst_all, wave * per), nst = min(nst_all, s_lo + per) - s_lo;
    PnRing3T<T, 2, PnF> rg;
    if (nst > 0) {                                             // (wave-uniform) in flight while the A block is staged
        const uint4* bf[T];
#pragma unroll
        for (int t = 0; t < T; ++t) bf[t] = B.f + ((size_t)min(tile0 + t, B.ntile - 1) * B.nst + s_lo) * PNF * 64;
        panel3_prefetch_t<T, 2, PnF>(rg, bf, nst, lane);
    }
    f32x4_t ts[T];
#pragma unroll
    for (int t = 0; t < T; ++t) ts[t] = PnF::tile_isc(B, tile0 + t, g);
    PnF::stage<512, 1, 6>(pn3_lds, 0, A, lda, m0, M, 0, K, kpad, tid);
    __syncthreads();
    PnF::Acc acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = PnF::zero();
    if (nst > 0) panel3_mma_rt<1, T, 2>(pn3_lds + (size_t)s_lo * 64, pstride, 0, rg, nst, acc, lane);
    const float rs = PnF::row_isc(pn3_lds, pstride, j);        // (behind the planes: the meeting place below does not reach it)
    __syncthreads();                                           // every wave is done with the image: its head becomes the meeting place
    float4* const red = (float4*)pn3_lds;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const f32x4_t v = PnF::value(acc[t], rs, ts[t]);
        red[(size_t)(wave * T + t) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    if (wave < T) {                                            // wave t adds tile t's eight parts in wave order
        float4 sum = red[(size_t)wave * 64 + lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            const float4 v = red[(size_t)(w * T + wave) * 64 + lane];
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        const int tile = tile0 + wave, n4 = tile * 16 + 4 * g, m = m0 + j;
        if (tile < B.ntile && n4 < N && m < M) {
            float* dst = C + (size_t)m * ldc + n4;
            if (n4 + 3 < N) *(f32x4u_t*)dst = f32x4u_t{sum.x, sum.y, sum.z, sum.w};
            else {
                const float r[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n4 + e < N) dst[e] = r[e];
            }
        }
    }
}
// when the form pays: a long K in one LDS image, few row blocks (FDCAP_PN_KSW=0: never)
static inline bool panel_gemm3_ksw_ok(int M, int K, const PanelB3& B) {
    static std::atomic<int> on{-1};
    if (on < 0) { const char* e = getenv("FDCAP_PN_KSW"); on = (e && e[0] == '0') ? 0 : 1; }
    const int kpad = (K + 31) & ~31;
    return on && kpad <= PN3_MAX_K && kpad >= 768 && ((M + 15) / 16) * ((B.ntile + 7) / 8) < 128;
}
static inline hipError_t panel_gemm3_ksw(const float* A, int lda, int M, int K, const PanelB3& B, float* C, int ldc, int N, hipStream_t st) {
    if (M <= 0 || N <= 0) return hipSuccess;
    const int kpad = (K + 31) & ~31;
    static std::atomic<uint64_t> attr{0};                      // per DEVICE: the attribute belongs to one device's code object
    if (fdc_attr_needed(attr)) {
        hipError_t e = hipFuncSetAttribute((const void*)panel_gemm3_ksw_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pnf_lds_bytes(PN3_MAX_K, 1));
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)panel_gemm3_ksw_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pnf_lds_bytes(PN3_MAX_K, 1));
        if (e != hipSuccess) return e;
        fdc_attr_done(attr);
    }
    // column tiles per workgroup: two while that still gives >= 192 workgroups, else one (FDCAP_PN_KSW_T pins it: A/B) -- r6: and two
    // whenever one tile per workgroup would mean more workgroups than CUs (one workgroup fits a CU: 129-176 rows ran in two rounds,
    // 14.7 us at 160 rows against 8.8 at 128 and 9.5 at 192: tools/launch_times.py sweep)
    static std::atomic<int> t_env{-1};
    if (t_env < 0) { const char* e = getenv("FDCAP_PN_KSW_T"); t_env = e ? atoi(e) : 0; }
    const int nrb = (M + 15) / 16;
    int T = (nrb * ((B.ntile + 1) / 2) >= 192 || nrb * B.ntile > 256) ? 2 : 1;
    if (t_env == 1 || t_env == 2) T = t_env;
    const PnMap mp = panel_map(nrb, (B.ntile + T - 1) / T, (size_t)M * K * 4, (size_t)B.ntile * B.nst * PNF * 1024);
    note_form("panel_gemm3_ksw_kernel");
    if (T == 2)
        hipLaunchKernelGGL(panel_gemm3_ksw_kernel<2>, dim3(8 * mp.rpg * mp.cpg), dim3(512), pnf_lds_bytes(kpad, 1), st, A, lda, M, K, B, C, ldc, N, mp);
    else
        hipLaunchKernelGGL(panel_gemm3_ksw_kernel<1>, dim3(8 * mp.rpg * mp.cpg), dim3(512), pnf_lds_bytes(kpad, 1), st, A, lda, M, K, B, C, ldc, N, mp);
    return hipGetLastError();
}

// ... and for K far beyond any LDS image (r5: the FULL mesh's data gradient, K = 3 V = 31 425 -- BASELINE config 5's contact set, mode
// 'local', the body-model operator's backward; it ran on the generic fp32 tiles, 387 us at 512 rows and 214 us at 128): K in `ks`
// parts, a workgroup = (part, column block of 8 tiles, row pair) walks its part slab by slab -- stage the slab's two 16-row images,
// stream the slab's fragments, accumulate in registers -- and leaves one partial product; panel_part_sum_kernel adds the parts in
// ascending order.  Workgroup b: role = b % (ks ncb) = (part, column block) -> consecutive workgroups = consecutive roles, so with
// ks ncb a multiple of 8 an XCD keeps streaming the same eighth of the static operand; row pair = b / (ks ncb).
template <int RB, int T>
__global__ __launch_bounds__(512) void panel_gemm3_kloop_kernel(const float* __restrict__ A, int lda, int M, int K, PanelB3 B,
                                                                float* __restrict__ Cpart, size_t part_stride, int ldc, int N, int ks,
                                                                int slab_steps) {
    extern __shared__ __attribute__((aligned(16))) uint4 pn3_lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int ncb = (B.ntile + 8 * T - 1) / (8 * T), nrole = ks * ncb;
    const int role = (int)blockIdx.x % nrole, m0 = ((int)blockIdx.x / nrole) * (16 * RB), part = role / ncb, cb = role % ncb;
    const int nst_all = (K + 31) >> 5, per = (nst_all + ks - 1) / ks, s_lo = part * per, s_hi = min(nst_all, s_lo + per);
    const int tile0 = (cb * 8 + wave) * T;                      // this wave's T consecutive column tiles
    const int kpad = 32 * slab_steps, pstride = (kpad >> 3) * 16, img = pnf_img_u4(kpad);
    f32x4_t tot[RB * T];                                        // the slabs' products at the tile's scale (PnH2: a row's scale is per slab)
#pragma unroll
    for (int i = 0; i < RB * T; ++i) tot[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    PnF::Acc acc[RB * T];
#pragma unroll
    for (int i = 0; i < RB * T; ++i) acc[i] = PnF::zero();
    f32x4_t ts[T];
#pragma unroll
    for (int t = 0; t < T; ++t) ts[t] = PnF::tile_isc(B, tile0 + t, g);
    for (int s0 = s_lo; s0 < s_hi; s0 += slab_steps) {
        const int nst = min(slab_steps, s_hi - s0), k0 = 32 * s0, kn = min(K, 32 * (s0 + nst)) - k0;
        PnRing3T<T, 2, PnF> rg;                                 // (requested before the staging: the first round trip hides behind it)
        const uint4* bf[T];
#pragma unroll
        for (int t = 0; t < T; ++t) bf[t] = B.f + ((size_t)min(tile0 + t, B.ntile - 1) * B.nst + s0) * PNF * 64;
        panel3_prefetch_t<T, 2, PnF>(rg, bf, nst, lane);
        if (s0 > s_lo) __syncthreads();                         // every wave is done with the previous slab's images
        PnF::stage<512, RB, 3>(pn3_lds, img, A, lda, m0, M, k0, kn, kpad, tid);
        __syncthreads();
        if (tile0 < B.ntile) panel3_mma_rt<RB, T, 2>(pn3_lds, pstride, img, rg, nst, acc, lane);
        if (PnF::SC_U4) {                                       // fold the slab in at its rows' scales
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float rs = PnF::row_isc(pn3_lds + (size_t)rb * img, pstride, j);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const f32x4_t v = PnF::value(acc[rb * T + t], rs, f32x4_t{1.f, 1.f, 1.f, 1.f});
#pragma unroll
                    for (int r = 0; r < 4; ++r) tot[rb * T + t][r] += v[r];
                    acc[rb * T + t] = PnF::zero();
                }
            }
        }
    }
    float* const C = Cpart + (size_t)part * part_stride;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int n4 = (tile0 + t) * 16 + 4 * g;
        if (tile0 + t < B.ntile && n4 < N) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int m = m0 + 16 * rb + j;
                if (m < M) {
                    f32x4_t v;
                    if (PnF::SC_U4) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = tot[rb * T + t][r] * ts[t][r];
                    } else v = PnF::value(acc[rb * T + t], 1.f, f32x4_t{1.f, 1.f, 1.f, 1.f});
                    pnf_store4(C, ldc, N, m, n4, v);
                }
            }
        }
    }
}
__global__ void panel_part_sum_kernel(const float* __restrict__ part, int ks, size_t part_stride, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = part[i];
    for (int p = 1; p < ks; ++p) s += part[(size_t)p * part_stride + i];
    out[i] = s;
}
constexpr int PN3_KLOOP_SLAB = 24;                               // steps (32 columns each) per slab: two 16-row images of 768 columns = 144 KB
// parts of K for M rows and B's tiles: enough workgroups for 256 CUs, ks x column blocks a multiple of 8 (XCD <-> slice of B), <= 32
inline std::atomic<int>& panel_gemm3_kloop_slab() {                          // FDCAP_KLOOP_SLAB (A/B): steps per slab, 4..24
    static std::atomic<int> v{-1};
    if (v < 0) { const char* e = getenv("FDCAP_KLOOP_SLAB"); v = e ? std::min(24, std::max(4, atoi(e))) : PN3_KLOOP_SLAB; }
    return v;
}
inline std::atomic<int>& panel_gemm3_kloop_rb() {                             // FDCAP_KLOOP_RB (A/B): 16-row blocks per fragment stream, 2 or 4
    static std::atomic<int> v{-1};
    if (v < 0) { const char* e = getenv("FDCAP_KLOOP_RB"); v = (e && atoi(e) == 4) ? 4 : 2; }
    return v;
}
inline std::atomic<int>& panel_gemm3_kloop_t() {                              // FDCAP_KLOOP_T (A/B): column tiles per wave, 1 or 2
    static std::atomic<int> v{-1};
    if (v < 0) { const char* e = getenv("FDCAP_KLOOP_T"); v = (e && atoi(e) == 1) ? 1 : 2; }
    return v;
}
static inline int panel_gemm3_kloop_parts(int M, const PanelB3& B) {
    const int rb = panel_gemm3_kloop_rb(), T = panel_gemm3_kloop_t(), ncb = (B.ntile + 8 * T - 1) / (8 * T), nrp = (M + 16 * rb - 1) / (16 * rb);
    static std::atomic<int> wgs{-1};                                         // FDCAP_KLOOP_WGS (A/B): workgroups to aim for
    if (wgs < 0) { const char* e = getenv("FDCAP_KLOOP_WGS"); wgs = e ? atoi(e) : 256; }
    // r6: the LARGEST admissible part count that still fits one round of workgroups (a workgroup fills a CU's LDS: one per CU).  The
    // count used to be rounded UP to the next admissible one -- 288-336 workgroups at 96 / 192 / 320 / 384 / 448 rows, two rounds:
    // 141.8 us at 384 rows against 113.9 at 512 (tools/launch_times.py --config c5 sweep).
    int ks = std::min(64, std::max(1, wgs / (ncb * nrp)));
    while (ks > 1 && (ks * ncb) % 8 != 0) --ks;
    if ((ks * ncb) % 8 != 0) { ks = 1; while ((ks * ncb) % 8 != 0 && ks < 64) ++ks; }       // (nothing admissible below: the smallest above)
    return std::min(ks, 64);
}
// C [M rows of ldc] = sum of the parts; `part` must hold ks x M x ldc floats (panel_gemm3_kloop_parts)
static inline hipError_t panel_gemm3_kloop(const float* A, int lda, int M, int K, const PanelB3& B, float* part, float* C, int ldc, int N,
                                           hipStream_t st) {
    if (M <= 0 || N <= 0) return hipSuccess;
    const int rb = panel_gemm3_kloop_rb(), T = panel_gemm3_kloop_t(), ks = panel_gemm3_kloop_parts(M, B), ncb = (B.ntile + 8 * T - 1) / (8 * T), nrp = (M + 16 * rb - 1) / (16 * rb);
    const size_t stride = (size_t)M * ldc;
    // steps per slab: RB images of 32 x slab columns, 6 bytes each, in <= 147 KB
    const int slab = std::min((int)panel_gemm3_kloop_slab(), rb == 4 ? 12 : 24);   // (measured at 512 / 128 rows: RB 4 T 1 192 / 55 us, RB 2 T 2 139 / 44, RB 4 T 2 166 / 57 with 55 spilled registers)
    const size_t lds = pnf_lds_bytes(32 * slab, rb);
    static std::atomic<uint64_t> attr{0};                      // per DEVICE: the attribute belongs to one device's code object
    if (fdc_attr_needed(attr)) {
        hipError_t e = hipSuccess;
        const void* fs[] = {(const void*)panel_gemm3_kloop_kernel<2, 1>, (const void*)panel_gemm3_kloop_kernel<4, 1>,
                            (const void*)panel_gemm3_kloop_kernel<2, 2>, (const void*)panel_gemm3_kloop_kernel<4, 2>};
        for (const void* f : fs)
            if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return e;
        fdc_attr_done(attr);
    }
    const dim3 grid(ks * ncb * nrp);
#define FDC_KLOOP(RBV, TV) hipLaunchKernelGGL((panel_gemm3_kloop_kernel<RBV, TV>), grid, dim3(512), lds, st, A, lda, M, K, B, part, stride, ldc, N, ks, slab)
    if (rb == 4 && T == 2) FDC_KLOOP(4, 2); else if (rb == 4) FDC_KLOOP(4, 1); else if (T == 2) FDC_KLOOP(2, 2); else FDC_KLOOP(2, 1);
#undef FDC_KLOOP
    note_form("panel_gemm3_kloop_kernel");
    hipLaunchKernelGGL(panel_part_sum_kernel, dim3((unsigned)((stride + 255) / 256)), dim3(256), 0, st, part, ks, stride, stride, C);
    return hipGetLastError();
}

// four consecutive columns n4 .. n4 + 3 of frame row j -> the planes of an LDS block (8 bytes per plane); sc: the row's scale (PnH2)
__device__ __forceinline__ void pn3_store4(uint4* __restrict__ sA3, int pstride, int n4, int j, float4 v) {
    unsigned h[2], m[2], l[2];
    pn3_split2(v.x, v.y, h[0], m[0], l[0]);
    pn3_split2(v.z, v.w, h[1], m[1], l[1]);
    const size_t it = (size_t)(n4 >> 3) * 16 + j;
    const int half = (n4 >> 2) & 1;
    ((uint2*)(sA3 + (size_t)0 * pstride + it))[half] = make_uint2(h[0], h[1]);
    ((uint2*)(sA3 + (size_t)1 * pstride + it))[half] = make_uint2(m[0], m[1]);
    ((uint2*)(sA3 + (size_t)2 * pstride + it))[half] = make_uint2(l[0], l[1]);
}
__device__ __forceinline__ void pnf_lds_store4(PnB3*, uint4* __restrict__ sA3, int pstride, int n4, int j, float4 v, float) { pn3_store4(sA3, pstride, n4, j, v); }
__device__ __forceinline__ void pnf_lds_store4(PnH2*, uint4* __restrict__ sA, int pstride, int n4, int j, float4 v, float sc) {
    const float x[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
    unsigned h[2], l[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        h[e] = pn2_pk2(x[2 * e], x[2 * e + 1]);
        const pn_f16x2 hh = __builtin_bit_cast(pn_f16x2, h[e]);
        l[e] = pn2_pk2((x[2 * e] - (float)hh[0]) * 2048.f, (x[2 * e + 1] - (float)hh[1]) * 2048.f);
    }
    const size_t it = (size_t)(n4 >> 3) * 16 + j;
    const int half = (n4 >> 2) & 1;
    ((uint2*)(sA + (size_t)0 * pstride + it))[half] = make_uint2(h[0], h[1]);
    ((uint2*)(sA + (size_t)1 * pstride + it))[half] = make_uint2(l[0], l[1]);
}
// one latent value per thread (row i = tid >> 5, column k = tid & 31) straight into a 32-column image (plane stride `pstride`)
__device__ __forceinline__ void pnf_put_latent(PnB3*, uint4* __restrict__ sZ, int pstride, int i, int k, float v) {
    const unsigned h = pn3_bf(v);
    const float r1 = v - pn3_bff(h);
    const unsigned m = pn3_bf(r1), l = pn3_bf(r1 - pn3_bff(m));
    unsigned short* const img = (unsigned short*)sZ;
    const int it = (k >> 3) * 16 + i, e = k & 7;
    img[(size_t)(0 * pstride + it) * 8 + e] = (unsigned short)h;
    img[(size_t)(1 * pstride + it) * 8 + e] = (unsigned short)m;
    img[(size_t)(2 * pstride + it) * 8 + e] = (unsigned short)l;
}
#define PN_DPP_MAX(m, ctrl) fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), ctrl, 0xF, 0xF, false)))
__device__ __forceinline__ void pnf_put_latent(PnH2*, uint4* __restrict__ sZ, int pstride, int i, int k, float v) {
    // the row's largest |z|: its 32 values sit in one half of a wave -- four DPP steps inside each row of 16 lanes, one exchange between rows
    float m = fabsf(v);
    m = PN_DPP_MAX(m, 0xB1);                                     // quad_perm [1,0,3,2]
    m = PN_DPP_MAX(m, 0x4E);                                     // quad_perm [2,3,0,1]
    m = PN_DPP_MAX(m, 0x141);                                    // row_half_mirror
    m = PN_DPP_MAX(m, 0x140);                                    // row_mirror
    m = fmaxf(m, __shfl_xor(m, 16));
    float sc, isc;
    PnH2::pow2(m, sc, isc);
    const float x = v * sc;
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)((x - (float)h) * 2048.f);
    unsigned short* const img = (unsigned short*)sZ;
    const int it = (k >> 3) * 16 + i, e = k & 7;
    img[(size_t)(0 * pstride + it) * 8 + e] = __builtin_bit_cast(unsigned short, h);
    img[(size_t)(1 * pstride + it) * 8 + e] = __builtin_bit_cast(unsigned short, l);
    if (k == 0) ((float*)(sZ + (size_t)PnH2::NP * pstride))[i] = isc;
}
#undef PN_DPP_MAX

// ---------------------------------------------------------------------------------------------------------------------
// VPoser decoder (fc1 32 -> 512, LeakyReLU(0.2), fc2 512 -> 512, LeakyReLU(0.2), out 512 -> 126)
constexpr int VP_Z = 32, VP_H = 512, VP_NQ = 4, VP_QW = VP_H / VP_NQ;
struct VPoserPanels {
    PanelB w1, w2, w3;          // forward:  B(k, n) = W[n][k]   (y = x W^T + b)
    PanelB w3t, w2t, w1t;       // backward: B(k, n) = W[k][n]   (dx = dy W)
    const float *b1 = nullptr, *b2 = nullptr, *b3 = nullptr;
};

__device__ __forceinline__ float vp_lrelu(float v) { return v > 0.f ? v : 0.2f * v; }

// dX[row, latent columns] += ((p0 + p1) + (p2 + p3))   (the order the Adam kernel uses when it folds the partials itself)
__device__ __forceinline__ float vp_sum_dz(const float* __restrict__ part, size_t part_stride, size_t e) {
    return (part[e] + part[part_stride + e]) + (part[2 * part_stride + e] + part[3 * part_stride + e]);
}

// DeferredStep (fdc_loss.h), decoder side: element (row r0 + tid / 32, latent column tid % 32) of the 16 x 32 block AFTER the
// pending Adam step -- one element per thread of the 512-thread workgroup, rows >= row_hi zero.  Nothing is written back: the
// four quarter-workgroups of a row block all do this, pose_fwd_kernel repeats it for its row and stores parameters and moments.
__device__ __forceinline__ float vp_deferred_latent(const DeferredStep& ds, int r0, int row_hi, int tid) {
    const int i = tid >> 5, col = tid & 31, row = r0 + i;
    float pp = 0.f;
    if (row < row_hi) {
        const size_t e = (size_t)(row - ds.row0) * XDIM + X_LATENT + col;
        float mm = ds.x.m[e], vv = ds.x.v[e], gg = ds.x.g[e];
        pp = ds.x.p[e];
        if (ds.dzpart) gg += vp_sum_dz(ds.dzpart, ds.dz_stride, (size_t)row * VP_Z + col);
        adam_update(pp, mm, vv, gg, ds.x.a);
    }
    return pp;
}

// grid = 4 * ceil(rows / 16): blockIdx & 3 = hidden-column quarter q (blocks of one quarter share an XCD's L2: the
// dispatcher deals consecutive blocks to the 8 XCDs, so XCD x only ever streams quarter x & 3 of W2), blockIdx >> 2 = row block.
// Z = X + latent column (row stride ldx); rows [row_lo, row_hi).  H1, H2 [*, 512] (kept for the backward's masks),
// Opart [4][part_stride]: partial decoder outputs (row-major [*, 126]); the bias rides on partial 0.
// A launch may cover TWO row ranges (a shard's halo rows on either side of its owned rows, VpRows): row blocks [0, nb1) belong to
// [row_lo, row_hi), the rest to [row2_lo, row2_hi).  A row's result does not depend on the block it is decoded in.
struct VpRows { int nb1 = 0x7fffffff, row2_lo = 0, row2_hi = 0; };
__global__ __launch_bounds__(512) void vposer_fwd_fused_kernel(VPoserPanels P, const float* __restrict__ Z, int ldx, int row_lo,
                                                               int row_hi, float* __restrict__ H1, float* __restrict__ H2,
                                                               float* __restrict__ Opart, size_t part_stride, VpRows two, DeferredStep ds) {
    __shared__ __attribute__((aligned(16))) float lds[VP_Z * 16 + VP_H * 16 + VP_QW * 16];
    float* const sZ = lds;
    float* const sH1 = lds + VP_Z * 16;
    float* const sH2 = sH1 + VP_H * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    int rblk = (int)(blockIdx.x >> 2);
    if (rblk >= two.nb1) { rblk -= two.nb1; row_lo = two.row2_lo; row_hi = two.row2_hi; }
    const int q = blockIdx.x & 3, r0 = row_lo + rblk * 16;
    PN_STAMP(0);
    PnRing<1, 4> rg2, rg3;                             // next layer's first fragments are requested before the barrier in front of it
    // all three layers' biases before the first barrier (see vposer_fwd_split3_kernel)
    float4 bias1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bias1[t] = *(const float4*)(P.b1 + (wave * 4 + t) * 16 + 4 * g);
    const float4 bias2 = *(const float4*)(P.b2 + (q * 8 + wave) * 16 + 4 * g);
    float bias3[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bias3[e] = P.b3[min(wave * 16 + 4 * g + e, ODIM - 1)];
    {   // layer 1, all 512 columns (four tiles per wave)
        f32x4_t acc[4];
        const float4* bf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; bf[t] = P.w1.f + (size_t)(wave * 4 + t) * P.w1.nss * 64; }
        PnRing<4, 2> rg1;
        panel_prefetch<4, 2>(rg1, bf, VP_Z / 16, lane);
        if (ds.on) {                                       // (wave-uniform) the latent rows after the pending optimiser step,
            const int i = tid >> 5, k = tid & 31;          // straight into panel_stage's k-blocked image: [k / 4][row][k % 4]
            sZ[((k >> 2) * 16 + i) * 4 + (k & 3)] = vp_deferred_latent(ds, r0, row_hi, tid);
        } else
            panel_stage<512>(sZ, Z, ldx, r0, row_hi, 0, VP_Z, VP_Z, tid);
        __syncthreads();
        PN_STAMP(1);
        panel_mma<4, 2>(sZ, rg1, VP_Z / 16, acc, lane);
        {
            const float4* bf2 = P.w2.f + (size_t)(q * 8 + wave) * P.w2.nss * 64;
            panel_prefetch<1, 4>(rg2, &bf2, VP_H / 16, lane);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n4 = (wave * 4 + t) * 16 + 4 * g;
            const float4 bias = bias1[t];
            const float4 v = make_float4(vp_lrelu(acc[t][0] + bias.x), vp_lrelu(acc[t][1] + bias.y), vp_lrelu(acc[t][2] + bias.z),
                                         vp_lrelu(acc[t][3] + bias.w));
            *(float4*)(sH1 + (size_t)((n4 >> 2) * 16 + j) * 4) = v;
            if ((n4 / VP_QW) == q && r0 + j < row_hi) *(float4*)(H1 + (size_t)(r0 + j) * VP_H + n4) = v;
        }
    }
    __syncthreads();
    PN_STAMP(2);
    {   // layer 2, this quarter's 128 columns (one tile per wave)
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        const int tile = q * 8 + wave;
        panel_mma<1, 4>(sH1, rg2, VP_H / 16, &acc, lane);
        {
            const float4* bf3 = P.w3.f + ((size_t)wave * P.w3.nss + q * (VP_QW / 16)) * 64;
            panel_prefetch<1, 4>(rg3, &bf3, VP_QW / 16, lane);
        }
        const int n4 = tile * 16 + 4 * g;
        const float4 bias = bias2;
        const float4 v = make_float4(vp_lrelu(acc[0] + bias.x), vp_lrelu(acc[1] + bias.y), vp_lrelu(acc[2] + bias.z), vp_lrelu(acc[3] + bias.w));
        *(float4*)(sH2 + (size_t)(((n4 - q * VP_QW) >> 2) * 16 + j) * 4) = v;
        if (r0 + j < row_hi) *(float4*)(H2 + (size_t)(r0 + j) * VP_H + n4) = v;
    }
    __syncthreads();
    PN_STAMP(3);
    {   // output layer: this quarter's K-slice of all 126 (128) columns
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        panel_mma<1, 4>(sH2, rg3, VP_QW / 16, &acc, lane);
        const int n4 = wave * 16 + 4 * g, row = r0 + j;              // ODIM = 126: rows are 8-byte aligned -> two float2 stores
        if (row < row_hi) {
            float* dst = Opart + (size_t)q * part_stride + (size_t)row * ODIM + n4;
            const float b0 = q == 0 ? bias3[0] : 0.f, b1 = q == 0 ? bias3[1] : 0.f;
            *(float2*)dst = make_float2(acc[0] + b0, acc[1] + b1);
            if (n4 + 2 < ODIM) {
                const float b2 = q == 0 ? bias3[2] : 0.f, b3 = q == 0 ? bias3[3] : 0.f;
                *(float2*)(dst + 2) = make_float2(acc[2] + b2, acc[3] + b3);
            }
        }
    }
    PN_STAMP(4);
}

// O = ((p0 + p1) + (p2 + p3)) -- the one summation order every consumer of the partial outputs uses
__device__ __forceinline__ float vp_sum_parts(const float* __restrict__ part, size_t part_stride, size_t e) {
    return (part[e] + part[part_stride + e]) + (part[2 * part_stride + e] + part[3 * part_stride + e]);
}
__global__ void vposer_sum_parts_kernel(const float* __restrict__ part, size_t part_stride, size_t e0, size_t n, float* __restrict__ O) {
    const size_t e = e0 + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e < e0 + n) O[e] = vp_sum_parts(part, part_stride, e);
}

// dO [*, 126] -> dZpart [4][part_stride] ([*, 32] row-major): partial latent gradients of rows [row_lo, row_hi).
__global__ __launch_bounds__(512) void vposer_bwd_fused_kernel(VPoserPanels P, const float* __restrict__ dO, int row_lo, int row_hi,
                                                               const float* __restrict__ H1, const float* __restrict__ H2,
                                                               float* __restrict__ dZpart, size_t part_stride, ScaleTail tail) {
    // (tail.block == 0: one extra workgroup, FIRST in the grid -- these kernels keep one workgroup per CU (LDS), a 261st at the
    //  end would wait for a CU to come free and then run alone; first, it is done in a microsecond and hands its CU to the rest)
    if ((int)blockIdx.x == tail.block) { scale_tail_block(tail); return; }
    const unsigned bid = blockIdx.x - (tail.block == 0 ? 1u : 0u);
    __shared__ __attribute__((aligned(16))) float lds[VP_QW * 16 + VP_QW * 16 + VP_H * 16 + 8 * 256];
    float* const sdO = lds;                       // K = 126 padded to 128
    float* const sdH2 = sdO + VP_QW * 16;         // this quarter's 128 columns of dH2
    float* const sdH1 = sdH2 + VP_QW * 16;        // partial dH1 (all 512 columns)
    float* const sred = sdH1 + VP_H * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int q = bid & 3, r0 = row_lo + (int)(bid >> 2) * 16;
    PnRing<4, 4> rgB;
    PnRing<1, 4> rgC;
    // the masking activations before the first barrier (see vposer_bwd_split3_kernel)
    const size_t hrow = (size_t)min(r0 + j, row_hi - 1) * VP_H;
    const float4 hm2 = *(const float4*)(H2 + hrow + (q * 8 + wave) * 16 + 4 * g);
    float4 hm1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) hm1[t] = *(const float4*)(H1 + hrow + (wave + 8 * t) * 16 + 4 * g);
    {   // dH2[:, quarter] = (dO x W3[:, quarter]) * mask(H2)
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        const int tile = q * 8 + wave;
        const float4* bf = P.w3t.f + (size_t)tile * P.w3t.nss * 64;
        PnRing<1, 4> rgA;
        panel_prefetch<1, 4>(rgA, &bf, 8, lane);
        panel_stage<512>(sdO, dO, ODIM, r0, row_hi, 0, ODIM, 128, tid);
        __syncthreads();
        panel_mma<1, 4>(sdO, rgA, 8, &acc, lane);
        {
            const float4* bfb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) bfb[t] = P.w2t.f + ((size_t)(wave + 8 * t) * P.w2t.nss + q * (VP_QW / 16)) * 64;
            panel_prefetch<4, 4>(rgB, bfb, VP_QW / 16, lane);
        }
        const int n4 = tile * 16 + 4 * g;
        const float4 h = hm2;
        *(float4*)(sdH2 + (size_t)(((n4 - q * VP_QW) >> 2) * 16 + j) * 4) =
            make_float4(acc[0] * (h.x > 0.f ? 1.f : 0.2f), acc[1] * (h.y > 0.f ? 1.f : 0.2f), acc[2] * (h.z > 0.f ? 1.f : 0.2f),
                        acc[3] * (h.w > 0.f ? 1.f : 0.2f));
    }
    __syncthreads();
    {   // partial dH1 = (dH2[:, quarter] x W2[quarter rows, :]) * mask(H1): tiles wave, wave + 8, wave + 16, wave + 24
        f32x4_t acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        panel_mma<4, 4>(sdH2, rgB, VP_QW / 16, acc, lane);
        {
            const float4* bfc = P.w1t.f + ((size_t)(wave & 1) * P.w1t.nss + (wave >> 1) * 8) * 64;
            panel_prefetch<1, 4>(rgC, &bfc, 8, lane);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n4 = (wave + 8 * t) * 16 + 4 * g;
            const float4 h = hm1[t];
            *(float4*)(sdH1 + (size_t)((n4 >> 2) * 16 + j) * 4) =
                make_float4(acc[t][0] * (h.x > 0.f ? 1.f : 0.2f), acc[t][1] * (h.y > 0.f ? 1.f : 0.2f),
                            acc[t][2] * (h.z > 0.f ? 1.f : 0.2f), acc[t][3] * (h.w > 0.f ? 1.f : 0.2f));
        }
    }
    __syncthreads();
    {   // partial d latent = partial dH1 x W1: 2 column tiles x 4 K-slices over the 8 waves, slices summed in order
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        const int tile = wave & 1, ks = wave >> 1;
        panel_mma<1, 4>(sdH1 + (size_t)ks * 8 * 256, rgC, 8, &acc, lane);
        *(float4*)(sred + (size_t)(ks * 2 + tile) * 256 + lane * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    __syncthreads();
    if (wave < 2) {
        const float4 s0 = *(const float4*)(sred + (size_t)(0 + wave) * 256 + lane * 4), s1 = *(const float4*)(sred + (size_t)(2 + wave) * 256 + lane * 4);
        const float4 s2 = *(const float4*)(sred + (size_t)(4 + wave) * 256 + lane * 4), s3 = *(const float4*)(sred + (size_t)(6 + wave) * 256 + lane * 4);
        const int n4 = wave * 16 + 4 * g, row = r0 + j;
        if (row < row_hi)
            *(float4*)(dZpart + (size_t)q * part_stride + (size_t)row * VP_Z + n4) =
                make_float4(((s0.x + s1.x) + s2.x) + s3.x, ((s0.y + s1.y) + s2.y) + s3.y, ((s0.z + s1.z) + s2.z) + s3.z, ((s0.w + s1.w) + s2.w) + s3.w);
    }
}

__global__ void vposer_fold_dz_kernel(const float* __restrict__ part, size_t part_stride, int row_lo, int nrows, float* __restrict__ dX) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nrows * VP_Z) return;
    const int row = row_lo + e / VP_Z, c = e % VP_Z;
    dX[(size_t)row * XDIM + X_LATENT + c] += vp_sum_dz(part, part_stride, (size_t)row * VP_Z + c);
}

// ---------------------------------------------------------------------------------------------------------------------
// The fused VPoser kernels on the split format VpF (= PnF: two scaled fp16 planes, r5; three bf16 planes until then) (same decomposition: 16 rows x one quarter of the hidden columns
// per workgroup, four partial outputs / four partial latent gradients; see vposer_fwd_fused_kernel / vposer_bwd_fused_kernel).
struct VPoserPanels3 {
    PanelB3 w1, w2, w3, w3t, w2t, w1t;
    const float *b1 = nullptr, *b2 = nullptr, *b3 = nullptr;
    // PnH2: what bounds a hidden row from the row before it -- |x W + b| <= c max|x| + bmax with c = the largest column 1-norm of W --
    // so every row's power-of-two scale follows from its latent's (its output gradient's) largest entry without a reduction
    float c1 = 0.f, b1max = 0.f, c2 = 0.f, b2max = 0.f, c3t = 0.f, c2t = 0.f;
};
typedef PnF VpF;                                  // operand format of the fused VPoser kernels
constexpr int VP3_PZ = (VP_Z / 8) * 16, VP3_PH = (VP_H / 8) * 16, VP3_PQ = (VP_QW / 8) * 16;     // plane strides (uint4)
// What the backward needs of the hidden activations is their SIGNS (LeakyReLU's slope): the split kernels keep one byte per four
// columns in H1 / H2 -- bit e = column 4 b + e is positive -- 128 bytes per row instead of 2 KB (late r4: what a kernel leaves dirty in
// L2 is written back when it ends, ~0.2 us per MB, tools/launch_overhead_probe.hip; the forward wrote 4 MB of activations per launch
// and the four quarter-workgroups of a row block read H1 four times over).  Indexed by absolute row: the forward's and the backward's
// 16-row blocks need not coincide (sharded runs).  The exact-fp32 twins keep floats in the same buffers.
constexpr int VP3_MROW = VP_H / 4;
#ifndef FDC_VP3_PF2
#define FDC_VP3_PF2 4
#endif
constexpr int VP3_PF2 = FDC_VP3_PF2;          // fragments of layer 2's stream in flight per wave (16 steps; the ring turns every 2 PF)
__device__ __forceinline__ unsigned char vp3_signs(const float4 v) {
    return (unsigned char)((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u));
}

__global__ __launch_bounds__(512) void vposer_fwd_split3_kernel(VPoserPanels3 P, const float* __restrict__ Z, int ldx, int row_lo,
                                                                int row_hi, float* __restrict__ H1, float* __restrict__ H2,
                                                                float* __restrict__ Opart, size_t part_stride, VpRows two, DeferredStep ds) {
    constexpr int NP = VpF::NP;
    __shared__ __attribute__((aligned(16))) uint4 lds3[NP * (VP3_PZ + VP3_PH + VP3_PQ) + VpF::SC_U4];
    uint4* const sZ = lds3;                                      // (PnH2: the latent rows' inverse scales right behind its planes)
    uint4* const sH1 = sZ + NP * VP3_PZ + VpF::SC_U4;
    uint4* const sH2 = sH1 + NP * VP3_PH;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    int rblk = (int)(blockIdx.x >> 2);
    if (rblk >= two.nb1) { rblk -= two.nb1; row_lo = two.row2_lo; row_hi = two.row2_hi; }
    const int q = blockIdx.x & 3, r0 = row_lo + rblk * 16;
    PnRing3T<1, VP3_PF2, VpF> rg2;
    PnRing3T<1, 2, VpF> rg3;
    // the biases (and column scales) of all three layers, requested before the first barrier (the compiler does not move a load across
    // one): fetched where they are used -- after each layer's products -- every tile's bias was a round trip of its own
    float4 bias1[4];
    f32x4_t cs1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { bias1[t] = *(const float4*)(P.b1 + (wave * 4 + t) * 16 + 4 * g); cs1[t] = VpF::tile_isc(P.w1, wave * 4 + t, g); }
    const float4 bias2 = *(const float4*)(P.b2 + (q * 8 + wave) * 16 + 4 * g);
    const f32x4_t cs2 = VpF::tile_isc(P.w2, q * 8 + wave, g), cs3 = VpF::tile_isc(P.w3, wave, g);
    float bias3[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bias3[e] = P.b3[min(wave * 16 + 4 * g + e, ODIM - 1)];
    float bnd1, s1, is1;
    {   // layer 1, all 512 columns (four tiles per wave), K = 32 = one step
        VpF::Acc acc[4];
        const uint4* bf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[t] = VpF::zero(); bf[t] = P.w1.f + (size_t)(wave * 4 + t) * P.w1.nst * NP * 64; }
        PnRing3T<4, 1, VpF> rg1;
        panel3_prefetch_t<4, 1, VpF>(rg1, bf, 1, lane);
        if (ds.on)                                         // (wave-uniform) the latent rows after the pending optimiser step, straight into the image
            pnf_put_latent((VpF*)nullptr, sZ, VP3_PZ, tid >> 5, tid & 31, vp_deferred_latent(ds, r0, row_hi, tid));
        else
            VpF::stage<512, 1, 1>(sZ, 0, Z, ldx, r0, row_hi, 0, VP_Z, VP_Z, tid);
        __syncthreads();
        panel3_mma_t<4, 1, VpF>(sZ, VP3_PZ, rg1, 1, acc, lane);
        {
            const uint4* bf2 = P.w2.f + (size_t)(q * 8 + wave) * P.w2.nst * NP * 64;
            panel3_prefetch_t<1, VP3_PF2, VpF>(rg2, &bf2, VP_H / 32, lane);
        }
        const float zis = VpF::row_isc(sZ, VP3_PZ, j);
        bnd1 = P.c1 * (16384.f * zis) + P.b1max;           // >= every |H1| of row j (2^14 zis >= its largest |z|)
        VpF::pow2(bnd1, s1, is1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n4 = (wave * 4 + t) * 16 + 4 * g;
            const float4 bias = bias1[t];
            const f32x4_t o = VpF::value(acc[t], zis, cs1[t]);
            const float4 v = make_float4(vp_lrelu(o[0] + bias.x), vp_lrelu(o[1] + bias.y), vp_lrelu(o[2] + bias.z), vp_lrelu(o[3] + bias.w));
            pnf_lds_store4((VpF*)nullptr, sH1, VP3_PH, n4, j, v, s1);
            if ((n4 / VP_QW) == q && r0 + j < row_hi) ((unsigned char*)H1)[(size_t)(r0 + j) * VP3_MROW + (n4 >> 2)] = vp3_signs(v);
        }
    }
    __syncthreads();
    float s2, is2;
    {   // layer 2, this quarter's 128 columns (one tile per wave), K = 512 = 16 steps
        VpF::Acc acc = VpF::zero();
        const int tile = q * 8 + wave;
        panel3_mma_t<1, VP3_PF2, VpF>(sH1, VP3_PH, rg2, VP_H / 32, &acc, lane);
        {
            const uint4* bf3 = P.w3.f + ((size_t)wave * P.w3.nst + q * (VP_QW / 32)) * NP * 64;
            panel3_prefetch_t<1, 2, VpF>(rg3, &bf3, VP_QW / 32, lane);
        }
        VpF::pow2(P.c2 * bnd1 + P.b2max, s2, is2);
        const int n4 = tile * 16 + 4 * g;
        const float4 bias = bias2;
        const f32x4_t o = VpF::value(acc, is1, cs2);
        const float4 v = make_float4(vp_lrelu(o[0] + bias.x), vp_lrelu(o[1] + bias.y), vp_lrelu(o[2] + bias.z), vp_lrelu(o[3] + bias.w));
        pnf_lds_store4((VpF*)nullptr, sH2, VP3_PQ, n4 - q * VP_QW, j, v, s2);
        if (r0 + j < row_hi) ((unsigned char*)H2)[(size_t)(r0 + j) * VP3_MROW + (n4 >> 2)] = vp3_signs(v);
    }
    __syncthreads();
    {   // output layer: this quarter's K-slice (128 = 4 steps) of all 126 (128) columns
        VpF::Acc acc = VpF::zero();
        panel3_mma_t<1, 2, VpF>(sH2, VP3_PQ, rg3, VP_QW / 32, &acc, lane);
        const f32x4_t o = VpF::value(acc, is2, cs3);
        const int n4 = wave * 16 + 4 * g, row = r0 + j;
        if (row < row_hi) {
            float* dst = Opart + (size_t)q * part_stride + (size_t)row * ODIM + n4;
            const float b0 = q == 0 ? bias3[0] : 0.f, b1 = q == 0 ? bias3[1] : 0.f;
            *(float2*)dst = make_float2(o[0] + b0, o[1] + b1);
            if (n4 + 2 < ODIM) {
                const float b2 = q == 0 ? bias3[2] : 0.f, b3 = q == 0 ? bias3[3] : 0.f;
                *(float2*)(dst + 2) = make_float2(o[2] + b2, o[3] + b3);
            }
        }
    }
}

__global__ __launch_bounds__(512) void vposer_bwd_split3_kernel(VPoserPanels3 P, const float* __restrict__ dO, int row_lo, int row_hi,
                                                                const float* __restrict__ H1, const float* __restrict__ H2,
                                                                float* __restrict__ dZpart, size_t part_stride, ScaleTail tail) {
    // (tail.block == 0: one extra workgroup, FIRST in the grid -- these kernels keep one workgroup per CU (LDS), a 261st at the
    //  end would wait for a CU to come free and then run alone; first, it is done in a microsecond and hands its CU to the rest)
    if ((int)blockIdx.x == tail.block) {
        ScaleTail t2 = tail;
        if (tail.lg_spread) t2.lg.rows = nullptr;            // (the logged sums: regular workgroups, below)
        scale_tail_block(t2);
        return;
    }
    const unsigned bid = blockIdx.x - (tail.block == 0 ? 1u : 0u);
    constexpr int NP = VpF::NP;
    __shared__ __attribute__((aligned(16))) uint4 lds3[NP * (VP3_PQ + VP3_PQ + VP3_PH) + VpF::SC_U4 + 8 * 64];
    uint4* const sdO = lds3;                      // K = 126 padded to 128 (PnH2: + the rows' inverse scales behind the planes)
    uint4* const sdH2 = sdO + NP * VP3_PQ + VpF::SC_U4;   // this quarter's 128 columns of dH2
    uint4* const sdH1 = sdH2 + NP * VP3_PQ;       // partial dH1 (all 512 columns)
    float* const sred = (float*)(sdH1 + NP * VP3_PH);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int q = bid & 3, r0 = row_lo + (int)(bid >> 2) * 16;
    PnRing3T<4, 2, VpF> rgB;
    PnRing3T<1, 2, VpF> rgC;
    // the forward activations whose signs mask this wave's tiles, requested before the first barrier (rows clamped:
    // unconditional loads).  Fetched after each layer's products they were five dependent round trips on cold data.
    const size_t hrow = (size_t)min(r0 + j, row_hi - 1) * VP3_MROW;          // (sign bytes: vp3_signs)
    const unsigned hm2 = ((const unsigned char*)H2)[hrow + (q * 8 + wave) * 4 + g];
    unsigned hm1[4];
    f32x4_t csB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { hm1[t] = ((const unsigned char*)H1)[hrow + (wave + 8 * t) * 4 + g]; csB[t] = VpF::tile_isc(P.w2t, wave + 8 * t, g); }
    const f32x4_t csA = VpF::tile_isc(P.w3t, q * 8 + wave, g), csC = VpF::tile_isc(P.w1t, wave & 1, g);
    float bd2, sd2, isd2;
    {   // dH2[:, quarter] = (dO x W3[:, quarter]) * mask(H2)
        VpF::Acc acc = VpF::zero();
        const int tile = q * 8 + wave;
        const uint4* bf = P.w3t.f + (size_t)tile * P.w3t.nst * NP * 64;
        PnRing3T<1, 2, VpF> rgA;
        panel3_prefetch_t<1, 2, VpF>(rgA, &bf, 4, lane);
        if (tail.lg_spread && bid < (unsigned)LROW && wave == 0)     // (wave-uniform) one logged term: ScaleTail::lg_spread
            loss_rows_reduce_slot(tail.lg.rows, tail.row0, tail.lg.n, tail.lg.mask, tail.lg.assign, tail.lg.losses, (int)bid, lane);
        VpF::stage<512, 1, 1>(sdO, 0, dO, ODIM, r0, row_hi, 0, ODIM, 128, tid);
        __syncthreads();
        panel3_mma_t<1, 2, VpF>(sdO, VP3_PQ, rgA, 4, &acc, lane);
        {
            const uint4* bfb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) bfb[t] = P.w2t.f + ((size_t)(wave + 8 * t) * P.w2t.nst + q * (VP_QW / 32)) * NP * 64;
            panel3_prefetch_t<4, 2, VpF>(rgB, bfb, VP_QW / 32, lane);
        }
        const float dois = VpF::row_isc(sdO, VP3_PQ, j);
        bd2 = P.c3t * (16384.f * dois);                   // >= every |dH2| of row j
        VpF::pow2(bd2, sd2, isd2);
        const f32x4_t o = VpF::value(acc, dois, csA);
        const int n4 = tile * 16 + 4 * g;
        const unsigned h = hm2;
        pnf_lds_store4((VpF*)nullptr, sdH2, VP3_PQ, n4 - q * VP_QW, j,
                       make_float4(o[0] * ((h & 1u) ? 1.f : 0.2f), o[1] * ((h & 2u) ? 1.f : 0.2f), o[2] * ((h & 4u) ? 1.f : 0.2f),
                                   o[3] * ((h & 8u) ? 1.f : 0.2f)), sd2);
    }
    __syncthreads();
    float sd1, isd1;
    {   // partial dH1 = (dH2[:, quarter] x W2[quarter rows, :]) * mask(H1): tiles wave, wave + 8, wave + 16, wave + 24
        VpF::Acc acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = VpF::zero();
        panel3_mma_t<4, 2, VpF>(sdH2, VP3_PQ, rgB, VP_QW / 32, acc, lane);
        {
            const uint4* bfc = P.w1t.f + ((size_t)(wave & 1) * P.w1t.nst + (wave >> 1) * 4) * NP * 64;
            panel3_prefetch_t<1, 2, VpF>(rgC, &bfc, 4, lane);
        }
        VpF::pow2(P.c2t * bd2, sd1, isd1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n4 = (wave + 8 * t) * 16 + 4 * g;
            const unsigned h = hm1[t];
            const f32x4_t o = VpF::value(acc[t], isd2, csB[t]);
            pnf_lds_store4((VpF*)nullptr, sdH1, VP3_PH, n4, j,
                           make_float4(o[0] * ((h & 1u) ? 1.f : 0.2f), o[1] * ((h & 2u) ? 1.f : 0.2f),
                                       o[2] * ((h & 4u) ? 1.f : 0.2f), o[3] * ((h & 8u) ? 1.f : 0.2f)), sd1);
        }
    }
    __syncthreads();
    {   // partial d latent = partial dH1 x W1: 2 column tiles x 4 K-slices (128 columns = 4 steps each), slices summed in order
        VpF::Acc acc = VpF::zero();
        const int tile = wave & 1, ks = wave >> 1;
        panel3_mma_t<1, 2, VpF>(sdH1 + (size_t)ks * 4 * 64, VP3_PH, rgC, 4, &acc, lane);
        const f32x4_t o = VpF::value(acc, isd1, csC);
        *(float4*)(sred + (size_t)(ks * 2 + tile) * 256 + lane * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
    if (wave < 2) {
        const float4 s0 = *(const float4*)(sred + (size_t)(0 + wave) * 256 + lane * 4), s1 = *(const float4*)(sred + (size_t)(2 + wave) * 256 + lane * 4);
        const float4 s2 = *(const float4*)(sred + (size_t)(4 + wave) * 256 + lane * 4), s3 = *(const float4*)(sred + (size_t)(6 + wave) * 256 + lane * 4);
        const int n4 = wave * 16 + 4 * g, row = r0 + j;
        if (row < row_hi)
            *(float4*)(dZpart + (size_t)q * part_stride + (size_t)row * VP_Z + n4) =
                make_float4(((s0.x + s1.x) + s2.x) + s3.x, ((s0.y + s1.y) + s2.y) + s3.y, ((s0.z + s1.z) + s2.z) + s3.z, ((s0.w + s1.w) + s2.w) + s3.w);
    }
}

}  // namespace fdc
