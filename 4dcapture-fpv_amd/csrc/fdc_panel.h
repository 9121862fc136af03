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

#include <vector>

#include "fdc_frame.h"

namespace fdc {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

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
template <int T, int PF>
__device__ __forceinline__ void panel_mma(const float* __restrict__ sA, const float4* const* bf, int nss, f32x4_t* acc, int lane) {
    float4 b[T][PF];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int t = 0; t < T; ++t) b[t][p] = (p < nss) ? bf[t][p * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < nss; s0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int s = s0 + p;
            if (s < nss) {                                           // wave-uniform
                const float4 a = *(const float4*)(sA + (size_t)s * 256 + lane * 4);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float4 bb = b[t][p];
                    if (s + PF < nss) b[t][p] = bf[t][(s + PF) * 64 + lane];
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bb.x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bb.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bb.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bb.w, acc[t], 0, 0, 0);
                }
            }
        }
    }
}

// RB row blocks of 16 share one B fragment stream (wide outputs: B traffic / RB).  sA: RB images, `img` floats apart.
template <int RB, int PF>
__device__ __forceinline__ void panel_mma_rows(const float* __restrict__ sA, int img, const float4* __restrict__ bf, int nss,
                                               f32x4_t* acc, int lane) {
    float4 b[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) b[p] = (p < nss) ? bf[p * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < nss; s0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int s = s0 + p;
            if (s < nss) {
                const float4 bb = b[p];
                if (s + PF < nss) b[p] = bf[(s + PF) * 64 + lane];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const float4 a = *(const float4*)(sA + (size_t)rb * img + (size_t)s * 256 + lane * 4);
                    acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bb.x, acc[rb], 0, 0, 0);
                    acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bb.y, acc[rb], 0, 0, 0);
                    acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bb.z, acc[rb], 0, 0, 0);
                    acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bb.w, acc[rb], 0, 0, 0);
                }
            }
        }
    }
}

// C[M, N] = A[M, K] x B.  Workgroup = 8 waves = 16 RB rows x 128 columns (a 16-column tile per wave); K in slabs of `kslab`
// columns (multiple of 16) so any K fits the LDS.  blockIdx.x = column block, blockIdx.y = row block.
// Dynamic LDS: RB * kslab * 16 floats.
template <int RB>
__global__ __launch_bounds__(512) void panel_gemm_kernel(const float* __restrict__ A, int lda, int M, int K, PanelB B, int kslab,
                                                         float* __restrict__ C, int ldc, int N) {
    extern __shared__ __attribute__((aligned(16))) float pn_lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x * 8 + wave;
    const int m0 = blockIdx.y * (16 * RB);
    const int img = kslab * 16;
    f32x4_t acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += kslab) {
        const int kn = min(kslab, K - k0), kpad = (kn + 15) & ~15;
        if (k0) __syncthreads();
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) panel_stage<512>(pn_lds + (size_t)rb * img, A, lda, m0 + 16 * rb, M, k0, kn, kpad, tid);
        __syncthreads();
        if (tile < B.ntile)
            panel_mma_rows<RB, 4>(pn_lds, img, B.f + ((size_t)tile * B.nss + (k0 >> 4)) * 64, kpad >> 4, acc, lane);
    }
    const int n = tile * 16 + j;
    if (tile < B.ntile && n < N) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * rb + 4 * g + r;
                if (m < M) C[(size_t)m * ldc + n] = acc[rb][r];
            }
    }
}

static inline hipError_t panel_gemm(const float* A, int lda, int M, int K, const PanelB& B, float* C, int ldc, int N, hipStream_t st) {
    if (M <= 0 || N <= 0) return hipSuccess;
    // one row block per workgroup while the operand re-reads stay inside an XCD's L2 (the loop's products: a 3 MB panel);
    // wide outputs (full-mesh blend, N = 31 425) take four row blocks per fragment stream
    const bool wide = (size_t)B.ntile * B.nss * 1024 > (size_t)(24u << 20) && M >= 64;
    const int kpad = (K + 15) & ~15;
    const dim3 grid((B.ntile + 7) / 8, wide ? (M + 63) / 64 : (M + 15) / 16);
    if (wide) {
        const int kslab = kpad <= 512 ? kpad : 512;                 // 4 x 32 KiB
        hipLaunchKernelGGL(panel_gemm_kernel<4>, grid, dim3(512), (size_t)4 * kslab * 16 * sizeof(float), st, A, lda, M, K, B, kslab, C, ldc, N);
    } else {
        const int kslab = kpad <= 1536 ? kpad : 1024;               // <= 96 KiB
        hipLaunchKernelGGL(panel_gemm_kernel<1>, grid, dim3(512), (size_t)kslab * 16 * sizeof(float), st, A, lda, M, K, B, kslab, C, ldc, N);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// VPoser decoder (fc1 32 -> 512, LeakyReLU(0.2), fc2 512 -> 512, LeakyReLU(0.2), out 512 -> 126)
constexpr int VP_Z = 32, VP_H = 512, VP_NQ = 4, VP_QW = VP_H / VP_NQ;
struct VPoserPanels {
    PanelB w1, w2, w3;          // forward:  B(k, n) = W[n][k]   (y = x W^T + b)
    PanelB w3t, w2t, w1t;       // backward: B(k, n) = W[k][n]   (dx = dy W)
    const float *b1 = nullptr, *b2 = nullptr, *b3 = nullptr;
};

__device__ __forceinline__ float vp_lrelu(float v) { return v > 0.f ? v : 0.2f * v; }

// grid = 4 * ceil(rows / 16): blockIdx & 3 = hidden-column quarter q (blocks of one quarter share an XCD's L2: the
// dispatcher deals consecutive blocks to the 8 XCDs, so an XCD only ever streams two quarters of W2), blockIdx >> 2 = row block.
// Z = X + latent column (row stride ldx); rows [row_lo, row_hi).  H1, H2 [*, 512] (kept for the backward's masks),
// Opart [4][part_stride]: partial decoder outputs (row-major [*, 126]); the bias rides on partial 0.
__global__ __launch_bounds__(512) void vposer_fwd_fused_kernel(VPoserPanels P, const float* __restrict__ Z, int ldx, int row_lo,
                                                               int row_hi, float* __restrict__ H1, float* __restrict__ H2,
                                                               float* __restrict__ Opart, size_t part_stride) {
    __shared__ __attribute__((aligned(16))) float lds[VP_Z * 16 + VP_H * 16 + VP_QW * 16];
    float* const sZ = lds;
    float* const sH1 = lds + VP_Z * 16;
    float* const sH2 = sH1 + VP_H * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int q = blockIdx.x & 3, r0 = row_lo + (int)(blockIdx.x >> 2) * 16;
    panel_stage<512>(sZ, Z, ldx, r0, row_hi, 0, VP_Z, VP_Z, tid);
    __syncthreads();
    {   // layer 1, all 512 columns (four tiles per wave)
        f32x4_t acc[4];
        const float4* bf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; bf[t] = P.w1.f + (size_t)(wave * 4 + t) * P.w1.nss * 64; }
        panel_mma<4, 2>(sZ, bf, VP_Z / 16, acc, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = (wave * 4 + t) * 16 + j;
            const float bias = P.b1[n];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * g + r;
                const float v = vp_lrelu(acc[t][r] + bias);
                sH1[pn_lds_index(i, n)] = v;
                if ((n / VP_QW) == q && r0 + i < row_hi) H1[(size_t)(r0 + i) * VP_H + n] = v;
            }
        }
    }
    __syncthreads();
    {   // layer 2, this quarter's 128 columns (one tile per wave)
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        const int tile = q * 8 + wave;
        const float4* bf = P.w2.f + (size_t)tile * P.w2.nss * 64;
        panel_mma<1, 4>(sH1, &bf, VP_H / 16, &acc, lane);
        const int n = tile * 16 + j;
        const float bias = P.b2[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * g + r;
            const float v = vp_lrelu(acc[r] + bias);
            sH2[pn_lds_index(i, n - q * VP_QW)] = v;
            if (r0 + i < row_hi) H2[(size_t)(r0 + i) * VP_H + n] = v;
        }
    }
    __syncthreads();
    {   // output layer: this quarter's K-slice of all 126 (128) columns
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        const float4* bf = P.w3.f + ((size_t)wave * P.w3.nss + q * (VP_QW / 16)) * 64;
        panel_mma<1, 4>(sH2, &bf, VP_QW / 16, &acc, lane);
        const int n = wave * 16 + j;
        if (n < ODIM) {
            const float bias = q == 0 ? P.b3[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + 4 * g + r;
                if (row < row_hi) Opart[(size_t)q * part_stride + (size_t)row * ODIM + n] = acc[r] + bias;
            }
        }
    }
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
                                                               float* __restrict__ dZpart, size_t part_stride) {
    __shared__ __attribute__((aligned(16))) float lds[VP_QW * 16 + VP_QW * 16 + VP_H * 16 + 8 * 256];
    float* const sdO = lds;                       // K = 126 padded to 128
    float* const sdH2 = sdO + VP_QW * 16;         // this quarter's 128 columns of dH2
    float* const sdH1 = sdH2 + VP_QW * 16;        // partial dH1 (all 512 columns)
    float* const sred = sdH1 + VP_H * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int q = blockIdx.x & 3, r0 = row_lo + (int)(blockIdx.x >> 2) * 16;
    panel_stage<512>(sdO, dO, ODIM, r0, row_hi, 0, ODIM, 128, tid);
    __syncthreads();
    {   // dH2[:, quarter] = (dO x W3[:, quarter]) * mask(H2)
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        const int tile = q * 8 + wave;
        const float4* bf = P.w3t.f + (size_t)tile * P.w3t.nss * 64;
        panel_mma<1, 4>(sdO, &bf, 8, &acc, lane);
        const int n = tile * 16 + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * g + r;
            const float h = (r0 + i < row_hi) ? H2[(size_t)(r0 + i) * VP_H + n] : 0.f;
            sdH2[pn_lds_index(i, n - q * VP_QW)] = acc[r] * (h > 0.f ? 1.f : 0.2f);
        }
    }
    __syncthreads();
    {   // partial dH1 = (dH2[:, quarter] x W2[quarter rows, :]) * mask(H1): tiles wave, wave + 8, wave + 16, wave + 24
        f32x4_t acc[4];
        const float4* bf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            bf[t] = P.w2t.f + ((size_t)(wave + 8 * t) * P.w2t.nss + q * (VP_QW / 16)) * 64;
        }
        panel_mma<4, 4>(sdH2, bf, VP_QW / 16, acc, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = (wave + 8 * t) * 16 + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * g + r;
                const float h = (r0 + i < row_hi) ? H1[(size_t)(r0 + i) * VP_H + n] : 0.f;
                sdH1[pn_lds_index(i, n)] = acc[t][r] * (h > 0.f ? 1.f : 0.2f);
            }
        }
    }
    __syncthreads();
    {   // partial d latent = partial dH1 x W1: 2 column tiles x 4 K-slices over the 8 waves, slices summed in order
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        const int tile = wave & 1, ks = wave >> 1;
        const float4* bf = P.w1t.f + ((size_t)tile * P.w1t.nss + ks * 8) * 64;
        panel_mma<1, 4>(sdH1 + (size_t)ks * 8 * 256, &bf, 8, &acc, lane);
        *(float4*)(sred + (size_t)(ks * 2 + tile) * 256 + lane * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    __syncthreads();
    if (wave < 2) {
        const float4 s0 = *(const float4*)(sred + (size_t)(0 + wave) * 256 + lane * 4), s1 = *(const float4*)(sred + (size_t)(2 + wave) * 256 + lane * 4);
        const float4 s2 = *(const float4*)(sred + (size_t)(4 + wave) * 256 + lane * 4), s3 = *(const float4*)(sred + (size_t)(6 + wave) * 256 + lane * 4);
        const float v[4] = {((s0.x + s1.x) + s2.x) + s3.x, ((s0.y + s1.y) + s2.y) + s3.y, ((s0.z + s1.z) + s2.z) + s3.z, ((s0.w + s1.w) + s2.w) + s3.w};
        const int n = wave * 16 + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + 4 * g + r;
            if (row < row_hi) dZpart[(size_t)q * part_stride + (size_t)row * VP_Z + n] = v[r];
        }
    }
}

// dX[row, latent columns] += ((p0 + p1) + (p2 + p3))   (the order the Adam kernel uses when it folds the partials itself)
__device__ __forceinline__ float vp_sum_dz(const float* __restrict__ part, size_t part_stride, size_t e) {
    return (part[e] + part[part_stride + e]) + (part[2 * part_stride + e] + part[3 * part_stride + e]);
}
__global__ void vposer_fold_dz_kernel(const float* __restrict__ part, size_t part_stride, int row_lo, int nrows, float* __restrict__ dX) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nrows * VP_Z) return;
    const int row = row_lo + e / VP_Z, c = e % VP_Z;
    dX[(size_t)row * XDIM + X_LATENT + c] += vp_sum_dz(part, part_stride, (size_t)row * VP_Z + c);
}

}  // namespace fdc
