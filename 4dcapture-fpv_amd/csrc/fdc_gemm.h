// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, bitwise an fmaf
// chain) used for every dense contraction of the hot path:
//   VPoser decoder layers forward / data-gradient (K3),
//   pose blendshapes  v_off[F,3V'] = pose_feature[F,486] x posedirs[486,3V']  (K8) and its
//   data-gradient     dPF[F,486]   = dv_off[F,3V'] x posedirs^T.
// C[M,N] = epi(A[M,K] x B), A row-major with leading dimension lda (the latent is read in place
// from the 78-wide parameter rows), B either [K,N] ("NN") or [N,K] ("NT").
// 64-wide wavefronts: a workgroup is 4 waves in a 2x2 arrangement, each wave owns TM x TN
// 32x32 accumulator tiles; operands are staged through LDS with a +1 padded K stride so both
// fragment reads (lanes walk rows, fixed k) are bank-conflict free.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

namespace fdc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum GemmEpi {
    EPI_STORE = 0,        // C = acc
    EPI_BIAS = 1,         // C = acc + bias[n]
    EPI_BIAS_LRELU = 2,   // C = leaky_relu(acc + bias[n], 0.2)
    EPI_MASK_LRELU = 3,   // C = acc * (aux[m,n] > 0 ? 1 : 0.2)   (aux = forward activation)
    EPI_ACCUM = 4         // C += acc
};

template <bool B_IS_NK, int EPI, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C,
    int ldc, int M, int N, int K, const float* __restrict__ aux, int ldaux) {
    constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32, LD = BK + 1;
    __shared__ float As[BM * LD];
    __shared__ float Bs[BN * LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
        for (int i = 0; i < (BM * BK) / 256; ++i) {
            int e = tid + i * 256, r = e >> 5, c = e & 31;
            int gm = m0 + r, gk = k0 + c;
            As[r * LD + c] = (gm < M && gk < K) ? A[(size_t)gm * lda + gk] : 0.f;
        }
        if (B_IS_NK) {
#pragma unroll
            for (int i = 0; i < (BN * BK) / 256; ++i) {
                int e = tid + i * 256, r = e >> 5, c = e & 31;
                int gn = n0 + r, gk = k0 + c;
                Bs[r * LD + c] = (gn < N && gk < K) ? B[(size_t)gn * ldb + gk] : 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < (BN * BK) / 256; ++i) {
                int e = tid + i * 256, kk = e / BN, nn = e % BN;
                int gn = n0 + nn, gk = k0 + kk;
                Bs[nn * LD + kk] = (gn < N && gk < K) ? B[(size_t)gk * ldb + gn] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[((wm * TM + i) * 32 + (lane & 31)) * LD + kk + (lane >> 5)];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[((wn * TN + j) * 32 + (lane & 31)) * LD + kk + (lane >> 5)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                int n = n0 + (wn * TN + j) * 32 + (lane & 31);
                if (m < M && n < N) {
                    float v = acc[i][j][r];
                    if (EPI == EPI_BIAS || EPI == EPI_BIAS_LRELU) v += aux[n];
                    if (EPI == EPI_BIAS_LRELU) v = v > 0.f ? v : 0.2f * v;
                    if (EPI == EPI_MASK_LRELU) v *= (aux[(size_t)m * ldaux + n] > 0.f) ? 1.f : 0.2f;
                    float* dst = C + (size_t)m * ldc + n;
                    if (EPI == EPI_ACCUM) v += *dst;
                    *dst = v;
                }
            }
}

// Variant for the small / skinny products of the optimiser (M ~ 1e3 rows, N from 32 to 1500,
// K up to 1500): one 32x32 output tile per workgroup, the four waves split each 128-deep K slab
// between them (intra-workgroup split-K) and are summed in a fixed order through LDS, so there are
// 4x more workgroups and 4x fewer barrier rounds than with 64x64 tiles, results stay reproducible,
// and the next slab's global loads are in flight while the current one is multiplied.
template <bool B_IS_NK, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_mfma_ksplit_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C,
    int ldc, int M, int N, int K, const float* __restrict__ aux, int ldaux) {
    constexpr int BM = 32, BN = 32, BK = 128, LD = BK + 1, NLD = (BM * BK) / 256;   // 16 loads per operand per thread
    __shared__ float As[BM * LD];
    __shared__ float Bs[BN * LD];
    __shared__ float Red[3][BM * BN];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float ra[NLD], rb[NLD];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int e = tid + i * 256, r = e >> 7, c = e & 127;
            int gm = m0 + r, gk = k0 + c;
            ra[i] = (gm < M && gk < K) ? A[(size_t)gm * lda + gk] : 0.f;
        }
        if (B_IS_NK) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                int e = tid + i * 256, r = e >> 7, c = e & 127;
                int gn = n0 + r, gk = k0 + c;
                rb[i] = (gn < N && gk < K) ? B[(size_t)gn * ldb + gk] : 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                int e = tid + i * 256, kk = e >> 5, nn = e & 31;
                int gn = n0 + nn, gk = k0 + kk;
                rb[i] = (gn < N && gk < K) ? B[(size_t)gk * ldb + gn] : 0.f;
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int e = tid + i * 256;
            As[(e >> 7) * LD + (e & 127)] = ra[i];
            if (B_IS_NK) Bs[(e >> 7) * LD + (e & 127)] = rb[i];
            else Bs[(e & 31) * LD + (e >> 5)] = rb[i];
        }
    };
    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        lstore();
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);                    // in flight during the MFMAs below
        const int kq = wave * 32;                           // this wave's quarter of the slab
#pragma unroll 4
        for (int kk = 0; kk < 32; kk += 2) {
            float a = As[(lane & 31) * LD + kq + kk + (lane >> 5)];
            float b = Bs[(lane & 31) * LD + kq + kk + (lane >> 5)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // fixed-order reduction of the four K-quarters: ((w0 + w1) + w2) + w3
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Red[wave - 1][r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r] + Red[0][r * 64 + lane];
            v += Red[1][r * 64 + lane];
            v += Red[2][r * 64 + lane];
            int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            int n = n0 + (lane & 31);
            if (m < M && n < N) {
                if (EPI == EPI_BIAS || EPI == EPI_BIAS_LRELU) v += aux[n];
                if (EPI == EPI_BIAS_LRELU) v = v > 0.f ? v : 0.2f * v;
                if (EPI == EPI_MASK_LRELU) v *= (aux[(size_t)m * ldaux + n] > 0.f) ? 1.f : 0.2f;
                float* dst = C + (size_t)m * ldc + n;
                if (EPI == EPI_ACCUM) v += *dst;
                *dst = v;
            }
        }
    }
}

// Vectorised staging for the split-K scheme (32x32 tile): 16-byte global loads and ds_write_b128 / ds_read_b128 instead
// of one dword per instruction -- a quarter of the vector-memory and LDS instructions per slab (a timing ablation of the
// scalar kernel put ~1/3 of its time in the load + LDS-store phase, none in the fragment reads).  Needs 16-byte aligned
// operand rows (lda, ldb multiples of 4, K -- and N for a [K,N] B -- multiples of 4, aligned bases); the dispatcher
// falls back to the scalar kernel otherwise.  K order inside a wave's 32-deep quarter is permuted so that one float4
// feeds four MFMAs: MFMA j of group q multiplies k = 8q + j (lanes 0-31) and k = 8q + 4 + j (lanes 32-63) -- any
// pairing is valid as long as A and B use the same one; the sum over k is the same set of products.
template <bool B_IS_NK, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_mfma_ksplit_v4_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C,
    int ldc, int M, int N, int K, const float* __restrict__ aux, int ldaux) {
    constexpr int BM = 32, BN = 32, BK = 128, LD = BK + 4, LDB = BN + 4, NV = (BM * BK) / (4 * 256);   // 4 float4 per operand per thread
    __shared__ __attribute__((aligned(16))) float As[BM * LD];
    __shared__ __attribute__((aligned(16))) float Bs[B_IS_NK ? BN * LD : BK * LDB];
    __shared__ float Red[3][BM * BN];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ra[NV], rb[NV];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + i * 256, r = e >> 5, c4 = e & 31;
            const int gm = m0 + r, gk = k0 + 4 * c4;
            ra[i] = z4;
            if (gm < M && gk < K) ra[i] = *reinterpret_cast<const float4*>(A + (size_t)gm * lda + gk);
        }
        if (B_IS_NK) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = tid + i * 256, r = e >> 5, c4 = e & 31;
                const int gn = n0 + r, gk = k0 + 4 * c4;
                rb[i] = z4;
                if (gn < N && gk < K) rb[i] = *reinterpret_cast<const float4*>(B + (size_t)gn * ldb + gk);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = tid + i * 256, kk = e >> 3, n4 = e & 7;
                const int gn = n0 + 4 * n4, gk = k0 + kk;
                rb[i] = z4;
                if (gn < N && gk < K) rb[i] = *reinterpret_cast<const float4*>(B + (size_t)gk * ldb + gn);
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + i * 256;
            *reinterpret_cast<float4*>(&As[(e >> 5) * LD + 4 * (e & 31)]) = ra[i];
            if (B_IS_NK) *reinterpret_cast<float4*>(&Bs[(e >> 5) * LD + 4 * (e & 31)]) = rb[i];
            else *reinterpret_cast<float4*>(&Bs[(e >> 3) * LDB + 4 * (e & 7)]) = rb[i];
        }
    };
    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        lstore();
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);                    // in flight during the MFMAs below
        const int kq = wave * 32;                           // this wave's quarter of the slab
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kb = kq + 8 * q + 4 * half;
            const float4 a4 = *reinterpret_cast<const float4*>(&As[col * LD + kb]);
            float4 b4;
            if (B_IS_NK) b4 = *reinterpret_cast<const float4*>(&Bs[col * LD + kb]);
            else b4 = make_float4(Bs[kb * LDB + col], Bs[(kb + 1) * LDB + col], Bs[(kb + 2) * LDB + col], Bs[(kb + 3) * LDB + col]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // fixed-order reduction of the four K-quarters: ((w0 + w1) + w2) + w3
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Red[wave - 1][r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r] + Red[0][r * 64 + lane];
            v += Red[1][r * 64 + lane];
            v += Red[2][r * 64 + lane];
            int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            int n = n0 + col;
            if (m < M && n < N) {
                if (EPI == EPI_BIAS || EPI == EPI_BIAS_LRELU) v += aux[n];
                if (EPI == EPI_BIAS_LRELU) v = v > 0.f ? v : 0.2f * v;
                if (EPI == EPI_MASK_LRELU) v *= (aux[(size_t)m * ldaux + n] > 0.f) ? 1.f : 0.2f;
                float* dst = C + (size_t)m * ldc + n;
                if (EPI == EPI_ACCUM) v += *dst;
                *dst = v;
            }
        }
    }
}

// The same scheme on 16x16 output tiles (v_mfma_f32_16x16x4_f32): four times the workgroups and a 19 KB LDS
// footprint, for products whose 32x32 grid leaves most CUs with one or two workgroups -- there every K slab
// costs a full, exposed global-load latency (one slab = 2 x TS x 128 floats in flight per workgroup), and only
// more resident workgroups per CU hide it.  Same fixed-order sum of the four K-quarters.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <bool B_IS_NK, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_mfma_ksplit16_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C,
    int ldc, int M, int N, int K, const float* __restrict__ aux, int ldaux) {
    constexpr int TS = 16, BK = 128, LD = BK + 1, NLD = (TS * BK) / 256;   // 8 loads per operand per thread
    __shared__ float As[TS * LD];
    __shared__ float Bs[TS * LD];
    __shared__ float Red[3][TS * TS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.y * TS, n0 = blockIdx.x * TS;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    float ra[NLD], rb[NLD];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int e = tid + i * 256, r = e >> 7, c = e & 127;
            int gm = m0 + r, gk = k0 + c;
            ra[i] = (gm < M && gk < K) ? A[(size_t)gm * lda + gk] : 0.f;
        }
        if (B_IS_NK) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                int e = tid + i * 256, r = e >> 7, c = e & 127;
                int gn = n0 + r, gk = k0 + c;
                rb[i] = (gn < N && gk < K) ? B[(size_t)gn * ldb + gk] : 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                int e = tid + i * 256, kk = e >> 4, nn = e & 15;
                int gn = n0 + nn, gk = k0 + kk;
                rb[i] = (gn < N && gk < K) ? B[(size_t)gk * ldb + gn] : 0.f;
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int e = tid + i * 256;
            As[(e >> 7) * LD + (e & 127)] = ra[i];
            if (B_IS_NK) Bs[(e >> 7) * LD + (e & 127)] = rb[i];
            else Bs[(e & 15) * LD + (e >> 4)] = rb[i];
        }
    };
    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        lstore();
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);                    // in flight during the MFMAs below
        const int kq = wave * 32;                           // this wave's quarter of the slab
#pragma unroll
        for (int kk = 0; kk < 32; kk += 4) {
            float a = As[(lane & 15) * LD + kq + kk + (lane >> 4)];
            float b = Bs[(lane & 15) * LD + kq + kk + (lane >> 4)];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Red[wave - 1][r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[r] + Red[0][r * 64 + lane];
            v += Red[1][r * 64 + lane];
            v += Red[2][r * 64 + lane];
            int m = m0 + 4 * (lane >> 4) + r;               // C/D layout of the 16x16 tile: col = lane & 15, row = 4 * (lane >> 4) + reg
            int n = n0 + (lane & 15);
            if (m < M && n < N) {
                if (EPI == EPI_BIAS || EPI == EPI_BIAS_LRELU) v += aux[n];
                if (EPI == EPI_BIAS_LRELU) v = v > 0.f ? v : 0.2f * v;
                if (EPI == EPI_MASK_LRELU) v *= (aux[(size_t)m * ldaux + n] > 0.f) ? 1.f : 0.2f;
                float* dst = C + (size_t)m * ldc + n;
                if (EPI == EPI_ACCUM) v += *dst;
                *dst = v;
            }
        }
    }
}

// 16x16-tile split-K with the vectorised staging of gemm_f32_mfma_ksplit_v4_kernel.  v_mfma_f32_16x16x4_f32: lane group
// g = lane >> 4 supplies k = 4 s + g of step s; with one float4 per lane, MFMA j of group q multiplies k = 16 q + 4 g + j.
template <bool B_IS_NK, int EPI, int BK>
__global__ __launch_bounds__(256) void gemm_f32_mfma_ksplit16_v4_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C,
    int ldc, int M, int N, int K, const float* __restrict__ aux, int ldaux) {
    constexpr int TS = 16, LD = BK + 4, LDB = TS + 4, NV = (TS * BK) / (4 * 256), C4 = BK / 4, KW = BK / 4;   // NV float4 per operand per thread; C4 float4 per row; KW k per wave and slab
    __shared__ __attribute__((aligned(16))) float As[TS * LD];
    __shared__ __attribute__((aligned(16))) float Bs[B_IS_NK ? TS * LD : BK * LDB];
    __shared__ float Red[3][TS * TS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = lane >> 4, col = lane & 15;
    const int m0 = blockIdx.y * TS, n0 = blockIdx.x * TS;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ra[NV], rb[NV];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + i * 256, r = e / C4, c4 = e % C4;
            const int gm = m0 + r, gk = k0 + 4 * c4;
            ra[i] = z4;
            if (gm < M && gk < K) ra[i] = *reinterpret_cast<const float4*>(A + (size_t)gm * lda + gk);
        }
        if (B_IS_NK) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = tid + i * 256, r = e / C4, c4 = e % C4;
                const int gn = n0 + r, gk = k0 + 4 * c4;
                rb[i] = z4;
                if (gn < N && gk < K) rb[i] = *reinterpret_cast<const float4*>(B + (size_t)gn * ldb + gk);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = tid + i * 256, kk = e >> 2, n4 = e & 3;
                const int gn = n0 + 4 * n4, gk = k0 + kk;
                rb[i] = z4;
                if (gn < N && gk < K) rb[i] = *reinterpret_cast<const float4*>(B + (size_t)gk * ldb + gn);
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + i * 256;
            *reinterpret_cast<float4*>(&As[(e / C4) * LD + 4 * (e % C4)]) = ra[i];
            if (B_IS_NK) *reinterpret_cast<float4*>(&Bs[(e / C4) * LD + 4 * (e % C4)]) = rb[i];
            else *reinterpret_cast<float4*>(&Bs[(e >> 2) * LDB + 4 * (e & 3)]) = rb[i];
        }
    };
    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        lstore();
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);                    // in flight during the MFMAs below
        const int kq = wave * KW;                           // this wave's quarter of the slab
#pragma unroll
        for (int q = 0; q < KW / 16; ++q) {
            const int kb = kq + 16 * q + 4 * grp;
            const float4 a4 = *reinterpret_cast<const float4*>(&As[col * LD + kb]);
            float4 b4;
            if (B_IS_NK) b4 = *reinterpret_cast<const float4*>(&Bs[col * LD + kb]);
            else b4 = make_float4(Bs[kb * LDB + col], Bs[(kb + 1) * LDB + col], Bs[(kb + 2) * LDB + col], Bs[(kb + 3) * LDB + col]);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Red[wave - 1][r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[r] + Red[0][r * 64 + lane];
            v += Red[1][r * 64 + lane];
            v += Red[2][r * 64 + lane];
            int m = m0 + 4 * grp + r;
            int n = n0 + col;
            if (m < M && n < N) {
                if (EPI == EPI_BIAS || EPI == EPI_BIAS_LRELU) v += aux[n];
                if (EPI == EPI_BIAS_LRELU) v = v > 0.f ? v : 0.2f * v;
                if (EPI == EPI_MASK_LRELU) v *= (aux[(size_t)m * ldaux + n] > 0.f) ? 1.f : 0.2f;
                float* dst = C + (size_t)m * ldc + n;
                if (EPI == EPI_ACCUM) v += *dst;
                *dst = v;
            }
        }
    }
}

// Wide "NN" product C[M,N] = A[M,K] x B[K,N] with N >> M (the full-mesh pose blendshapes:
// M = frames ~ 1e3, K = 486, N = 3V = 31 425).  B (61 MB) is the only operand that does not fit in
// L2, so the blockIdx -> tile map keeps all M-tiles of one 128-column B panel on ONE XCD, back to
// back (blocks b, b+8, ..., b+8*(MT-1) share n-tile): the panel is fetched from HBM once per XCD L2
// and reused by the other M-tiles.  128x128 tiles, 2x2 waves x 2x2 MFMA accumulators, next K slab
// prefetched into registers during the MFMAs, coalesced loads along N for B.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_mfma_wide_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C,
    int ldc, int M, int N, int K, const float* __restrict__ aux, int ldaux) {
    constexpr int BM = 128, BN = 128, BK = 32, LD = BK + 1, NL = (BM * BK) / 256;   // 16 loads per operand per thread
    __shared__ float As[BM * LD];
    __shared__ float Bs[BN * LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int MT = (M + BM - 1) / BM, NT = (N + BN - 1) / BN;
    // XCD-aware map: xcd = b % 8 owns n-tiles {xcd, xcd + 8, ...}; within an XCD consecutive slots walk M
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int nt = (slot / MT) * 8 + xcd, mt = slot % MT;
    if (nt >= NT) return;
    const int m0 = mt * BM, n0 = nt * BN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float ra[NL], rb[NL];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            int e = tid + i * 256, r = e >> 5, c = e & 31;          // A: 32 consecutive k per row
            int gm = m0 + r, gk = k0 + c;
            ra[i] = (gm < M && gk < K) ? A[(size_t)gm * lda + gk] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            int e = tid + i * 256, kk = e >> 7, nn = e & 127;       // B: 128 consecutive n per k row
            int gn = n0 + nn, gk = k0 + kk;
            rb[i] = (gn < N && gk < K) ? B[(size_t)gk * ldb + gn] : 0.f;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            int e = tid + i * 256;
            As[(e >> 5) * LD + (e & 31)] = ra[i];
            Bs[(e & 127) * LD + (e >> 7)] = rb[i];
        }
    };
    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        lstore();
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);
#pragma unroll 4
        for (int kk = 0; kk < BK; kk += 2) {
            float a[2], bb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[((wm * 2 + i) * 32 + (lane & 31)) * LD + kk + (lane >> 5)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bb[j] = Bs[((wn * 2 + j) * 32 + (lane & 31)) * LD + kk + (lane >> 5)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                int n = n0 + (wn * 2 + j) * 32 + (lane & 31);
                if (m < M && n < N) {
                    float v = acc[i][j][r];
                    if (EPI == EPI_ACCUM) v += C[(size_t)m * ldc + n];
                    C[(size_t)m * ldc + n] = v;
                }
            }
}

// gemm_f32_mfma_wide_kernel with vectorised staging (see gemm_f32_mfma_ksplit_v4_kernel): 16-byte global loads, A
// fragments as ds_read_b128, B stored transposed ([n][k]) so its fragments are ds_read_b128 too; the K order of a 32-deep
// slab is permuted as there (MFMA j of group q: k = 8q + j | 8q + 4 + j).  Needs lda, ldb multiples of 4 and 16-byte
// aligned bases; B rows may be read up to ldb (the caller pads posedirs rows to a multiple of 4 with zeros).
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_mfma_wide_v4_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C,
    int ldc, int M, int N, int K, const float* __restrict__ aux, int ldaux) {
    constexpr int BM = 128, BN = 128, BK = 32, LD = BK + 4, NV = (BM * BK) / (4 * 256);   // 4 float4 per operand per thread
    __shared__ __attribute__((aligned(16))) float As[BM * LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int MT = (M + BM - 1) / BM, NT = (N + BN - 1) / BN;
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int nt = (slot / MT) * 8 + xcd, mt = slot % MT;
    if (nt >= NT) return;
    const int m0 = mt * BM, n0 = nt * BN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ra[NV], rb[NV];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + i * 256, r = e >> 3, c4 = e & 7;              // A: 8 float4 (32 k) per row
            const int gm = m0 + r, gk = k0 + 4 * c4;
            ra[i] = z4;
            if (gm < M && gk < K) ra[i] = *reinterpret_cast<const float4*>(A + (size_t)gm * lda + gk);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + i * 256, kk = e >> 5, n4 = e & 31;            // B: 32 float4 (128 n) per k row
            const int gn = n0 + 4 * n4, gk = k0 + kk;
            rb[i] = z4;
            if (gn < ldb && gk < K) rb[i] = *reinterpret_cast<const float4*>(B + (size_t)gk * ldb + gn);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + i * 256;
            *reinterpret_cast<float4*>(&As[(e >> 3) * LD + 4 * (e & 7)]) = ra[i];
            const int kk = e >> 5, n4 = e & 31;                               // transposed: Bs[n][k]
            Bs[(4 * n4) * LD + kk] = rb[i].x; Bs[(4 * n4 + 1) * LD + kk] = rb[i].y;
            Bs[(4 * n4 + 2) * LD + kk] = rb[i].z; Bs[(4 * n4 + 3) * LD + kk] = rb[i].w;
        }
    };
    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        lstore();
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kb = 8 * q + 4 * half;
            float4 a4[2], b4[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a4[i] = *reinterpret_cast<const float4*>(&As[((wm * 2 + i) * 32 + col) * LD + kb]);
#pragma unroll
            for (int j = 0; j < 2; ++j) b4[j] = *reinterpret_cast<const float4*>(&Bs[((wn * 2 + j) * 32 + col) * LD + kb]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, b4[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, b4[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, b4[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, b4[j].w, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                int n = n0 + (wn * 2 + j) * 32 + col;
                if (m < M && n < N) {
                    float v = acc[i][j][r];
                    if (EPI == EPI_ACCUM) v += C[(size_t)m * ldc + n];
                    C[(size_t)m * ldc + n] = v;
                }
            }
}

template <bool NK, int EPI>
static inline hipError_t gemm_dispatch_tile(const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                            int M, int N, int K, const float* aux, int ldaux, hipStream_t st) {
    if (M <= 0 || N <= 0) return hipSuccess;
    // skinny / small outputs (every product of the optimiser loop): 32x32 tiles, intra-workgroup split-K
    if ((long long)M * N < 64LL * 64 * 1024 && K >= 32) {
        // 16x16 tiles read twice the operand bytes per MAC: they pay off while the grid is small (latency-bound) and
        // K is short; the K = 1500 pose-blend data-gradient at 1024 frames (528 32x32 tiles) is L2-bandwidth-bound
        // with them (35.9 us vs 31.3).  FDCAP_GEMM_T16 overrides the tile-count threshold (A/B).
        static std::atomic<int> t16{-1};
        if (t16 < 0) { const char* e = getenv("FDCAP_GEMM_T16"); t16 = e ? atoi(e) : 1024; }
        const long long tiles32 = (long long)((N + 31) / 32) * ((M + 31) / 32);
        static std::atomic<int> v4{-1};                                  // FDCAP_GEMM_V4=0: scalar staging only (A/B)
        if (v4 < 0) { const char* e = getenv("FDCAP_GEMM_V4"); v4 = (e && e[0] == '0') ? 0 : 1; }
        const bool aligned = (lda % 4 == 0) && (ldb % 4 == 0) && (K % 4 == 0) && (NK || N % 4 == 0) &&
                             ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
        if (tiles32 < t16 / 4 || (tiles32 < t16 && K <= 768)) {
            dim3 grid((N + 15) / 16, (M + 15) / 16);
            static std::atomic<int> bk256{-1};                           // FDCAP_GEMM_BK256=0: 128-deep slabs only (A/B)
            if (bk256 < 0) { const char* e = getenv("FDCAP_GEMM_BK256"); bk256 = (e && e[0] == '0') ? 0 : 1; }
            // deep products on small grids: half the slab rounds (barriers + exposed load latencies); with >= 1024 workgroups
            // in flight the shorter slabs overlap better (measured: N = 512 layers 14.1 vs 15.0 us, N = 126 / 32 layers 6.8 / 7.1 vs 5.8 / 6.7)
            if (v4 && aligned && bk256 && K >= 512 && (long long)grid.x * grid.y < 1024)
                hipLaunchKernelGGL((gemm_f32_mfma_ksplit16_v4_kernel<NK, EPI, 256>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc,
                                   M, N, K, aux, ldaux);
            else if (v4 && aligned)
                hipLaunchKernelGGL((gemm_f32_mfma_ksplit16_v4_kernel<NK, EPI, 128>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc,
                                   M, N, K, aux, ldaux);
            else
                hipLaunchKernelGGL((gemm_f32_mfma_ksplit16_kernel<NK, EPI>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K,
                                   aux, ldaux);
            return hipGetLastError();
        }
        dim3 grid((N + 31) / 32, (M + 31) / 32);
        if (v4 && aligned)
            hipLaunchKernelGGL((gemm_f32_mfma_ksplit_v4_kernel<NK, EPI>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K,
                               aux, ldaux);
        else
            hipLaunchKernelGGL((gemm_f32_mfma_ksplit_kernel<NK, EPI>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K,
                               aux, ldaux);
        return hipGetLastError();
    }
    // wide NN products (full-mesh pose blendshapes): XCD-aware B-panel reuse + register prefetch
    if (!NK && (EPI == EPI_STORE || EPI == EPI_ACCUM) && (long long)M * N >= 128LL * 128 * 512 && N >= 8 * 128) {
        const int MT = (M + 127) / 128, NT = (N + 127) / 128;
        const int blocks = (NT + 7) / 8 * 8 * MT;
        static std::atomic<int> wv4{-1};                                 // FDCAP_GEMM_V4=0: scalar staging only (A/B)
        if (wv4 < 0) { const char* e = getenv("FDCAP_GEMM_V4"); wv4 = (e && e[0] == '0') ? 0 : 1; }
        if (wv4 && lda % 4 == 0 && ldb % 4 == 0 && K % 4 == 0 &&
            ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0)
            hipLaunchKernelGGL((gemm_f32_mfma_wide_v4_kernel<EPI>), dim3(blocks), dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K,
                               aux, ldaux);
        else
            hipLaunchKernelGGL((gemm_f32_mfma_wide_kernel<EPI>), dim3(blocks), dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K,
                               aux, ldaux);
        return hipGetLastError();
    }
    // large outputs: 128x128 workgroup tiles (4 accumulators per wave); medium ones: 64x64
    if ((long long)M * N >= 128LL * 128 * 512) {
        dim3 grid((N + 127) / 128, (M + 127) / 128);
        hipLaunchKernelGGL((gemm_f32_mfma_kernel<NK, EPI, 2, 2>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc,
                           M, N, K, aux, ldaux);
    } else {
        dim3 grid((N + 63) / 64, (M + 63) / 64);
        hipLaunchKernelGGL((gemm_f32_mfma_kernel<NK, EPI, 1, 1>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc,
                           M, N, K, aux, ldaux);
    }
    return hipGetLastError();
}

// b_is_nk: B stored [N,K] (C = A B^T) else [K,N]
static inline hipError_t gemm_f32(bool b_is_nk, int epi, const float* A, int lda, const float* B, int ldb, float* C,
                                  int ldc, int M, int N, int K, const float* aux, int ldaux, hipStream_t st) {
#define FDC_GEMM_CASE(NK, E) \
    if (b_is_nk == NK && epi == E) return gemm_dispatch_tile<NK, E>(A, lda, B, ldb, C, ldc, M, N, K, aux, ldaux, st);
    FDC_GEMM_CASE(true, EPI_STORE) FDC_GEMM_CASE(true, EPI_BIAS) FDC_GEMM_CASE(true, EPI_BIAS_LRELU)
    FDC_GEMM_CASE(true, EPI_ACCUM)
    FDC_GEMM_CASE(false, EPI_STORE) FDC_GEMM_CASE(false, EPI_MASK_LRELU) FDC_GEMM_CASE(false, EPI_ACCUM)
#undef FDC_GEMM_CASE
    return hipErrorInvalidValue;
}

}  // namespace fdc
