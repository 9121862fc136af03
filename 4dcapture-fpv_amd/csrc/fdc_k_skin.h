// Skinning kernels: linear-blend skinning + scale + world transform (skin_fwd_kernel) and the backward with the contact
// robustifier's gradient in its three forms -- any vertex set (skin_bwd_kernel), contact-set sizes (skin_bwd_small_kernel), and the
// packed 16-byte form the loop runs (skin_bwd_vec_kernel).  Part of the single translation unit csrc/fdcap.hip.
#pragma once

namespace {

// thread per (frame, vertex), 256-thread workgroups.  Vout layout [rows, nv, 3].  world = 0: body frame (+transl only)
__global__ __launch_bounds__(256) void skin_fwd_kernel(SkinModel sm, int nv, const float* __restrict__ X, int ldx, int beta_off, int transl_off,
                                const float* __restrict__ Voff, const float* __restrict__ A,
                                const float* __restrict__ M, const float* __restrict__ scale, int row0, int world,
                                float* __restrict__ Vout) {
    // the frame's 55 skinning transforms staged in LDS once per block: per vertex they are reached through its joint
    // ids (a dependent load chain from global memory otherwise).  By LDS-DMA, with the vertex's own loads issued before
    // the barrier: one cold round trip (the copy loop that was here waited for each of its three trips, then the
    // per-vertex loads made a fourth)
    __shared__ __attribute__((aligned(16))) float sA[NJ * 12];
    const int c = blockIdx.x * 256 + threadIdx.x, cc = min(c, nv - 1);
    const int r = row0 + blockIdx.y;
    glds_wg<1, 4, 16>(A + (size_t)r * NJ * 12, sA, NJ * 3);
    const float* x = X + (size_t)r * ldx;
    const V3 transl = v3(x[transl_off], x[transl_off + 1], x[transl_off + 2]);
    float Mr[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};      // registers: a pointer that is either global or a local array turns into flat loads
    if (world) {
#pragma unroll
        for (int e = 0; e < 12; ++e) Mr[e] = M[(size_t)r * 12 + e];
    }
    const float sc_v = world ? *scale : 1.f;
    float* o = Vout + ((size_t)r * nv + c) * 3;
    if (sm.vpack && !FDC_SKIN_HAS_S(sm)) {
        // packed per-vertex constants (two 16-byte loads instead of eleven 4-byte ones); same terms, same order
        const float4* const vp4 = (const float4*)sm.vpack;
        const int G = (sm.K + 3) >> 2;                                       // (wave-uniform)
        const float4 p0 = vp4[cc], p1 = vp4[nv + cc];
        float4 p2 = make_float4(0.f, 0.f, 0.f, 0.f), p3 = p2, pj = p2;       // K > 4: more weight planes + the ids' plane
        if (G >= 2) { p2 = vp4[(size_t)2 * nv + cc]; pj = vp4[(size_t)(G + 1) * nv + cc]; }
        if (G >= 3) p3 = vp4[(size_t)3 * nv + cc];
        const float* vo = Voff + ((size_t)r * nv + cc) * 3;
        const float v0 = vo[0], v1 = vo[1], v2 = vo[2];
        __syncthreads();
        if (c >= nv) return;
        const float px = p0.x + v0, py = p0.y + v1, pz = p0.z + v2;
        const unsigned jb[3] = {__float_as_uint(p0.w), __float_as_uint(pj.x), __float_as_uint(pj.y)};
        const float w12[12] = {p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w, p3.x, p3.y, p3.z, p3.w};
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k)
            if (k < sm.K) {
                const float* a = sA + 12 * ((jb[k >> 2] >> (8 * (k & 3))) & 255u);
#pragma unroll
                for (int e = 0; e < 12; ++e) T[e] += w12[k] * a[e];
            }
        const V3 vb = v3(T[0] * px + T[1] * py + T[2] * pz + T[3], T[4] * px + T[5] * py + T[6] * pz + T[7],
                         T[8] * px + T[9] * py + T[10] * pz + T[11]) + transl;
        const V3 sv = sc_v * vb;
        o[0] = Mr[0] * sv.x + Mr[1] * sv.y + Mr[2] * sv.z + Mr[3];
        o[1] = Mr[4] * sv.x + Mr[5] * sv.y + Mr[6] * sv.z + Mr[7];
        o[2] = Mr[8] * sv.x + Mr[9] * sv.y + Mr[10] * sv.z + Mr[11];
        return;
    }
    __syncthreads();
    if (c >= nv) return;
    SkinFwd f = skin_forward_vertex(sm, c, x + beta_off, Voff + ((size_t)r * nv + c) * 3, sA, transl, Mr, sc_v);
    o[0] = f.vw.x; o[1] = f.vw.y; o[2] = f.vw.z;
}


// Blend product + skinning of the contact set in ONE launch (r5): panel_gemm3_rb2_kernel's product for a block of 32 frames x 64
// vertices, then skin_fwd_kernel's arithmetic on the block while it is still in LDS -- the offsets leave once (Voff, for the
// backward) next to the world vertices (Vw), and the separate skinning launch (5.5 us: launch, a cold read of the 6 MB just
// written, its own 6 MB left dirty) is gone.  For that a lane must see x, y and z of a vertex: the static operand's columns are
// permuted on the host (SkinSet::pn_fwdS) -- column block b = vertices 64 b .. 64 b + 63, tile 3 g + c of the block = component c of
// vertices 64 b + 16 g .. + 15 -- and the three component tiles of a vertex group meet in LDS.  Same products in the same order
// per column, same skinning expressions: Voff and Vw bit for bit those of the two launches (tests/test_gpu_parity.py).
// Grid 8 x ceil(M / 32) (blockIdx & 7 = column block = XCD: nv <= 512), 768 threads; K <= 4 weights per vertex (two vpack planes),
// joints < ja staged per frame.  Dynamic LDS: 2 images | sA [32][ja][12] | sM [32][12] | sT [32][4] | sVP [2][64][4] | sX [3][32][64].
__global__ __launch_bounds__(768) void blend_skin_fwd_kernel(const float* __restrict__ PF, int M, PanelB3 B, SkinModel sm, int nv, int ja,
                                                             const float* __restrict__ X, int ldx, int transl_off,
                                                             const float* __restrict__ A, const float* __restrict__ Mw,
                                                             const float* __restrict__ scale, int row0, float* __restrict__ Voff,
                                                             float* __restrict__ Vw) {
    extern __shared__ __attribute__((aligned(16))) uint4 bs_lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int cb = blockIdx.x & 7, m0 = (int)(blockIdx.x >> 3) * 32;
    const int ncb = (B.ntile + 11) / 12;
    if (cb >= ncb) return;
    constexpr int K = NPFX, kpad = (NPFX + 31) & ~31, nst = kpad >> 5, pstride = (kpad >> 3) * 16;
    const int img = pnf_img_u4(kpad);
    float* const sA = (float*)(bs_lds + 2 * img);
    float* const sM = sA + 32 * ja * 12;
    float* const sT = sM + 32 * 12;
    float* const sVP = sT + 32 * 4;                             // [2][64] float4
    float* const sX = sVP + 2 * 64 * 4;
    const float* const PFr = PF + (size_t)row0 * NPFX;
    PnRing3<2> rg;
    const int tile = cb * 12 + wave;
    panel3_prefetch<2>(rg, B.f + (size_t)min(tile, B.ntile - 1) * B.nst * PNF * 64, nst, lane);
    const f32x4_t ts = PnF::tile_isc(B, tile, g);
    // the same batch: the frames' skinning transforms (joints < ja), world transforms and translations by LDS-DMA, this thread's
    // vertex constants into registers (a thread serves vertex tid & 63 of the block for every frame it is dealt)
    {
        const int nu = 32 * ja * 3;                              // 16-byte units of sA: unit u = frame f (u / (3 ja)), word w
        for (int u0 = 0; u0 < nu; u0 += 768) {
            const int u = u0 + tid;
            if (u < nu) {
                const int f = u / (3 * ja), w = u - f * 3 * ja;
                const float* src = A + (size_t)(row0 + min(m0 + f, M - 1)) * NJ * 12 + 4 * w;
                __builtin_amdgcn_global_load_lds((fdc_gptr_t)src, (fdc_lptr_t)((char*)sA + 16 * (u0 + 64 * wave)), 16, 0, 0);
            }
        }
        if (tid < 96) {                                          // (waves 0, 1: wave-uniform destinations)
            const int f = tid / 3, w = tid - 3 * f;
            const float* src = Mw + (size_t)(row0 + min(m0 + f, M - 1)) * 12 + 4 * w;
            __builtin_amdgcn_global_load_lds((fdc_gptr_t)src, (fdc_lptr_t)((char*)sM + 16 * 64 * wave), 16, 0, 0);
        } else if (tid >= 128 && tid < 128 + 128) {              // (waves 2, 3) translations: [32][4] words, the fourth unused
            const int q = tid - 128, f = q >> 2, w = min(q & 3, 2);
            const float* src = X + (size_t)(row0 + min(m0 + f, M - 1)) * ldx + transl_off + w;
            __builtin_amdgcn_global_load_lds((fdc_gptr_t)src, (fdc_lptr_t)((char*)sT + 4 * 64 * (wave - 2)), 4, 0, 0);
        } else if (tid >= 256 && tid < 256 + 128) {              // (waves 4, 5) the block's 64 vertices' packed constants, two planes
            const int q = tid - 256;
            const float* src = sm.vpack + 4 * ((size_t)(q >> 6) * nv + min(cb * 64 + (q & 63), nv - 1));
            __builtin_amdgcn_global_load_lds((fdc_gptr_t)src, (fdc_lptr_t)((char*)sVP + 16 * 64 * (wave - 4)), 16, 0, 0);
        }
    }
    const int vv = tid & 63, v = cb * 64 + vv;
    const float sc_v = *scale;
    PnF::stage<768, 2, 2>(bs_lds, img, PFr, NPFX, m0, M, 0, K, kpad, tid);
    __syncthreads();
    PnF::Acc acc[2] = {PnF::zero(), PnF::zero()};
    if (tile < B.ntile) panel3_mma<2, 2>(bs_lds, pstride, img, rg, nst, acc, lane);
    {   // component c = wave % 3 of vertices 16 (wave / 3) + 4 g .. + 3, frames 16 rb + j
        const int c = wave % 3, vg = wave / 3;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const f32x4_t o = PnF::value(acc[rb], PnF::row_isc(bs_lds + (size_t)rb * img, pstride, j), ts);
            *(float4*)(sX + (size_t)((c * 32 + 16 * rb + j) * 64 + vg * 16 + 4 * g)) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int p = tid + 768 * k, f = p >> 6;                 // pair (frame f of the block, vertex vv): 2048 pairs
        if (f < 32 && m0 + f < M && v < nv) {
            const size_t r = (size_t)(row0 + m0 + f);
            const float v0 = sX[(0 * 32 + f) * 64 + vv], v1 = sX[(1 * 32 + f) * 64 + vv], v2 = sX[(2 * 32 + f) * 64 + vv];
            const float4 p0 = ((const float4*)sVP)[vv], p1 = ((const float4*)sVP)[64 + vv];
            float* const vo = Voff + (r * nv + v) * 3;
            vo[0] = v0; vo[1] = v1; vo[2] = v2;
            // skin_fwd_kernel's packed path, K <= 4, term by term
            const float px = p0.x + v0, py = p0.y + v1, pz = p0.z + v2;
            const unsigned jb = __float_as_uint(p0.w);
            const float w4[4] = {p1.x, p1.y, p1.z, p1.w};
            float T[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                if (kk < sm.K) {
                    const float* a = sA + (size_t)(f * ja + (int)((jb >> (8 * kk)) & 255u)) * 12;
#pragma unroll
                    for (int e = 0; e < 12; ++e) T[e] += w4[kk] * a[e];
                }
            const V3 transl = v3(sT[4 * f], sT[4 * f + 1], sT[4 * f + 2]);
            const V3 vb = v3(T[0] * px + T[1] * py + T[2] * pz + T[3], T[4] * px + T[5] * py + T[6] * pz + T[7],
                             T[8] * px + T[9] * py + T[10] * pz + T[11]) + transl;
            const V3 sv = sc_v * vb;
            const float* Mr = sM + 12 * f;
            float* const o = Vw + (r * nv + v) * 3;
            o[0] = Mr[0] * sv.x + Mr[1] * sv.y + Mr[2] * sv.z + Mr[3];
            o[1] = Mr[4] * sv.x + Mr[5] * sv.y + Mr[6] * sv.z + Mr[7];
            o[2] = Mr[8] * sv.x + Mr[9] * sv.y + Mr[10] * sv.z + Mr[11];
        }
    }
}
static inline size_t blend_skin_lds_bytes(int ja) { return pnf_lds_bytes((NPFX + 31) & ~31, 2) + (size_t)(32 * ja * 12 + 32 * 12 + 32 * 4 + 2 * 64 * 4 + 3 * 32 * 64) * sizeof(float); }

__device__ __forceinline__ float wave_sum(float v) { return wave_sum64(v); }

// workgroup per frame: skinning + world-transform backward of d loss / d world vertices, reduced over
// the frame's vertex set.  dVw / dVoff are [rows, nc, 3] and may alias (each thread reads its vertex's
// gradient before it writes the vertex's pose-blend gradient).
// CONTACT: d loss / d world vertex is the contact robustifier's gradient (:295), formed here from the NN
// result (Vw, dist, idx -> scene point) instead of being read from dVw; its un-weighted sum goes to
// loss_rows[r][3] when that is non-null (logging iterations only; see ParamLossIn).
// nnpt (optional): the neighbours' coordinates as the NN kernel keeps them ([q] {x, y, z, -}, coalesced) instead of the
// dependent gather scene[idx[q]].
struct ContactGradIn { const float* Vw; const float* dist; const int* idx; const float4* scene; const float4* nnpt; float coef; float* loss_rows; };
constexpr int SKB_NACC = NBETA + 3 + 12 + 1;   // dbeta, dtransl, dM, ds
constexpr int SKB_VCH = 1024;                  // vertices per LDS chunk
constexpr int SKP_STRIDE = 688;                // floats of a (frame, chunk) partial of the split form: dA [660] | the SKB_NACC sums | contact term | pad
static_assert(NJ * 12 + SKB_NACC + 1 <= SKP_STRIDE, "partial record");
// SPLIT (r5; vertex sets of more than one chunk: the full mesh -- BASELINE config 5's contact set, mode 'local', the body-model
// operator's backward): grid (frames, chunks), a workgroup takes ONE chunk of one frame and leaves its sums in `part`
// [frame][chunk][SKP_STRIDE]; skin_bwd_reduce_kernel adds the chunks in ascending order.  One workgroup per frame walking all
// 10 475 vertices (41 per thread, eleven dA reductions in a row) put two workgroups on a CU for half a millisecond: 541 us per
// launch at 512 frames, 17 % of a config-5 iteration (profiles/r5_c5_kernel_trace_stats.txt).
#ifndef FDC_SKB_WFQ
#define FDC_SKB_WFQ 4
#endif
constexpr int SKB_WFQ = FDC_SKB_WFQ;                   // quads (of four steps) per block of the matrix-form dA (measured at config 5: 4 -> 231 us, 8 -> 239, 16 -> 261)
// skin_forward_vertex for a vertex whose constants come packed (SkinModel::vpack planes 0 and 1: K <= 4), the skinning transforms
// read as 16-byte LDS rows: the same terms in the same order (padding weights are zeros on joint 0, as in the unpacked lists)
__device__ __forceinline__ SkinFwd skin_forward_vertex_packed(const float4 p0, const float4 p1, int K, const float* __restrict__ voff,
                                                              const float* sA /* LDS, 16-byte aligned */, V3 transl, const float* M, float scale) {
    SkinFwd r;
    const float px = p0.x + voff[0], py = p0.y + voff[1], pz = p0.z + voff[2];
    r.vp = v3(px, py, pz);
    const unsigned jb = __float_as_uint(p0.w);
    const float w4[4] = {p1.x, p1.y, p1.z, p1.w};
#pragma unroll
    for (int e = 0; e < 12; ++e) r.T[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < K) {
            const float4* a = (const float4*)(sA + 12 * ((jb >> (8 * k)) & 255u));
            const float4 a0 = a[0], a1 = a[1], a2 = a[2];
            const float w = w4[k];
            r.T[0] += w * a0.x; r.T[1] += w * a0.y; r.T[2] += w * a0.z; r.T[3] += w * a0.w;
            r.T[4] += w * a1.x; r.T[5] += w * a1.y; r.T[6] += w * a1.z; r.T[7] += w * a1.w;
            r.T[8] += w * a2.x; r.T[9] += w * a2.y; r.T[10] += w * a2.z; r.T[11] += w * a2.w;
        }
    const V3 vl = v3(r.T[0] * px + r.T[1] * py + r.T[2] * pz + r.T[3], r.T[4] * px + r.T[5] * py + r.T[6] * pz + r.T[7],
                     r.T[8] * px + r.T[9] * py + r.T[10] * pz + r.T[11]);
    r.vb = vl + transl;
    const V3 sv = scale * r.vb;
    r.vw = v3(M[0] * sv.x + M[1] * sv.y + M[2] * sv.z + M[3], M[4] * sv.x + M[5] * sv.y + M[6] * sv.z + M[7],
              M[8] * sv.x + M[9] * sv.y + M[10] * sv.z + M[11]);
    return r;
}

#ifndef FDC_SKB_OCC
#define FDC_SKB_OCC 5
#endif
#ifndef FDC_SKB_PACKED
#define FDC_SKB_PACKED 1
#endif
constexpr int SKB_ROW = 6;                             // floats per vertex of the factored dT rows (matrix-form dA): gv[3] | vp[3]
template <bool CONTACT, bool SPLIT = false>
__global__ __launch_bounds__(256, FDC_SKB_OCC) void skin_bwd_kernel(SkinModel sm, int nc, const float* __restrict__ X,
                                                       const float* __restrict__ Voff, const float* __restrict__ A,
                                                       const float* __restrict__ M, const float* __restrict__ scale,
                                                       int row0, const float* dVw, float* dVoff,
                                                       float* __restrict__ dA, float* __restrict__ dbeta_v,
                                                       float* __restrict__ dtransl_v, float* __restrict__ dMv,
                                                       float* __restrict__ dsv, ContactGradIn cg, float* __restrict__ part = nullptr) {
    constexpr int VCH = SKB_VCH;
    extern __shared__ float sdT[];                 // [min(nc, VCH) * 12] (dynamic: 500 contact vertices leave room for 6 workgroups per CU)
    __shared__ float sdA[NJ * 12];
    __shared__ float sred[4][SKB_NACC];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = row0 + blockIdx.x;
    FDC_FR_STAMP(2, 0);
    const float* x = X + (size_t)r * XDIM;
    const float s = *scale;
    V3 transl = v3(x[X_TRANSL], x[X_TRANSL + 1], x[X_TRANSL + 2]);
    float acc[SKB_NACC];
    float cterm = 0.f;
    // lane j of every wave: joint j's range in the transposed weight list (loaded once, while the vertex phase runs)
    const int jlo = lane < NJ ? sm.csc_start[lane] : 0, jhi = lane < NJ ? sm.csc_start[lane + 1] : 0;
    __shared__ __attribute__((aligned(16))) float sAf[NJ * 12];   // this frame's skinning transforms (see skin_fwd_kernel)
    __shared__ float sM[12];                        // ... and its world matrix (r6: every vertex read it from global memory twice)
    for (int i = tid; i < NJ * 12; i += 256) sAf[i] = A[(size_t)r * NJ * 12 + i];
    if (tid < 12) sM[tid] = M[(size_t)r * 12 + tid];
    const bool packed = FDC_SKB_PACKED && sm.vpack && sm.K <= 4;             // (kernel-uniform)
    const float4* const vp4 = (const float4*)sm.vpack;
#pragma unroll
    for (int i = 0; i < SKB_NACC; ++i) acc[i] = 0.f;
    for (int i = tid; i < NJ * 12; i += 256) sdA[i] = 0.f;
    __syncthreads();
    FDC_FR_STAMP(2, 1);
    const int c_lo = SPLIT ? (int)blockIdx.y * VCH : 0, c_hi = SPLIT ? min(nc, c_lo + VCH) : nc;
    for (int c0 = c_lo; c0 < c_hi; c0 += VCH) {
        const int c1 = min(c_hi, c0 + VCH);
#ifdef FDC_SKB_VUNROLL
#pragma unroll FDC_SKB_VUNROLL
#endif
        for (int c = c0 + tid; c < c1; c += 256) {
            size_t qi = (size_t)r * nc + c;
            // every global load of this vertex goes out before the first use (the kernel is a chain of latencies: four
            // workgroups per CU, nothing else to run meanwhile): the NN result first, un-branched, then the skinning inputs
            float dq = 0.f, vwx = 0.f, vwy = 0.f, vwz = 0.f;
            int jq = -1;
            float4 pq = make_float4(0.f, 0.f, 0.f, 0.f);
            if (CONTACT) {
                dq = cg.dist[qi];
                jq = cg.idx[qi];
                if (cg.nnpt) pq = cg.nnpt[qi];                                   // kernel-uniform
                vwx = cg.Vw[3 * qi]; vwy = cg.Vw[3 * qi + 1]; vwz = cg.Vw[3 * qi + 2];
            }
            SkinFwd f;
            if (packed) f = skin_forward_vertex_packed(vp4[c], vp4[nc + c], sm.K, Voff + 3 * qi, sAf, transl, sM, s);
            else f = skin_forward_vertex(sm, c, x + X_BETAS, Voff + 3 * qi, sAf, transl, sM, s);
            V3 g;
            if (CONTACT) {
                float dterm;
                cterm += contact_term(dq, &dterm);
                const float gg = jq >= 0 ? 2.f * cg.coef * dterm : 0.f;          // no neighbour (NaN query): zero gradient
                if (!cg.nnpt && jq >= 0) pq = cg.scene[jq];
                if (jq < 0) pq = make_float4(0.f, 0.f, 0.f, 0.f);
                g = v3(gg * (vwx - pq.x), gg * (vwy - pq.y), gg * (vwz - pq.z));
            } else {
                g = v3(dVw[3 * qi], dVw[3 * qi + 1], dVw[3 * qi + 2]);
            }
            SkinBwd b = skin_backward_vertex(f, sM, s, g);
            dVoff[3 * qi] = b.dvp.x; dVoff[3 * qi + 1] = b.dvp.y; dVoff[3 * qi + 2] = b.dvp.z;
            if (FDC_SKIN_HAS_S(sm))                         // else: d betas = dVoff x shapedirs, columns 486.. of the blend data-gradient GEMM
                for (int l = 0; l < NBETA; ++l)
                    acc[l] += sm.S[(3 * c) * 10 + l] * b.dvp.x + sm.S[(3 * c + 1) * 10 + l] * b.dvp.y +
                              sm.S[(3 * c + 2) * 10 + l] * b.dvp.z;
            acc[NBETA] += b.gv.x; acc[NBETA + 1] += b.gv.y; acc[NBETA + 2] += b.gv.z;
#pragma unroll
            for (int e = 0; e < 12; ++e) acc[NBETA + 3 + e] += b.dM[e];
            acc[NBETA + 15] += b.ds;
            if (sm.wf_tab) {                                 // (kernel-uniform) matrix-form dA: dT_v = gv (x) [vp ; 1] stays factored -- 32 B per vertex
                float* const t = sdT + (c - c0) * SKB_ROW;      // instead of 48: five workgroups per CU instead of two (r6)
                t[0] = b.gv.x; t[1] = b.gv.y; t[2] = b.gv.z;
                t[3] = f.vp.x; t[4] = f.vp.y; t[5] = f.vp.z;
            } else {
#pragma unroll
                for (int e = 0; e < 12; ++e) sdT[(c - c0) * 12 + e] = b.dT[e];
            }
        }
        // lane j: where joint j's list enters / leaves this chunk (loaded while the vertex phase's stores drain)
        int clo = 0, chi = 0;
        if (nc > VCH && lane < NJ) {
            const int* t = sm.csc_chunk + (size_t)lane * (sm.nch + 1) + c0 / VCH;
            clo = t[0]; chi = t[1];
        }
        __syncthreads();
        FDC_FR_STAMP(2, 2);
        // dA_j += sum_v w_vj dT_v, ordered and atomic-free (run-to-run reproducible): the joints are dealt
        // to the 4 waves; a wave's lanes stride over joint j's vertex list (ascending, restricted to this
        // chunk by two binary searches) and are combined by a butterfly
        // the non-empty joints (one ballot over the preloaded list bounds) are dealt to the four waves in turn
        // r5: the lists live in global memory here (any vertex set, any size), and a joint's walk was two dependent round trips -- the
        // entries, then the LDS rows they name -- in front of its twelve wave sums: 40 of a chunk workgroup's 64 k cycles at 10 475
        // vertices.  The first 128 entries of the NEXT joint of this wave are requested before the current joint is summed.
        if (sm.wf_tab) {
            // r5 (late): dA = W^T dT as a matrix product -- v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation), four vertices
            // per step, wave w = the chunk's w-th quarter, the 16-joint tiles one after the other over the steps that touch them
            // (SkinModel::wf_*: one 16-byte load per lane = four steps' fragments, their step numbers through the scalar cache), the four
            // waves' partial tiles added in wave order.  The list form spends 34 k of a chunk workgroup's 55 k cycles here (55 joints x
            // twelve wave sums per chunk, lists fetched joint by joint); a first matrix form with one 4-byte load per step took as
            // long -- 1000 vector-memory instructions per workgroup: their NUMBER binds, not their bytes.  Same terms, another order.
            // r6: the LDS rows hold the FACTORS of dT_v = gv (x) [vp ; 1] (24 B per vertex; a lane forms its entry gv[e >> 2] * [vp ; 1][e & 3]
            // -- the product skin_backward_vertex forms, same bits -- when it reads it), and the waves' partial tiles go where the rows
            // were once every wave is done with them: 30 KB of LDS per workgroup instead of 70, and with the register budget of five waves
            // per SIMD (86 registers, no spills: the ten dead d-beta accumulators went with SkinModel::S) FIVE workgroups per CU instead
            // of two (the phase is a chain of latencies: fragment loads from L2, dependent LDS reads, one accumulator per tile).
            float* const sPart = sdT;                                          // [4 waves][64 joints][16], after the barrier below
            const int e = lane & 15, kk = lane >> 4;
            const int eg = min(e >> 2, 2), ep = 3 + min(e & 3, 2);
            const bool one = (e & 3) == 3;                                     // dT[4 r + 3] = gv[r] * 1
            const int tb = ((c0 / VCH) * 4 + wave) * 4;
            typedef unsigned wf_u2 __attribute__((ext_vector_type(2)));
            typedef const wf_u2 __attribute__((address_space(4)))* wf_sp_t;    // (wave-uniform: s_load)
            const wf_sp_t steps_c = (wf_sp_t)(const void*)sm.wf_step;
            f32x4_t dacc[4];
            // (r6) what a lane adds to a step's row address is constant: its vertex of the step (kk) and its entry's two factors.  Result
            // columns 12-15 of a tile are never read (dT has 12 entries), so lanes e >= 12 need no masking; a FULL chunk (all but a
            // frame's last) needs no bounds test either; a padded quad (wave-uniform) contributes through a zeroed weight fragment.
            const int og = kk * SKB_ROW + eg, op = kk * SKB_ROW + ep;
            const bool full = c1 - c0 == VCH;                                  // (workgroup-uniform)
#define FDC_SKB_MMA(MASKED)                                                                                                          \
            _Pragma("unroll")                                                                                                        \
            for (int jt = 0; jt < 4; ++jt) {                                                                                         \
                const int g_lo = sm.wf_tab[tb + jt], g_hi = sm.wf_tab[tb + jt + 1];     /* (wave-uniform) quads */                   \
                dacc[jt] = f32x4_t{0.f, 0.f, 0.f, 0.f};                                                                              \
                for (int g0 = g_lo; g0 < g_hi; g0 += SKB_WFQ) {                  /* blocks of SKB_WFQ quads: all their loads first */ \
                    float4 a[SKB_WFQ];                                                                                               \
                    wf_u2 st[SKB_WFQ];                                                                                               \
                    _Pragma("unroll")                                                                                                \
                    for (int u = 0; u < SKB_WFQ; ++u) {                                                                              \
                        const int gc = min(g0 + u, g_hi - 1);                                                                        \
                        a[u] = ((const float4*)sm.wf_frag)[(size_t)gc * 64 + lane];                                                  \
                        st[u] = steps_c[gc];                                                                                         \
                    }                                                                                                                \
                    _Pragma("unroll")                                                                                                \
                    for (int u = 0; u < SKB_WFQ; ++u) {                                                                              \
                        const bool on = g0 + u < g_hi;                          /* (wave-uniform) */                                 \
                        const float4 w = on ? a[u] : make_float4(0.f, 0.f, 0.f, 0.f);                                                \
                        const int s0 = (int)(st[u].x & 0xFFFFu), s1 = (int)(st[u].x >> 16), s2 = (int)(st[u].y & 0xFFFFu), s3 = (int)(st[u].y >> 16); \
                        dacc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, FDC_DT(s0, MASKED), dacc[jt], 0, 0, 0);                 \
                        dacc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, FDC_DT(s1, MASKED), dacc[jt], 0, 0, 0);                 \
                        dacc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, FDC_DT(s2, MASKED), dacc[jt], 0, 0, 0);                 \
                        dacc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, FDC_DT(s3, MASKED), dacc[jt], 0, 0, 0);                 \
                    }                                                                                                                \
                }                                                                                                                    \
            }
            // a step's entry: rows of the step's four vertices start at 4 s SKB_ROW; MASKED: vertices beyond the chunk's end read as zero
#define FDC_DT(s, MASKED) ((!(MASKED) || c0 + 4 * (s) + kk < c1) ? sdT[4 * SKB_ROW * (s) + og] * (one ? 1.f : sdT[4 * SKB_ROW * (s) + op]) : 0.f)
            if (full) { FDC_SKB_MMA(false) } else { FDC_SKB_MMA(true) }
#undef FDC_DT
#undef FDC_SKB_MMA
            __syncthreads();                                                   // every wave is done reading the rows: their space takes the tiles
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int i = 0; i < 4; ++i) sPart[(size_t)((wave * 64 + jt * 16 + 4 * kk + i) * 16) + e] = dacc[jt][i];
            __syncthreads();
            for (int i = tid; i < NJ * 12; i += 256) {
                const int j = i / 12, o = j * 16 + (i - 12 * j);
                sdA[i] += ((sPart[o] + sPart[1024 + o]) + sPart[2048 + o]) + sPart[3072 + o];
            }
        } else {
            unsigned long long jact = __ballot(nc > VCH ? chi > clo : jhi > jlo);
            int kact = 0;
            auto next_joint = [&](int* lo, int* hi) -> int {    // this wave's next non-empty joint (wave-uniform), -1: none
                while (jact) {
                    const int j = __ffsll((long long)jact) - 1;
                    jact &= jact - 1;
                    if ((kact++ & 3) != wave) continue;
                    *lo = __builtin_amdgcn_readlane(nc > VCH ? clo : jlo, j);
                    *hi = __builtin_amdgcn_readlane(nc > VCH ? chi : jhi, j);
                    return j;
                }
                return -1;
            };
            auto fetch = [&](int lo, int hi, float* w, int* v) {  // entries lo + lane, lo + 64 + lane (clamped: unconditional loads)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int i = min(lo + 64 * k + lane, max(hi - 1, lo));
                    w[k] = sm.csc_w[i];
                    v[k] = sm.csc_v[i];
                }
            };
            int lo = 0, hi = 0, lo_n = 0, hi_n = 0;
            float wn[2] = {0.f, 0.f};
            int vn[2] = {0, 0};
            int j = next_joint(&lo, &hi);
            if (j >= 0) fetch(lo, hi, wn, vn);
            while (j >= 0) {
                const float w0 = wn[0], w1 = wn[1];
                const int v0 = vn[0], v1 = vn[1];
                const int jn = next_joint(&lo_n, &hi_n);
                if (jn >= 0) fetch(lo_n, hi_n, wn, vn);
                float pa[12];
#pragma unroll
                for (int e = 0; e < 12; ++e) pa[e] = 0.f;
                if (lo + lane < hi) {
                    const float* t = sdT + (v0 - c0) * 12;
#pragma unroll
                    for (int e = 0; e < 12; ++e) pa[e] += w0 * t[e];
                }
                if (lo + 64 + lane < hi) {
                    const float* t = sdT + (v1 - c0) * 12;
#pragma unroll
                    for (int e = 0; e < 12; ++e) pa[e] += w1 * t[e];
                }
                for (int i = lo + 128 + lane; i < hi; i += 64) {
                    const float w = sm.csc_w[i];
                    const float* t = sdT + (sm.csc_v[i] - c0) * 12;
#pragma unroll
                    for (int e = 0; e < 12; ++e) pa[e] += w * t[e];
                }
#pragma unroll
                for (int e = 0; e < 12; ++e) {
                    float v = wave_sum(pa[e]);
                    if (lane == 0) sdA[j * 12 + e] += v;
                }
                j = jn; lo = lo_n; hi = hi_n;
            }
        }
        __syncthreads();
    }
    FDC_FR_STAMP(2, 3);
#pragma unroll
    for (int i = 0; i < SKB_NACC; ++i) {
        float v = wave_sum(acc[i]);
        if ((tid & 63) == 0) sred[tid >> 6][i] = v;
    }
    __syncthreads();
    if (SPLIT) {                                        // this chunk's sums: added up (ascending chunks) by skin_bwd_reduce_kernel
        __shared__ float scon[4];
        float* const rec = part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * SKP_STRIDE;
        const float v = wave_sum(cterm);
        if ((tid & 63) == 0) scon[tid >> 6] = v;
        __syncthreads();
        for (int i = tid; i < NJ * 12; i += 256) rec[i] = sdA[i];
        if (tid < SKB_NACC) rec[NJ * 12 + tid] = sred[0][tid] + sred[1][tid] + sred[2][tid] + sred[3][tid];
        if (tid == 0) rec[NJ * 12 + SKB_NACC] = (scon[0] + scon[1]) + (scon[2] + scon[3]);
        FDC_FR_STAMP(2, 4);
        return;
    }
    if (CONTACT && cg.loss_rows) {                      // wave-uniform
        __shared__ float scon[4];
        const float v = wave_sum(cterm);
        if ((tid & 63) == 0) scon[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) cg.loss_rows[(size_t)r * LROW + 3] = (scon[0] + scon[1]) + (scon[2] + scon[3]);
    }
    for (int i = tid; i < NJ * 12; i += 256) dA[(size_t)r * NJ * 12 + i] = sdA[i];
    if (tid < SKB_NACC) {
        float v = sred[0][tid] + sred[1][tid] + sred[2][tid] + sred[3][tid];
        if (tid < NBETA) { if (dbeta_v) dbeta_v[(size_t)r * NBETA + tid] = v; }
        else if (tid < NBETA + 3) dtransl_v[(size_t)r * 3 + tid - NBETA] = v;
        else if (tid < NBETA + 15) dMv[(size_t)r * 12 + tid - NBETA - 3] = v;
        else dsv[r] = v;
    }
    FDC_FR_STAMP(2, 4);
}

// the split form's second launch: per frame, the chunk partials added in ascending chunk order (run-to-run reproducible)
__global__ __launch_bounds__(256) void skin_bwd_reduce_kernel(const float* __restrict__ part, int nch, int row0, float* __restrict__ dA,
                                                              float* __restrict__ dbeta_v, float* __restrict__ dtransl_v,
                                                              float* __restrict__ dMv, float* __restrict__ dsv, float* __restrict__ loss_rows) {
    const int r = row0 + blockIdx.x;
    const float* p = part + (size_t)blockIdx.x * nch * SKP_STRIDE;
    for (int i = threadIdx.x; i < NJ * 12 + SKB_NACC + 1; i += 256) {
        float s = 0.f;
        for (int ch = 0; ch < nch; ++ch) s += p[(size_t)ch * SKP_STRIDE + i];
        const int t = i - NJ * 12;
        if (t < 0) dA[(size_t)r * NJ * 12 + i] = s;
        else if (t < NBETA) { if (dbeta_v) dbeta_v[(size_t)r * NBETA + t] = s; }
        else if (t < NBETA + 3) dtransl_v[(size_t)r * 3 + t - NBETA] = s;
        else if (t < NBETA + 15) dMv[(size_t)r * 12 + t - NBETA - 3] = s;
        else if (t < SKB_NACC) dsv[r] = s;
        else if (loss_rows) loss_rows[(size_t)r * LROW + 3] = s;
    }
}

// Contact-set form of skin_bwd_kernel<true> (vertex sets of at most SKS_MAXV vertices / SKS_MAXNNZ skinning weights: the
// optimiser loop's 500 contact vertices).  Same arithmetic per vertex; what differs is where the time went (s_memtime,
// 41 k cycles per frame: vertex loop 16.5 k, dA reduction 16.2 k):
//   * the transposed weight lists (static) are staged in LDS at kernel start -- the reduction read them from global memory
//     joint by joint, a dependent L2 round trip per 64 entries in front of every wave sum;
//   * dT_v = [gv (x) vp | gv] has rank one: the vertex phase leaves gv and vp (6 floats) in LDS, not the 12 products;
//   * a thread's vertices (nc / 256 <= 4) are loaded in one batch before the first is processed.
// The dA sums run joint by joint over ascending vertices with the same wave-sum tree: run-to-run reproducible.
// (VPT vertices and KC weight-list entries per thread in registers: 130 VGPRs for 4 / 16 cost a wave per SIMD -- the launch
// then needs a second generation of workgroups; the loop's 500 vertices / 2000 weights take the 2 / 8 instance)
constexpr int SKS_MAXV = 1024, SKS_MAXNNZ = 6144;
template <int SKS_VPT, int SKS_KC>
__global__ __launch_bounds__(256) void skin_bwd_small_kernel(SkinModel sm, int nc, int nnz, const float* __restrict__ X,
                                                             const float* __restrict__ Voff, const float* __restrict__ A,
                                                             const float* __restrict__ M, const float* __restrict__ scale,
                                                             int row0, float* __restrict__ dVoff, float* __restrict__ dA,
                                                             float* __restrict__ dtransl_v, float* __restrict__ dMv,
                                                             float* __restrict__ dsv, ContactGradIn cg) {
    extern __shared__ __attribute__((aligned(16))) float sk_lds[];
    // dynamic: sGV [nc][3] | sVP [nc][3] | csc_w [nnz] | csc_v [nnz] (ushort)
    float* const sGV = sk_lds;
    float* const sVP = sGV + 3 * nc;
    float* const sCW = sVP + 3 * nc;
    unsigned short* const sCV = (unsigned short*)(sCW + nnz);
    __shared__ float sAf[NJ * 12];
    __shared__ float sdA[NJ * 12];
    __shared__ float sred[4][SKB_NACC];
    __shared__ int sCS[NJ + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = row0 + blockIdx.x;
    FDC_FR_STAMP(2, 0);
    const float* x = X + (size_t)r * XDIM;
    const float s = *scale;
    const V3 transl = v3(x[X_TRANSL], x[X_TRANSL + 1], x[X_TRANSL + 2]);
    float Mr[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) Mr[e] = M[(size_t)r * 12 + e];
    // batch 1 of loads: this thread's vertices (NN result, world vertex, pose offsets, template, weights)
    float dq[SKS_VPT], vwx[SKS_VPT], vwy[SKS_VPT], vwz[SKS_VPT], vox[SKS_VPT], voy[SKS_VPT], voz[SKS_VPT], vtx[SKS_VPT], vty[SKS_VPT], vtz[SKS_VPT];
    int jq[SKS_VPT];
    float4 pq[SKS_VPT];
#pragma unroll
    for (int u = 0; u < SKS_VPT; ++u) {
        const int c = tid + 256 * u;
        const bool ok = c < nc;
        const size_t qi = (size_t)r * nc + (ok ? c : 0);
        dq[u] = ok ? cg.dist[qi] : 0.f;
        jq[u] = ok ? cg.idx[qi] : -1;
        pq[u] = (ok && cg.nnpt) ? cg.nnpt[qi] : make_float4(0.f, 0.f, 0.f, 0.f);
        vwx[u] = ok ? cg.Vw[3 * qi] : 0.f; vwy[u] = ok ? cg.Vw[3 * qi + 1] : 0.f; vwz[u] = ok ? cg.Vw[3 * qi + 2] : 0.f;
        vox[u] = ok ? Voff[3 * qi] : 0.f; voy[u] = ok ? Voff[3 * qi + 1] : 0.f; voz[u] = ok ? Voff[3 * qi + 2] : 0.f;
        vtx[u] = ok ? sm.vt[3 * c] : 0.f; vty[u] = ok ? sm.vt[3 * c + 1] : 0.f; vtz[u] = ok ? sm.vt[3 * c + 2] : 0.f;
    }
    // batch 2: the frame's transforms and the static transposed weight lists -> LDS.  Two unrolled passes (all loads, then
    // all LDS writes): as a load-store loop every trip waits for its own load (measured 16 k cycles for this prologue)
    {
        constexpr int KA = (NJ * 12 + 255) / 256, KC = SKS_KC;
        float va[KA], vw[KC];
        int vv[KC];
#pragma unroll
        for (int k = 0; k < KA; ++k) { const int i = tid + 256 * k; va[k] = i < NJ * 12 ? A[(size_t)r * NJ * 12 + i] : 0.f; }
#pragma unroll
        for (int k = 0; k < KC; ++k) { const int i = tid + 256 * k; vw[k] = i < nnz ? sm.csc_w[i] : 0.f; vv[k] = i < nnz ? sm.csc_v[i] : 0; }
        const int cs = tid <= NJ ? sm.csc_start[tid] : 0;
#pragma unroll
        for (int k = 0; k < KA; ++k) { const int i = tid + 256 * k; if (i < NJ * 12) { sAf[i] = va[k]; sdA[i] = 0.f; } }
#pragma unroll
        for (int k = 0; k < KC; ++k) { const int i = tid + 256 * k; if (i < nnz) { sCW[i] = vw[k]; sCV[i] = (unsigned short)vv[k]; } }
        if (tid <= NJ) sCS[tid] = cs;
    }
    __syncthreads();
    FDC_FR_STAMP(2, 1);
    float acc[SKB_NACC];
#pragma unroll
    for (int i = 0; i < SKB_NACC; ++i) acc[i] = 0.f;
    float cterm = 0.f;
#pragma unroll
    for (int u = 0; u < SKS_VPT; ++u) {
        const int c = tid + 256 * u;
        if (c < nc) {
            const size_t qi = (size_t)r * nc + c;
            // (skin_forward_vertex with its inputs already in registers)
            SkinFwd f;
            const float p0 = vtx[u] + vox[u], p1 = vty[u] + voy[u], p2 = vtz[u] + voz[u];
            f.vp = v3(p0, p1, p2);
#pragma unroll
            for (int e = 0; e < 12; ++e) f.T[e] = 0.f;
            for (int k = 0; k < sm.K; ++k) {
                const float w = sm.ww[c * sm.K + k];
                const float* a = sAf + 12 * sm.wj[c * sm.K + k];
#pragma unroll
                for (int e = 0; e < 12; ++e) f.T[e] += w * a[e];
            }
            const V3 vl = v3(f.T[0] * p0 + f.T[1] * p1 + f.T[2] * p2 + f.T[3], f.T[4] * p0 + f.T[5] * p1 + f.T[6] * p2 + f.T[7],
                             f.T[8] * p0 + f.T[9] * p1 + f.T[10] * p2 + f.T[11]);
            f.vb = vl + transl;
            float dterm;
            cterm += contact_term(dq[u], &dterm);
            const float gg = jq[u] >= 0 ? 2.f * cg.coef * dterm : 0.f;          // no neighbour (NaN query): zero gradient
            float4 pt = pq[u];
            if (!cg.nnpt && jq[u] >= 0) pt = cg.scene[jq[u]];
            if (jq[u] < 0) pt = make_float4(0.f, 0.f, 0.f, 0.f);
            const V3 g = v3(gg * (vwx[u] - pt.x), gg * (vwy[u] - pt.y), gg * (vwz[u] - pt.z));
            const SkinBwd b = skin_backward_vertex(f, Mr, s, g);
            dVoff[3 * qi] = b.dvp.x; dVoff[3 * qi + 1] = b.dvp.y; dVoff[3 * qi + 2] = b.dvp.z;
            acc[NBETA] += b.gv.x; acc[NBETA + 1] += b.gv.y; acc[NBETA + 2] += b.gv.z;
#pragma unroll
            for (int e = 0; e < 12; ++e) acc[NBETA + 3 + e] += b.dM[e];
            acc[NBETA + 15] += b.ds;
            sGV[3 * c] = b.gv.x; sGV[3 * c + 1] = b.gv.y; sGV[3 * c + 2] = b.gv.z;
            sVP[3 * c] = p0; sVP[3 * c + 1] = p1; sVP[3 * c + 2] = p2;
        }
    }
    __syncthreads();
    FDC_FR_STAMP(2, 2);
    {   // dA_j = sum_v w_vj [gv (x) vp | gv]: the non-empty joints are dealt to the four waves in turn
        const int jlo = lane < NJ ? sCS[lane] : 0, jhi = lane < NJ ? sCS[lane + 1] : 0;
        unsigned long long jact = __ballot(jhi > jlo);
        for (int kact = 0; jact; ++kact) {
            const int j = __ffsll((long long)jact) - 1;
            jact &= jact - 1;
            if ((kact & 3) != wave) continue;               // wave-uniform
            const int lo = __builtin_amdgcn_readlane(jlo, j), hi = __builtin_amdgcn_readlane(jhi, j);
            float pa[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) pa[e] = 0.f;
            for (int i = lo + lane; i < hi; i += 64) {
                const float w = sCW[i];
                const int v = sCV[i];
                const float gx = w * sGV[3 * v], gy = w * sGV[3 * v + 1], gz = w * sGV[3 * v + 2];
                const float px = sVP[3 * v], py = sVP[3 * v + 1], pz = sVP[3 * v + 2];
                pa[0] += gx * px; pa[1] += gx * py; pa[2] += gx * pz; pa[3] += gx;
                pa[4] += gy * px; pa[5] += gy * py; pa[6] += gy * pz; pa[7] += gy;
                pa[8] += gz * px; pa[9] += gz * py; pa[10] += gz * pz; pa[11] += gz;
            }
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const float v = wave_sum(pa[e]);
                if (lane == 0) sdA[j * 12 + e] = v;
            }
        }
    }
    FDC_FR_STAMP(2, 3);
#pragma unroll
    for (int i = NBETA; i < SKB_NACC; ++i) {
        const float v = wave_sum(acc[i]);
        if (lane == 0) sred[wave][i] = v;
    }
    const float ct = (cg.loss_rows != nullptr) ? wave_sum(cterm) : 0.f;
    if (cg.loss_rows && lane == 0) sred[wave][0] = ct;
    __syncthreads();
    if (cg.loss_rows && tid == 0) cg.loss_rows[(size_t)r * LROW + 3] = (sred[0][0] + sred[1][0]) + (sred[2][0] + sred[3][0]);
    for (int i = tid; i < sm.ja_hi * 12; i += 256) dA[(size_t)r * NJ * 12 + i] = sdA[i];   // (rows >= ja_hi: zero, never read -- SkinModel::ja_hi)
    if (tid >= NBETA && tid < SKB_NACC) {
        const float v = sred[0][tid] + sred[1][tid] + sred[2][tid] + sred[3][tid];
        if (tid < NBETA + 3) dtransl_v[(size_t)r * 3 + tid - NBETA] = v;
        else if (tid < NBETA + 15) dMv[(size_t)r * 12 + tid - NBETA - 3] = v;
        else dsv[r] = v;
    }
    FDC_FR_STAMP(2, 4);
}

// skin_bwd_small_kernel<2, 8> with every input staged by 16-byte loads (contact sets with nc % 4 == 0, nc <= 512, nnz <= 2048
// and the packed constants of SkinModel).  s_memtime had put 13 k of the scalar kernel's 30 k cycles in its prologue: ~57
// four-byte loads per thread (stride-12 x/y/z components, weight lists entry by entry) keep the CU's address unit busy for
// that long with four workgroups resident -- the same bytes as float4 are 18 loads.  The frame's world vertices and pose
// offsets are copied into the LDS regions that later hold gv / vp (a thread reads and overwrites only its own vertices),
// the pose-blend gradient leaves through LDS as float4 rows.  Arithmetic, summation order and results: unchanged.
// G = ceil(K / 4) weight groups per vertex (1..3: up to 12 skinning weights per vertex; the transposed lists grow with it)
template <int G>
__global__ __launch_bounds__(256) void skin_bwd_vec_kernel(SkinModel sm, int nc, int nnz, const float* __restrict__ X,
                                                           const float* __restrict__ Voff, const float* __restrict__ A,
                                                           const float* __restrict__ M, const float* __restrict__ scale,
                                                           int row0, float* __restrict__ dVoff, float* __restrict__ dA,
                                                           float* __restrict__ dtransl_v, float* __restrict__ dMv,
                                                           float* __restrict__ dsv, ContactGradIn cg) {
    extern __shared__ __attribute__((aligned(16))) float sk_lds[];
    // dynamic: sGV [nc][3] (Vw, then gv) | sVP [nc][3] (Voff, then vp) | sDV [nc][3] | csc_w [nnz4] | csc_v [nnz8] (ushort)
    const int nnz4 = (nnz + 3) & ~3, nnz8 = (nnz + 7) & ~7;
    float* const sGV = sk_lds;
    float* const sVP = sGV + 3 * nc;
    float* const sDV = sVP + 3 * nc;
    float* const sCW = sDV + 3 * nc;
    unsigned short* const sCV = (unsigned short*)(sCW + nnz4);
    __shared__ __attribute__((aligned(16))) float sAf[NJ * 12];
    __shared__ float sdA[NJ * 12];
    __shared__ float sred[4][SKB_NACC];
    __shared__ int sCS[NJ + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = row0 + blockIdx.x;
    FDC_FR_STAMP(2, 0);
    const float* x = X + (size_t)r * XDIM;
    const float s = *scale;
    const V3 transl = v3(x[X_TRANSL], x[X_TRANSL + 1], x[X_TRANSL + 2]);
    float Mr[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) Mr[e] = M[(size_t)r * 12 + e];
    // one batch: the rows and lists go global -> LDS by LDS-DMA (stage_pose_issue's comment), this thread's per-vertex
    // constants and NN results into registers (indices clamped: unconditional loads)
    const int n4 = (3 * nc) >> 2, nw4 = nnz4 >> 2, nv8 = nnz8 >> 3;
    glds_wg<2, 4, 16>(cg.Vw + (size_t)r * nc * 3, sGV, n4);
    glds_wg<2, 4, 16>(Voff + (size_t)r * nc * 3, sVP, n4);
    glds_wg<2 * G, 4, 16>(sm.csc_w, sCW, nw4);                 // (nnz <= 2048 G)
    glds_wg<G, 4, 16>(sm.csc_v16, sCV, nv8);
    glds_wg<1, 4, 16>(A + (size_t)r * NJ * 12, sAf, NJ * 3);
    glds_wg<1, 4, 4>(sm.csc_start, sCS, NJ + 1);
    float4 lvp0[2], lvp1[2][G], lvpj[2], lpq[2];
    float ldq[2];
    int ljq[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = min(tid + 256 * k, nc - 1);
        const size_t qi = (size_t)r * nc + c;
        lvp0[k] = ((const float4*)sm.vpack)[c];
#pragma unroll
        for (int g = 0; g < G; ++g) lvp1[k][g] = ((const float4*)sm.vpack)[(size_t)(1 + g) * nc + c];
        lvpj[k] = G > 1 ? ((const float4*)sm.vpack)[(size_t)(1 + G) * nc + c] : make_float4(0.f, 0.f, 0.f, 0.f);
        ldq[k] = cg.dist[qi];
        // (with the NN launch's own neighbour records -- {x, y, z, bits(position)}, position -1: none -- idx is not needed)
        lpq[k] = cg.nnpt ? cg.nnpt[qi] : make_float4(0.f, 0.f, 0.f, 0.f);
        ljq[k] = cg.nnpt ? __float_as_int(lpq[k].w) : cg.idx[qi];
    }
    for (int i = tid; i < NJ * 12; i += 256) sdA[i] = 0.f;
    __syncthreads();
    FDC_FR_STAMP(2, 1);
    float acc[SKB_NACC];
#pragma unroll
    for (int i = 0; i < SKB_NACC; ++i) acc[i] = 0.f;
    float cterm = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = tid + 256 * u;
        if (c < nc) {
            SkinFwd f;
            const float p0 = lvp0[u].x + sVP[3 * c], p1 = lvp0[u].y + sVP[3 * c + 1], p2 = lvp0[u].z + sVP[3 * c + 2];
            const float vwx = sGV[3 * c], vwy = sGV[3 * c + 1], vwz = sGV[3 * c + 2];
            f.vp = v3(p0, p1, p2);
#pragma unroll
            for (int e = 0; e < 12; ++e) f.T[e] = 0.f;
            const unsigned jb[3] = {__float_as_uint(lvp0[u].w), __float_as_uint(lvpj[u].x), __float_as_uint(lvpj[u].y)};
#pragma unroll
            for (int k = 0; k < 4 * G; ++k) {
                if (k < sm.K) {                                             // (same terms in the same order as the scalar kernel)
                    const float4 wq = lvp1[u][k >> 2];
                    const float w = (k & 3) == 0 ? wq.x : (k & 3) == 1 ? wq.y : (k & 3) == 2 ? wq.z : wq.w;
                    const float* a = sAf + 12 * ((jb[k >> 2] >> (8 * (k & 3))) & 255u);
#pragma unroll
                    for (int e = 0; e < 12; ++e) f.T[e] += w * a[e];
                }
            }
            const V3 vl = v3(f.T[0] * p0 + f.T[1] * p1 + f.T[2] * p2 + f.T[3], f.T[4] * p0 + f.T[5] * p1 + f.T[6] * p2 + f.T[7],
                             f.T[8] * p0 + f.T[9] * p1 + f.T[10] * p2 + f.T[11]);
            f.vb = vl + transl;
            float dterm;
            cterm += contact_term(ldq[u], &dterm);
            const float gg = ljq[u] >= 0 ? 2.f * cg.coef * dterm : 0.f;         // no neighbour (NaN query): zero gradient
            float4 pt = lpq[u];
            if (!cg.nnpt && ljq[u] >= 0) pt = cg.scene[ljq[u]];
            if (ljq[u] < 0) pt = make_float4(0.f, 0.f, 0.f, 0.f);
            const V3 g = v3(gg * (vwx - pt.x), gg * (vwy - pt.y), gg * (vwz - pt.z));
            const SkinBwd b = skin_backward_vertex(f, Mr, s, g);
            sDV[3 * c] = b.dvp.x; sDV[3 * c + 1] = b.dvp.y; sDV[3 * c + 2] = b.dvp.z;
            acc[NBETA] += b.gv.x; acc[NBETA + 1] += b.gv.y; acc[NBETA + 2] += b.gv.z;
#pragma unroll
            for (int e = 0; e < 12; ++e) acc[NBETA + 3 + e] += b.dM[e];
            acc[NBETA + 15] += b.ds;
            sGV[3 * c] = b.gv.x; sGV[3 * c + 1] = b.gv.y; sGV[3 * c + 2] = b.gv.z;
            sVP[3 * c] = p0; sVP[3 * c + 1] = p1; sVP[3 * c + 2] = p2;
        }
    }
    __syncthreads();
    FDC_FR_STAMP(2, 2);
    {   // the pose-blend gradient row leaves as float4s
        float4* const gD = (float4*)(dVoff + (size_t)r * nc * 3);
        for (int i = tid; i < n4; i += 256) gD[i] = ((const float4*)sDV)[i];
    }
    {   // dA_j = sum_v w_vj [gv (x) vp | gv]: the non-empty joints are dealt to the four waves in turn
        const int jlo = lane < NJ ? sCS[lane] : 0, jhi = lane < NJ ? sCS[lane + 1] : 0;
        unsigned long long jact = __ballot(jhi > jlo);
        for (int kact = 0; jact; ++kact) {
            const int j = __ffsll((long long)jact) - 1;
            jact &= jact - 1;
            if ((kact & 3) != wave) continue;               // wave-uniform
            const int lo = __builtin_amdgcn_readlane(jlo, j), hi = __builtin_amdgcn_readlane(jhi, j);
            float pa[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) pa[e] = 0.f;
            for (int i = lo + lane; i < hi; i += 64) {
                const float w = sCW[i];
                const int v = sCV[i];
                const float gx = w * sGV[3 * v], gy = w * sGV[3 * v + 1], gz = w * sGV[3 * v + 2];
                const float px = sVP[3 * v], py = sVP[3 * v + 1], pz = sVP[3 * v + 2];
                pa[0] += gx * px; pa[1] += gx * py; pa[2] += gx * pz; pa[3] += gx;
                pa[4] += gy * px; pa[5] += gy * py; pa[6] += gy * pz; pa[7] += gy;
                pa[8] += gz * px; pa[9] += gz * py; pa[10] += gz * pz; pa[11] += gz;
            }
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const float v = wave_sum(pa[e]);
                if (lane == 0) sdA[j * 12 + e] = v;
            }
        }
    }
    FDC_FR_STAMP(2, 3);
#pragma unroll
    for (int i = NBETA; i < SKB_NACC; ++i) {
        const float v = wave_sum(acc[i]);
        if (lane == 0) sred[wave][i] = v;
    }
    const float ct = (cg.loss_rows != nullptr) ? wave_sum(cterm) : 0.f;
    if (cg.loss_rows && lane == 0) sred[wave][0] = ct;
    __syncthreads();
    if (cg.loss_rows && tid == 0) cg.loss_rows[(size_t)r * LROW + 3] = (sred[0][0] + sred[1][0]) + (sred[2][0] + sred[3][0]);
    for (int i = tid; i < sm.ja_hi * 12; i += 256) dA[(size_t)r * NJ * 12 + i] = sdA[i];   // (rows >= ja_hi: zero, never read -- SkinModel::ja_hi)
    if (tid >= NBETA && tid < SKB_NACC) {
        const float v = sred[0][tid] + sred[1][tid] + sred[2][tid] + sred[3][tid];
        if (tid < NBETA + 3) dtransl_v[(size_t)r * 3 + tid - NBETA] = v;
        else if (tid < NBETA + 15) dMv[(size_t)r * 12 + tid - NBETA - 3] = v;
        else dsv[r] = v;
    }
    FDC_FR_STAMP(2, 4);
}

}  // namespace
