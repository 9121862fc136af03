// Per-vertex shape blend + linear-blend skinning + world transform, forward and backward.
// Restates smplx.lbs.lbs steps v_shaped / v_posed / T = W·A / verts, `+transl`,
// `*scale` (/root/reference/global_optimization.py:284) and verts_transform (:119-127, :285).
// The per-vertex 4x4 transform T[B,V,4,4] the library materialises (670 KB/frame) only ever
// exists in registers here.
#pragma once
#include "fdc_math.h"

namespace fdc {

struct SkinModel {
    const float* vt;      // [V,3]  v_template (+ constant expression offset folded in)
    const float* S;       // [V,3,10] shapedirs (betas part); null: the shape blend is folded into voff by the blend GEMM
                          // (its operand rows are [pose feature | betas], its matrix [posedirs ; shapedirs^T])
    const int* wj;        // [V,K] joint ids of the non-zero skinning weights (padded with 0)
    const float* ww;      // [V,K] weights (padded with 0.0)
    int K;
    // transpose of the sparse weights (per joint: vertices ascending) for the ordered, atomic-free
    // reduction of d loss / d A_j in the backward
    const int* csc_start; // [56]
    const int* csc_v;     // [nnz]
    const float* csc_w;   // [nnz] (allocated and zero-padded to a multiple of 4)
    // 16-byte forms of the same constants (K <= 12, V <= 65535; null otherwise) for kernels that stage them with vector loads:
    // [skin_vpack_planes(K)][V][4], 16-byte aligned.  Plane 0: {vt.x, vt.y, vt.z, bits(j0 | j1 << 8 | j2 << 16 | j3 << 24)} of every
    // vertex; plane 1: {w0, w1, w2, w3}; K > 4: planes 2 (, 3): {w4..w7} (, {w8..w11}), last plane {bits(j4..j7), bits(j8..j11), 0, 0}.
    // (K <= 4 is the two-plane layout of r2; a real SMPLX_NEUTRAL.npz is not promised to be 4-sparse.)
    const float* vpack = nullptr;
    const unsigned short* csc_v16 = nullptr; // [nnz rounded up to a multiple of 8] csc_v as 16-bit ids
    // [55][nch + 1], nch = ceil(V / 1024): where joint j's list enters each 1024-vertex chunk (r5: the chunked backward of large
    // vertex sets looked these up with two binary searches per joint and chunk -- ~26 dependent global round trips each, 80 % of
    // its 540 us at 10 475 vertices)
    const int* csc_chunk = nullptr;
    int nch = 0;
    // 1 + the highest joint any vertex of the set is skinned to (r5).  The backward's dA rows [ja_hi, 55) are zero by construction:
    // the contact-set kernels do not write them and pose_bwd_kernel does not read them (a leg-only contact set: 12 of 55 rows,
    // 2.1 MB less written and 2.1 MB less read per iteration at 1024 frames)
    int ja_hi = 55;
    // (r5, late) the weights as v_mfma_f32_16x16x4_f32 A operands for the chunked backward's dA = W^T dT: per 1024-vertex chunk, per
    // quarter q of it (256 vertices = 64 steps of four vertices: one wave's share) and per 16-joint tile jt the steps whose vertices touch
    // the tile, in QUADS of four (padded with zero fragments): wf_tab[(chunk 4 + q) 4 + jt] .. [+1) = range of quads; wf_step[quad] = the
    // four steps (within the chunk, 0..255) as 16-bit fields; wf_frag[64 quad + lane] = {W(vertex 1024 chunk + 4 step_i + (lane >> 4),
    // joint 16 jt + (lane & 15)), i = 0..3}: ONE 16-byte load per lane and four products -- the phase is bound by the number of
    // vector-memory instructions, not by their bytes.  null: the ordered list form.
    const int* wf_tab = nullptr;
    const unsigned* wf_step = nullptr;      // [quads][2]
    const float* wf_frag = nullptr;         // [quads][64][4]
};

// The per-vertex shapedirs rows (SkinModel::S) are only ever set by the HOST harness (tests/cpu_harness): every device-side vertex set
// folds the shape blend into the blend product (build_skin_set leaves S null).  Device code therefore drops the branch at compile
// time -- ten accumulator registers and their arithmetic in the skinning backward.
#ifdef __HIP_DEVICE_COMPILE__
#define FDC_SKIN_HAS_S(sm) false
#else
#define FDC_SKIN_HAS_S(sm) ((sm).S != nullptr)
#endif

FDC_HD int skin_vpack_planes(int K) { const int G = (K + 3) / 4; return G <= 1 ? 2 : G + 2; }

struct SkinFwd { V3 vp, vb, vw; float T[12]; };

// v: vertex id in the full mesh; voff: this vertex's 3 pose-blend offsets (PF @ posedirs)
FDC_HD SkinFwd skin_forward_vertex(const SkinModel& sm, int v, const float* beta, const float* voff,
                                   const float* A, V3 transl, const float* M, float scale) {
    SkinFwd r;
    float p[3];
    for (int c = 0; c < 3; ++c) {
        float acc = sm.vt[3 * v + c];
        if (FDC_SKIN_HAS_S(sm)) {
            const float* s = sm.S + (3 * v + c) * 10;
            for (int l = 0; l < 10; ++l) acc += s[l] * beta[l];
        }
        p[c] = acc + voff[c];
    }
    r.vp = v3(p[0], p[1], p[2]);
    for (int e = 0; e < 12; ++e) r.T[e] = 0.f;
    for (int k = 0; k < sm.K; ++k) {
        float w = sm.ww[v * sm.K + k];
        const float* a = A + 12 * sm.wj[v * sm.K + k];
        for (int e = 0; e < 12; ++e) r.T[e] += w * a[e];
    }
    V3 vl = v3(r.T[0] * p[0] + r.T[1] * p[1] + r.T[2] * p[2] + r.T[3],
               r.T[4] * p[0] + r.T[5] * p[1] + r.T[6] * p[2] + r.T[7],
               r.T[8] * p[0] + r.T[9] * p[1] + r.T[10] * p[2] + r.T[11]);
    r.vb = vl + transl;
    V3 sv = scale * r.vb;
    r.vw = v3(M[0] * sv.x + M[1] * sv.y + M[2] * sv.z + M[3],
              M[4] * sv.x + M[5] * sv.y + M[6] * sv.z + M[7],
              M[8] * sv.x + M[9] * sv.y + M[10] * sv.z + M[11]);
    return r;
}

// Gradient of one vertex.  g = d loss / d vw.  Emits
//   dT[12]  : d loss / d T_v  (caller scatters w_vj * dT into dA_j)
//   dvp     : d loss / d v_posed  (= d voff; caller contracts with shapedirs for d beta)
//   gv      : d loss / d (v + transl)  (= d transl contribution)
//   dM[12]  : contribution to d loss / d M
//   ds      : contribution to d loss / d scale
struct SkinBwd { float dT[12]; V3 dvp, gv; float dM[12]; float ds; };

FDC_HD SkinBwd skin_backward_vertex(const SkinFwd& f, const float* M, float scale, V3 g) {
    SkinBwd b;
    V3 sv = scale * f.vb;
    // vw = M.R (s vb) + M.t
    b.dM[0] = g.x * sv.x; b.dM[1] = g.x * sv.y; b.dM[2] = g.x * sv.z;  b.dM[3] = g.x;
    b.dM[4] = g.y * sv.x; b.dM[5] = g.y * sv.y; b.dM[6] = g.y * sv.z;  b.dM[7] = g.y;
    b.dM[8] = g.z * sv.x; b.dM[9] = g.z * sv.y; b.dM[10] = g.z * sv.z; b.dM[11] = g.z;
    V3 gs = v3(M[0] * g.x + M[4] * g.y + M[8] * g.z,
               M[1] * g.x + M[5] * g.y + M[9] * g.z,
               M[2] * g.x + M[6] * g.y + M[10] * g.z);       // d loss / d (s vb)
    b.ds = dot(gs, f.vb);
    b.gv = scale * gs;
    // vl = T.R vp + T.t
    b.dT[0] = b.gv.x * f.vp.x; b.dT[1] = b.gv.x * f.vp.y; b.dT[2] = b.gv.x * f.vp.z;  b.dT[3] = b.gv.x;
    b.dT[4] = b.gv.y * f.vp.x; b.dT[5] = b.gv.y * f.vp.y; b.dT[6] = b.gv.y * f.vp.z;  b.dT[7] = b.gv.y;
    b.dT[8] = b.gv.z * f.vp.x; b.dT[9] = b.gv.z * f.vp.y; b.dT[10] = b.gv.z * f.vp.z; b.dT[11] = b.gv.z;
    b.dvp = v3(f.T[0] * b.gv.x + f.T[4] * b.gv.y + f.T[8] * b.gv.z,
               f.T[1] * b.gv.x + f.T[5] * b.gv.y + f.T[9] * b.gv.z,
               f.T[2] * b.gv.x + f.T[6] * b.gv.y + f.T[10] * b.gv.z);
    return b;
}

// contact robustifier (:295): per-query value and d value / d dist, before the mean and weights
FDC_HD float contact_term(float d, float* dterm_dd) {
    float r = sqrtf(d + 1e-4f);
    float q = 1.f / (r + 1.f);
    *dterm_dd = q * q * (0.5f / r);
    return r * q;
}

}  // namespace fdc
