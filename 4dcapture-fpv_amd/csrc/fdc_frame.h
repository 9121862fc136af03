// Per-frame body-model math: joint rotations, kinematic chain, world transform -- forward and
// hand-derived backward -- written as block-cooperative code.  A "team" of `nthr` threads runs
// one frame; `sync()` separates data-parallel phases.  On the GPU the team is one 64-thread
// workgroup (one wavefront) and the scratch arrays live in LDS; the test-only host harness runs
// the same source with nthr = 1 and a no-op sync.
//
// Restates (forward): smplx.SMPLX.forward + lbs.batch_rigid_transform as configured at
// /root/reference/global_optimization.py:154-168 and called at :280-283, the
// VPoser 6D->rotation tail (A.2), body2world (:191-206) and verts_transform on joints (:298-299).
// The R -> angle-axis -> Rodrigues round trip the reference makes for global_orient and the 21
// VPoser joints is the identity on SO(3), including its gradient after composition with
// Gram-Schmidt, so rotation matrices feed the chain directly.
#pragma once
#include "fdc_math.h"

#ifndef FDC_FR_STAMP
#define FDC_FR_STAMP(w, i)      // instrumentation hook (timing builds of the GPU library only)
#endif

namespace fdc {

constexpr int NJ = 55;          // SMPL-X joints
constexpr int NJW = 23;         // world joints the reference reads (joints[:, 0:23], :298)
constexpr int NBETA = 10;
constexpr int NPF = (NJ - 1) * 9;   // 486 pose-feature columns
constexpr int NPFX = NPF + NBETA;   // 496: row of the blend GEMMs' operand [pose feature | betas] (shape blend folded in)
constexpr int XDIM = 78;        // optimised parameter row
constexpr int X_TRANSL = 0, X_SIXD = 3, X_BETAS = 9, X_LATENT = 19, X_LH = 51, X_RH = 63, X_CAMT = 75;
constexpr int ODIM = 126;       // VPoser decoder output (21 joints x 6D)
constexpr int MAX_LEVELS = 16;

struct PoseModel {
    const float* Jt;          // [55,3]   J_regressor @ v_template
    const float* Jd;          // [55,3,10] J_regressor @ shapedirs[:, :, :10]
    const int* parents;       // [55], root -1
    const int* order;         // [55] joints sorted by depth
    const int* level_start;   // [nlevels+1] ranges into order
    const int* child_start;   // [56] CSR of children
    const int* child_list;    // [54]
    const int* depth = nullptr;   // [55] level of each joint (optional: enables the lane-resident chains of the GPU kernels)
    const float* tab = nullptr;   // GPU library: every table above as one image in the staging struct's layout (fdcap.hip PoseStage)
    const float* hand_comp;   // [2,12,45]
    const float* hand_mean;   // [2,45]
    int nlevels;
};

// scratch a team needs (LDS on device)
struct alignas(16) PoseScratch {   // rows padded to 16 B multiples so a joint's matrix moves as ds_read/write_b128
    float R[NJ][12];      // 3x3 in the first 9
    float J[NJ][4];
    float G[NJ][12];      // [R | t] row-major 3x4
    float dG[NJ][12];
    float dR[NJ][12];
    float drel[NJ][4];
    float dJ[NJ][4];
    float dMj[NJW][12];
    float dTj[NJW][3];
    float daa[2][45];
    float red[6][NBETA];  // partial d betas (joints j = part, part + 6, ...)
    float dMs[12];        // d world matrix, summed over the joints
};

FDC_HD M3 load_m3(const float* p) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = p[i]; return r; }
FDC_HD void store_m3(float* p, const M3& a) { for (int i = 0; i < 9; ++i) p[i] = a.m[i]; }
FDC_HD M3 g_rot(const float* g) { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[3 * i + j] = g[4 * i + j]; return r; }
FDC_HD V3 g_trn(const float* g) { return v3(g[3], g[7], g[11]); }
FDC_HD void g_store(float* g, const M3& R, V3 t) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) g[4 * i + j] = R.m[3 * i + j];
    g[3] = t.x; g[7] = t.y; g[11] = t.z;
}
// 12 floats at a 16-byte aligned address (rows of the per-frame [55,12] arrays): three 16-byte accesses on the GPU
FDC_HD void store12(float* dst, const float* v) {
#if defined(__HIP_DEVICE_COMPILE__)
    ((float4*)dst)[0] = make_float4(v[0], v[1], v[2], v[3]);
    ((float4*)dst)[1] = make_float4(v[4], v[5], v[6], v[7]);
    ((float4*)dst)[2] = make_float4(v[8], v[9], v[10], v[11]);
#else
    for (int e = 0; e < 12; ++e) dst[e] = v[e];
#endif
}
FDC_HD void load12(float* v, const float* src) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float4 a = ((const float4*)src)[0], b = ((const float4*)src)[1], c = ((const float4*)src)[2];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
#else
    for (int e = 0; e < 12; ++e) v[e] = src[e];
#endif
}

FDC_HD V3 hand_aa(const PoseModel& pm, const float* x, int j) {
    int h = (j - 25) / 15, f = (j - 25) % 15;
    const float* pca = x + (h == 0 ? X_LH : X_RH);
    const float* comp = pm.hand_comp + h * 12 * 45;
    float a[3];
    for (int c = 0; c < 3; ++c) {
        float acc = 0.f;
        for (int i = 0; i < 12; ++i) acc += pca[i] * comp[i * 45 + 3 * f + c];
        a[c] = acc + pm.hand_mean[h * 45 + 3 * f + c];
    }
    return v3(a[0], a[1], a[2]);
}

// body2world (:191-206): M = cam_ext[:3,:] @ [[I, cam_t*s],[0,1]]
FDC_HD void world_matrix(const float* cam_ext, const float* x, float scale, M3* MR, V3* Mt) {
    M3 E; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) E.m[3 * i + j] = cam_ext[4 * i + j];
    V3 Et = v3(cam_ext[3], cam_ext[7], cam_ext[11]);
    V3 ct = v3(x[X_CAMT] * scale, x[X_CAMT + 1] * scale, x[X_CAMT + 2] * scale);
    *MR = E;
    *Mt = m3_vec(E, ct) + Et;
}

// world position of a joint: M (G.t + transl); joints are NOT multiplied by scale (:298-299).  One function for pose_forward's
// tail and for the rows that only refresh M / Jw after `scale` changed (pose_fwd_kernel's world-only rows): same bits
FDC_HD V3 world_joint(const M3& MR, V3 Mt, V3 Gt, V3 transl) { return m3_vec(MR, Gt + transl) + Mt; }

// Forward for one frame.  Global outputs (any may be null):
//   Rm[55*9], PF[486], Jrest[55*3], G[55*12], A[55*12], M[12], Jw[23*3]
template <class Sync>
FDC_HD void pose_forward(const PoseModel& pm, const float* x, const float* o, const float* cam_ext,
                         float scale, PoseScratch& sc, float* Rm, float* PF, float* Jrest, float* G,
                         float* A, float* M, float* Jw, int tid, int nthr, Sync sync,
                         const float* aa22 = nullptr, int split = 0) {
    // split = 1 (the staged GPU kernels, r4): FOUR waves call this, tid 0..255 with nthr = 64.  The kinematic chain stays one
    // wave's job; around it the independent pieces run side by side on the workgroup's other SIMDs instead of one after the
    // other in one lone wave (~5.4 cycles per instruction whatever the SIMD could issue): before the chain the 6D / axis-angle
    // rotations of the body joints, the hand joints' PCA pose + Rodrigues, and the joint regression; during the chain the
    // outputs that only need those (Rm, PF, Jrest); after it A / G on one wave and the world joints on another.  Every number
    // is computed by the same expression as before: same bits.
    const int wv = split ? (tid >> 6) : 0;
    const int ln = split ? (tid & 63) : tid;
    auto joint_rotation = [&](int j) {
        M3 R;
        // aa22 (operator-level API only): global_orient + 21 body joints given as axis-angle
        if (aa22 && j <= 21) R = rodrigues_forward(v3(aa22[3 * j], aa22[3 * j + 1], aa22[3 * j + 2]));
        else if (j == 0) R = gs_forward(x + X_SIXD, 1, nullptr);
        else if (j <= 21) R = gs_forward(o + 6 * (j - 1), 1, nullptr);
        else if (j < 25) R = m3_identity();          // jaw / eyes: zero Parameters, never optimised
        else R = rodrigues_forward(hand_aa(pm, x, j));
        store_m3(sc.R[j], R);
    };
    auto joint_rest = [&](int j) {
        for (int c = 0; c < 3; ++c) {
            float acc = pm.Jt[3 * j + c];
            for (int l = 0; l < NBETA; ++l) acc += pm.Jd[(3 * j + c) * NBETA + l] * x[X_BETAS + l];
            sc.J[j][c] = acc;
        }
    };
    FDC_FR_STAMP(0, 1);
    if (split) {                                         // (global_orient on the fourth wave: its code path is not the body joints', whose wave then walks one path only)
        if (wv == 0) { if (ln >= 1 && ln < 25) joint_rotation(ln); }
        else if (wv == 1) { if (ln + 25 < NJ) joint_rotation(ln + 25); }
        else if (wv == 2) { if (ln < NJ) joint_rest(ln); }
        else if (ln == 0) joint_rotation(0);
    } else {
        for (int j = tid; j < NJ; j += nthr) { joint_rotation(j); joint_rest(j); }
    }
    sync();
    if (split) {
        // while three waves walk the chain, the fourth writes what needs only the rotations and rest joints
        if (wv == 3 && ln < NJ) {
            const int j = ln;
            if (Rm) for (int e = 0; e < 9; ++e) Rm[9 * j + e] = sc.R[j][e];
            if (PF && j >= 1)
                for (int e = 0; e < 9; ++e) PF[9 * (j - 1) + e] = sc.R[j][e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
            if (Jrest) for (int c = 0; c < 3; ++c) Jrest[3 * j + c] = sc.J[j][c];
        }
        if (wv != 0) tid = 1 << 20;                      // (the generic loops below are the first wave's)
    }
    FDC_FR_STAMP(0, 2);
#if defined(__HIP_DEVICE_COMPILE__)
    if (split && pm.depth != nullptr) {
        // The chain on three waves (late r4): wave i keeps ROW i of every joint's 3x4 transform -- row i of G_c = (row i of G_p) [R | rel]
        // + (0, 0, 0, t_p[i]) needs row i of the parent only, so the waves never meet inside the chain, and a level is one 16-byte
        // read, twelve multiply-adds and one 16-byte write per lane instead of three, thirty-six and three.  Same expressions as
        // m3_mul / m3_vec element by element.
        const int j = ln;
        const bool act = wv < 3 && j < NJ;
        const int p = act ? pm.parents[j] : -1, dep = act ? pm.depth[j] : -1;
        const M3 R = act ? load_m3(sc.R[j]) : m3_identity();
        const V3 Jj = act ? v3(sc.J[j][0], sc.J[j][1], sc.J[j][2]) : v3(0.f, 0.f, 0.f);
        const V3 rel = (act && p >= 0) ? Jj - v3(sc.J[p][0], sc.J[p][1], sc.J[p][2]) : Jj;
        const int row = wv < 3 ? wv : 0;
        for (int L = 0; L < pm.nlevels; ++L) {
            if (dep == L) {
                float4 out;
                if (p < 0) {
                    out.x = row == 0 ? R.m[0] : row == 1 ? R.m[3] : R.m[6];
                    out.y = row == 0 ? R.m[1] : row == 1 ? R.m[4] : R.m[7];
                    out.z = row == 0 ? R.m[2] : row == 1 ? R.m[5] : R.m[8];
                    out.w = row == 0 ? Jj.x : row == 1 ? Jj.y : Jj.z;
                } else {
                    const float4 g = *(const float4*)&sc.G[p][4 * row];
                    out.x = g.x * R.m[0] + g.y * R.m[3] + g.z * R.m[6];
                    out.y = g.x * R.m[1] + g.y * R.m[4] + g.z * R.m[7];
                    out.z = g.x * R.m[2] + g.y * R.m[5] + g.z * R.m[8];
                    out.w = (g.x * rel.x + g.y * rel.y + g.z * rel.z) + g.w;
                }
                *(float4*)&sc.G[j][4 * row] = out;
            }
            __builtin_amdgcn_wave_barrier();
        }
        sync();
    } else if (nthr == 64 && pm.depth != nullptr) {
        // One wave per frame, lane j = joint j: the joint's rotation, rest position, parent and level stay in registers and
        // a level costs one hand-over through LDS (read the parent's transform, write one's own) -- the generic loop below
        // re-reads level_start -> order -> parents -> R / J from LDS at every level (measured 970 cycles per level, ten levels).
        // A wave's LDS operations execute in order, so between levels a compiler-only barrier is enough.
        const int j = tid;
        const bool act = j < NJ;
        const int p = act ? pm.parents[j] : -1, dep = act ? pm.depth[j] : -1;
        const M3 R = act ? load_m3(sc.R[j]) : m3_identity();
        const V3 Jj = act ? v3(sc.J[j][0], sc.J[j][1], sc.J[j][2]) : v3(0.f, 0.f, 0.f);
        const V3 rel = (act && p >= 0) ? Jj - v3(sc.J[p][0], sc.J[p][1], sc.J[p][2]) : Jj;
        for (int L = 0; L < pm.nlevels; ++L) {
            if (dep == L) {
                if (p < 0) g_store(sc.G[j], R, Jj);
                else {
                    const M3 Rp = g_rot(sc.G[p]);
                    g_store(sc.G[j], m3_mul(Rp, R), m3_vec(Rp, rel) + g_trn(sc.G[p]));
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        sync();
    } else
#endif
    for (int L = 0; L < pm.nlevels; ++L) {
        for (int k = pm.level_start[L] + tid; k < pm.level_start[L + 1]; k += nthr) {
            int j = pm.order[k];
            int p = pm.parents[j];
            M3 R = load_m3(sc.R[j]);
            V3 Jj = v3(sc.J[j][0], sc.J[j][1], sc.J[j][2]);
            if (p < 0) {
                g_store(sc.G[j], R, Jj);
            } else {
                M3 Rp = g_rot(sc.G[p]);
                V3 rel = Jj - v3(sc.J[p][0], sc.J[p][1], sc.J[p][2]);
                g_store(sc.G[j], m3_mul(Rp, R), m3_vec(Rp, rel) + g_trn(sc.G[p]));
            }
        }
        sync();
    }
    FDC_FR_STAMP(0, 3);
    M3 MR; V3 Mt;
    world_matrix(cam_ext, x, scale, &MR, &Mt);
    V3 transl = v3(x[X_TRANSL], x[X_TRANSL + 1], x[X_TRANSL + 2]);
    if (split) {
        if (wv == 0 && ln < NJ) {
            const int j = ln;
            M3 GR = g_rot(sc.G[j]);
            V3 Gt = g_trn(sc.G[j]);
            V3 Jj = v3(sc.J[j][0], sc.J[j][1], sc.J[j][2]);
            if (A) { float a12[12]; g_store(a12, GR, Gt - m3_vec(GR, Jj)); store12(A + 12 * j, a12); }
            if (G) store12(G + 12 * j, sc.G[j]);
        }
        if (wv == 1) {
            if (Jw && ln < NJW) {
                V3 w = world_joint(MR, Mt, g_trn(sc.G[ln]), transl);
                Jw[3 * ln] = w.x; Jw[3 * ln + 1] = w.y; Jw[3 * ln + 2] = w.z;
            }
            if (M && ln == 0) g_store(M, MR, Mt);
        }
        FDC_FR_STAMP(0, 4);
        return;
    }
    for (int j = tid; j < NJ; j += nthr) {
        M3 GR = g_rot(sc.G[j]);
        V3 Gt = g_trn(sc.G[j]);
        V3 Jj = v3(sc.J[j][0], sc.J[j][1], sc.J[j][2]);
        if (A) { float a12[12]; g_store(a12, GR, Gt - m3_vec(GR, Jj)); store12(A + 12 * j, a12); }
        if (G) store12(G + 12 * j, sc.G[j]);
        if (Rm) for (int e = 0; e < 9; ++e) Rm[9 * j + e] = sc.R[j][e];
        if (Jrest) for (int c = 0; c < 3; ++c) Jrest[3 * j + c] = sc.J[j][c];
        if (PF && j >= 1)
            for (int e = 0; e < 9; ++e) PF[9 * (j - 1) + e] = sc.R[j][e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
        if (Jw && j < NJW) {
            V3 w = world_joint(MR, Mt, Gt, transl);
            Jw[3 * j] = w.x; Jw[3 * j + 1] = w.y; Jw[3 * j + 2] = w.z;
        }
    }
    if (M && tid == 0) g_store(M, MR, Mt);
    FDC_FR_STAMP(0, 4);
}

// Backward for one frame.
// In : x, o, cam_ext, scale; stored Jrest / G from the forward (Rm: no longer read -- the subtree-sum form of the reverse chain works
//      on the world transforms alone; the parameter stays for the callers' sake and may be null);
//      dA[55*12] (d loss / d skinning transforms, may be null), dPF[486] (may be null),
//      dJw[23*3] (may be null), dMv[12], dsv, dbeta_v[10], dtransl_v[3]: vertex-side sums (null -> 0)
// Out: dx[78] += (transl, 6D, betas, hands, cam_t),  dO[126] =,  dcam_ext[16] =,  *dscale =
// Operator-level extras (fdcap_smplx_backward; all default to null): aa22 -- the forward took global_orient + the 21 body
// joints as axis-angle (Rodrigues), their gradient goes to daa22[66] instead of dx's 6D slot / dO; dJb[55*3] -- gradient
// of the posed body-frame joints (G.t + transl) as the body-model operator returns them.
template <class Sync>
FDC_HD void pose_backward(const PoseModel& pm, const float* x, const float* o, const float* cam_ext,
                          float scale, const float* Rm, const float* Jrest, const float* G,
                          const float* dA, const float* dPF, const float* dJw, const float* dMv,
                          const float* dsv, const float* dbeta_v, const float* dtransl_v,
                          PoseScratch& sc, float* dx, float* dO, float* dcam_ext, float* dscale,
                          int tid, int nthr, Sync sync, const float* aa22 = nullptr, float* daa22 = nullptr,
                          const float* dJb = nullptr, int split = 0) {
    // split = 1 (the optimiser's staged kernel, r4): FOUR waves call this, tid 0..255 with nthr = 64.  The second wave idles
    // through the chain (it only meets the barriers) and then forms everything of the tail reductions that does not need the
    // rotation gradients -- d betas, d M, d transl, the camera row -- WHILE the first wave runs the body joints' rotation backward
    // and the third the fingers' (6D and Rodrigues: two code paths one wave would walk one after the other):
    // independent instruction streams of one lone wave each, now side by side (a lone wave issues one VALU
    // instruction per ~5.4 cycles: a second wave on the workgroup's next SIMD costs the first nothing).  Same terms in the same
    // order: same bits.
    const int tid_all = tid;
    if (split && tid >= 64) tid = 1 << 20;              // (the second wave: no joint, no row -- every loop below is empty for it)
    FDC_FR_STAMP(1, 1);
    M3 MR; V3 Mt;
    world_matrix(cam_ext, x, scale, &MR, &Mt);
    V3 transl = v3(x[X_TRANSL], x[X_TRANSL + 1], x[X_TRANSL + 2]);
    for (int j = tid; j < NJ; j += nthr) {
        if (G) { float g12[12]; load12(g12, G + 12 * j); for (int e = 0; e < 12; ++e) sc.G[j][e] = g12[e]; }   // null: the caller has put it into sc.G
        for (int c = 0; c < 3; ++c) sc.J[j][c] = Jrest[3 * j + c];
        M3 GR = g_rot(sc.G[j]);
        V3 Jj = v3(sc.J[j][0], sc.J[j][1], sc.J[j][2]);
        M3 dGR = m3_zero();
        V3 dGt = v3(0, 0, 0), dJ = v3(0, 0, 0);
        if (dA) {
            float a12[12];
            load12(a12, dA + 12 * j);
            dGR = g_rot(a12);
            V3 dAt = g_trn(a12);
            m3_add_outer(dGR, -1.f * dAt, Jj);          // A.t = G.t - G.R J
            dGt = dAt;
            dJ = -1.f * m3t_vec(GR, dAt);
        }
        if (j < NJW) {
            M3 dMR = m3_zero();
            V3 q = v3(0, 0, 0), g = v3(0, 0, 0);
            if (dJw) {
                g = v3(dJw[3 * j], dJw[3 * j + 1], dJw[3 * j + 2]);
                m3_add_outer(dMR, g, g_trn(sc.G[j]) + transl);
                q = m3t_vec(MR, g);
                dGt = dGt + q;
            }
            if (dJb) q = q + v3(dJb[3 * j], dJb[3 * j + 1], dJb[3 * j + 2]);   // (+ transl: summed with the world joints' share below)
            g_store(sc.dMj[j], dMR, g);
            sc.dTj[j][0] = q.x; sc.dTj[j][1] = q.y; sc.dTj[j][2] = q.z;
        }
        if (dJb) dGt = dGt + v3(dJb[3 * j], dJb[3 * j + 1], dJb[3 * j + 2]);
        g_store(sc.dG[j], dGR, dGt);
        sc.dJ[j][0] = dJ.x; sc.dJ[j][1] = dJ.y; sc.dJ[j][2] = dJ.z;
    }
    sync();
    FDC_FR_STAMP(1, 2);
    // Reverse chain.  With G_c.R = G_p.R R_c and G_c.t = G_p.R (J_c - J_p) + G_p.t, a joint's total gradient is a sum
    // over its subtree that factors through the WORLD transforms of the forward pass:
    //     dG_p.t = sum_{d in sub(p)} g_d.t
    //     dG_p.R = [ sum_{d in sub(p)} (g_d.R G_d.R^T + g_d.t (x) G_d.t)  -  (sum_d g_d.t) (x) G_p.t ] G_p.R
    // (g = the joints' own gradients formed above; R(p->d)^T = G_d.R^T G_p.R and a (x) (R^T v) = (a (x) v) R).
    // So the level-ordered part is a plain subtree sum of 12 numbers per joint -- three b128 reads and 12 adds per child
    // instead of a 3x3 product, an outer product and 24 scalar reads -- and the products run once, for all joints at once.
#if defined(__HIP_DEVICE_COMPILE__)
    const bool rows_split = split && pm.depth != nullptr;
#else
    const bool rows_split = false;
#endif
    if (!rows_split) {
        for (int d = tid; d < NJ; d += nthr) {
            M3 U = m3_mul_bt(g_rot(sc.dG[d]), g_rot(sc.G[d]));
            V3 gt = g_trn(sc.dG[d]);
            m3_add_outer(U, gt, g_trn(sc.G[d]));
            g_store(sc.dG[d], U, gt);
        }
        sync();
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (split && pm.depth != nullptr) {
        // The subtree sums on three waves (late r4): they are twelve independent sums per joint, wave w takes numbers 4w .. 4w + 3 --
        // one 16-byte read and four adds per child instead of three and twelve; the waves never meet inside the chain.
        const int cw = tid_all >> 6, p = tid_all & 63;
        const bool act = cw < 3 && p < NJ;
        const int dep = act ? pm.depth[p] : -1;
        const int c_lo = act ? pm.child_start[p] : 0, nch = act ? pm.child_start[p + 1] - c_lo : 0;
        const int c0 = nch > 0 ? pm.child_list[c_lo] : 0, c1 = nch > 1 ? pm.child_list[c_lo + 1] : 0, c2 = nch > 2 ? pm.child_list[c_lo + 2] : 0;
        const int c3 = nch > 3 ? pm.child_list[c_lo + 3] : 0, c4 = nch > 4 ? pm.child_list[c_lo + 4] : 0;
        const int e0 = 4 * (cw < 3 ? cw : 0);
        auto add4 = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
        if (act) {
            // the products above, row cw of [U | gt] only (m3_mul_bt / m3_add_outer element by element): this wave's chain reads
            // nothing else, so no barrier between the products and the chain
            const float4 a = *(const float4*)&sc.dG[p][e0];
            const float4 b0 = *(const float4*)&sc.G[p][0], b1 = *(const float4*)&sc.G[p][4], b2 = *(const float4*)&sc.G[p][8];
            float4 u;
            u.x = a.x * b0.x + a.y * b0.y + a.z * b0.z; u.x += a.w * b0.w;
            u.y = a.x * b1.x + a.y * b1.y + a.z * b1.z; u.y += a.w * b1.w;
            u.z = a.x * b2.x + a.y * b2.y + a.z * b2.z; u.z += a.w * b2.w;
            u.w = a.w;
            *(float4*)&sc.dG[p][e0] = u;
        }
        __builtin_amdgcn_wave_barrier();
        for (int L = pm.nlevels - 1; L >= 1; --L) {
            if (dep == L - 1 && nch > 0) {
                float4 a = *(const float4*)&sc.dG[p][e0];
                add4(a, *(const float4*)&sc.dG[c0][e0]);
                if (nch > 1) add4(a, *(const float4*)&sc.dG[c1][e0]);
                if (nch > 2) add4(a, *(const float4*)&sc.dG[c2][e0]);
                if (nch > 3) add4(a, *(const float4*)&sc.dG[c3][e0]);
                if (nch > 4) add4(a, *(const float4*)&sc.dG[c4][e0]);
                for (int ci = c_lo + 5; ci < c_lo + nch; ++ci) add4(a, *(const float4*)&sc.dG[pm.child_list[ci]][e0]);
                *(float4*)&sc.dG[p][e0] = a;
            }
            __builtin_amdgcn_wave_barrier();
        }
        sync();
    } else if (nthr == 64 && pm.depth != nullptr) {
        // lane p = joint p with its (at most five) children in registers: a level is "add the children's sums to one's own"
        const int p = tid;
        const bool act = p < NJ;
        const int dep = act ? pm.depth[p] : -1;
        const int c_lo = act ? pm.child_start[p] : 0, nch = act ? pm.child_start[p + 1] - c_lo : 0;
        const int c0 = nch > 0 ? pm.child_list[c_lo] : 0, c1 = nch > 1 ? pm.child_list[c_lo + 1] : 0, c2 = nch > 2 ? pm.child_list[c_lo + 2] : 0;
        const int c3 = nch > 3 ? pm.child_list[c_lo + 3] : 0, c4 = nch > 4 ? pm.child_list[c_lo + 4] : 0;
        for (int L = pm.nlevels - 1; L >= 1; --L) {
            if (dep == L - 1 && nch > 0) {
                float acc[12];
                for (int e = 0; e < 12; ++e) acc[e] = sc.dG[p][e];
                for (int e = 0; e < 12; ++e) acc[e] += sc.dG[c0][e];
                if (nch > 1) for (int e = 0; e < 12; ++e) acc[e] += sc.dG[c1][e];
                if (nch > 2) for (int e = 0; e < 12; ++e) acc[e] += sc.dG[c2][e];
                if (nch > 3) for (int e = 0; e < 12; ++e) acc[e] += sc.dG[c3][e];
                if (nch > 4) for (int e = 0; e < 12; ++e) acc[e] += sc.dG[c4][e];
                for (int ci = c_lo + 5; ci < c_lo + nch; ++ci)            // (no SMPL-X joint has more than five children)
                    for (int e = 0; e < 12; ++e) acc[e] += sc.dG[pm.child_list[ci]][e];
                for (int e = 0; e < 12; ++e) sc.dG[p][e] = acc[e];
            }
            __builtin_amdgcn_wave_barrier();
        }
        sync();
    } else
#endif
    for (int L = pm.nlevels - 1; L >= 1; --L) {
        for (int k = pm.level_start[L - 1] + tid; k < pm.level_start[L]; k += nthr) {
            int p = pm.order[k];
            float acc[12];
            for (int e = 0; e < 12; ++e) acc[e] = sc.dG[p][e];
            for (int ci = pm.child_start[p]; ci < pm.child_start[p + 1]; ++ci) {
                int c = pm.child_list[ci];
                for (int e = 0; e < 12; ++e) acc[e] += sc.dG[c][e];
            }
            for (int e = 0; e < 12; ++e) sc.dG[p][e] = acc[e];
        }
        sync();
    }
    FDC_FR_STAMP(1, 3);
    // subtree sums -> the joint's total gradient -> what it hands to its local rotation and offset (dR, drel)
    for (int c = tid; c < NJ; c += nthr) {
        M3 SU = g_rot(sc.dG[c]);
        V3 dGt = g_trn(sc.dG[c]);
        m3_add_outer(SU, -1.f * dGt, g_trn(sc.G[c]));
        M3 dGR = m3_mul(SU, g_rot(sc.G[c]));
        int p = pm.parents[c];
        if (p >= 0) {
            M3 Rp = g_rot(sc.G[p]);
            store_m3(sc.dR[c], m3_mul_at(Rp, dGR));
            V3 dr = m3t_vec(Rp, dGt);
            sc.drel[c][0] = dr.x; sc.drel[c][1] = dr.y; sc.drel[c][2] = dr.z;
        } else {
            store_m3(sc.dR[c], dGR);
            sc.drel[c][0] = dGt.x; sc.drel[c][1] = dGt.y; sc.drel[c][2] = dGt.z;
        }
    }
    for (int i = tid; i < 90; i += nthr) (&sc.daa[0][0])[i] = 0.f;
    sync();
    FDC_FR_STAMP(1, 4);
    // rel_j = J_j - J_parent: d J_j = own + drel_j - the children's drel (split: the second wave's job, below)
    auto joint_offset_grad = [&](int j) {
        V3 dJ = v3(sc.dJ[j][0] + sc.drel[j][0], sc.dJ[j][1] + sc.drel[j][1], sc.dJ[j][2] + sc.drel[j][2]);
        for (int ci = pm.child_start[j]; ci < pm.child_start[j + 1]; ++ci) {
            int c = pm.child_list[ci];
            dJ = dJ - v3(sc.drel[c][0], sc.drel[c][1], sc.drel[c][2]);
        }
        sc.dJ[j][0] = dJ.x; sc.dJ[j][1] = dJ.y; sc.dJ[j][2] = dJ.z;
    };
    // the small per-frame reductions, in two steps so that no thread walks more than ~30 terms (one output element per thread made
    // thread t < 10 sum 165 products for d betas and one thread 276 terms for d M); fixed summation order
    auto reduce_step1 = [&](int t) {
        if (t < 6 * NBETA) {
            const int part = t / NBETA, b = t % NBETA;
            float acc = 0.f;
            for (int j = part; j < NJ; j += 6)
                for (int c = 0; c < 3; ++c) acc += pm.Jd[(3 * j + c) * NBETA + b] * sc.dJ[j][c];
            sc.red[part][b] = acc;
        } else {
            const int e = t - 6 * NBETA;
            float acc = dMv ? dMv[e] : 0.f;
            for (int j = 0; j < NJW; ++j) acc += sc.dMj[j][e];
            sc.dMs[e] = acc;
        }
    };
    auto reduce_step2 = [&](int t) {
        if (t < NBETA) {
            float acc = dbeta_v ? dbeta_v[t] : 0.f;
            for (int part = 0; part < 6; ++part) acc += sc.red[part][t];
            dx[X_BETAS + t] += acc;
        } else if (t < NBETA + 24) {                      // (needs the fingers' rotation gradients)
            int i = t - NBETA, h = i / 12, ii = i % 12;
            const float* comp = pm.hand_comp + (h * 12 + ii) * 45;
            float acc = 0.f;
            for (int k = 0; k < 45; ++k) acc += comp[k] * sc.daa[h][k];
            dx[(h == 0 ? X_LH : X_RH) + ii] += acc;
        } else if (t < NBETA + 24 + 3) {
            int c = t - NBETA - 24;
            float acc = dtransl_v ? dtransl_v[c] : 0.f;
            for (int j = 0; j < NJW; ++j) acc += sc.dTj[j][c];
            if (dJb) for (int j = NJW; j < NJ; ++j) acc += dJb[3 * j + c];
            dx[X_TRANSL + c] += acc;
        } else if (t == NBETA + 24 + 3) {
            M3 dMR = g_rot(sc.dMs);
            V3 dMt = g_trn(sc.dMs);
            V3 ct = v3(x[X_CAMT], x[X_CAMT + 1], x[X_CAMT + 2]);
            M3 dER = dMR;
            m3_add_outer(dER, dMt, scale * ct);           // M.t = E.R (s ct) + E.t
            V3 q = m3t_vec(MR, dMt);
            dx[X_CAMT] += scale * q.x; dx[X_CAMT + 1] += scale * q.y; dx[X_CAMT + 2] += scale * q.z;
            *dscale = (dsv ? *dsv : 0.f) + dot(q, ct);
            g_store(dcam_ext, dER, dMt);
            dcam_ext[12] = dcam_ext[13] = dcam_ext[14] = dcam_ext[15] = 0.f;   // bottom row never used
        }
    };
#if defined(__HIP_DEVICE_COMPILE__)
    if (split && tid_all >= 64 && tid_all < 128) {
        // the second wave, beside the first wave's rotation backward: a wave's LDS operations execute in order, so its own
        // hand-overs need a compiler barrier only
        const int l = tid_all - 64;                     // d betas: needs the joints' offset gradients
        if (l < NJ) joint_offset_grad(l);
        __builtin_amdgcn_wave_barrier();
        if (l < 6 * NBETA) reduce_step1(l);
        __builtin_amdgcn_wave_barrier();
        if (l < NBETA) reduce_step2(l);
    }
    if (split && tid_all >= 192) {
        // the fourth wave, after global_orient's lane (below) or rather around it: d M, d transl and the camera row need nothing
        // of this phase
        const int l = tid_all - 192;
        if (l < 12) reduce_step1(6 * NBETA + l);
        __builtin_amdgcn_wave_barrier();
        if (l < 4) reduce_step2(NBETA + 24 + l);
    }
#endif
    auto rot_backward = [&](int j) {
        M3 dR = load_m3(sc.dR[j]);
        if (dPF && j >= 1) for (int e = 0; e < 9; ++e) dR.m[e] += dPF[9 * (j - 1) + e];
        if (aa22 && j <= 21) {
            V3 d = rodrigues_backward(v3(aa22[3 * j], aa22[3 * j + 1], aa22[3 * j + 2]), dR);
            daa22[3 * j] = d.x; daa22[3 * j + 1] = d.y; daa22[3 * j + 2] = d.z;
        } else if (j == 0) {
            GsCache c; gs_forward(x + X_SIXD, 1, &c);
            float d6[6]; gs_backward(c, dR, d6, 1);
            for (int e = 0; e < 6; ++e) dx[X_SIXD + e] += d6[e];
        } else if (j <= 21) {
            GsCache c; gs_forward(o + 6 * (j - 1), 1, &c);
            gs_backward(c, dR, dO + 6 * (j - 1), 1);
        } else if (j >= 25) {
            V3 d = rodrigues_backward(hand_aa(pm, x, j), dR);
            int h = (j - 25) / 15, f = (j - 25) % 15;
            sc.daa[h][3 * f] = d.x; sc.daa[h][3 * f + 1] = d.y; sc.daa[h][3 * f + 2] = d.z;
        }
    };
    if (split) {
        // the 21 body joints' 6D backward on the first wave, the fingers' Rodrigues backward on the third, global_orient (the body
        // joints' arithmetic on other pointers: another code path, which their wave would walk after its own) on the fourth
        if (tid_all >= 1 && tid_all < 25) rot_backward(tid_all);
        else if (tid_all >= 128 && tid_all < 192) { if (tid_all - 128 + 25 < NJ) rot_backward(tid_all - 128 + 25); }
        else if (tid_all == 192) rot_backward(0);
    } else {
        for (int j = tid; j < NJ; j += nthr) { joint_offset_grad(j); rot_backward(j); }
    }
    sync();
    FDC_FR_STAMP(1, 5);
    if (split) {                                         // only the hand components are left: 24 sums of 45 products
        const int t = tid_all - 64 + NBETA;
        if (tid_all >= 64 && tid_all < 128 && t < NBETA + 24) reduce_step2(t);
    } else {
        for (int t = tid; t < 6 * NBETA + 12; t += nthr) reduce_step1(t);
        sync();
        for (int t = tid; t < 64; t += nthr) reduce_step2(t);
    }
    FDC_FR_STAMP(1, 6);
}

}  // namespace fdc
