// Per-frame pose kernels of the optimiser loop (one workgroup per frame, four waves): 6D / hand PCA -> rotations, joint regression,
// kinematic chain, world joints (pose_fwd_kernel) and their backward with the fused parameter-space losses (pose_bwd_kernel).
// Part of the single translation unit csrc/fdcap.hip (included there, in this order; not a stand-alone header).
#pragma once

namespace {

struct SyncBlock {
    __device__ void operator()() const { __syncthreads(); }
};

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------

// Everything a frame's pose kernels read besides their per-joint state, staged in LDS by ONE batch of loads at kernel start:
// the kinematic tree's index arrays (the level loops chase level_start -> order -> parents -> child lists: ~30 dependent
// hops per frame from global memory), the collapsed joint regressor Jt / Jd, the hand PCA basis, and this frame's parameter
// row, camera row and scale.  Measured per frame (s_memtime): the phases that read these tables straight from global memory
// took 5.5 k (forward: rotations + joints), 6.5 k (backward: rotation gradients) and 14 k cycles (backward: the serial
// reductions over Jd / the PCA basis) of 25 k / 42 k-cycle kernels.
struct alignas(16) PoseStage {
    // the static part: ONE contiguous image (fdcap_ctx::pose_tab holds it in exactly this layout, built once per context), so that
    // it arrives as 13 sixteen-byte copy instructions instead of 24 of mixed sizes (late r4: the batch is bound by the CU's rate
    // of copy INSTRUCTIONS, ~24 cycles each with four workgroups staging at once, not by bytes)
    float Jd[NJ * 3 * NBETA + 2];   // 1650 (+ padding: every array stays 16-byte aligned)
    float hand_comp[2 * 12 * 45];   // 1080
    float Jt[NJ * 3 + 3];           // 165
    float hand_mean[90 + 2];
    int parents[NJ + 1], order[NJ + 1], level_start[MAX_LEVELS + 4], child_start[NJ + 1], child_list[NJ + 1], depth[NJ + 1];
    // the frame's own rows
    float x[XDIM + 2];
    float cam[16];
};
constexpr int PS_STATIC_FLOATS = (NJ * 3 * NBETA + 2) + 2 * 12 * 45 + (NJ * 3 + 3) + 92 + 5 * (NJ + 1) + (MAX_LEVELS + 4);
static_assert(PS_STATIC_FLOATS % 4 == 0 && offsetof(PoseStage, x) == PS_STATIC_FLOATS * 4, "the static image must end where the frame's rows begin");
constexpr int PS_STATIC_U16 = PS_STATIC_FLOATS / 4;          // 16-byte units: 823
// row strides of the forward's per-frame state that the backward stages: padded to 16-byte multiples so a row is 2 / 1 / 1 copy
// instructions instead of 8 / 3 / 2 (the optimiser's own buffers only: the operator-level workspaces keep the dense strides)
constexpr int RM_LD = NJ * 9 + 1, JR_LD = NJ * 3 + 3, O_LD = ODIM + 2;       // 496, 168, 128
static_assert(RM_LD % 4 == 0 && JR_LD % 4 == 0 && O_LD % 4 == 0, "16-byte rows");
// Staging by LDS-DMA (global_load_lds: global -> LDS without passing through registers; destination = wave-uniform LDS
// address + lane x size, source per lane).  A freshly launched kernel finds none of its inputs in its L2 and every DEPENDENT
// round trip at its start costs ~1-2.5 k cycles (s_memtime); with the copies issued back to back and ONE wait in front of the
// barrier the whole prologue is a single round trip, whatever else the kernel adds to the batch.  What this replaced, each
// measured: load-store loops (the compiler waits for each trip's load: 16 k cycles); two unrolled passes through registers
// (3 k alone, but loads under lane masks are branches whose merges -- and waits -- land between the loads once other code
// follows, the scheduler pairs unconditional loads with their stores, and any fence that would pin them sends the
// staging arrays to scratch).
typedef __attribute__((address_space(1))) const void* fdc_gptr_t;
typedef __attribute__((address_space(3))) void* fdc_lptr_t;
// one wave copies n units of 16 / 4 bytes: unit i = 64 k + lane.  g and lds 16- / 4-byte aligned; K = ceil(n / 64) trips.
template <int K>
__device__ __forceinline__ void glds16(const void* g, void* lds, int n) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (lane + 64 * k < n)
            __builtin_amdgcn_global_load_lds((fdc_gptr_t)((const char*)g + 16 * (lane + 64 * k)), (fdc_lptr_t)((char*)lds + 1024 * k), 16, 0, 0);
}
template <int K>
__device__ __forceinline__ void glds4(const void* g, void* lds, int n) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (lane + 64 * k < n)
            __builtin_amdgcn_global_load_lds((fdc_gptr_t)((const char*)g + 4 * (lane + 64 * k)), (fdc_lptr_t)((char*)lds + 256 * k), 4, 0, 0);
}
// the same for a workgroup of NW waves: unit i = 64 NW k + threadIdx.x (each wave's destination is wave-uniform)
template <int K, int NW, int SZ>
__device__ __forceinline__ void glds_wg(const void* g, void* lds, int n) {
    static_assert(SZ == 4 || SZ == 16, "unit size");
    const int tid = threadIdx.x, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (tid + 64 * NW * k < n) {
            const fdc_gptr_t src = (fdc_gptr_t)((const char*)g + SZ * (tid + 64 * NW * k));
            const fdc_lptr_t dst = (fdc_lptr_t)((char*)lds + SZ * 64 * (NW * k + wave));
            if constexpr (SZ == 16) __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
            else __builtin_amdgcn_global_load_lds(src, dst, 4, 0, 0);
        }
}
#ifdef FDC_DEBUG_BUFFERS
// instrumentation build only: does what the LDS-DMA batch left in LDS equal the global source?  [0] mismatches seen,
// then up to 15 records {table id, index, LDS bits, global bits, row, blockIdx, lane-of-index, 0}
__device__ unsigned g_stage_bad[8 * 16];
__device__ __forceinline__ void stage_check(int id, const void* g, const void* l, int n, int row) {
    const unsigned* gp = (const unsigned*)g; const unsigned* lp = (const unsigned*)l;
    for (int i = threadIdx.x; i < n; i += 64) {             // (called by the first wave only)
        const unsigned a = lp[i], b = gp[i];
        if (a != b) {
            const unsigned k = atomicAdd(&g_stage_bad[0], 1u);
            if (k < 15) {
                unsigned* r = g_stage_bad + 8 * (k + 1);
                r[0] = id; r[1] = i; r[2] = a; r[3] = b; r[4] = row; r[5] = blockIdx.x; r[6] = i & 63; r[7] = 0;
            }
        }
    }
}
#endif
constexpr int PS_NJD4 = (NJ * 3 * NBETA) / 4, PS_NHC4 = (2 * 12 * 45) / 4;      // 412, 270
// Issue the copies of the pose tables and this frame's rows (no wait).  A one-wave kernel issues its whole batch alone at
// ~100 cycles per copy instruction (s_memtime: 37 instructions = 5.1 k cycles in pose_fwd_kernel, 74 = 9.1 k in
// pose_bwd_kernel, linear in the count), so the pose kernels run POSE_NW = 4 waves per frame: each issues one PART of the
// batch under ONE wave-uniform branch (dealing single copies to waves by a running index makes hipcc wait after every copy),
// three of them only for that.
constexpr int POSE_NW = 4;
template <int PART>
__device__ __forceinline__ void stage_pose_part(const PoseModel& pm, PoseStage& t, const float* __restrict__ xrow,
                                                const float* __restrict__ camrow, bool rows = true) {
    // this wave's quarter of the static image (256 units of 16 bytes; the last quarter is short)
    static_assert(PS_STATIC_U16 <= 4 * 256, "four waves x four trips");
    constexpr int n = PS_STATIC_U16 - 256 * PART < 256 ? PS_STATIC_U16 - 256 * PART : 256;
    glds16<(n + 63) / 64>((const char*)pm.tab + 4096 * PART, (char*)&t + 4096 * PART, n);
    if constexpr (PART == 0) {
        if (rows) glds4<1>(camrow, t.cam, 16);
    } else if constexpr (PART == 2) {
        if (rows) glds4<2>(xrow, t.x, XDIM);
    }
}
// after the barrier that follows the copies: the model with its tables in LDS
__device__ __forceinline__ PoseModel stage_pose_model(const PoseModel& pm, PoseStage& t) {
    PoseModel l = pm;
    l.Jd = t.Jd; l.Jt = t.Jt; l.hand_comp = t.hand_comp; l.hand_mean = t.hand_mean;
    l.parents = t.parents; l.order = t.order; l.level_start = t.level_start; l.child_start = t.child_start; l.child_list = t.child_list;
    l.depth = t.depth;
    return l;
}

// One workgroup per frame: POSE_NW waves issue the staging copies, the first one does the frame's arithmetic.
// PARTS: the decoder output arrives as the four partial sums of vposer_fwd_fused_kernel (Opart, part_stride apart); they are
// added here in the fixed order of vp_sum_parts, kept in LDS for this frame and written to O for the backward.
template <bool PARTS>
__global__ __launch_bounds__(64 * POSE_NW) void pose_fwd_kernel(PoseModel pm, const float* __restrict__ X, float* __restrict__ O,
                                                      const float* __restrict__ CAM, const float* __restrict__ scale,
                                                      int row0, float* Rm, float* PF, float* Jrest, float* G, float* A,
                                                      float* M, float* Jw, const float* AA, const float* __restrict__ Opart,
                                                      size_t part_stride, int wo_lo = 0, int wo_hi = 0, DeferredStep ds = DeferredStep()) {
    __shared__ PoseScratch sc;
    __shared__ PoseStage stg;
    __shared__ float s_O[ODIM + 2];
    __shared__ float s_Op[PARTS ? VP_NQ : 1][ODIM + 2];
    FDC_FR_STAMP(0, 0);
    int r = row0 + blockIdx.x;
    if (r >= wo_lo && r < wo_hi) {
        // world-only rows (fdcap_opt_forward_ahead): the pose state of this row was computed before `scale` was stepped;
        // only M and the world joints depend on it -- refreshed from the stored joint transforms, pose_forward's own tail
        if (threadIdx.x >= 64) return;
        const float* x = X + (size_t)r * XDIM;
        M3 MR; V3 Mt;
        world_matrix(CAM + (size_t)r * 16, x, *scale, &MR, &Mt);
        const V3 transl = v3(x[X_TRANSL], x[X_TRANSL + 1], x[X_TRANSL + 2]);
        const int j = threadIdx.x;
        if (j < NJW) {
            const V3 w = world_joint(MR, Mt, g_trn(G + ((size_t)r * NJ + j) * 12), transl);
            float* o = Jw + ((size_t)r * NJW + j) * 3;
            o[0] = w.x; o[1] = w.y; o[2] = w.z;
        }
        if (j == 0) g_store(M + (size_t)r * 12, MR, Mt);
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const float* const xrow = X + (size_t)r * XDIM;
    const float* const camrow = CAM + (size_t)r * 16;
    // (ds.on: the frame's own rows are not copied -- the deferred step below writes the stepped rows into their LDS places)
    if (wave == 0) stage_pose_part<0>(pm, stg, xrow, camrow, !ds.on);
    else if (wave == 1) stage_pose_part<1>(pm, stg, xrow, camrow, !ds.on);
    else if (wave == 2) stage_pose_part<2>(pm, stg, xrow, camrow, !ds.on);
    else {
        stage_pose_part<3>(pm, stg, xrow, camrow, !ds.on);
        if (PARTS) {                                         // the decoder's partial sums ride in the same batch of copies
#pragma unroll
            for (int q = 0; q < VP_NQ; ++q) glds4<2>(Opart + (size_t)q * part_stride + (size_t)r * ODIM, s_Op[q], ODIM);
        }
    }
    const float sc_v = *scale;
    // A deferred optimiser step (DeferredStep, fdc_loss.h): this frame's row of body_rotation_rec / camera_ext takes its pending
    // Adam update here -- the loads ride in the staging batch, the stepped row goes to LDS (where the copy of the old one would
    // have gone) and back to global memory with both moments.
    if (ds.on) {                                             // (wave-uniform)
        const int t = (int)threadIdx.x;
        if (t < XDIM) {
            const size_t e = (size_t)(r - ds.row0) * XDIM + t;
            float pp = ds.x.p[e], mm = ds.x.m[e], vv = ds.x.v[e], gg = ds.x.g[e];
            const int col = t - X_LATENT;
            if (ds.dzpart && col >= 0 && col < VP_Z) gg += vp_sum_dz(ds.dzpart, ds.dz_stride, (size_t)r * VP_Z + col);
            adam_update(pp, mm, vv, gg, ds.x.a);
            ds.x.p[e] = pp; ds.x.m[e] = mm; ds.x.v[e] = vv;
            stg.x[t] = pp;
        } else if (t < XDIM + 16) {
            const int ec = t - XDIM;
            if (ds.cam.p) {
                const size_t e = (size_t)(r - ds.row0) * 16 + ec;
                float pp = ds.cam.p[e], mm = ds.cam.m[e], vv = ds.cam.v[e];
                adam_update(pp, mm, vv, ds.cam.g[e], ds.cam.a);
                ds.cam.p[e] = pp; ds.cam.m[e] = mm; ds.cam.v[e] = vv;
                stg.cam[ec] = pp;
            } else
                stg.cam[ec] = camrow[ec];
        }
    }
    __syncthreads();                                         // (every wave waits for its copies: vmcnt(0) in front of the barrier)
    // (all four waves stay: they share the frame's arithmetic -- pose_forward, split)
#ifdef FDC_DEBUG_BUFFERS
    if (!ds.on && threadIdx.x < 64) {
    stage_check(1, pm.Jd, stg.Jd, NJ * 3 * NBETA, r); stage_check(2, pm.hand_comp, stg.hand_comp, 2 * 12 * 45, r);
    stage_check(3, pm.Jt, stg.Jt, NJ * 3, r); stage_check(4, pm.hand_mean, stg.hand_mean, 90, r);
    stage_check(5, X + (size_t)r * XDIM, stg.x, XDIM, r); stage_check(6, CAM + (size_t)r * 16, stg.cam, 16, r);
    stage_check(7, pm.parents, stg.parents, NJ, r); stage_check(8, pm.order, stg.order, NJ, r);
    stage_check(9, pm.child_list, stg.child_list, NJ - 1, r); stage_check(10, pm.depth, stg.depth, NJ, r);
    stage_check(11, pm.child_start, stg.child_start, NJ + 1, r); stage_check(12, pm.level_start, stg.level_start, min(pm.nlevels, MAX_LEVELS) + 1, r);
    if (PARTS) for (int q = 0; q < VP_NQ; ++q) stage_check(20 + q, Opart + (size_t)q * part_stride + (size_t)r * ODIM, s_Op[q], ODIM, r);
    }
    __syncthreads();
#endif
    const PoseModel pml = stage_pose_model(pm, stg);
    if (PARTS) {
        for (int e = threadIdx.x; e < ODIM; e += 256) {
            const float v = (s_Op[0][e] + s_Op[1][e]) + (s_Op[2][e] + s_Op[3][e]);     // vp_sum_parts' order
            s_O[e] = v;
            O[(size_t)r * O_LD + e] = v;
        }
        __syncthreads();
    }
    if (PF && threadIdx.x < NBETA) PF[(size_t)r * NPFX + NPF + threadIdx.x] = stg.x[X_BETAS + threadIdx.x];
    if (PARTS) {
        pose_forward(pml, stg.x, s_O, stg.cam, sc_v, sc,
                     Rm ? Rm + (size_t)r * RM_LD : nullptr, PF ? PF + (size_t)r * NPFX : nullptr,
                     Jrest ? Jrest + (size_t)r * JR_LD : nullptr, G ? G + (size_t)r * NJ * 12 : nullptr,
                     A ? A + (size_t)r * NJ * 12 : nullptr, M ? M + (size_t)r * 12 : nullptr,
                     Jw ? Jw + (size_t)r * NJW * 3 : nullptr, threadIdx.x, 64, SyncBlock(), nullptr, 1);
    } else {
        pose_forward(pml, stg.x, O ? O + (size_t)r * ODIM : nullptr, stg.cam, sc_v, sc,
                     Rm ? Rm + (size_t)r * NJ * 9 : nullptr, PF ? PF + (size_t)r * NPFX : nullptr,
                     Jrest ? Jrest + (size_t)r * NJ * 3 : nullptr, G ? G + (size_t)r * NJ * 12 : nullptr,
                     A ? A + (size_t)r * NJ * 12 : nullptr, M ? M + (size_t)r * 12 : nullptr,
                     Jw ? Jw + (size_t)r * NJW * 3 : nullptr, threadIdx.x, 64, SyncBlock(),
                     AA ? AA + (size_t)r * 66 : nullptr, 1);
    }
}

// optional fused prologue of pose_bwd_kernel (X0 == nullptr: off)
// loss_rows (optional, logging iterations): this frame's partial sums of the printed terms, [row][LROW] floats in the slots of
// losses_d (0 rec, 1 z^2, 2 smoothing, 3 contact -- written by the skinning backward --, 4 world smoothing); summed over the
// rows in a fixed order by loss_rows_reduce_kernel.  (Atomics on the eight doubles serialise: 1024 frames x 4 adds made the
// separate param_loss_kernel 15 us and the skinning backward 8 us slower on logging iterations.)
struct ParamLossIn { const float* X0; const float* mask; const float* Jw; int frame0, n_total; float w_rec, w_sm, w_ws; int world_grad; float* loss_rows;
                     // (logging phase 2) the contact term that is only printed: this frame's sum of the robustified distances goes
                     // to slot 3 of loss_rows -- contact_loss_rows_kernel's 256-thread sum, thread for thread, without its launch
                     const float* cdist = nullptr; int cnc = 0; };

__global__ __launch_bounds__(64 * POSE_NW) void pose_bwd_kernel(PoseModel pm, const float* __restrict__ X, const float* __restrict__ O,
                                                      const float* __restrict__ CAM, const float* __restrict__ scale,
                                                      int row0, const float* Rm, const float* Jrest, const float* G,
                                                      const float* dA, const float* dPF, const float* dJw,
                                                      const float* dMv, const float* dsv, const float* dbeta_v,
                                                      int dbeta_stride, const float* dtransl_v, float* dX, float* dO,
                                                      float* dCAM, float* dscale_row, ParamLossIn pl, const float* dPF2, int dA_nj = NJ) {
    // dA_nj: rows of dA that were written (the rest are zero: SkinModel::ja_hi)
    // dPF2 (optional): second partial of dPF -- the data-gradient product split over K (panel_gemm3_rb2k_kernel); the row is
    // dPF + dPF2, d betas its columns NPF.. (dbeta_v must then be dPF + NPF, stride NPFX)
    __shared__ PoseScratch sc;
    __shared__ PoseStage stg;
    __shared__ float s_dJw[NJW * 3];
    // Everything this frame reads from global memory arrives in ONE batch of LDS-DMA copies (stage_pose_issue's comment): the
    // pose tables, the forward pass's per-joint state, the incoming gradient rows, and what the fused parameter-loss prologue
    // needs (neighbouring rows: two halo rows exist on either side of every owned row).  Fetched phase by phase -- as
    // pose_backward does for its generic callers -- they were ~8 dependent cold round trips.
    __shared__ __attribute__((aligned(16))) float s_dPF[NPFX];
    __shared__ __attribute__((aligned(16))) float s_O[O_LD], s_Jr[JR_LD];
    __shared__ float s_xn[4][XDIM + 2], s_x0[XDIM + 2], s_jw[3][NJW * 3 + 3], s_misc[32];
    __shared__ float s_dx[XDIM + 2];      // the parameter-gradient row: accumulated here (pose_backward adds to it from several
                                          // phases -- read-modify-write round trips on the global row), stored once at the end
    FDC_FR_STAMP(1, 0);
    const int r = row0 + blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const float* const xrow = X + (size_t)r * XDIM;
    const float* const camrow = CAM + (size_t)r * 16;
    if (wave == 0) {
        stage_pose_part<0>(pm, stg, xrow, camrow);
        if (pl.X0) {                                         // (the copies are dealt so that every wave issues 12-13 of them)
            const float* j = pl.Jw + (size_t)r * NJW * 3;
            glds4<2>(j - NJW * 3, s_jw[0], NJW * 3); glds4<2>(j, s_jw[1], NJW * 3); glds4<2>(j + NJW * 3, s_jw[2], NJW * 3);
        }
    } else if (wave == 1) {
        stage_pose_part<1>(pm, stg, xrow, camrow);
        glds16<1>(Jrest + (size_t)r * JR_LD, s_Jr, JR_LD / 4);
        glds16<(NJ * 3 + 63) / 64>(G + (size_t)r * NJ * 12, &sc.G[0][0], NJ * 3);              // rows of sc.G are 12 floats: a flat copy
        glds16<1>(O + (size_t)r * O_LD, s_O, O_LD / 4);
        if (dMv) glds4<1>(dMv + (size_t)r * 12, s_misc, 12);
        if (dsv) glds4<1>(dsv + r, s_misc + 12, 1);
        if (dtransl_v) glds4<1>(dtransl_v + (size_t)r * 3, s_misc + 13, 3);
        if (dbeta_v) glds4<1>(dbeta_v + (size_t)r * dbeta_stride, s_misc + 16, NBETA);
    } else if (wave == 2) {
        stage_pose_part<2>(pm, stg, xrow, camrow);
        if (dA) {                                          // waits in sc.dG: lane j reads row j, then overwrites it
            glds16<(NJ * 3 + 63) / 64>(dA + (size_t)r * NJ * 12, &sc.dG[0][0], dA_nj * 3);
            for (int e = dA_nj * 12 + (int)(threadIdx.x & 63); e < NJ * 12; e += 64) (&sc.dG[0][0])[e] = 0.f;   // rows nobody wrote
        }
        if (dPF) glds16<(NPFX / 4 + 63) / 64>(dPF + (size_t)r * NPFX, s_dPF, NPFX / 4);
        if (dPF2) glds16<(NPFX / 4 + 63) / 64>(dPF2 + (size_t)r * NPFX, &sc.dR[0][0], NPFX / 4);   // parked in sc.dR (written much later)
    } else if (pl.X0) {
        stage_pose_part<3>(pm, stg, xrow, camrow);
        const float* x = xrow;
        glds4<2>(x - 2 * XDIM, s_xn[0], XDIM); glds4<2>(x - XDIM, s_xn[1], XDIM);
        glds4<2>(x + XDIM, s_xn[2], XDIM); glds4<2>(x + 2 * XDIM, s_xn[3], XDIM);
        glds4<2>(pl.X0 + (size_t)r * XDIM, s_x0, XDIM);
        glds4<1>(pl.mask + r, s_misc + 27, 1);
    } else {
        stage_pose_part<3>(pm, stg, xrow, camrow);
        glds4<2>(dX + (size_t)r * XDIM, s_dx, XDIM);         // the row a separate param_loss_kernel launch initialised
    }
    const float sc_v = *scale;
    __shared__ float s_csum[POSE_NW];
    if (pl.cdist) {                                          // (kernel-uniform) contact_loss_rows_kernel's sum, same threads, same order
        float v = 0.f;
        for (int c = threadIdx.x; c < pl.cnc; c += 256) { float d; v += contact_term(pl.cdist[(size_t)r * pl.cnc + c], &d); }
        v = wave_sum64(v);
        if ((threadIdx.x & 63) == 0) s_csum[threadIdx.x >> 6] = v;
    }
    __syncthreads();                                         // (every wave waits for its copies: vmcnt(0) in front of the barrier)
    if (pl.cdist && threadIdx.x == 0) pl.loss_rows[(size_t)r * LROW + 3] = (s_csum[0] + s_csum[1]) + (s_csum[2] + s_csum[3]);
    // (all four waves stay for their share of pose_backward, split)
    FDC_FR_STAMP(1, 7);
    const PoseModel pml = stage_pose_model(pm, stg);
    const bool w0 = threadIdx.x < 64;
    if (dPF2) {
        const float* p2 = &sc.dR[0][0];
        for (int e = threadIdx.x; e < NPFX; e += 64 * POSE_NW) s_dPF[e] += p2[e];
        if (!pl.X0) __syncthreads();                         // (else: the barrier behind the loss prologue covers it)
    }
    if (pl.X0) {
        // param_loss_kernel's gradients formed here: dX row (=) data + temporal terms on the raw rows, world-smoothing
        // gradient of this frame's joints into LDS instead of a round trip through dJw
        const int g = pl.frame0 + blockIdx.x;
        const float lmask = s_misc[27];
        float l_rec = 0.f, l_vp = 0.f, l_sm = 0.f, l_ws = 0.f;
        // (two waves side by side: the first takes the parameter row's terms, the second the world joints')
        if (w0) {
            for (int e = threadIdx.x; e < XDIM; e += 64) {
                const float xc = stg.x[e];
                float rec = 0.f, sm = 0.f;
                s_dx[e] = param_loss_grad(g, pl.n_total, g >= 2 ? s_xn[0][e] : 0.f, g >= 1 ? s_xn[1][e] : 0.f, xc,
                                          g + 1 < pl.n_total ? s_xn[2][e] : 0.f, g + 2 < pl.n_total ? s_xn[3][e] : 0.f,
                                          s_x0[e], lmask, pl.w_rec, pl.w_sm, &rec, &sm);
                l_rec += rec; l_sm += sm;
                if (e >= X_LATENT && e < X_LATENT + 32) l_vp += xc * xc;
            }
            if (pl.loss_rows) {                              // kernel-uniform: logging iterations only
                l_rec = wave_sum64(l_rec); l_vp = wave_sum64(l_vp); l_sm = wave_sum64(l_sm);
                if (threadIdx.x == 0) {
                    float* lr = pl.loss_rows + (size_t)r * LROW;
                    lr[0] = l_rec; lr[1] = l_vp; lr[2] = l_sm;
                }
            }
        } else if (threadIdx.x < 128 && (pl.world_grad || pl.loss_rows)) {
            for (int e = threadIdx.x - 64; e < NJW * 3; e += 64) {
                float ws = 0.f;
                s_dJw[e] = world_smooth_grad(g, pl.n_total, g >= 1 ? s_jw[0][e] : 0.f, s_jw[1][e], g + 1 < pl.n_total ? s_jw[2][e] : 0.f,
                                             pl.w_ws, &ws);
                l_ws += ws;
            }
            if (pl.loss_rows) {
                l_ws = wave_sum64(l_ws);
                if (threadIdx.x == 64) pl.loss_rows[(size_t)r * LROW + 4] = l_ws;
            }
        }
        __syncthreads();
    }
    const float* dJw_row = (pl.X0 && pl.world_grad) ? s_dJw : (dJw ? dJw + (size_t)r * NJW * 3 : nullptr);
    pose_backward(pml, stg.x, s_O, stg.cam, sc_v,
                  (const float*)nullptr, s_Jr, (const float*)nullptr,     // (Rm: not read any more; G: already in sc.G)
                  dA ? &sc.dG[0][0] : nullptr, dPF ? s_dPF : nullptr,
                  dJw_row, dMv ? s_misc : nullptr,
                  dsv ? s_misc + 12 : nullptr, dbeta_v ? (dPF2 ? s_dPF + NPF : s_misc + 16) : nullptr,
                  dtransl_v ? s_misc + 13 : nullptr, sc, s_dx,
                  dO + (size_t)r * ODIM, dCAM + (size_t)r * 16, dscale_row + r, threadIdx.x, 64, SyncBlock(),
                  nullptr, nullptr, nullptr, 1);
    __syncthreads();
    for (int e = threadIdx.x; e < XDIM; e += 256) dX[(size_t)r * XDIM + e] = s_dx[e];
}

}  // namespace
