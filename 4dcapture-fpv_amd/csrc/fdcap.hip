// libfdcap_hip.so -- kernels + C-ABI (include/fdcap.h) of the MI355X global-optimisation path.
// gfx950 only.  See DESIGN.md for the data layout and the per-kernel rooflines.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

// Build requirement: this file is compiled WITHOUT packed fp32 instructions
//   -Xclang -target-feature -Xclang -packed-fp32-ops -DFDC_BUILD_NO_PK_F32
// (on the MI355X boxes this was developed on, v_pk_{fma,mul,add}_f32 now and then return a wrong low element while another
// kernel's MFMAs share the CU: tools/pk_f32_mfma_repro.hip).  The define travels with the flag so that a build script which
// drops one drops both and stops here; capi.load_library() reads it back through fdcap_build_info(), and
// tests/test_io_and_abi.py disassembles the library to check the flag really took effect.
#if !defined(FDC_BUILD_NO_PK_F32) && !defined(FDC_BUILD_ALLOW_PK_F32)
#error "build fdcap.hip with: -Xclang -target-feature -Xclang -packed-fp32-ops -DFDC_BUILD_NO_PK_F32 (see __graft_entry__.build)"
#endif

#ifdef FDC_PN_TIMING
// instrumentation build only: per-frame s_memtime stamps of the pose kernels [which][block][8]
__device__ unsigned long long g_fr_times[3][2048 * 8];
#define FDC_FR_STAMP(w, i) do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_fr_times[w][blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#ifndef FDC_FR_STAMP
#define FDC_FR_STAMP(w, i)
#endif
#include "../../include/fdcap.h"
#include "fdc_chamfer.h"
#include "fdc_comm.h"
#include "fdc_dct.h"
#include "fdc_fit2d.h"
#include "fdc_frame.h"
#include "fdc_gemm.h"
#include "fdc_host_setup.h"
#include "fdc_lbfgs.h"
#include "fdc_loss.h"
#include "fdc_math.h"
#include "fdc_panel.h"
#include "fdc_skin.h"
#include "fdc_trace.h"

using namespace fdc;

#define HIP_TRY(expr)                              \
    do {                                           \
        hipError_t _e = (expr);                    \
        if (_e != hipSuccess) return (int)_e;      \
    } while (0)

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
                                                      float* dCAM, float* dscale_row, ParamLossIn pl, const float* dPF2) {
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
        if (dA) glds16<(NJ * 3 + 63) / 64>(dA + (size_t)r * NJ * 12, &sc.dG[0][0], NJ * 3);   // waits in sc.dG: lane j reads row j, then overwrites it
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
    if (sm.vpack && !sm.S) {
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
template <bool CONTACT>
__global__ __launch_bounds__(256) void skin_bwd_kernel(SkinModel sm, int nc, const float* __restrict__ X,
                                                       const float* __restrict__ Voff, const float* __restrict__ A,
                                                       const float* __restrict__ M, const float* __restrict__ scale,
                                                       int row0, const float* dVw, float* dVoff,
                                                       float* __restrict__ dA, float* __restrict__ dbeta_v,
                                                       float* __restrict__ dtransl_v, float* __restrict__ dMv,
                                                       float* __restrict__ dsv, ContactGradIn cg) {
    constexpr int VCH = 1024;                      // vertices per LDS chunk
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
    __shared__ float sAf[NJ * 12];                  // this frame's skinning transforms (see skin_fwd_kernel)
    for (int i = tid; i < NJ * 12; i += 256) sAf[i] = A[(size_t)r * NJ * 12 + i];
#pragma unroll
    for (int i = 0; i < SKB_NACC; ++i) acc[i] = 0.f;
    for (int i = tid; i < NJ * 12; i += 256) sdA[i] = 0.f;
    __syncthreads();
    FDC_FR_STAMP(2, 1);
    for (int c0 = 0; c0 < nc; c0 += VCH) {
        const int c1 = min(nc, c0 + VCH);
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
            SkinFwd f = skin_forward_vertex(sm, c, x + X_BETAS, Voff + 3 * qi, sAf, transl, M + (size_t)r * 12, s);
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
            SkinBwd b = skin_backward_vertex(f, M + (size_t)r * 12, s, g);
            dVoff[3 * qi] = b.dvp.x; dVoff[3 * qi + 1] = b.dvp.y; dVoff[3 * qi + 2] = b.dvp.z;
            if (sm.S)                                       // else: d betas = dVoff x shapedirs, columns 486.. of the blend data-gradient GEMM
                for (int l = 0; l < NBETA; ++l)
                    acc[l] += sm.S[(3 * c) * 10 + l] * b.dvp.x + sm.S[(3 * c + 1) * 10 + l] * b.dvp.y +
                              sm.S[(3 * c + 2) * 10 + l] * b.dvp.z;
            acc[NBETA] += b.gv.x; acc[NBETA + 1] += b.gv.y; acc[NBETA + 2] += b.gv.z;
#pragma unroll
            for (int e = 0; e < 12; ++e) acc[NBETA + 3 + e] += b.dM[e];
            acc[NBETA + 15] += b.ds;
#pragma unroll
            for (int e = 0; e < 12; ++e) sdT[(c - c0) * 12 + e] = b.dT[e];
        }
        __syncthreads();
        FDC_FR_STAMP(2, 2);
        // dA_j += sum_v w_vj dT_v, ordered and atomic-free (run-to-run reproducible): the joints are dealt
        // to the 4 waves; a wave's lanes stride over joint j's vertex list (ascending, restricted to this
        // chunk by two binary searches) and are combined by a butterfly
        // the non-empty joints (one ballot over the preloaded list bounds) are dealt to the four waves in turn
        unsigned long long jact = __ballot(jhi > jlo);
        for (int kact = 0; jact; ++kact) {
            const int j = __ffsll((long long)jact) - 1;
            jact &= jact - 1;
            if ((kact & 3) != wave) continue;               // wave-uniform
            int lo = __builtin_amdgcn_readlane(jlo, j), hi = __builtin_amdgcn_readlane(jhi, j);
            if (nc > VCH) {
                int a = lo, bnd = hi;
                while (a < bnd) { int m = (a + bnd) >> 1; if (sm.csc_v[m] < c0) a = m + 1; else bnd = m; }
                lo = a; bnd = hi;
                while (a < bnd) { int m = (a + bnd) >> 1; if (sm.csc_v[m] < c1) a = m + 1; else bnd = m; }
                hi = a;
                if (lo == hi) continue;
            }
            float pa[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) pa[e] = 0.f;
            for (int i = lo + lane; i < hi; i += 64) {
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
    for (int i = tid; i < NJ * 12; i += 256) dA[(size_t)r * NJ * 12 + i] = sdA[i];
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
    for (int i = tid; i < NJ * 12; i += 256) dA[(size_t)r * NJ * 12 + i] = sdA[i];
    if (tid >= NBETA && tid < SKB_NACC) {
        const float v = sred[0][tid] + sred[1][tid] + sred[2][tid] + sred[3][tid];
        if (tid < NBETA + 3) dtransl_v[(size_t)r * 3 + tid - NBETA] = v;
        else if (tid < NBETA + 15) dMv[(size_t)r * 12 + tid - NBETA - 3] = v;
        else dsv[r] = v;
    }
    FDC_FR_STAMP(2, 4);
}

// mode 'local', cal_loss2 (:404-405): d/dV of mean |second difference over frames| of ALL world vertices.
// V is [rows, nv3] (nv3 = 3 * vertices); owned rows start at row0, global frame = frame0 + blockIdx.y.
__global__ void vert_smooth_kernel(const float* __restrict__ V, size_t nv3, int row0, int frame0, int n_total,
                                   float w_over_cnt, float* __restrict__ dV, double* __restrict__ loss_sum) {
    __shared__ float sred[4];
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int r = row0 + blockIdx.y, g = frame0 + blockIdx.y;
    float ab = 0.f;
    if (e < nv3) {
        const float* v = V + (size_t)r * nv3 + e;
        const float x0 = v[0];
        const float xm2 = g >= 2 ? v[-2 * (ptrdiff_t)nv3] : 0.f, xm1 = g >= 1 ? v[-(ptrdiff_t)nv3] : 0.f;
        const float xp1 = g + 1 < n_total ? v[nv3] : 0.f, xp2 = g + 2 < n_total ? v[2 * nv3] : 0.f;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        if (g <= n_total - 3) { float d = second_diff(x0, xp1, xp2); s0 = sgn(d); ab = fabsf(d); }
        if (g >= 1 && g <= n_total - 2) s1 = sgn(second_diff(xm1, x0, xp1));
        if (g >= 2) s2 = sgn(second_diff(xm2, xm1, x0));
        dV[(size_t)r * nv3 + e] = (s0 - 2.f * s1 + s2) * w_over_cnt;
    }
    ab = wave_sum(ab);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = ab;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, (double)((sred[0] + sred[1]) + (sred[2] + sred[3])));
}

// mode 'local', cal_loss2 (:415-429): foot-skate term  mean|dL * w_left| + mean|dR * w_right| on the first
// difference over frames of the left / right contact vertices; adds its gradient into dV (full-mesh layout).
// vid[c] = mesh vertex of contact slot c in the CALLER's order (first n_left = left part); wgt[N] per frame.
__global__ void foot_skate_kernel(const float* __restrict__ V, size_t nv3, const int* __restrict__ vid, int nc,
                                  int n_left, const float* __restrict__ wgt, int row0, int frame0, int n_total,
                                  float* __restrict__ dV, double* __restrict__ loss_sum) {
    __shared__ float sred[4];
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int r = row0 + blockIdx.y, g = frame0 + blockIdx.y;
    float ab = 0.f;
    if (t < nc * 3) {
        const int c = t / 3, k = t % 3;
        const bool is_left = c < n_left;
        const int npart = is_left ? n_left : nc - n_left;
        const float inv = 1.f / ((float)(n_total - 1) * (float)npart * 3.f);
        const size_t e = (size_t)vid[c] * 3 + k;
        const float* v = V + (size_t)r * nv3 + e;
        float grad = 0.f;
        // weight_right = w, weight_left = 1 - w, both zeroed below 0.5 (:418-422); pair (i, i+1) uses w[i+1]
        if (g + 1 < n_total) {
            float w = wgt[g + 1];
            w = is_left ? 1.f - w : w;
            w = w < 0.5f ? 0.f : w;
            float d = (v[0] - v[nv3]) * w;
            ab = fabsf(d);
            grad += sgn(d) * w;
        }
        if (g >= 1) {
            float w = wgt[g];
            w = is_left ? 1.f - w : w;
            w = w < 0.5f ? 0.f : w;
            grad -= sgn((v[-(ptrdiff_t)nv3] - v[0]) * w) * w;
        }
        dV[(size_t)r * nv3 + e] += grad * inv;
        ab *= inv;                                          // the two parts have different denominators
    }
    ab = wave_sum(ab);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = ab;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, (double)((sred[0] + sred[1]) + (sred[2] + sred[3])));
}

// detect_contact (:355-364): per frame, mean squared NN distance of the left part and left / (left + left)
__global__ void detect_contact_kernel(const float* __restrict__ dist, const int* __restrict__ perm, int nc, int n_left,
                                      int row0, float* __restrict__ weight_left) {
    __shared__ float sred[4];
    const int r = row0 + blockIdx.x;
    float a = 0.f;
    for (int c = threadIdx.x; c < nc; c += 256)
        if (perm[c] < n_left) a += dist[(size_t)r * nc + c];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        float left = ((sred[0] + sred[1]) + (sred[2] + sred[3])) / (float)n_left;
        weight_left[blockIdx.x] = left / (left + left);
    }
}

// sum of the contact robustifier only (phase-2 logging): block per frame -> loss_rows[r][3]
__global__ __launch_bounds__(256) void contact_loss_rows_kernel(const float* __restrict__ dist, int nc, int row0, float* __restrict__ loss_rows) {
    __shared__ float sred[4];
    const int r = row0 + blockIdx.x;
    float v = 0.f;
    for (int c = threadIdx.x; c < nc; c += 256) { float d; v += contact_term(dist[(size_t)r * nc + c], &d); }
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) loss_rows[(size_t)r * LROW + 3] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
}

// block (128 threads) per owned frame: data + temporal terms on the raw rows, optional world
// smoothing on joints.  Initialises dX (=) and dJw (=).
__global__ __launch_bounds__(128) void param_loss_kernel(const float* __restrict__ X, const float* __restrict__ X0,
                                                         const float* __restrict__ mask, const float* __restrict__ Jw,
                                                         int row0, int frame0, int n_total, float w_rec_over_cnt,
                                                         float w_sm_over_cnt, float w_ws_over_cnt, int world_grad,
                                                         float* __restrict__ dX, float* __restrict__ dJw,
                                                         double* __restrict__ losses) {
    __shared__ float sred[2][4];
    const int tid = threadIdx.x;
    const int r = row0 + blockIdx.x;
    const int g = frame0 + blockIdx.x;
    float rec = 0.f, sm = 0.f, ws = 0.f, vp = 0.f;
    if (tid < XDIM) {
        const float* x = X + (size_t)r * XDIM + tid;
        float xm2 = (g >= 2) ? x[-2 * XDIM] : 0.f;
        float xm1 = (g >= 1) ? x[-XDIM] : 0.f;
        float xp1 = (g + 1 < n_total) ? x[XDIM] : 0.f;
        float xp2 = (g + 2 < n_total) ? x[2 * XDIM] : 0.f;
        dX[(size_t)r * XDIM + tid] = param_loss_grad(g, n_total, xm2, xm1, x[0], xp1, xp2, X0[(size_t)r * XDIM + tid],
                                                     mask[r], w_rec_over_cnt, w_sm_over_cnt, &rec, &sm);
        if (tid >= X_LATENT && tid < X_LATENT + 32) vp = x[0] * x[0];
    }
    if (tid < NJW * 3) {
        const float* j = Jw + (size_t)r * NJW * 3 + tid;
        float jm1 = (g >= 1) ? j[-NJW * 3] : 0.f;
        float jp1 = (g + 1 < n_total) ? j[NJW * 3] : 0.f;
        float gr = world_smooth_grad(g, n_total, jm1, j[0], jp1, w_ws_over_cnt, &ws);
        if (world_grad) dJw[(size_t)r * NJW * 3 + tid] = gr;
    }
    if (!losses) return;                                   // partial sums only on logging iterations (block-uniform)
    float vals[4] = {rec, vp, sm, ws};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = wave_sum(vals[i]);
        if ((tid & 63) == 0) sred[tid >> 6][i] = v;
    }
    __syncthreads();
    if (tid < 4) {
        const int slot[4] = {0, 1, 2, 4};
        atomicAdd(&losses[slot[tid]], (double)(sred[0][tid] + sred[1][tid]));
    }
}

// End of a logging backward, one block: the per-frame partials of the slots in `mask` summed over rows [row0, row0 + n) in
// double, in a fixed order (thread t: rows t, t + 256, ...; then a tree) -- deterministic, unlike the atomics it replaces --
// and stored to (assign != 0: every slot, the others zero) or added to losses[]; then the sum of the per-frame
// d loss / d scale partials (thread t: rows t, t + 256, ...; butterfly; the order of the step kernels' own reduction).
// (dscale_out may be null: the launch that also steps `scale` forms that sum itself, in the same order)
__global__ __launch_bounds__(256) void loss_rows_reduce_kernel(const float* __restrict__ rows, int row0, int n, unsigned mask, int assign,
                                                               double* __restrict__ losses, const float* __restrict__ dscale_row,
                                                               float* __restrict__ dscale_out) {
    loss_rows_reduce_block(rows, row0, n, mask, assign, losses, dscale_row, dscale_out);
}

// dzpart != nullptr: p is body_rotation_rec from row `row0` on, and the latent columns' gradient still lacks the VPoser backward's
// four partials (vp_sum_dz: the sum the fold kernel would have added to g first, in its order)
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ g, size_t n, AdamScalars a, int zero_grad,
                            const float* __restrict__ dzpart = nullptr, size_t dz_stride = 0, int row0 = 0) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float pp = p[i], mm = m[i], vv = v[i];
    float gg = zero_grad ? 0.f : g[i];
    if (dzpart) {
        const int row = (int)(i / XDIM), col = (int)(i % XDIM) - X_LATENT;
        if (col >= 0 && col < VP_Z) gg += vp_sum_dz(dzpart, dz_stride, (size_t)(row0 + row) * VP_Z + col);
    }
    adam_update(pp, mm, vv, gg, a);
    p[i] = pp; m[i] = mm; v[i] = vv;
}

// One launch for the whole optimizer.step() of an iteration (:592): blocks [0, nb_x) update body_rotation_rec,
// [nb_x, nb_x + nb_cam) camera_ext, the last block `scale` -- first reducing the per-frame d loss / d scale
// partials in loss_rows_reduce_kernel's fixed order when `reduce_n` > 0 (single-GPU; a sharded run gets the sum
// from the exchange instead).
// Sharded runs: the message of the iteration's one collective -- [first 2 | last 2 owned rows] of (x | camera_ext) +
// this rank's d loss / d scale -- is written by the same launch (xch != nullptr): every thread that updates a boundary-row
// element also stores it into its slot, the last block adds the reduced scale gradient (and the camera_ext rows while
// camera_ext is not being stepped).
constexpr int XCH_ROW = XDIM + 16;                 // 94 floats
constexpr int XCH_LEN = 4 * XCH_ROW + 8;           // + dscale partial (+ padding to 32 B)
__global__ __launch_bounds__(256) void adam_step_kernel(AdamTensor x, AdamTensor cam, AdamTensor sc, int nb_x, int nb_cam,
                                                        const float* __restrict__ dscale_row, int row0, int reduce_n,
                                                        float* __restrict__ dscale, int scale_zero_grad,
                                                        float* __restrict__ xch, int n_local, const float* __restrict__ cam_rows,
                                                        const float* __restrict__ dzpart, size_t dz_stride, LogReduceIn lg) {
    const int b = blockIdx.x;
    if (b == nb_x + nb_cam + 1) {                      // (only launched when a logging backward left its sums to this launch)
        loss_rows_reduce_block(lg.rows, row0, lg.n, lg.mask, lg.assign, lg.losses, dscale_row, nullptr);
        return;
    }
    if (b < nb_x + nb_cam) {
        const bool is_x = b < nb_x;
        // (field by field: a reference selected between two by-value kernel arguments is an address into the argument
        // segment, and its fields then arrive one dependent scalar load after the other -- four cold round trips)
        float* const tp = is_x ? x.p : cam.p;
        float* const tm = is_x ? x.m : cam.m;
        float* const tv = is_x ? x.v : cam.v;
        const float* const tg = is_x ? x.g : cam.g;
        const size_t tn = is_x ? x.n : cam.n;
        const AdamScalars ta = is_x ? x.a : cam.a;
        const size_t i = (size_t)(is_x ? b : b - nb_x) * 256 + threadIdx.x;
        if (i >= tn) return;
        float pp = tp[i], mm = tm[i], vv = tv[i];
        float gg = tg[i];
        if (is_x && dzpart) {                          // latent columns: + the four partials of vposer_bwd_fused_kernel (row0 = first owned row)
            const int lr = (int)(i / XDIM), col = (int)(i % XDIM) - X_LATENT;
            if (col >= 0 && col < VP_Z) gg += vp_sum_dz(dzpart, dz_stride, (size_t)(row0 + lr) * VP_Z + col);
        }
        adam_update(pp, mm, vv, gg, ta);
        tp[i] = pp; tm[i] = mm; tv[i] = vv;
        if (xch) {
            const int w = is_x ? XDIM : 16, lr = (int)(i / w), e = (int)(i % w) + (is_x ? 0 : XDIM);
            if (lr < 2) xch[lr * XCH_ROW + e] = pp;                                    // first two owned rows: slots 0, 1
            if (lr >= n_local - 2) xch[(2 + lr - (n_local - 2)) * XCH_ROW + e] = pp;  // last two: slots 2, 3
        }
        return;
    }
    __shared__ float sred[4];
    float g = 0.f;
    if (reduce_n > 0) {
        float a = 0.f;
        for (int i = threadIdx.x; i < reduce_n; i += 256) a += dscale_row[row0 + i];
        a = wave_sum(a);
        if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = a;
        __syncthreads();
        g = (sred[0] + sred[1]) + (sred[2] + sred[3]);
        if (threadIdx.x == 0) *dscale = g;
    } else {
        g = *dscale;
    }
    if (xch) {
        if (threadIdx.x < 8) xch[4 * XCH_ROW + threadIdx.x] = threadIdx.x == 0 ? g : 0.f;
        if (nb_cam == 0 && threadIdx.x < 64) {                                         // camera_ext unchanged this iteration
            const int slot = threadIdx.x >> 4, e = threadIdx.x & 15;
            const int row = slot < 2 ? 2 + slot : n_local + slot - 2;                  // buffer rows (owned rows start at 2)
            xch[slot * XCH_ROW + XDIM + e] = cam_rows[(size_t)row * 16 + e];
        }
    }
    if (threadIdx.x == 0 && sc.p) {
        float pp = *sc.p, mm = *sc.m, vv = *sc.v;
        adam_update(pp, mm, vv, scale_zero_grad ? 0.f : g, sc.a);
        *sc.p = pp; *sc.m = mm; *sc.v = vv;
    }
}

// halo rows <- neighbours' boundary rows; scale gradient = sum over ranks in rank order (same bits everywhere), then
// Adam on `scale` when sc.p is set (same launch: the sharded iteration tail is latency-bound)
__global__ void unpack_exchange_kernel(const float* __restrict__ all, int rank, int world, int n_local, float* __restrict__ X,
                                       float* __restrict__ CAM, float* __restrict__ dscale, AdamTensor sc, int scale_zero_grad) {
    int t = threadIdx.x;
    if (t < 4 * XCH_ROW) {
        int k = t / XCH_ROW, e = t % XCH_ROW;
        // k = 0,1: left halo rows 0,1 <- last two rows of rank-1 (its slots 2,3); k = 2,3: right halo <- first two of rank+1
        int src_rank = (k < 2) ? rank - 1 : rank + 1;
        if (src_rank >= 0 && src_rank < world) {
            int slot = (k < 2) ? 2 + k : k - 2;
            float v = all[(size_t)src_rank * XCH_LEN + slot * XCH_ROW + e];
            int row = (k < 2) ? k : n_local + k;                 // rows 0,1 and n_local+2, n_local+3
            if (e < XDIM) X[(size_t)row * XDIM + e] = v; else CAM[(size_t)row * 16 + e - XDIM] = v;
        }
    } else if (t == 4 * XCH_ROW) {
        float s = 0.f;
        for (int r = 0; r < world; ++r) s += all[(size_t)r * XCH_LEN + 4 * XCH_ROW];
        if (dscale) *dscale = s;                                 // (null: halo rows only, fdcap_opt_halo_exchange)
        if (sc.p) {
            float pp = *sc.p, mm = *sc.m, vv = *sc.v;
            adam_update(pp, mm, vv, scale_zero_grad ? 0.f : s, sc.a);
            *sc.p = pp; *sc.m = mm; *sc.v = vv;
        }
    }
}

__global__ void p75_to_78_kernel(const float* __restrict__ in, int B, float* __restrict__ out) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* p = in + (size_t)b * 75;
    float* x = out + (size_t)b * XDIM;
    for (int i = 0; i < 3; ++i) x[i] = p[i];
    M3 R = tgm_aa_to_rotmat(v3(p[3], p[4], p[5]));
    // first two COLUMNS, flattened row-major (global_optimization.py:101-102)
    x[3] = R.m[0]; x[4] = R.m[1]; x[5] = R.m[3]; x[6] = R.m[4]; x[7] = R.m[6]; x[8] = R.m[7];
    for (int i = 6; i < 75; ++i) x[i + 3] = p[i];
}

__global__ void p78_to_75_kernel(const float* __restrict__ in, int B, float* __restrict__ out) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* x = in + (size_t)b * XDIM;
    float* p = out + (size_t)b * 75;
    for (int i = 0; i < 3; ++i) p[i] = x[i];
    V3 aa = tgm_rotmat_to_aa(gs_forward(x + X_SIXD, 1, nullptr));
    p[3] = aa.x; p[4] = aa.y; p[5] = aa.z;
    for (int i = 9; i < XDIM; ++i) p[i - 3] = x[i];
}

// O[B,126] -> rot[B,21,9] (+ optional aa[B,63])
__global__ void sixd_to_rot_kernel(const float* __restrict__ O, int n, float* __restrict__ rot, float* __restrict__ aa) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    M3 R = gs_forward(O + (size_t)i * 6, 1, nullptr);
    if (rot) for (int e = 0; e < 9; ++e) rot[(size_t)i * 9 + e] = R.m[e];
    if (aa) { V3 a = tgm_rotmat_to_aa(R); aa[(size_t)i * 3] = a.x; aa[(size_t)i * 3 + 1] = a.y; aa[(size_t)i * 3 + 2] = a.z; }
}

__global__ void joints_out_kernel(const float* __restrict__ G, const float* __restrict__ X, int ldx, int B, float* __restrict__ J) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * NJ) return;
    int b = i / NJ;
    const float* g = G + (size_t)i * 12;
    const float* x = X + (size_t)b * ldx;
    J[(size_t)i * 3] = g[3] + x[0]; J[(size_t)i * 3 + 1] = g[7] + x[1]; J[(size_t)i * 3 + 2] = g[11] + x[2];
}

// operator-level inputs -> a 78-wide row (6D / latent slots unused) + the 22 axis-angle joints
__global__ void assemble_rows_kernel(const float* __restrict__ go, const float* __restrict__ bp, const float* __restrict__ betas,
                                     const float* __restrict__ lh, const float* __restrict__ rh, const float* __restrict__ transl,
                                     int B, float* __restrict__ X, float* __restrict__ AA) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float* x = X + (size_t)b * XDIM;
    for (int i = 0; i < XDIM; ++i) x[i] = 0.f;
    for (int i = 0; i < 3; ++i) x[X_TRANSL + i] = transl[3 * b + i];
    for (int i = 0; i < NBETA; ++i) x[X_BETAS + i] = betas[NBETA * b + i];
    for (int i = 0; i < 12; ++i) { x[X_LH + i] = lh[12 * b + i]; x[X_RH + i] = rh[12 * b + i]; }
    float* a = AA + (size_t)b * 66;
    for (int i = 0; i < 3; ++i) a[i] = go[3 * b + i];
    for (int i = 0; i < 63; ++i) a[3 + i] = bp[63 * b + i];
}

// ---- operator-level backward helpers (fdcap_vposer_decode_bwd / fdcap_smplx_backward; not on the optimiser's path) ----------
// decoder output O[B,126] + gradients of its rotation matrices (g_rot [n,9], may be null) and / or of their tgm angle-axis
// form (g_aa [n,3], may be null) -> dO[n,6]: through tgm's R -> aa (fdc_math.h) and the Gram-Schmidt step
__global__ void vposer_out_bwd_kernel(const float* __restrict__ O, int n, const float* __restrict__ g_rot, const float* __restrict__ g_aa,
                                      float* __restrict__ dO) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    GsCache c;
    const M3 R = gs_forward(O + (size_t)i * 6, 1, &c);
    M3 dR = m3_zero();
    if (g_rot) for (int e = 0; e < 9; ++e) dR.m[e] = g_rot[(size_t)i * 9 + e];
    if (g_aa) m3_add(dR, tgm_rotmat_to_aa_backward(R, v3(g_aa[(size_t)i * 3], g_aa[(size_t)i * 3 + 1], g_aa[(size_t)i * 3 + 2])));
    gs_backward(c, dR, dO + (size_t)i * 6, 1);
}
__global__ void vposer_fold_dz_rows_kernel(const float* __restrict__ part, size_t part_stride, int nrows, float* __restrict__ gz) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < nrows * VP_Z) gz[e] = vp_sum_dz(part, part_stride, (size_t)e);
}
// one wave per frame: pose_backward on global pointers (generic form; the optimiser's pose_bwd_kernel is the staged one)
__global__ __launch_bounds__(64) void pose_bwd_op_kernel(PoseModel pm, const float* __restrict__ X, const float* __restrict__ AA,
                                                         const float* Rm, const float* Jrest, const float* G, const float* dA,
                                                         const float* dPF, const float* dtransl_v, const float* dJb,
                                                         float* dX, float* dAA) {
    __shared__ PoseScratch sc;
    __shared__ float s_cam[16], s_dO[ODIM], s_dcam[16], s_ds[1];
    const int r = blockIdx.x;
    if (threadIdx.x < 16) s_cam[threadIdx.x] = 0.f;
    __syncthreads();
    pose_backward(pm, X + (size_t)r * XDIM, (const float*)nullptr, s_cam, 0.f, Rm + (size_t)r * NJ * 9, Jrest + (size_t)r * NJ * 3,
                  G + (size_t)r * NJ * 12, dA ? dA + (size_t)r * NJ * 12 : nullptr, dPF ? dPF + (size_t)r * NPFX : nullptr,
                  (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, dPF ? dPF + (size_t)r * NPFX + NPF : nullptr,
                  dtransl_v ? dtransl_v + (size_t)r * 3 : nullptr, sc, dX + (size_t)r * XDIM, s_dO, s_dcam, s_ds, threadIdx.x, 64,
                  SyncBlock(), AA + (size_t)r * 66, dAA + (size_t)r * 66, dJb ? dJb + (size_t)r * NJ * 3 : nullptr);
}
__global__ void smplx_bwd_split_kernel(const float* __restrict__ dX, const float* __restrict__ dAA, int B, float* g_go, float* g_bp,
                                       float* g_betas, float* g_lh, float* g_rh, float* g_transl) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* x = dX + (size_t)b * XDIM;
    const float* a = dAA + (size_t)b * 66;
    if (g_go) for (int i = 0; i < 3; ++i) g_go[3 * b + i] = a[i];
    if (g_bp) for (int i = 0; i < 63; ++i) g_bp[63 * b + i] = a[3 + i];
    if (g_betas) for (int i = 0; i < NBETA; ++i) g_betas[NBETA * b + i] = x[X_BETAS + i];
    if (g_lh) for (int i = 0; i < 12; ++i) g_lh[12 * b + i] = x[X_LH + i];
    if (g_rh) for (int i = 0; i < 12; ++i) g_rh[12 * b + i] = x[X_RH + i];
    if (g_transl) for (int i = 0; i < 3; ++i) g_transl[3 * b + i] = x[X_TRANSL + i];
}
__global__ void identity_rows_kernel(float* __restrict__ M, int B, float* __restrict__ one) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *one = 1.f;
    if (i < B * 12) { const int e = i % 12; M[i] = (e == 0 || e == 5 || e == 10) ? 1.f : 0.f; }
}

// count of non-finite values in p[0, n) added to *count (optional --check-finite hook; never on by default)
__global__ void count_nonfinite_kernel(const float* __restrict__ p, size_t n, int* __restrict__ count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // (exponent bits all ones -- tested on the bit pattern: the library is built with -fno-honor-nans, which lets isfinite() fold)
    const bool bad = i < n && (__float_as_uint(p[i]) & 0x7f800000u) == 0x7f800000u;
    const unsigned long long m = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, __popcll(m));
}

// dst[r, perm[c], :] = src[r, c, :]   (internal contact-slot order -> caller's order)
template <class T>
__global__ void unpermute_kernel(const T* __restrict__ src, const int* __restrict__ perm, int rows, int nc, int w,
                                 T* __restrict__ dst) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * nc * w) return;
    int k = i % w;
    size_t rc = i / w;
    int c = rc % nc, r = rc / nc;
    dst[((size_t)r * nc + perm[c]) * w + k] = src[i];
}


// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t ensure(size_t count) {
        if (count <= n) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    hipError_t upload(const T* h, size_t count) {
        hipError_t e = ensure(count);
        if (e != hipSuccess) return e;
        return count ? hipMemcpy(p, h, count * sizeof(T), hipMemcpyHostToDevice) : hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

struct SkinSet {          // skinning constants for a vertex set (all V, or the contact subset)
    int nv = 0, K = 0, nnz = 0;
    int ldp = 0;                                // row stride of posedirs: 3*nv rounded up to a multiple of 4 (16-byte rows)
    DevBuf<float> vt, ww, posedirs, csc_w;      // posedirs [496, ldp] = [posedirs ; shapedirs^T], zero padding
    DevBuf<int> wj, csc_start, csc_v;
    DevBuf<float4> vpack;                       // SkinModel::vpack / csc_v16 (K <= 4 and nv <= 65535 only)
    DevBuf<unsigned short> csc_v16;
    // the blend matrix in MFMA fragment order (fdc_panel.h), built for vertex sets whose K = 3 nv image fits the LDS slabs:
    // pn_fwd: B(k, n) = posedirs[k, n] (offsets = [pose feature | betas] x B), pn_bwd: B(k, n) = posedirs[n, k] (data gradient)
    DevBuf<float> pn_fwd_f, pn_bwd_f;
    PanelB pn_fwd, pn_bwd;
    DevBuf<unsigned> pn_fwd3_f, pn_bwd3_f;      // the same two operands as three bf16 planes (panel_gemm3_kernel)
    PanelB3 pn_fwd3, pn_bwd3;
    SkinModel model() const {
        SkinModel m; m.vt = vt.p; m.S = nullptr; m.wj = wj.p; m.ww = ww.p; m.K = K;
        m.csc_start = csc_start.p; m.csc_v = csc_v.p; m.csc_w = csc_w.p;
        m.vpack = (const float*)vpack.p; m.csc_v16 = csc_v16.p;
        return m;
    }
    void release() { vt.release(); ww.release(); posedirs.release(); wj.release(); csc_w.release(); csc_start.release(); csc_v.release();
                     vpack.release(); csc_v16.release();
                     pn_fwd_f.release(); pn_bwd_f.release(); pn_fwd = PanelB(); pn_bwd = PanelB();
                     pn_fwd3_f.release(); pn_bwd3_f.release(); pn_fwd3 = PanelB3(); pn_bwd3 = PanelB3(); }
};

struct OptState {
    fdcap_opt_config cfg;
    int R = 0;            // rows = n_local + 4
    bool contact_on = false;
    int nsplit = 8;           // scene splits of the in-loop NN launch
    int nsplit_bf = 8;        // ... of a brute-force launch (timing API)
    struct Ext { float* p = nullptr; } X, CAM, scale, dscale;   // caller-owned, registered
    struct ExtD { double* p = nullptr; } losses;
    DevBuf<float> X0, mask, mX, vX, mCAM, vCAM, mS, vS;
    // fdcap_opt_backward_and_step: the rows' part of iteration ii's optimiser step, to be applied by the next forward's first two
    // launches (DeferredStep, fdc_loss.h; `scale` was stepped by the backward's last launch) -- or by opt_sync()
    struct { bool on = false; int ii = 0, P = 0; } pend;
    DevBuf<float> H1, H2, O, dO;
    DevBuf<float> Opart, dZpart;       // [4][R*126] partial decoder outputs, [4][R*32] partial latent gradients (fdc_panel.h)
    bool dz_pending = false;           // the last backward left the latent gradient as partials: the next Adam launch (or
                                       // fdcap_opt_get_grads) folds them into dX
    DevBuf<float> Rm, PF, Jrest, G, A, M, Jw;
    DevBuf<float> Voff, Vw, dist, pd, dVoff;
    DevBuf<int> idx, pi;
    DevBuf<float> kp2d;       // per-frame inner fit: 2D keypoints [n_local,23,3] (u, v, confidence)
    DevBuf<float> floss;      // ... and the per-frame objective of its L-BFGS variant
    fdcap_lbfgs* lbfgs = nullptr;
    DevBuf<float4> seedpt;    // coordinates (+ position in the sorted scene) of each query's current neighbour: next launch's seed
    // work-list cache of the in-loop NN launch (fdc_chamfer.h NNCache): ids [groups * 4][64], hdr [groups * 4], anchors [4][nq]
    bool skin_vec = true;          // FDCAP_SKIN_VEC=0 (read by fdcap_opt_create; A/B): the scalar-load skinning backward
    DevBuf<float> loss_rows;       // [R][LROW] per-frame partial sums of the printed loss terms (logging iterations)
    bool log_pending = false;      // a logging backward (log_terms = 2) left the reduction of loss_rows to the next step launch
    unsigned log_mask = 0;
    int log_assign = 0;
    double* log_dst = nullptr;
    DevBuf<unsigned short> nnc_ids;
    DevBuf<int> nnc_hdr;
    DevBuf<float4> nnc_anchor;
    NNOrder nn_order;                      // launch order of the in-loop NN launch (fdc_chamfer.h NNOrder; its tables sit behind nnc_hdr); FDCAP_NN_ORDER=0 turns it off, =k re-sorts every k launches
    float nnc_slack = 0.03f;  // metres; FDCAP_NN_CACHE_SLACK overrides, 0 disables the cache
    // fdcap_opt_nn_timing: HIP events around every in-loop NN launch of a fit (the bench's roofline figure)
    bool nn_timing = false;
    std::vector<hipEvent_t> nn_ev;
    int nn_ev_used = 0;
    NNCache nn_cache(int) { return NNCache{nnc_slack > 0.f ? nnc_ids.p : nullptr, nnc_slack > 0.f ? nnc_hdr.p : nullptr, nnc_anchor.p, nnc_slack}; }
    DevBuf<float> dA, dtransl_v, dMv, dsv, dPF, dJw, dX, dCAM, dscale_row;     // d betas: columns 486.. of dPF
    DevBuf<float> VoffF, VwF, dVF;      // mode 'local' second loop: full-mesh pose offsets / world vertices / gradient
    int cam_steps = 0;
    // mode 'dct': basis [T,C], coefficients + Adam moments [W,69,C] (W = n_total / T windows of the whole clip)
    DevBuf<float> dctD, dctCoef, dctM, dctV;
    DevBuf<AdamScalars> adam_tab;
    std::vector<AdamScalars> adam_tab_h;
    int dctT = 0, dctC = 0, dctW = 0;
    bool dct_grad = false;    // the last backward gave `scale` a gradient through the DCT term
    bool seeded = false;      // a contact forward has run since fdcap_opt_create (idx holds neighbours)
    // fdcap_opt_forward_ahead ran for the owned rows and nothing has touched them since: the next backward only adds the halo
    // rows and the scale-dependent outputs (ahead_blend: the contact set's pose-blend product is done as well)
    bool ahead = false, ahead_blend = false;
    bool nnpt_valid = false;  // the last contact forward left the neighbours' coordinates in seedpt
    bool use_seed = true;     // last iteration's neighbours seed the NN bound (pruning only)
    bool use_cull = true;     // skip k-d cells whose box is out of every query's reach
};

}  // namespace

struct fdcap_lbfgs {          // batched L-BFGS (csrc/fdc_lbfgs.h): n independent problems
    int n = 0;
    LbfgsCfg cf{};
    DevBuf<LbfgsScalars> S;
    DevBuf<float> W, RO;      // per problem: vector workspace, 1 / (y . s) of the history pairs
    DevBuf<int> active;
    int* active_h = nullptr;  // pinned
    int round = 0;            // rounds since the last reset: active[round & 1] is the counter of the current one
};

struct fdcap_ctx {
    int V = 0;
    // host copies needed to build vertex subsets
    std::vector<float> h_vt, h_S10, h_posedirs, h_lbs;
    // device constants
    DevBuf<float> Jt, Jd, hand_comp, hand_mean;
    DevBuf<int> parents, order, level_start, child_start, child_list, depth;
    DevBuf<float> pose_tab;        // all of the above as ONE image in PoseStage's layout (what the staged pose kernels copy)
    int nlevels = 0;
    DevBuf<float> W1, b1, W2, b2, W3, b3;
    DevBuf<float> vp_pn[6];            // decoder weights in MFMA fragment order: forward w1 w2 w3, backward w3t w2t w1t
    VPoserPanels vp;
    DevBuf<unsigned> vp_pn3[6];        // ... and as three bf16 planes each (the default form of the products)
    VPoserPanels3 vp3;
    SkinSet full, contact;
    bool full_ready = false;
    DevBuf<float4> scene;          // original order {x,y,z,bits(i)}: gradient gather by index
    DevBuf<float4> scene_sorted;   // k-d cell order {x,y,z,bits(original index)}: what the NN scan streams
    DevBuf<float4> scene_bounds;   // axis-aligned box {lo},{hi} of each MF_CH-point chunk of scene_sorted
    DevBuf<float4> scene_sbounds;  // ... of each run of ST4_SUPER chunks
    DevBuf<float4> scene_qbounds;  // ... of each quarter chunk (128 points = four MFMA tiles, one k-d node): [chunk][4]{lo},{hi}
    DevBuf<int> scene_inv;         // original index -> position in scene_sorted
    DevBuf<uint4> scene_frags;     // precomputed chunk-centred bf16 MFMA A fragments of scene_sorted
    DevBuf<float4> scene_centers;  // chunk centres {x,y,z,radius}
    int64_t ns = 0;
    NNTarget nn_target(bool cull) const {
        NNTarget t; t.pts = scene_sorted.p; t.n = (int)ns; t.bounds = cull ? scene_bounds.p : nullptr; t.inv_perm = scene_inv.p;
        t.sbounds = cull ? scene_sbounds.p : nullptr; t.qbounds = cull ? scene_qbounds.p : nullptr;
        t.frags = cull ? scene_frags.p : nullptr; t.centers = scene_centers.p;
        return t;
    }
    int nc = 0;
    DevBuf<int> contact_vid;       // mesh vertex of each contact id (caller's order)
    DevBuf<int> contact_perm;      // internal contact slot -> position in the caller's id array
    // growable workspaces for the stand-alone operators
    DevBuf<AdamScalars> ws_adam;
    std::vector<AdamScalars> ws_adam_h;
    DevBuf<float> ws_f[12];
    DevBuf<float> ws_b[12];         // ... of their backward passes (fdcap_vposer_decode_bwd, fdcap_smplx_backward)
    DevBuf<float> ws_part;          // partial decoder outputs of the stand-alone operators
    DevBuf<int> ws_i[2];
    DevBuf<float4> ws_p;
    // Op 1 against the registered scene (fdcap_chamfer_fwd_scene): the previous call's neighbours = the next call's seeds
    struct SceneOp {
        DevBuf<float> dist;
        DevBuf<int> idx, hdr;
        DevBuf<float4> seedpt, anchor;
        DevBuf<unsigned short> ids;
        int nq = 0;                 // queries of the call that left the state (0: none)
        void release() { dist.release(); idx.release(); hdr.release(); seedpt.release(); anchor.release(); ids.release(); nq = 0; }
    } sop;
    OptState* opt = nullptr;
    Comm comm;                      // fdcap_comm_create: RCCL communicator of the sharded optimiser
    DevBuf<float> xch_send, xch_all;
    std::string comm_err;

    PoseModel pose_model() const {
        PoseModel pm;
        pm.tab = pose_tab.p;
        pm.Jt = Jt.p; pm.Jd = Jd.p; pm.parents = parents.p; pm.order = order.p; pm.level_start = level_start.p;
        pm.child_start = child_start.p; pm.child_list = child_list.p; pm.hand_comp = hand_comp.p;
        pm.hand_mean = hand_mean.p; pm.nlevels = nlevels; pm.depth = depth.p;
        return pm;
    }
};

namespace {

// largest K = 3 nv for which the blend products run on the fragment-ordered panels (both copies: 2 x 496 x K floats)
constexpr int PANEL_MAX_K = 6144;

int build_skin_set(fdcap_ctx* c, const std::vector<int64_t>& ids, SkinSet* out) {
    const int V = c->V;
    const int nv = (int)ids.size();
    int K = 1;
    for (int v : ids) {
        int k = 0;
        for (int j = 0; j < NJ; ++j) k += c->h_lbs[(size_t)v * NJ + j] != 0.f;
        K = std::max(K, k);
    }
    const int ldp = (3 * nv + 3) & ~3;
    std::vector<float> vt((size_t)nv * 3), ww((size_t)nv * K, 0.f), pd((size_t)NPFX * ldp, 0.f);
    std::vector<int> wj((size_t)nv * K, 0);
    for (int i = 0; i < nv; ++i) {
        int64_t v = ids[i];
        for (int k = 0; k < 3; ++k) vt[3 * i + k] = c->h_vt[3 * v + k];
        int k = 0;
        for (int j = 0; j < NJ; ++j) {
            float w = c->h_lbs[(size_t)v * NJ + j];
            if (w != 0.f) { wj[(size_t)i * K + k] = j; ww[(size_t)i * K + k] = w; ++k; }
        }
    }
    for (int r = 0; r < NPF; ++r)
        for (int i = 0; i < nv; ++i)
            for (int k = 0; k < 3; ++k)
                pd[(size_t)r * ldp + 3 * i + k] = c->h_posedirs[(size_t)r * 3 * V + 3 * ids[i] + k];
    // rows 486..495: shapedirs^T (betas part), so [pose feature | betas] x this matrix = pose offsets + shape offsets
    for (int l = 0; l < NBETA; ++l)
        for (int i = 0; i < nv; ++i)
            for (int k = 0; k < 3; ++k)
                pd[(size_t)(NPF + l) * ldp + 3 * i + k] = c->h_S10[((size_t)3 * ids[i] + k) * 10 + l];
    std::vector<int> csc_start(NJ + 1, 0), csc_v;
    std::vector<float> csc_w;
    for (int j = 0; j < NJ; ++j) {
        csc_start[j] = (int)csc_v.size();
        for (int i = 0; i < nv; ++i) {
            float w = c->h_lbs[(size_t)ids[i] * NJ + j];
            if (w != 0.f) { csc_v.push_back(i); csc_w.push_back(w); }
        }
    }
    csc_start[NJ] = (int)csc_v.size();
    out->nnz = (int)csc_v.size();
    if (csc_v.empty()) { csc_v.push_back(0); csc_w.push_back(0.f); }
    while (csc_w.size() & 3) csc_w.push_back(0.f);           // 16-byte staging reads whole float4s
    out->nv = nv; out->K = K; out->ldp = ldp;
    out->vpack.release(); out->csc_v16.release();
    if (K <= 12 && nv > 0 && nv <= 65535) {
        // planes of float4 per vertex (SkinModel::vpack): {template xyz, ids 0-3 as bytes}, {w0..w3}; for K > 4 also {w4..w7}
        // (, {w8..w11}) and last {bits(ids 4-7), bits(ids 8-11), 0, 0} -- skin_vpack_planes(K) planes in all
        const int G = (K + 3) / 4, NP = skin_vpack_planes(K);
        std::vector<float4> vp((size_t)nv * NP);
        for (int i = 0; i < nv; ++i) {
            unsigned jb[3] = {0, 0, 0};
            float w12[12] = {0.f};
            for (int k = 0; k < K; ++k) { jb[k >> 2] |= (unsigned)wj[(size_t)i * K + k] << (8 * (k & 3)); w12[k] = ww[(size_t)i * K + k]; }
            float jf[3]; memcpy(jf, jb, 12);
            vp[(size_t)i] = make_float4(vt[3 * i], vt[3 * i + 1], vt[3 * i + 2], jf[0]);
            for (int g = 0; g < G; ++g) vp[(size_t)(1 + g) * nv + i] = make_float4(w12[4 * g], w12[4 * g + 1], w12[4 * g + 2], w12[4 * g + 3]);
            if (G > 1) vp[(size_t)(1 + G) * nv + i] = make_float4(jf[1], jf[2], 0.f, 0.f);
        }
        std::vector<unsigned short> v16((csc_v.size() + 7) & ~(size_t)7, 0);
        for (size_t i = 0; i < csc_v.size(); ++i) v16[i] = (unsigned short)csc_v[i];
        HIP_TRY(out->vpack.upload(vp.data(), vp.size()));
        HIP_TRY(out->csc_v16.upload(v16.data(), v16.size()));
    }
    HIP_TRY(out->csc_start.upload(csc_start.data(), csc_start.size()));
    HIP_TRY(out->csc_v.upload(csc_v.data(), csc_v.size()));
    HIP_TRY(out->csc_w.upload(csc_w.data(), csc_w.size()));
    HIP_TRY(out->vt.upload(vt.data(), vt.size()));
    HIP_TRY(out->ww.upload(ww.data(), ww.size()));
    HIP_TRY(out->wj.upload(wj.data(), wj.size()));
    HIP_TRY(out->posedirs.upload(pd.data(), pd.size()));
    out->pn_fwd = PanelB(); out->pn_bwd = PanelB();
    if (nv > 0) {                                  // forward panel for every set (the full mesh takes the wide form of the kernel)
        std::vector<float> pf;
        int nt = 0, ns = 0;
        panel_pack(pd.data(), ldp, 1, NPFX, 3 * nv, pf, &nt, &ns);
        HIP_TRY(out->pn_fwd_f.upload(pf.data(), pf.size()));
        out->pn_fwd.f = (const float4*)out->pn_fwd_f.p; out->pn_fwd.ntile = nt; out->pn_fwd.nss = ns;
    }
    if (nv > 0 && 3 * nv <= PANEL_MAX_K) {         // data-gradient panel while a 16-row block of K = 3 nv columns fits the LDS slabs
        std::vector<float> pf;
        int nt = 0, ns = 0;
        panel_pack(pd.data(), 1, ldp, 3 * nv, NPFX, pf, &nt, &ns);
        HIP_TRY(out->pn_bwd_f.upload(pf.data(), pf.size()));
        out->pn_bwd.f = (const float4*)out->pn_bwd_f.p; out->pn_bwd.ntile = nt; out->pn_bwd.nss = ns;
    }
    out->pn_fwd3 = PanelB3(); out->pn_bwd3 = PanelB3();
    if (nv > 0) {                                 // the forward operand of every set also as three bf16 planes
        std::vector<unsigned> p3;
        panel_pack3(pd.data(), ldp, 1, NPFX, 3 * nv, p3, &out->pn_fwd3.ntile, &out->pn_fwd3.nst);
        HIP_TRY(out->pn_fwd3_f.upload(p3.data(), p3.size()));
        out->pn_fwd3.f = (const uint4*)out->pn_fwd3_f.p;
    }
    if (nv > 0 && panel_gemm3_fits(3 * nv)) {     // ... and the data-gradient operand of small sets (K = 3 nv in one LDS image <= 160 KB)
        std::vector<unsigned> p3;
        panel_pack3(pd.data(), 1, ldp, 3 * nv, NPFX, p3, &out->pn_bwd3.ntile, &out->pn_bwd3.nst);
        HIP_TRY(out->pn_bwd3_f.upload(p3.data(), p3.size()));
        out->pn_bwd3.f = (const uint4*)out->pn_bwd3_f.p;
    }
    return 0;
}

// dense products on the three-way bf16 split (FDCAP_GEMM_SPLIT3=0: exact-fp32 MFMA chains instead)
inline bool gemm_split3_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FDCAP_GEMM_SPLIT3"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}
// pose + shape blend offsets of a vertex set: Voff[M, 3 nv] = PF[M, 496] x [posedirs ; shapedirs^T]
hipError_t blend_forward(const SkinSet& ss, const float* PF, int M, float* Voff, hipStream_t st) {
    TraceRange tr_("fdcap:blend_fwd(K8)");
    if (gemm_split3_enabled() && ss.pn_fwd3.f) return panel_gemm3(PF, NPFX, M, NPFX, ss.pn_fwd3, Voff, 3 * ss.nv, 3 * ss.nv, st);
    if (ss.pn_fwd.f) return panel_gemm(PF, NPFX, M, NPFX, ss.pn_fwd, Voff, 3 * ss.nv, 3 * ss.nv, st);
    return gemm_f32(false, EPI_STORE, PF, NPFX, ss.posedirs.p, ss.ldp, Voff, 3 * ss.nv, M, 3 * ss.nv, NPFX, nullptr, 0, st);
}

// VPoser decoder forward for rows [row_lo, row_hi) of X (latent read in place at column latent_off): H1, H2 and the four
// partial outputs Opart (fdc_panel.h); O != nullptr: also the summed output (one more small launch -- the optimiser's
// pose_fwd_kernel<true> adds the partials itself instead)
// (row2_lo < row2_hi: a second row range in the same launch -- the halo rows on the far side of a shard's owned rows)
int vposer_forward(fdcap_ctx* c, const float* X, int ldx, int latent_off, int row_lo, int row_hi, float* H1, float* H2,
                   float* Opart, size_t part_stride, float* O, hipStream_t st, int row2_lo = 0, int row2_hi = 0,
                   const DeferredStep& ds = DeferredStep()) {
    const int rows = row_hi - row_lo, rows2 = std::max(row2_hi - row2_lo, 0);
    if (rows <= 0 && rows2 <= 0) return 0;
    if (rows <= 0) { row_lo = row2_lo; row_hi = row2_hi; return vposer_forward(c, X, ldx, latent_off, row_lo, row_hi, H1, H2, Opart, part_stride, O, st, 0, 0, ds); }
    VpRows two;
    const int nb1 = (rows + 15) / 16, nb2 = (rows2 + 15) / 16;
    if (rows2 > 0) { two.nb1 = nb1; two.row2_lo = row2_lo; two.row2_hi = row2_hi; }
    if (O && rows2 > 0) return FDCAP_E_ARG;                   // (the summed output is only formed for one range)
    if (gemm_split3_enabled())
        hipLaunchKernelGGL(vposer_fwd_split3_kernel, dim3(4 * (nb1 + nb2)), dim3(512), 0, st, c->vp3, X + latent_off, ldx, row_lo,
                           row_hi, H1, H2, Opart, part_stride, two, ds);
    else
        hipLaunchKernelGGL(vposer_fwd_fused_kernel, dim3(4 * (nb1 + nb2)), dim3(512), 0, st, c->vp, X + latent_off, ldx, row_lo,
                           row_hi, H1, H2, Opart, part_stride, two, ds);
    if (O) {
        const size_t n = (size_t)rows * ODIM;
        hipLaunchKernelGGL(vposer_sum_parts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Opart, part_stride,
                           (size_t)row_lo * ODIM, n, O);
    }
    return (int)hipGetLastError();
}

// rows of the optimiser's buffers whose pose is needed: the owned frames plus `halo` frames on each side that has a neighbour
void opt_row_range(const OptState* o, int halo, int* lo, int* hi) {
    const fdcap_opt_config& cf = o->cfg;
    *lo = cf.frame0 > 0 ? 2 - halo : 2;
    *hi = cf.n_local + 2 + (cf.frame0 + cf.n_local < cf.n_total ? halo : 0);
}

// The optimiser step of iteration ii as tensors for the Adam kernels (global_optimization.py:563-568, :577-580, :592 restated as
// arithmetic): which parameters are stepped, with which bias corrections.
struct StepPlan { AdamTensor x = {}, cam = {}, sc = {}; int nb_x = 0, nb_cam = 0; bool step_scale = false; };
StepPlan opt_step_plan(const OptState* o, int ii, int P, bool do_rows, bool do_scale) {
    const fdcap_opt_config& cf = o->cfg;
    const int nl = cf.n_local;
    StepPlan sp;
    // body_rotation_rec: every iteration, its own step counter = ii + 1
    if (do_rows) {
        sp.x = AdamTensor{o->X.p + 2 * XDIM, o->mX.p + 2 * XDIM, o->vX.p + 2 * XDIM, o->dX.p + 2 * XDIM, (size_t)nl * XDIM, adam_scalars(cf.lr, ii + 1)};
        sp.nb_x = (int)((sp.x.n + 255) / 256);
    }
    // camera_ext: first gradient at ii = P + 1 (flag flips after the forward of ii = P); mode 'local': the late-phase
    // loss has no camera_ext path -> grad None, never stepped
    if (do_rows && ii >= P + 1 && cf.phase2_world != 0.f) {
        sp.cam = AdamTensor{o->CAM.p + 2 * 16, o->mCAM.p + 2 * 16, o->vCAM.p + 2 * 16, o->dCAM.p + 2 * 16, (size_t)nl * 16, adam_scalars(cf.lr, ii - P)};
        sp.nb_cam = (int)((sp.cam.n + 255) / 256);
    }
    // scale: receives a gradient while ii < P (and only if a term that reaches it exists)
    sp.step_scale = do_scale && (o->contact_on || o->dct_grad) && (ii < P || cf.legacy_zero_grad);
    if (sp.step_scale) sp.sc = AdamTensor{o->scale.p, o->mS.p, o->vS.p, o->dscale.p, 1, adam_scalars(cf.lr, ii + 1)};
    return sp;
}
int opt_step_launch(fdcap_ctx* c, int32_t ii, int32_t P, bool do_rows, bool do_scale, bool reduce_scale, hipStream_t st, float* xch);

// Everything the caller registered (rows_x_d, rows_cam_d) and the Adam moments are current after this: the rows' part of a
// deferred step that no forward has consumed is applied by the ordinary Adam launch (`scale` was stepped with the backward).
int opt_sync(fdcap_ctx* c, hipStream_t st) {
    OptState* o = c->opt;
    if (!o || !o->pend.on) return 0;
    o->pend.on = false;
    return opt_step_launch(c, o->pend.ii, o->pend.P, true, false, false, st, nullptr);
}

// decoder + per-frame pose state (Rm, PF, Jrest, G, A, M, Jw) of rows [lo, hi): two launches.  A deferred optimiser step is
// applied by these two launches when they cover exactly the frames it steps (no halo rows: one rank), else by its own launch first.
// contact_state = false: the pose feature PF and the skinning transforms A -- read by the contact forward only -- are not written
// (phase 2 of a fit that does not log: 4.7 MB less for the end of the launch to write back, tools/launch_overhead_probe.hip)
int opt_pose_forward(fdcap_ctx* c, int lo, int hi, hipStream_t st, bool contact_state = true) {
    OptState* o = c->opt;
    o->ahead = false;                                       // (whatever ran ahead is recomputed here)
    const size_t ps = (size_t)o->R * ODIM;
    const int nl = o->cfg.n_local;
    DeferredStep ds;
    if (o->pend.on) {
        if (lo == 2 && hi == 2 + nl && o->cfg.frame0 == 0 && nl == o->cfg.n_total && !o->log_pending) {
            const StepPlan sp = opt_step_plan(o, o->pend.ii, o->pend.P, true, false);
            ds.on = 1; ds.x = sp.x; ds.cam = sp.cam; ds.row0 = 2;
            ds.dzpart = o->dz_pending ? o->dZpart.p : nullptr; ds.dz_stride = (size_t)o->R * VP_Z;
        } else {
            int e = opt_sync(c, st);
            if (e) return e;
        }
    }
    int e = vposer_forward(c, o->X.p, XDIM, X_LATENT, lo, hi, o->H1.p, o->H2.p, o->Opart.p, ps, nullptr, st, 0, 0, ds);
    if (e) return e;
    hipLaunchKernelGGL(pose_fwd_kernel<true>, dim3(hi - lo), dim3(64 * POSE_NW), 0, st, c->pose_model(), o->X.p, o->O.p, o->CAM.p, o->scale.p, lo,
#ifdef FDC_DEBUG_BUFFERS
                       o->Rm.p,                                // (the per-joint rotations: nobody reads them back but fdcap_debug_rows)
#else
                       (float*)nullptr,
#endif
                       contact_state ? o->PF.p : (float*)nullptr, o->Jrest.p, o->G.p, contact_state ? o->A.p : (float*)nullptr, o->M.p, o->Jw.p,
                       (const float*)nullptr, (const float*)o->Opart.p, ps, 0, 0, ds);
    if (ds.on) {                                            // the step has been issued: the launches that follow see its results
        o->pend.on = false;
        o->dz_pending = false;
    }
    return (int)hipGetLastError();
}

// After fdcap_opt_forward_ahead: what is left of opt_pose_forward(lo, hi) -- the halo rows on either side of the owned rows in
// full (their parameters arrived with the exchange), and M / Jw of the owned rows (`scale` was stepped by the exchange's tail).
// Two launches, nearly empty.
int opt_pose_forward_rest(fdcap_ctx* c, int lo, int hi, hipStream_t st) {
    OptState* o = c->opt;
    const int nl = o->cfg.n_local;
    const size_t ps = (size_t)o->R * ODIM;
    int e = vposer_forward(c, o->X.p, XDIM, X_LATENT, lo, 2, o->H1.p, o->H2.p, o->Opart.p, ps, nullptr, st, 2 + nl, hi);
    if (e) return e;
    hipLaunchKernelGGL(pose_fwd_kernel<true>, dim3(hi - lo), dim3(64 * POSE_NW), 0, st, c->pose_model(), o->X.p, o->O.p, o->CAM.p, o->scale.p, lo,
                       (float*)nullptr /* Rm: see opt_pose_forward */, o->PF.p, o->Jrest.p, o->G.p, o->A.p, o->M.p, o->Jw.p, (const float*)nullptr, (const float*)o->Opart.p, ps,
                       2, 2 + nl);
    return (int)hipGetLastError();
}

// VPoser data-gradient dO -> d latent of the owned rows, left as four partials in dZpart (fold = true: added into dX here)
// (tail.block >= 0: one more workgroup that steps `scale` -- ScaleTail, fdc_loss.h)
int opt_vposer_backward(fdcap_ctx* c, bool fold, hipStream_t st, ScaleTail tail = ScaleTail()) {
    OptState* o = c->opt;
    const int nl = o->cfg.n_local;
    const size_t ps = (size_t)o->R * VP_Z;
    const int nb = 4 * ((nl + 15) / 16);
    if (tail.block >= 0) tail.block = 0;                   // (first in the grid: fdc_panel.h)
    if (gemm_split3_enabled())
        hipLaunchKernelGGL(vposer_bwd_split3_kernel, dim3(nb + (tail.block >= 0 ? 1 : 0)), dim3(512), 0, st, c->vp3, o->dO.p, 2, 2 + nl, o->H1.p, o->H2.p,
                           o->dZpart.p, ps, tail);
    else
        hipLaunchKernelGGL(vposer_bwd_fused_kernel, dim3(nb + (tail.block >= 0 ? 1 : 0)), dim3(512), 0, st, c->vp, o->dO.p, 2, 2 + nl, o->H1.p, o->H2.p,
                           o->dZpart.p, ps, tail);
    if (fold) {
        hipLaunchKernelGGL(vposer_fold_dz_kernel, dim3((nl * VP_Z + 255) / 256), dim3(256), 0, st, o->dZpart.p, ps, 2, nl, o->dX.p);
        o->dz_pending = false;
    } else {
        o->dz_pending = true;
    }
    return (int)hipGetLastError();
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char* fdcap_version(void) { return "fdcap-hip 0.3 (gfx950)"; }
const char* fdcap_build_info(void) {
#ifdef FDC_BUILD_NO_PK_F32
    return "packed_fp32=off";
#else
    return "packed_fp32=on";
#endif
}

#ifdef FDC_DEBUG_BUFFERS
int fdcap_debug_stage_bad(unsigned* out) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stage_bad), sizeof(unsigned) * 8 * 16));
    return 0;
}
// instrumentation build only (tools/pk_where.py): row r0.. of an internal per-frame buffer, `per_row` floats per row
// which: 0 O [126], 1 PF [NPFX], 2 A [55*12], 3 M [12], 4 Voff [3 nc], 5 Vw [3 nc], 6 G [55*12], 7 Opart q=0 [126], 8 H2 [512], 9 Jw [69], 10 Rm [55*9], 11 Jrest [55*3]
int fdcap_debug_rows(fdcap_ctx* c, int which, float* dst, void* stream) {
    if (!c || !c->opt) return FDCAP_E_STATE;
    OptState* o = c->opt;
    const int nl = o->cfg.n_local, nc = c->nc;
    const float* src = nullptr; size_t w = 0;
    switch (which) {
        case 0: src = o->O.p; w = O_LD; break;       // (padded rows)
        case 1: src = o->PF.p; w = NPFX; break;
        case 2: src = o->A.p; w = NJ * 12; break;
        case 3: src = o->M.p; w = 12; break;
        case 4: src = o->Voff.p; w = (size_t)3 * nc; break;
        case 5: src = o->Vw.p; w = (size_t)3 * nc; break;
        case 6: src = o->G.p; w = NJ * 12; break;
        case 7: src = o->Opart.p; w = ODIM; break;
        case 8: src = o->H2.p; w = VP_H; break;
        case 9: src = o->Jw.p; w = NJW * 3; break;
        case 10: src = o->Rm.p; w = RM_LD; break;
        case 11: src = o->Jrest.p; w = JR_LD; break;
        default: return FDCAP_E_ARG;
    }
    HIP_TRY(hipMemcpyAsync(dst, src + 2 * w, (size_t)nl * w * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return (int)w;
}
#endif
#ifdef FDC_NN_STATS
int fdcap_debug_nn_hist(unsigned long long* out) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nn_hist), 96 * sizeof(unsigned long long)));
    unsigned long long z[96] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_nn_hist), z, sizeof(z)));
    return 0;
}
int fdcap_debug_nn_stats(unsigned long long* out) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nn_stats), 8 * sizeof(unsigned long long)));
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_nn_stats), z, sizeof(z)));
    return 0;
}
#endif

#ifdef FDC_NN_TIMELINE
int fdcap_debug_nn_timeline(unsigned long long* out, int n) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nn_timeline), (size_t)n * sizeof(unsigned long long)));
    return 0;
}
#endif
#ifdef FDC_PN_TIMING
int fdcap_debug_frame_times(unsigned long long* out) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fr_times), sizeof(unsigned long long) * 3 * 2048 * 8));
    return 0;
}
int fdcap_debug_panel_reset(void) {
    HIP_TRY(hipDeviceSynchronize());
    static std::vector<unsigned long long> z(8192 * 8, 0ull);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_pn_times), z.data(), z.size() * sizeof(unsigned long long)));
    return 0;
}
int fdcap_debug_panel_times(unsigned long long* out, int n) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pn_times), (size_t)n * sizeof(unsigned long long)));
    return 0;
}
#endif

int fdcap_set_nn_kernel(int32_t mode) {
    if (mode < 0 || mode > 2) return FDCAP_E_ARG;
    nn_mode_ref() = mode;
    return FDCAP_OK;
}

int fdcap_ctx_create(const fdcap_model_desc* md, fdcap_ctx** out) {
    if (!md || !out || md->num_verts <= 0 || md->num_shape < NBETA) return FDCAP_E_ARG;
    if (!md->v_template || !md->shapedirs || !md->posedirs || !md->J_regressor || !md->parents ||
        !md->lbs_weights || !md->hands_componentsl || !md->hands_componentsr || !md->hands_meanl ||
        !md->hands_meanr || !md->vp_fc1_w || !md->vp_fc1_b || !md->vp_fc2_w || !md->vp_fc2_b || !md->vp_out_w ||
        !md->vp_out_b)
        return FDCAP_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return FDCAP_E_NODEVICE;
    fdcap_ctx* c = new fdcap_ctx();
    const int V = c->V = md->num_verts;
    c->h_vt.assign(md->v_template, md->v_template + (size_t)V * 3);
    c->h_S10.resize((size_t)V * 30);
    for (size_t i = 0; i < (size_t)V * 3; ++i)
        for (int l = 0; l < NBETA; ++l) c->h_S10[i * NBETA + l] = md->shapedirs[i * md->num_shape + l];
    c->h_posedirs.assign(md->posedirs, md->posedirs + (size_t)NPF * 3 * V);
    c->h_lbs.assign(md->lbs_weights, md->lbs_weights + (size_t)V * NJ);
    HostPoseSetup hs;
    if (!host_pose_setup(V, c->h_vt.data(), c->h_S10.data(), md->J_regressor, md->parents, &hs)) { delete c; return FDCAP_E_ARG; }
    std::vector<float>&Jt = hs.Jt, &Jd = hs.Jd;
    std::vector<int>&parents = hs.parents, &order = hs.order, &level_start = hs.level_start,
                    &child_start = hs.child_start, &child_list = hs.child_list;
    c->nlevels = hs.nlevels;
    std::vector<float> hc(2 * 12 * 45), hm(90);
    memcpy(hc.data(), md->hands_componentsl, 12 * 45 * sizeof(float));
    memcpy(hc.data() + 12 * 45, md->hands_componentsr, 12 * 45 * sizeof(float));
    memcpy(hm.data(), md->hands_meanl, 45 * sizeof(float));
    memcpy(hm.data() + 45, md->hands_meanr, 45 * sizeof(float));
    int err = 0;
#define UP(buf, ptr, cnt) if (!err) { hipError_t e_ = (buf).upload(ptr, cnt); if (e_ != hipSuccess) err = (int)e_; }
    UP(c->Jt, Jt.data(), Jt.size()) UP(c->Jd, Jd.data(), Jd.size())
    UP(c->parents, parents.data(), parents.size()) UP(c->order, order.data(), order.size())
    UP(c->level_start, level_start.data(), level_start.size())
    UP(c->child_start, child_start.data(), child_start.size()) UP(c->child_list, child_list.data(), child_list.size())
    UP(c->depth, hs.depth.data(), hs.depth.size())
    UP(c->hand_comp, hc.data(), hc.size()) UP(c->hand_mean, hm.data(), hm.size())
    {   // the same tables once more, laid out as PoseStage's static part
        std::vector<float> img((size_t)PS_STATIC_FLOATS, 0.f);
        auto put = [&](size_t off_bytes, const void* src, size_t n_words) { memcpy((char*)img.data() + off_bytes, src, n_words * 4); };
        put(offsetof(PoseStage, Jd), Jd.data(), Jd.size());
        put(offsetof(PoseStage, hand_comp), hc.data(), hc.size());
        put(offsetof(PoseStage, Jt), Jt.data(), Jt.size());
        put(offsetof(PoseStage, hand_mean), hm.data(), hm.size());
        put(offsetof(PoseStage, parents), parents.data(), parents.size());
        put(offsetof(PoseStage, order), order.data(), order.size());
        put(offsetof(PoseStage, level_start), level_start.data(), std::min<size_t>(level_start.size(), MAX_LEVELS + 4));
        put(offsetof(PoseStage, child_start), child_start.data(), child_start.size());
        put(offsetof(PoseStage, child_list), child_list.data(), child_list.size());
        put(offsetof(PoseStage, depth), hs.depth.data(), hs.depth.size());
        UP(c->pose_tab, img.data(), img.size())
    }
    UP(c->W1, md->vp_fc1_w, 512 * 32) UP(c->b1, md->vp_fc1_b, 512)
    UP(c->W2, md->vp_fc2_w, 512 * 512) UP(c->b2, md->vp_fc2_b, 512)
    UP(c->W3, md->vp_out_w, ODIM * 512) UP(c->b3, md->vp_out_b, ODIM)
#undef UP
    if (!err) {
        // decoder weights in MFMA fragment order.  torch Linear weights are [out, in]: forward y = x W^T -> B(k, n) = W[n][k];
        // data gradient dx = dy W -> B(k, n) = W[k][n]
        struct { const float* w; long sk, sn; int K, N; PanelB* dst; } pk[6] = {
            {md->vp_fc1_w, 1, VP_Z, VP_Z, VP_H, &c->vp.w1}, {md->vp_fc2_w, 1, VP_H, VP_H, VP_H, &c->vp.w2},
            {md->vp_out_w, 1, VP_H, VP_H, ODIM, &c->vp.w3}, {md->vp_out_w, VP_H, 1, ODIM, VP_H, &c->vp.w3t},
            {md->vp_fc2_w, VP_H, 1, VP_H, VP_H, &c->vp.w2t}, {md->vp_fc1_w, VP_Z, 1, VP_H, VP_Z, &c->vp.w1t}};
        std::vector<float> pf;
        for (int i = 0; i < 6 && !err; ++i) {
            int nt = 0, ns = 0;
            panel_pack(pk[i].w, pk[i].sk, pk[i].sn, pk[i].K, pk[i].N, pf, &nt, &ns);
            hipError_t e_ = c->vp_pn[i].upload(pf.data(), pf.size());
            if (e_ != hipSuccess) err = (int)e_;
            pk[i].dst->f = (const float4*)c->vp_pn[i].p; pk[i].dst->ntile = nt; pk[i].dst->nss = ns;
        }
        c->vp.b1 = c->b1.p; c->vp.b2 = c->b2.p; c->vp.b3 = c->b3.p;
        PanelB3* dst3[6] = {&c->vp3.w1, &c->vp3.w2, &c->vp3.w3, &c->vp3.w3t, &c->vp3.w2t, &c->vp3.w1t};
        std::vector<unsigned> p3;
        for (int i = 0; i < 6 && !err; ++i) {
            panel_pack3(pk[i].w, pk[i].sk, pk[i].sn, pk[i].K, pk[i].N, p3, &dst3[i]->ntile, &dst3[i]->nst);
            hipError_t e_ = c->vp_pn3[i].upload(p3.data(), p3.size());
            if (e_ != hipSuccess) err = (int)e_;
            dst3[i]->f = (const uint4*)c->vp_pn3[i].p;
        }
        c->vp3.b1 = c->b1.p; c->vp3.b2 = c->b2.p; c->vp3.b3 = c->b3.p;
    }
    if (err) { fdcap_ctx_destroy(c); return err; }
    *out = c;
    return FDCAP_OK;
}

void fdcap_ctx_destroy(fdcap_ctx* c) {
    if (!c) return;
    fdcap_opt_destroy(c);
    (void)fdcap_comm_destroy(c);
    c->xch_send.release(); c->xch_all.release();
    c->ws_adam.release();
    c->Jt.release(); c->Jd.release(); c->hand_comp.release(); c->hand_mean.release();
    c->parents.release(); c->order.release(); c->level_start.release(); c->child_start.release(); c->child_list.release(); c->depth.release();
    c->pose_tab.release();
    c->W1.release(); c->b1.release(); c->W2.release(); c->b2.release(); c->W3.release(); c->b3.release();
    for (auto& b : c->vp_pn) b.release();
    for (auto& b : c->vp_pn3) b.release();
    c->full.release(); c->contact.release(); c->contact_vid.release(); c->contact_perm.release(); c->scene.release(); c->scene_sorted.release(); c->scene_bounds.release(); c->scene_sbounds.release(); c->scene_qbounds.release(); c->scene_inv.release(); c->scene_frags.release(); c->scene_centers.release(); c->sop.release();
    for (auto& b : c->ws_f) b.release();
    for (auto& b : c->ws_b) b.release();
    c->ws_part.release();
    for (auto& b : c->ws_i) b.release();
    c->ws_p.release();
    delete c;
}

int fdcap_set_scene(fdcap_ctx* c, const float* xyz, int64_t ns) {
    // FDCAP_MAX_SCENE_POINTS: the in-loop NN launch streams the scene's MFMA fragments (32 B per point) through one buffer
    // resource with 32-bit byte offsets; beyond 2 GiB of fragments its loads would silently return zeros
    if (!c || ns < 0 || (ns > 0 && !xyz) || ns > FDCAP_MAX_SCENE_POINTS) return FDCAP_E_ARG;
    // a live optimiser holds buffers sized for, and pruning state (seeds, kept work lists) valid for, the registered scene
    if (c->opt) return FDCAP_E_STATE;
    c->sop.nq = 0;                                          // (the scene operator's seeds / kept lists were for the old scene)
    std::vector<float4> orig((size_t)ns), sorted((size_t)ns);
    for (int64_t i = 0; i < ns; ++i) {
        orig[i] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 0.f);
        int ii = (int)i;
        memcpy(&orig[i].w, &ii, 4);
    }
    // Spatial order by recursive median splits (k-d cells): every MF_CH-point chunk is one cell and every
    // 32-point MFMA tile inside it a sub-cell, so the chunk boxes the NN scan culls with are compact and
    // disjoint (runs of a Morton curve jump across quadrant borders and give long, overlapping boxes).
    // A node of n points is cut at a multiple of the unit below it (chunks above MF_CH, tiles below), along
    // its longest axis; ties by index, so the order is deterministic.  Results never depend on this order.
    std::vector<int> order((size_t)ns), inv((size_t)ns);
    for (int64_t i = 0; i < ns; ++i) order[i] = (int)i;
    {
        std::vector<std::pair<int64_t, int64_t>> stack;
        if (ns > 0) stack.push_back({0, ns});
        while (!stack.empty()) {
            const int64_t a = stack.back().first, b = stack.back().second;
            stack.pop_back();
            const int64_t n = b - a;
            if (n <= 32) continue;
            const int64_t unit = n > MF_CH ? MF_CH : 32;
            const int64_t units = (n + unit - 1) / unit;
            const int64_t nleft = std::min(n - 1, (units / 2) * unit);
            if (nleft <= 0) continue;
            float blo[3] = {1e30f, 1e30f, 1e30f}, bhi[3] = {-1e30f, -1e30f, -1e30f};
            for (int64_t p = a; p < b; ++p)
                for (int k = 0; k < 3; ++k) { float v = xyz[3 * (int64_t)order[p] + k]; blo[k] = std::min(blo[k], v); bhi[k] = std::max(bhi[k], v); }
            int ax = 0;
            for (int k = 1; k < 3; ++k) if (bhi[k] - blo[k] > bhi[ax] - blo[ax]) ax = k;
            std::nth_element(order.begin() + a, order.begin() + a + nleft, order.begin() + b, [&](int i, int j) {
                const float vi = xyz[3 * (int64_t)i + ax], vj = xyz[3 * (int64_t)j + ax];
                return vi < vj || (vi == vj && i < j);
            });
            stack.push_back({a, a + nleft});
            stack.push_back({a + nleft, b});
        }
    }
    for (int64_t p = 0; p < ns; ++p) { sorted[p] = orig[order[p]]; inv[order[p]] = (int)p; }
    const int64_t nchunk = (ns + MF_CH - 1) / MF_CH;
    std::vector<float4> bounds((size_t)nchunk * 2);           // axis-aligned box per chunk, slightly inflated
    for (int64_t ch = 0; ch < nchunk; ++ch) {
        int64_t a = ch * MF_CH, b = std::min<int64_t>(ns, a + MF_CH);
        float blo[3] = {1e30f, 1e30f, 1e30f}, bhi[3] = {-1e30f, -1e30f, -1e30f};
        for (int64_t p = a; p < b; ++p) {
            const float v[3] = {sorted[p].x, sorted[p].y, sorted[p].z};
            for (int k = 0; k < 3; ++k) { blo[k] = std::min(blo[k], v[k]); bhi[k] = std::max(bhi[k], v[k]); }
        }
        for (int k = 0; k < 3; ++k) {
            float pad = 1e-6f + 1e-6f * std::max(fabsf(blo[k]), fabsf(bhi[k]));
            blo[k] -= pad; bhi[k] += pad;
        }
        bounds[2 * ch] = make_float4(blo[0], blo[1], blo[2], 0.f);
        bounds[2 * ch + 1] = make_float4(bhi[0], bhi[1], bhi[2], 0.f);
    }
    // boxes of the quarter chunks (the k-d recursion goes on below the chunk, so 128 consecutive points are one node);
    // a quarter past the end of the scene gets the empty box (+inf, +inf): at infinite distance from every query (box_d2)
    std::vector<float4> qbounds((size_t)nchunk * 8);
    for (int64_t qc = 0; qc < nchunk * 4; ++qc) {
        int64_t a = qc * (MF_CH / 4), b = std::min<int64_t>(ns, a + MF_CH / 4);
        float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int64_t p = a; p < b; ++p) {
            const float v[3] = {sorted[p].x, sorted[p].y, sorted[p].z};
            for (int k = 0; k < 3; ++k) { blo[k] = std::min(blo[k], v[k]); bhi[k] = std::max(bhi[k], v[k]); }
        }
        if (a < b)
            for (int k = 0; k < 3; ++k) {
                float pad = 1e-6f + 1e-6f * std::max(fabsf(blo[k]), fabsf(bhi[k]));
                blo[k] -= pad; bhi[k] += pad;
            }
        else
            for (int k = 0; k < 3; ++k) bhi[k] = INFINITY;
        qbounds[2 * qc] = make_float4(blo[0], blo[1], blo[2], 0.f);
        qbounds[2 * qc + 1] = make_float4(bhi[0], bhi[1], bhi[2], 0.f);
    }
    // chunk-centred bf16 hi/lo fragments in MFMA A layout (see nn_stream4_kernel)
    auto bf = [](float f) -> uint32_t { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return u >> 16; };   // RNE
    auto bff = [](uint32_t h) -> float { uint32_t u = h << 16; float f; memcpy(&f, &u, 4); return f; };
    std::vector<uint4> frags((size_t)nchunk * (MF_CH / 32) * 64);
    std::vector<float4> centers((size_t)nchunk);
    for (int64_t ch = 0; ch < nchunk; ++ch) {
        const float4 lo = bounds[2 * ch], hi = bounds[2 * ch + 1];
        const float cx = 0.5f * (lo.x + hi.x), cy = 0.5f * (lo.y + hi.y), cz = 0.5f * (lo.z + hi.z);
        float r2 = 0.f;
        for (int j = 0; j < MF_CH; ++j) {
            const int64_t p = ch * MF_CH + j;
            float yx = 0.f, yy = 0.f, yz = 0.f, n2 = 1e30f;      // padding rows: score 1e30
            if (p < ns) {
                yx = sorted[p].x - cx; yy = sorted[p].y - cy; yz = sorted[p].z - cz;
                n2 = yz * yz + (yy * yy + yx * yx);
                r2 = std::max(r2, n2);
            }
            uint32_t hx = bf(yx), hy = bf(yy), hz = bf(yz);
            uint32_t lx = bf(yx - bff(hx)), ly = bf(yy - bff(hy)), lz = bf(yz - bff(hz));
            // the score's factor -2 (|y|^2 - 2 x.y) rides on the static side: exact in bf16, and the per-chunk query
            // fragment is the plain hi | lo split
            hx = bf(-2.f * bff(hx)); hy = bf(-2.f * bff(hy)); hz = bf(-2.f * bff(hz));
            lx = bf(-2.f * bff(lx)); ly = bf(-2.f * bff(ly)); lz = bf(-2.f * bff(lz));
            const uint32_t nh = bf(n2);
            const float r1 = n2 - bff(nh);
            const uint32_t nm = bf(r1), nl = bf(r1 - bff(nm));
            const int tile = j >> 5, pt = j & 31;
            uint4* t = frags.data() + ((size_t)ch * (MF_CH / 32) + tile) * 64;
            t[pt] = make_uint4(hx | (hx << 16), lx | (lx << 16), hy | (hy << 16), ly | (ly << 16));
            t[32 + pt] = make_uint4(hz | (hz << 16), lz | (lz << 16), nh | (nm << 16), nl);
        }
        centers[ch] = make_float4(cx, cy, cz, sqrtf(r2) * 1.00001f + 1e-6f);
    }
    HIP_TRY(c->scene_frags.upload(frags.data(), frags.size()));
    HIP_TRY(c->scene_centers.upload(centers.data(), centers.size()));
    HIP_TRY(c->scene.upload(orig.data(), orig.size()));
    HIP_TRY(c->scene_sorted.upload(sorted.data(), sorted.size()));
    HIP_TRY(c->scene_bounds.upload(bounds.data(), bounds.size()));
    {
        const int64_t nsuper = (nchunk + ST4_SUPER - 1) / ST4_SUPER;
        std::vector<float4> sb((size_t)std::max<int64_t>(nsuper, 1) * 2, make_float4(0.f, 0.f, 0.f, 0.f));
        for (int64_t su = 0; su < nsuper; ++su) {
            float4 lo = make_float4(1e30f, 1e30f, 1e30f, 0.f), hi = make_float4(-1e30f, -1e30f, -1e30f, 0.f);
            for (int64_t ch = su * ST4_SUPER; ch < std::min(nchunk, (su + 1) * ST4_SUPER); ++ch) {
                lo.x = std::min(lo.x, bounds[2 * ch].x); lo.y = std::min(lo.y, bounds[2 * ch].y); lo.z = std::min(lo.z, bounds[2 * ch].z);
                hi.x = std::max(hi.x, bounds[2 * ch + 1].x); hi.y = std::max(hi.y, bounds[2 * ch + 1].y); hi.z = std::max(hi.z, bounds[2 * ch + 1].z);
            }
            sb[2 * su] = lo; sb[2 * su + 1] = hi;
        }
        HIP_TRY(c->scene_sbounds.upload(sb.data(), sb.size()));
        HIP_TRY(c->scene_qbounds.upload(qbounds.data(), qbounds.size()));
    }
    HIP_TRY(c->scene_inv.upload(inv.data(), inv.size()));
    c->ns = ns;
    return FDCAP_OK;
}

int fdcap_set_contact_ids(fdcap_ctx* c, const int64_t* vid, int32_t nc) {
    if (!c || nc < 0 || (nc > 0 && !vid)) return FDCAP_E_ARG;
    if (c->opt) return FDCAP_E_STATE;                  // (see fdcap_set_scene)
    for (int i = 0; i < nc; ++i) if (vid[i] < 0 || vid[i] >= c->V) return FDCAP_E_ARG;
    // Internal slot order = Morton order of the template positions: the 256 consecutive queries of an NN
    // workgroup are then spatially compact, so far fewer scene chunks survive its bound test (with all
    // 10 475 vertices as contacts a workgroup would otherwise span the whole body).  The loss is a mean
    // over the contact set, so the order is free; outputs go back in the caller's order via contact_perm.
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (int i = 0; i < nc; ++i)
        for (int k = 0; k < 3; ++k) {
            float v = c->h_vt[3 * vid[i] + k];
            lo[k] = std::min(lo[k], v); hi[k] = std::max(hi[k], v);
        }
    auto spread = [](uint32_t v) { v &= 1023; v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F;
                                   v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v; };
    std::vector<std::pair<uint32_t, int>> key((size_t)nc);
    for (int i = 0; i < nc; ++i) {
        uint32_t code = 0;
        for (int k = 0; k < 3; ++k) {
            float ext = hi[k] - lo[k];
            float u = ext > 0.f ? (c->h_vt[3 * vid[i] + k] - lo[k]) / ext : 0.f;
            code |= spread((uint32_t)std::min(1023.f, std::max(0.f, u * 1023.f))) << k;
        }
        key[i] = {code, i};
    }
    std::sort(key.begin(), key.end());
    std::vector<int64_t> ids((size_t)nc);
    std::vector<int> perm((size_t)std::max(nc, 1), 0), v32((size_t)std::max(nc, 1), 0);
    for (int sl = 0; sl < nc; ++sl) { ids[sl] = vid[key[sl].second]; perm[sl] = key[sl].second; }
    for (int i = 0; i < nc; ++i) v32[i] = (int)vid[i];
    int e = build_skin_set(c, ids, &c->contact);
    if (e) return e;
    HIP_TRY(c->contact_vid.upload(v32.data(), v32.size()));
    HIP_TRY(c->contact_perm.upload(perm.data(), perm.size()));
    c->nc = nc;
    return FDCAP_OK;
}

// ---- Op 1 ----------------------------------------------------------------------------------
int fdcap_chamfer_fwd(fdcap_ctx* c, const float* xyz1, const float* xyz2, int32_t B, int32_t n, int32_t m,
                      int64_t stride2, float* dist1, int32_t* idx1, float* dist2, int32_t* idx2, void* stream) {
    if (!c || !xyz1 || !xyz2 || B <= 0 || n <= 0 || m <= 0) return FDCAP_E_ARG;
    if ((dist1 && !idx1) || (dist2 && !idx2) || (!dist1 && !dist2)) return FDCAP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool shared = (stride2 == 0);
    size_t big = (size_t)std::max(n, m);
    HIP_TRY(c->ws_p.ensure(shared ? (size_t)m + (dist2 ? big : 0) : big));
    if (shared && dist1) {
        hipLaunchKernelGGL(pack_points_kernel, dim3((m + 255) / 256), dim3(256), 0, st, xyz2, m, c->ws_p.p);
        int nq = B * n;
        int nsplit = nn_pick_nsplit(nq, m);
        HIP_TRY(c->ws_f[0].ensure((size_t)nsplit * nq));
        HIP_TRY(c->ws_i[0].ensure((size_t)nsplit * nq));
        { NNTarget T{c->ws_p.p, m, nullptr, nullptr, nullptr, nullptr, nullptr}; HIP_TRY(nn_search(xyz1, nq, T, dist1, idx1, c->ws_f[0].p, c->ws_i[0].p, nsplit, st)); }
    }
    for (int b = 0; b < B && (!shared || dist2); ++b) {
        const float* x1 = xyz1 + (size_t)b * n * 3;
        const float* x2 = xyz2 + (size_t)b * stride2;
        float4* pk = c->ws_p.p + (shared ? m : 0);
        if (!shared && dist1) {
            hipLaunchKernelGGL(pack_points_kernel, dim3((m + 255) / 256), dim3(256), 0, st, x2, m, pk);
            int nsplit = nn_pick_nsplit(n, m);
            HIP_TRY(c->ws_f[0].ensure((size_t)nsplit * n));
            HIP_TRY(c->ws_i[0].ensure((size_t)nsplit * n));
            { NNTarget T{pk, m, nullptr, nullptr, nullptr, nullptr, nullptr}; HIP_TRY(nn_search(x1, n, T, dist1 + (size_t)b * n, idx1 + (size_t)b * n, c->ws_f[0].p, c->ws_i[0].p, nsplit, st)); }
        }
        if (dist2) {
            hipLaunchKernelGGL(pack_points_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x1, n, pk);
            int nsplit = nn_pick_nsplit(m, n);
            HIP_TRY(c->ws_f[1].ensure((size_t)nsplit * m));
            HIP_TRY(c->ws_i[1].ensure((size_t)nsplit * m));
            { NNTarget T{pk, n, nullptr, nullptr, nullptr, nullptr, nullptr}; HIP_TRY(nn_search(x2, m, T, dist2 + (size_t)b * m, idx2 + (size_t)b * m, c->ws_f[1].p, c->ws_i[1].p, nsplit, st)); }
        }
    }
    return (int)hipGetLastError();
}

int fdcap_chamfer_bwd(fdcap_ctx* c, const float* xyz1, const float* xyz2, int32_t B, int32_t n, int32_t m,
                      int64_t stride2, const float* gdist1, const int32_t* idx1, float* gxyz1, void* stream) {
    if (!c || !xyz1 || !xyz2 || !gdist1 || !idx1 || !gxyz1 || B <= 0 || n <= 0 || m <= 0) return FDCAP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(c->ws_p.ensure(m));
    if (stride2 == 0) {
        hipLaunchKernelGGL(pack_points_kernel, dim3((m + 255) / 256), dim3(256), 0, st, xyz2, m, c->ws_p.p);
        int nq = B * n;
        hipLaunchKernelGGL(nn_grad_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, xyz1, c->ws_p.p, gdist1, idx1, nq, gxyz1);
    } else {
        for (int b = 0; b < B; ++b) {
            hipLaunchKernelGGL(pack_points_kernel, dim3((m + 255) / 256), dim3(256), 0, st, xyz2 + (size_t)b * stride2, m, c->ws_p.p);
            hipLaunchKernelGGL(nn_grad_kernel, dim3((n + 255) / 256), dim3(256), 0, st, xyz1 + (size_t)b * n * 3, c->ws_p.p,
                               gdist1 + (size_t)b * n, idx1 + (size_t)b * n, n, gxyz1 + (size_t)b * n * 3);
        }
    }
    return (int)hipGetLastError();
}


// Op 1 against the REGISTERED scene.  The operator API's call site (:292-294) passes the same scene in every iteration of the
// caller's loop; fdcap_chamfer_fwd has to treat it as a foreign point set (unsorted: every pair visited, nn_mfma_kernel).  A caller
// that says "xyz2 is the scene I registered" gets the optimiser loop's search: the k-d-sorted scene with its cell boxes and
// precomputed fragments, seeds from nn_seed_kernel in the first call and from the previous call's neighbours afterwards (while
// B * n stays the same), kept work lists in between.  Results: the same (dist, lowest index among ties) bit for bit.
int fdcap_chamfer_fwd_scene(fdcap_ctx* c, const float* xyz1, int32_t B, int32_t n, float* dist1, int32_t* idx1, int32_t forget,
                            void* stream) {
    if (!c || !xyz1 || !dist1 || !idx1 || B <= 0 || n <= 0 || (int64_t)B * n > 0x7fffffff) return FDCAP_E_ARG;
    if (c->ns <= 0 || !c->scene_sorted.p) return FDCAP_E_STATE;
    hipStream_t st = (hipStream_t)stream;
    fdcap_ctx::SceneOp& so = c->sop;
    const int nq = B * n;
    const bool fresh = forget || so.nq != nq;
    if (so.nq != nq) {
        const size_t ng = ((size_t)nq + 31) / 32, ng4 = 4 * ng;
        HIP_TRY(so.dist.ensure(nq)); HIP_TRY(so.idx.ensure(nq)); HIP_TRY(so.seedpt.ensure(nq));
        HIP_TRY(so.ids.ensure(ng4 * NN_CACHE_CAP)); HIP_TRY(so.hdr.ensure(ng4 + 3 * ng)); HIP_TRY(so.anchor.ensure((size_t)4 * nq));
    }
    if (fresh) {
        const size_t ng = ((size_t)nq + 31) / 32, ng4 = 4 * ng;
        HIP_TRY(hipMemsetAsync(so.idx.p, 0xFF, (size_t)nq * sizeof(int), st));                 // -1: no seed
        HIP_TRY(hipMemsetAsync(so.hdr.p, 0xFF, ng4 * sizeof(int), st));                        // -1: nothing kept
        HIP_TRY(hipMemsetAsync(so.hdr.p + ng4, 0, 3 * ng * sizeof(int), st));
        HIP_TRY(hipMemsetAsync(so.anchor.p, 0, (size_t)4 * nq * sizeof(float4), st));
    }
    so.nq = nq;
    const NNTarget T = c->nn_target(true);
    const int nsplit = nn_pick_nsplit(nq, (int)c->ns, true);
    HIP_TRY(c->ws_f[0].ensure((size_t)nsplit * nq));
    HIP_TRY(c->ws_i[0].ensure((size_t)nsplit * nq));
    static float slack = -1.f;
    if (slack < 0.f) { const char* e = getenv("FDCAP_NN_CACHE_SLACK"); slack = e ? (float)atof(e) : 0.03f; }
    const NNCache cache{slack > 0.f ? so.ids.p : nullptr, slack > 0.f ? so.hdr.p : nullptr, so.anchor.p, slack};
    bool pt_written = false;
    HIP_TRY(nn_search(xyz1, nq, T, so.dist.p, so.idx.p, c->ws_f[0].p, c->ws_i[0].p, nsplit, st, so.idx.p, fresh, so.seedpt.p, &pt_written,
                      &cache, nullptr));
    if (!pt_written) so.nq = 0;                             // (a size the streaming search does not take: the next call re-seeds)
    HIP_TRY(hipMemcpyAsync(dist1, so.dist.p, (size_t)nq * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(idx1, so.idx.p, (size_t)nq * sizeof(int), hipMemcpyDeviceToDevice, st));
    return (int)hipGetLastError();
}

// ... and its gradient wrt the queries: the scene points come from the library's own copy (no pack pass per call)
int fdcap_chamfer_bwd_scene(fdcap_ctx* c, const float* xyz1, int32_t B, int32_t n, const float* gdist1, const int32_t* idx1, float* gxyz1,
                            void* stream) {
    if (!c || !xyz1 || !gdist1 || !idx1 || !gxyz1 || B <= 0 || n <= 0) return FDCAP_E_ARG;
    if (c->ns <= 0 || !c->scene.p) return FDCAP_E_STATE;
    const int nq = B * n;
    hipLaunchKernelGGL(nn_grad_kernel, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream, xyz1, c->scene.p, gdist1, idx1, nq, gxyz1);
    return (int)hipGetLastError();
}

// ---- Op 3 ----------------------------------------------------------------------------------
int fdcap_vposer_decode(fdcap_ctx* c, const float* z, int32_t ldz, int32_t B, float* rot, float* aa, void* stream) {
    if (!c || !z || B <= 0 || ldz < 32 || (!rot && !aa)) return FDCAP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(c->ws_f[2].ensure((size_t)B * 512));
    HIP_TRY(c->ws_f[3].ensure((size_t)B * 512));
    HIP_TRY(c->ws_f[4].ensure((size_t)B * ODIM));
    HIP_TRY(c->ws_part.ensure((size_t)4 * B * ODIM));
    int e = vposer_forward(c, z, ldz, 0, 0, B, c->ws_f[2].p, c->ws_f[3].p, c->ws_part.p, (size_t)B * ODIM, c->ws_f[4].p, st);
    if (e) return e;
    int n = B * 21;
    hipLaunchKernelGGL(sixd_to_rot_kernel, dim3((n + 255) / 256), dim3(256), 0, st, c->ws_f[4].p, n, rot, aa);
    return (int)hipGetLastError();
}

int fdcap_vposer_decode_bwd(fdcap_ctx* c, const float* z, int32_t ldz, int32_t B, const float* g_rot, const float* g_aa, float* g_z,
                            void* stream) {
    if (!c || !z || !g_z || B <= 0 || ldz < 32 || (!g_rot && !g_aa)) return FDCAP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    DevBuf<float>* w = c->ws_b;
    HIP_TRY(w[0].ensure((size_t)B * 512)); HIP_TRY(w[1].ensure((size_t)B * 512)); HIP_TRY(w[2].ensure((size_t)B * ODIM));
    HIP_TRY(w[3].ensure((size_t)B * ODIM)); HIP_TRY(w[4].ensure((size_t)4 * B * VP_Z));
    HIP_TRY(c->ws_part.ensure((size_t)4 * B * ODIM));
    // recompute the forward's activations (the operator keeps no state between calls), then the data-gradient chain
    int e = vposer_forward(c, z, ldz, 0, 0, B, w[0].p, w[1].p, c->ws_part.p, (size_t)B * ODIM, w[2].p, st);
    if (e) return e;
    const int n = B * 21;
    hipLaunchKernelGGL(vposer_out_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w[2].p, n, g_rot, g_aa, w[3].p);
    const size_t ps = (size_t)B * VP_Z;
    if (gemm_split3_enabled())
        hipLaunchKernelGGL(vposer_bwd_split3_kernel, dim3(4 * ((B + 15) / 16)), dim3(512), 0, st, c->vp3, w[3].p, 0, B, w[0].p, w[1].p, w[4].p, ps, ScaleTail());
    else
        hipLaunchKernelGGL(vposer_bwd_fused_kernel, dim3(4 * ((B + 15) / 16)), dim3(512), 0, st, c->vp, w[3].p, 0, B, w[0].p, w[1].p, w[4].p, ps, ScaleTail());
    hipLaunchKernelGGL(vposer_fold_dz_rows_kernel, dim3((B * VP_Z + 255) / 256), dim3(256), 0, st, w[4].p, ps, B, g_z);
    return (int)hipGetLastError();
}

// ---- parameter conversions -------------------------------------------------------------------
int fdcap_params_75_to_78(const float* p75, int32_t B, float* x78, void* stream) {
    if (!p75 || !x78 || B <= 0) return FDCAP_E_ARG;
    hipLaunchKernelGGL(p75_to_78_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, p75, B, x78);
    return (int)hipGetLastError();
}
int fdcap_params_78_to_75(const float* x78, int32_t B, float* p75, void* stream) {
    if (!p75 || !x78 || B <= 0) return FDCAP_E_ARG;
    hipLaunchKernelGGL(p78_to_75_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, x78, B, p75);
    return (int)hipGetLastError();
}

// ---- Op 2 ----------------------------------------------------------------------------------
// shared by fdcap_body_forward (body frame) and fdcap_world_mesh (scale + camera_ext @ T(cam_t * scale))
static int body_forward_impl(fdcap_ctx* c, const float* params, int32_t B, const float* cam_ext, const float* scale,
                             float* vertices, float* joints, hipStream_t st) {
    if (vertices && !c->full_ready) {
        std::vector<int64_t> all(c->V);
        for (int i = 0; i < c->V; ++i) all[i] = i;
        int e = build_skin_set(c, all, &c->full);
        if (e) return e;
        c->full_ready = true;
    }
    const int V = c->V;
    const bool world = cam_ext != nullptr && scale != nullptr;
    DevBuf<float>* w = c->ws_f;
    HIP_TRY(w[2].ensure((size_t)B * 512)); HIP_TRY(w[3].ensure((size_t)B * 512)); HIP_TRY(w[4].ensure((size_t)B * ODIM));
    HIP_TRY(w[5].ensure((size_t)B * XDIM)); HIP_TRY(w[6].ensure((size_t)B * NPFX)); HIP_TRY(w[7].ensure((size_t)B * NJ * 12));
    HIP_TRY(w[8].ensure((size_t)B * NJ * 12)); HIP_TRY(w[9].ensure((size_t)B * 16)); HIP_TRY(w[10].ensure(1));
    HIP_TRY(w[1].ensure((size_t)B * 12));
    float* X = w[5].p;
    hipLaunchKernelGGL(p75_to_78_kernel, dim3((B + 127) / 128), dim3(128), 0, st, params, B, X);
    HIP_TRY(c->ws_part.ensure((size_t)4 * B * ODIM));
    int e = vposer_forward(c, X, XDIM, X_LATENT, 0, B, w[2].p, w[3].p, c->ws_part.p, (size_t)B * ODIM, w[4].p, st);
    if (e) return e;
    if (!world) {
        HIP_TRY(hipMemsetAsync(w[9].p, 0, (size_t)B * 16 * sizeof(float), st));
        HIP_TRY(hipMemsetAsync(w[10].p, 0, sizeof(float), st));
    }
    const float* CAM = world ? cam_ext : w[9].p;
    const float* S = world ? scale : w[10].p;
    hipLaunchKernelGGL(pose_fwd_kernel<false>, dim3(B), dim3(64 * POSE_NW), 0, st, c->pose_model(), X, w[4].p, CAM, S, 0,
                       (float*)nullptr, w[6].p, (float*)nullptr, w[7].p, w[8].p, w[1].p, (float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (size_t)0);
    if (joints) hipLaunchKernelGGL(joints_out_kernel, dim3((B * NJ + 255) / 256), dim3(256), 0, st, w[7].p, X, XDIM, B, joints);
    if (vertices) {
        HIP_TRY(w[11].ensure((size_t)B * 3 * V));
        HIP_TRY(blend_forward(c->full, w[6].p, B, w[11].p, st));
        hipLaunchKernelGGL(skin_fwd_kernel, dim3((V + 255) / 256, B), dim3(256), 0, st, c->full.model(), V, X, XDIM, X_BETAS,
                           X_TRANSL, w[11].p, w[8].p, (const float*)w[1].p, S, 0, world ? 1 : 0, vertices);
    }
    return (int)hipGetLastError();
}

int fdcap_body_forward(fdcap_ctx* c, const float* params, int32_t B, float* vertices, float* joints, void* stream) {
    if (!c || !params || B <= 0 || (!vertices && !joints)) return FDCAP_E_ARG;
    return body_forward_impl(c, params, B, nullptr, nullptr, vertices, joints, (hipStream_t)stream);
}

int fdcap_world_mesh(fdcap_ctx* c, const float* params, int32_t B, const float* cam_ext, const float* scale, float* vertices,
                     void* stream) {
    if (!c || !params || B <= 0 || !cam_ext || !scale || !vertices) return FDCAP_E_ARG;
    return body_forward_impl(c, params, B, cam_ext, scale, vertices, nullptr, (hipStream_t)stream);
}

int fdcap_smplx_forward(fdcap_ctx* c, const float* go, const float* bp, const float* betas, const float* lh, const float* rh,
                        const float* transl, int32_t B, float* vertices, float* joints, void* stream) {
    if (!c || !go || !bp || !betas || !lh || !rh || !transl || B <= 0 || (!vertices && !joints)) return FDCAP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (vertices && !c->full_ready) {
        std::vector<int64_t> all(c->V);
        for (int i = 0; i < c->V; ++i) all[i] = i;
        int e = build_skin_set(c, all, &c->full);
        if (e) return e;
        c->full_ready = true;
    }
    const int V = c->V;
    DevBuf<float>* w = c->ws_f;
    HIP_TRY(w[4].ensure((size_t)B * 66));
    HIP_TRY(w[5].ensure((size_t)B * XDIM)); HIP_TRY(w[6].ensure((size_t)B * NPFX)); HIP_TRY(w[7].ensure((size_t)B * NJ * 12));
    HIP_TRY(w[8].ensure((size_t)B * NJ * 12)); HIP_TRY(w[9].ensure((size_t)B * 16)); HIP_TRY(w[10].ensure(1));
    float* X = w[5].p;
    hipLaunchKernelGGL(assemble_rows_kernel, dim3((B + 127) / 128), dim3(128), 0, st, go, bp, betas, lh, rh, transl, B, X, w[4].p);
    HIP_TRY(hipMemsetAsync(w[9].p, 0, (size_t)B * 16 * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(w[10].p, 0, sizeof(float), st));
    hipLaunchKernelGGL(pose_fwd_kernel<false>, dim3(B), dim3(64 * POSE_NW), 0, st, c->pose_model(), X, (float*)nullptr, w[9].p, w[10].p, 0,
                       (float*)nullptr, w[6].p, (float*)nullptr, w[7].p, w[8].p, (float*)nullptr, (float*)nullptr, (const float*)w[4].p,
                       (const float*)nullptr, (size_t)0);
    if (joints) hipLaunchKernelGGL(joints_out_kernel, dim3((B * NJ + 255) / 256), dim3(256), 0, st, w[7].p, X, XDIM, B, joints);
    if (vertices) {
        HIP_TRY(w[11].ensure((size_t)B * 3 * V));
        HIP_TRY(blend_forward(c->full, w[6].p, B, w[11].p, st));
        hipLaunchKernelGGL(skin_fwd_kernel, dim3((V + 255) / 256, B), dim3(256), 0, st, c->full.model(), V, X, XDIM, X_BETAS,
                           X_TRANSL, w[11].p, w[8].p, (const float*)nullptr, (const float*)nullptr, 0, 0, vertices);
    }
    return (int)hipGetLastError();
}

int fdcap_smplx_backward(fdcap_ctx* c, const float* go, const float* bp, const float* betas, const float* lh, const float* rh,
                         const float* transl, int32_t B, const float* g_vertices, const float* g_joints, float* g_go, float* g_bp,
                         float* g_betas, float* g_lh, float* g_rh, float* g_transl, void* stream) {
    if (!c || !go || !bp || !betas || !lh || !rh || !transl || B <= 0 || (!g_vertices && !g_joints)) return FDCAP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (g_vertices && !c->full_ready) {
        std::vector<int64_t> all(c->V);
        for (int i = 0; i < c->V; ++i) all[i] = i;
        int e = build_skin_set(c, all, &c->full);
        if (e) return e;
        c->full_ready = true;
    }
    const int V = c->V;
    const size_t nv3 = (size_t)3 * V;
    DevBuf<float>* w = c->ws_b;
    // forward state (recomputed: the operator keeps none): X rows, AA, PF, Rm, Jrest, G, A
    HIP_TRY(w[0].ensure((size_t)B * XDIM)); HIP_TRY(w[1].ensure((size_t)B * 66)); HIP_TRY(w[2].ensure((size_t)B * NPFX));
    HIP_TRY(w[3].ensure((size_t)B * NJ * 9)); HIP_TRY(w[4].ensure((size_t)B * NJ * 3)); HIP_TRY(w[5].ensure((size_t)B * NJ * 12));
    HIP_TRY(w[6].ensure((size_t)B * NJ * 12));
    // gradients: dA, [dtransl_v 3 | dMv 12 | dsv 1 | identity M 12 | cam 16] per row + scale, dPF, dX, dAA
    HIP_TRY(w[7].ensure((size_t)B * NJ * 12)); HIP_TRY(w[8].ensure((size_t)B * 44 + 4)); HIP_TRY(w[9].ensure((size_t)B * NPFX));
    HIP_TRY(w[10].ensure((size_t)B * XDIM)); HIP_TRY(w[11].ensure((size_t)B * 66));
    float* X = w[0].p; float* AA = w[1].p; float* PF = w[2].p;
    float* dtv = w[8].p; float* dMv = dtv + (size_t)B * 3; float* dsv = dMv + (size_t)B * 12; float* Mid = dsv + B;
    float* cam0 = Mid + (size_t)B * 12; float* one = cam0 + (size_t)B * 16;
    hipLaunchKernelGGL(assemble_rows_kernel, dim3((B + 127) / 128), dim3(128), 0, st, go, bp, betas, lh, rh, transl, B, X, AA);
    HIP_TRY(hipMemsetAsync(cam0, 0, ((size_t)B * 16 + 4) * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(w[10].p, 0, (size_t)B * XDIM * sizeof(float), st));
    hipLaunchKernelGGL(pose_fwd_kernel<false>, dim3(B), dim3(64 * POSE_NW), 0, st, c->pose_model(), X, (float*)nullptr, cam0, one /* = 0 here */, 0,
                       w[3].p, PF, w[4].p, w[5].p, w[6].p, (float*)nullptr, (float*)nullptr, (const float*)AA, (const float*)nullptr,
                       (size_t)0);
    const float* dA = nullptr; const float* dPF = nullptr; const float* dtr = nullptr;
    if (g_vertices) {
        // body-frame vertices = the world form with M = [I | 0] and scale = 1
        hipLaunchKernelGGL(identity_rows_kernel, dim3((B * 12 + 255) / 256), dim3(256), 0, st, Mid, B, one);
        DevBuf<float>* wf = c->ws_f;
        HIP_TRY(wf[11].ensure((size_t)B * nv3));                 // pose + shape blend offsets
        HIP_TRY(wf[0].ensure((size_t)B * nv3));                  // d offsets
        HIP_TRY(blend_forward(c->full, PF, B, wf[11].p, st));
        hipLaunchKernelGGL(skin_bwd_kernel<false>, dim3(B), dim3(256), (size_t)std::min(V, 1024) * 12 * sizeof(float), st, c->full.model(), V,
                           X, wf[11].p, w[6].p, Mid, one, 0, g_vertices, wf[0].p, w[7].p, (float*)nullptr, dtv, dMv, dsv, ContactGradIn());
        HIP_TRY(gemm_f32(true, EPI_STORE, wf[0].p, 3 * V, c->full.posedirs.p, c->full.ldp, w[9].p, NPFX, B, NPFX, 3 * V, nullptr, 0, st));
        dA = w[7].p; dPF = w[9].p; dtr = dtv;
    }
    hipLaunchKernelGGL(pose_bwd_op_kernel, dim3(B), dim3(64), 0, st, c->pose_model(), X, AA, w[3].p, w[4].p, w[5].p, dA, dPF, dtr,
                       g_joints, w[10].p, w[11].p);
    hipLaunchKernelGGL(smplx_bwd_split_kernel, dim3((B + 127) / 128), dim3(128), 0, st, w[10].p, w[11].p, B, g_go, g_bp, g_betas, g_lh,
                       g_rh, g_transl);
    return (int)hipGetLastError();
}

// ---- optimiser -------------------------------------------------------------------------------
void fdcap_opt_destroy(fdcap_ctx* c) {
    if (!c || !c->opt) return;
    OptState* o = c->opt;
    DevBuf<float>* fb[] = {&o->X0, &o->mask, &o->mX, &o->vX, &o->mCAM, &o->vCAM, &o->mS, &o->vS,
                           &o->H1, &o->H2, &o->O, &o->dO, &o->Opart, &o->dZpart, &o->Rm, &o->PF, &o->Jrest, &o->G, &o->A, &o->M,
                           &o->Jw, &o->Voff, &o->Vw, &o->dist, &o->pd, &o->dVoff, &o->dA, &o->dtransl_v, &o->dMv,
                           &o->dsv, &o->dPF, &o->dJw, &o->dX, &o->dCAM, &o->dscale_row, &o->loss_rows, &o->VoffF, &o->VwF, &o->dVF};
    for (auto* b : fb) b->release();
    o->dctD.release(); o->dctCoef.release(); o->dctM.release(); o->dctV.release(); o->adam_tab.release();
    o->idx.release(); o->pi.release(); o->seedpt.release(); o->kp2d.release(); o->floss.release();
    if (o->lbfgs) { fdcap_lbfgs_destroy(o->lbfgs); o->lbfgs = nullptr; }
    o->nnc_ids.release(); o->nnc_hdr.release(); o->nnc_anchor.release();
    for (hipEvent_t e : o->nn_ev) (void)hipEventDestroy(e);
    o->nn_ev.clear();
    delete o;
    c->opt = nullptr;
}

int fdcap_opt_create(fdcap_ctx* c, const fdcap_opt_config* cfg, float* rows_x, float* rows_cam, float* scale_d,
                     float* dscale_d, double* losses_d) {
    if (!c || !cfg || !rows_x || !rows_cam || !scale_d || !dscale_d || !losses_d || cfg->n_local <= 0 || cfg->n_total < cfg->n_local || cfg->frame0 < 0 ||
        cfg->frame0 + cfg->n_local > cfg->n_total)
        return FDCAP_E_ARG;
    // a second clip of the same shape reuses the scratch allocations (every buffer is re-zeroed below)
    OptState* o = c->opt ? c->opt : new OptState();
    c->opt = o;
    o->cfg = *cfg;
    o->cam_steps = 0;
    o->dz_pending = false;
    o->log_pending = false;
    o->seeded = false;
    o->dctT = o->dctC = o->dctW = 0;
    o->dct_grad = false;
    const int R = o->R = cfg->n_local + 4;
    o->contact_on = c->ns > 0 && c->nc > 0 && cfg->weight_contact != 0.f;
    const size_t nq = (size_t)R * std::max(c->nc, 1);
    {
        const char* e1 = getenv("FDCAP_NN_SEED");
        const char* e2 = getenv("FDCAP_NN_CULL");
        o->use_seed = !(e1 && e1[0] == '0');
        o->use_cull = !(e2 && e2[0] == '0');
    }
    const int nq_all = (int)((size_t)cfg->n_local * c->nc);
    o->nsplit = o->contact_on ? nn_pick_nsplit(nq_all, (int)c->ns, o->use_seed && o->use_cull) : 1;
    o->nsplit_bf = o->contact_on ? nn_pick_nsplit(nq_all, (int)c->ns, false) : 1;
    if (const char* e = getenv("FDCAP_NN_NSPLIT")) o->nsplit = o->nsplit_bf = std::max(1, atoi(e));      // tuning knob
    int err = 0;
#define AL(buf, cnt) if (!err) { hipError_t e_ = (buf).ensure(cnt); if (e_ != hipSuccess) err = (int)e_; else e_ = hipMemset((buf).p, 0, (size_t)(cnt) * sizeof(*(buf).p)); }
    o->X.p = rows_x; o->CAM.p = rows_cam; o->scale.p = scale_d; o->dscale.p = dscale_d; o->losses.p = losses_d;
    o->pend.on = false;
    AL(o->X0, (size_t)R * XDIM) AL(o->mask, R)
    AL(o->mX, (size_t)R * XDIM) AL(o->vX, (size_t)R * XDIM) AL(o->mCAM, (size_t)R * 16) AL(o->vCAM, (size_t)R * 16)
    AL(o->mS, 1) AL(o->vS, 1)
    AL(o->H1, (size_t)R * 512) AL(o->H2, (size_t)R * 512) AL(o->O, (size_t)R * O_LD) AL(o->dO, (size_t)R * ODIM)
    AL(o->Opart, (size_t)4 * R * ODIM) AL(o->dZpart, (size_t)4 * R * VP_Z)
    AL(o->Rm, (size_t)R * RM_LD) AL(o->PF, (size_t)R * NPFX) AL(o->Jrest, (size_t)R * JR_LD) AL(o->G, (size_t)R * NJ * 12)
    AL(o->A, (size_t)R * NJ * 12) AL(o->M, (size_t)R * 12) AL(o->Jw, (size_t)R * NJW * 3)
    AL(o->dA, (size_t)R * NJ * 12) AL(o->dtransl_v, (size_t)R * 3) AL(o->dMv, (size_t)R * 12)
    AL(o->dsv, R) AL(o->dPF, (size_t)2 * R * NPFX)   /* [2][R, 496]: the second half only as the K-split product's second partial */
    AL(o->dJw, (size_t)R * NJW * 3) AL(o->dX, (size_t)R * XDIM)
    AL(o->dCAM, (size_t)R * 16) AL(o->dscale_row, R) AL(o->loss_rows, (size_t)R * LROW)
    if (o->contact_on) {
        AL(o->Voff, nq * 3) AL(o->Vw, nq * 3) AL(o->dist, nq) AL(o->idx, nq) AL(o->dVoff, nq * 3) AL(o->seedpt, nq)
        AL(o->pd, (size_t)std::max(o->nsplit, o->nsplit_bf) * nq) AL(o->pi, (size_t)std::max(o->nsplit, o->nsplit_bf) * nq)
    }
#undef AL
    if (!err && o->contact_on) {
        hipError_t e_ = hipMemset(o->idx.p, 0xFF, nq * sizeof(int));      // -1: no seed yet
        if (e_ != hipSuccess) err = (int)e_;
    }
    if (!err && o->contact_on) {
        if (const char* e = getenv("FDCAP_SKIN_VEC")) o->skin_vec = e[0] != '0';
        if (const char* e = getenv("FDCAP_NN_CACHE_SLACK")) o->nnc_slack = (float)atof(e);
        int every = 32;
        if (const char* e = getenv("FDCAP_NN_ORDER")) every = atoi(e);
        o->nn_order = NNOrder{};
        if (o->nnc_slack > 0.f) {                                        // groups of 32 queries x up to 4 waves per group
            const size_t ng = ((size_t)nq_all + 31) / 32, ng4 = ng * 4;
            hipError_t e_ = o->nnc_ids.ensure(ng4 * NN_CACHE_CAP);
            if (e_ == hipSuccess) e_ = o->nnc_hdr.ensure(ng4 + 3 * ng);                                        // + work counts [ng] + two launch-order tables [ng]
            if (e_ == hipSuccess) e_ = o->nnc_anchor.ensure((size_t)4 * nq_all);
            if (e_ == hipSuccess) e_ = hipMemset(o->nnc_hdr.p, 0xFF, ng4 * sizeof(int));                       // -1: nothing kept
            if (e_ == hipSuccess) e_ = hipMemset(o->nnc_hdr.p + ng4, 0, 3 * ng * sizeof(int));
            if (e_ == hipSuccess) e_ = hipMemset(o->nnc_anchor.p, 0, (size_t)4 * nq_all * sizeof(float4));
            if (e_ != hipSuccess) err = (int)e_;
            if (every > 0) { o->nn_order.on = true; o->nn_order.every = every; }
        }
    }
    if (!err) {
        float s = cfg->scale_init;
        hipError_t e_ = hipMemcpy(o->scale.p, &s, sizeof(float), hipMemcpyHostToDevice);
        if (e_ != hipSuccess) err = (int)e_;
    }
    if (err) { fdcap_opt_destroy(c); return err; }
    return FDCAP_OK;
}

int fdcap_opt_set_inputs(fdcap_ctx* c, const float* data78, const float* init78, const float* mask, const float* cam,
                         void* stream) {
    if (!c || !c->opt || !data78 || !init78 || !mask || !cam) return FDCAP_E_ARG;
    { int es_ = opt_sync(c, (hipStream_t)stream); if (es_) return es_; }
    OptState* o = c->opt;
    hipStream_t st = (hipStream_t)stream;
    const size_t n = o->cfg.n_local;
    o->log_pending = false; o->log_dst = nullptr;
    HIP_TRY(hipMemcpyAsync(o->X0.p + 2 * XDIM, data78, n * XDIM * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(o->X.p + 2 * XDIM, init78, n * XDIM * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(o->mask.p + 2, mask, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(o->CAM.p + 2 * 16, cam, n * 16 * sizeof(float), hipMemcpyDeviceToDevice, st));
    return FDCAP_OK;
}

static int opt_contact_forward(fdcap_ctx* c, hipStream_t st, bool blend_done = false) {
    OptState* o = c->opt;
    const int nl = o->cfg.n_local, nc = c->nc;
    const size_t off = (size_t)2 * nc * 3;
    if (!blend_done) HIP_TRY(blend_forward(c->contact, o->PF.p + 2 * NPFX, nl, o->Voff.p + off, st));
    hipLaunchKernelGGL(skin_fwd_kernel, dim3((nc + 255) / 256, nl), dim3(256), 0, st, c->contact.model(), nc, o->X.p, XDIM,
                       X_BETAS, X_TRANSL, o->Voff.p, o->A.p, o->M.p, o->scale.p, 2, 1, o->Vw.p);
    const int nq = nl * nc;
    // the first contact forward of a fit has no neighbours from a previous iteration yet (idx = -1)
    const NNCache cache = o->nn_cache(0);
    const bool timed = o->nn_timing && o->nn_ev_used + 2 <= (int)o->nn_ev.size();
    if (timed) HIP_TRY(hipEventRecord(o->nn_ev[o->nn_ev_used], st));
    {
        TraceRange tr_("fdcap:chamfer_nn(K14)");
        HIP_TRY(nn_search(o->Vw.p + off, nq, c->nn_target(o->use_cull), o->dist.p + 2 * nc, o->idx.p + 2 * nc, o->pd.p, o->pi.p,
                          o->nsplit, st, o->use_seed ? o->idx.p + 2 * nc : nullptr, !o->seeded, o->seedpt.p + 2 * nc, &o->nnpt_valid,
                          &cache, &o->nn_order));
    }
    if (timed) { HIP_TRY(hipEventRecord(o->nn_ev[o->nn_ev_used + 1], st)); o->nn_ev_used += 2; }
    o->seeded = true;
    return 0;
}

namespace {
// weights of the loss total of one iteration (multipliers of the lossconfig weights, :570 / :582 / :620)
struct LossWeights { float rec, smooth, contact, world, dct; bool world_on; };
}

// fuse_ii >= 0 (fdcap_opt_backward_and_step): this backward is followed by the optimiser step of iteration fuse_ii -- `scale`
// is stepped by one more workgroup of the last launch, the rows' part is left pending for the next forward (DeferredStep)
static int opt_backward_impl(fdcap_ctx* c, const LossWeights& lw, int32_t log_terms, hipStream_t st, int fuse_ii = -1, int fuse_P = 0) {
    OptState* o = c->opt;
    const fdcap_opt_config& cf = o->cfg;
    const int nl = cf.n_local, nc = c->nc, N = cf.n_total;
    const bool dct_on = lw.dct != 0.f && o->dctW > 0;
    PoseModel pm = c->pose_model();
    TraceRange tr_(fuse_ii >= 0 ? "fdcap:backward_and_step" : "fdcap:backward");
    if (o->log_pending) {                               // a deferred reduction nobody stepped after: deliver it before loss_rows is rewritten
        hipLaunchKernelGGL(loss_rows_reduce_kernel, dim3(1), dim3(256), 0, st, o->loss_rows.p, 2, nl, o->log_mask, o->log_assign, o->log_dst,
                           o->dscale_row.p, o->dscale.p);
        o->log_pending = false;
    }
    double* const losses = log_terms ? o->losses.p : nullptr;       // the partial sums are only formed on logging iterations
    // Logging without a DCT term: every printed term leaves per-frame partials in loss_rows (inside the kernels that run
    // anyway), one small launch sums them.  With the DCT term the separate param_loss_kernel / dct kernel add into losses[].
    const bool rows_log = losses && o->dctW == 0;
    if (losses && !rows_log) HIP_TRY(hipMemsetAsync(losses, 0, FDCAP_NUM_LOSSES * sizeof(double), st));
    int row_lo, row_hi;
    opt_row_range(o, 1, &row_lo, &row_hi);
    const bool ahead = o->ahead, blend_done = o->ahead && o->ahead_blend;
    const bool contact_grad = o->contact_on && lw.contact != 0.f;
    const bool contact_fwd = o->contact_on && (contact_grad || log_terms);
    int e = ahead ? opt_pose_forward_rest(c, row_lo, row_hi, st) : opt_pose_forward(c, row_lo, row_hi, st, contact_fwd || (dct_on || o->dctW > 0));
    o->ahead = false;
    if (e) return e;
    if (contact_fwd) { e = opt_contact_forward(c, st, blend_done); if (e) return e; }
    const float w_rec = lw.rec * cf.weight_loss_rec / ((float)N * XDIM);
    const float w_sm = (N >= 3) ? lw.smooth / ((float)(N - 2) * XDIM) : 0.f;
    const float w_ws = (lw.world_on && N >= 2) ? lw.world / ((float)(N - 1) * NJW * 3) : 0.f;
    // the parameter-space terms: their own kernel when the loss sums are wanted or the DCT term also writes dJw, else
    // formed inside pose_bwd_kernel (one launch less per iteration)
    const bool fuse_pl = (!losses || rows_log) && !(o->dctW > 0 && dct_on);
    ParamLossIn pli = {};
    if (fuse_pl) pli = ParamLossIn{o->X0.p, o->mask.p, o->Jw.p, cf.frame0, N, w_rec, w_sm, w_ws, lw.world_on ? 1 : 0,
                                   rows_log ? o->loss_rows.p : nullptr};
    else
        hipLaunchKernelGGL(param_loss_kernel, dim3(nl), dim3(128), 0, st, o->X.p, o->X0.p, o->mask.p, o->Jw.p, 2, cf.frame0, N,
                           w_rec, w_sm, w_ws, lw.world_on ? 1 : 0, o->dX.p, o->dJw.p, losses);
    if (o->dctW > 0 && (dct_on || log_terms))
        hipLaunchKernelGGL(dct_joint_grad_kernel, dim3((nl * 69 + 255) / 256), dim3(256), 0, st, o->Jw.p, 2, cf.frame0, nl, o->dctT,
                           o->dctC, o->dctW, o->dctD.p, o->dctCoef.p, dct_on ? lw.dct / (69.f * (float)o->dctW) : 0.f,
                           lw.world_on ? 1 : 0, o->dJw.p, losses ? losses + 7 : nullptr);
    o->dct_grad = dct_on;
    bool dpf_split = false;
    if (contact_grad) {
        ContactGradIn cg;
        cg.Vw = o->Vw.p; cg.dist = o->dist.p; cg.idx = o->idx.p; cg.scene = c->scene.p;
        cg.nnpt = o->nnpt_valid ? o->seedpt.p : nullptr;
        cg.coef = lw.contact * cf.weight_contact / ((float)N * nc);
        cg.loss_rows = losses ? o->loss_rows.p : nullptr;
        const size_t lds_small = (size_t)6 * nc * sizeof(float) + (size_t)c->contact.nnz * sizeof(float) + (((size_t)c->contact.nnz * 2 + 15) & ~(size_t)15);
        if (nc <= SKS_MAXV && c->contact.nnz <= SKS_MAXNNZ && lds_small <= 57000) {      // (+ 6.4 KB of static LDS <= 64 KB)
            const int nnz = c->contact.nnz;
            const size_t lds = lds_small;
            const SkinModel smc = c->contact.model();
            const int G = (smc.K + 3) / 4;                           // weight groups per vertex: the packed layout covers K <= 12
            const size_t ldsv = (size_t)9 * nc * sizeof(float) + (size_t)((nnz + 3) & ~3) * sizeof(float) + (size_t)((nnz + 7) & ~7) * 2;
            if (o->skin_vec && nc <= 512 && G <= 3 && nnz <= 2048 * G && ldsv <= 60000 && (nc & 3) == 0 && smc.vpack && smc.csc_v16 &&
                (((size_t)o->Vw.p | (size_t)o->Voff.p | (size_t)o->dVoff.p | (size_t)o->A.p) & 15) == 0) {
#define FDC_SKV(GG) hipLaunchKernelGGL(skin_bwd_vec_kernel<GG>, dim3(nl), dim3(256), ldsv, st, smc, nc, nnz, o->X.p, o->Voff.p, o->A.p, \
                                   o->M.p, o->scale.p, 2, o->dVoff.p, o->dA.p, o->dtransl_v.p, o->dMv.p, o->dsv.p, cg)
                if (G == 1) FDC_SKV(1); else if (G == 2) FDC_SKV(2); else FDC_SKV(3);
#undef FDC_SKV
            } else if (nc <= 512 && nnz <= 2048)
                hipLaunchKernelGGL((skin_bwd_small_kernel<2, 8>), dim3(nl), dim3(256), lds, st, c->contact.model(), nc, nnz, o->X.p, o->Voff.p, o->A.p,
                                   o->M.p, o->scale.p, 2, o->dVoff.p, o->dA.p, o->dtransl_v.p, o->dMv.p, o->dsv.p, cg);
            else if (nc <= 512)                                       // (K > 4 at the loop's contact-set size: up to 6144 list entries)
                hipLaunchKernelGGL((skin_bwd_small_kernel<2, 24>), dim3(nl), dim3(256), lds, st, c->contact.model(), nc, nnz, o->X.p, o->Voff.p, o->A.p,
                                   o->M.p, o->scale.p, 2, o->dVoff.p, o->dA.p, o->dtransl_v.p, o->dMv.p, o->dsv.p, cg);
            else if (nnz <= 4096)
                hipLaunchKernelGGL((skin_bwd_small_kernel<4, 16>), dim3(nl), dim3(256), lds, st, c->contact.model(), nc, nnz, o->X.p, o->Voff.p, o->A.p,
                                   o->M.p, o->scale.p, 2, o->dVoff.p, o->dA.p, o->dtransl_v.p, o->dMv.p, o->dsv.p, cg);
            else
                hipLaunchKernelGGL((skin_bwd_small_kernel<4, 24>), dim3(nl), dim3(256), lds, st, c->contact.model(), nc, nnz, o->X.p, o->Voff.p, o->A.p,
                                   o->M.p, o->scale.p, 2, o->dVoff.p, o->dA.p, o->dtransl_v.p, o->dMv.p, o->dsv.p, cg);
        } else
        hipLaunchKernelGGL(skin_bwd_kernel<true>, dim3(nl), dim3(256), (size_t)std::min(nc, 1024) * 12 * sizeof(float), st, c->contact.model(), nc, o->X.p, o->Voff.p, o->A.p,
                           o->M.p, o->scale.p, 2, (const float*)nullptr, o->dVoff.p, o->dA.p, (float*)nullptr, o->dtransl_v.p,
                           o->dMv.p, o->dsv.p, cg);
        TraceRange tr_b("fdcap:blend_bwd(K8)");
        if (gemm_split3_enabled() && c->contact.pn_bwd3.f && panel_gemm3_rb2k_ok(nl, 3 * nc, c->contact.pn_bwd3)) {
            // two partial products (K halves), added by pose_bwd_kernel: [2][R, 496] in o->dPF
            dpf_split = true;
            HIP_TRY(panel_gemm3_rb2k(o->dVoff.p + (size_t)2 * nc * 3, 3 * nc, nl, 3 * nc, c->contact.pn_bwd3, o->dPF.p + 2 * NPFX,
                                     (size_t)o->R * NPFX, NPFX, NPFX, st));
        } else if (gemm_split3_enabled() && c->contact.pn_bwd3.f)
            HIP_TRY(panel_gemm3(o->dVoff.p + (size_t)2 * nc * 3, 3 * nc, nl, 3 * nc, c->contact.pn_bwd3, o->dPF.p + 2 * NPFX, NPFX, NPFX, st));
        else if (c->contact.pn_bwd.f)
            HIP_TRY(panel_gemm(o->dVoff.p + (size_t)2 * nc * 3, 3 * nc, nl, 3 * nc, c->contact.pn_bwd, o->dPF.p + 2 * NPFX, NPFX, NPFX, st));
        else
            HIP_TRY(gemm_f32(true, EPI_STORE, o->dVoff.p + (size_t)2 * nc * 3, 3 * nc, c->contact.posedirs.p, c->contact.ldp,
                             o->dPF.p + 2 * NPFX, NPFX, nl, NPFX, 3 * nc, nullptr, 0, st));
    } else if (contact_fwd && losses) {
        if (fuse_pl && rows_log) { pli.cdist = o->dist.p; pli.cnc = nc; }        // (rides in pose_bwd_kernel's prologue: one launch less)
        else hipLaunchKernelGGL(contact_loss_rows_kernel, dim3(nl), dim3(256), 0, st, o->dist.p, nc, 2, o->loss_rows.p);
    }
    const bool joint_grad = lw.world_on || dct_on;
    hipLaunchKernelGGL(pose_bwd_kernel, dim3(nl), dim3(64 * POSE_NW), 0, st, pm, o->X.p, o->O.p, o->CAM.p, o->scale.p, 2, o->Rm.p,
                       o->Jrest.p, o->G.p, contact_grad ? o->dA.p : nullptr, contact_grad ? o->dPF.p : nullptr,
                       joint_grad ? o->dJw.p : nullptr, contact_grad ? o->dMv.p : nullptr, contact_grad ? o->dsv.p : nullptr,
                       contact_grad ? o->dPF.p + NPF : nullptr, NPFX, contact_grad ? o->dtransl_v.p : nullptr, o->dX.p, o->dO.p,
                       o->dCAM.p, o->dscale_row.p, pli, (contact_grad && dpf_split) ? (const float*)(o->dPF.p + (size_t)o->R * NPFX) : (const float*)nullptr);
    const unsigned log_mask = (rows_log ? 0x17u : 0u) | (contact_fwd ? 0x8u : 0u);     // 0 rec, 1 z^2, 2 smoothing, 4 world | 3 contact
    bool log_in_tail = false;
    {
        ScaleTail tail;
        if (fuse_ii >= 0) {
            const StepPlan sp = opt_step_plan(o, fuse_ii, fuse_P, false, true);
            tail.dscale_row = o->dscale_row.p; tail.row0 = 2;
            if (sp.step_scale) {
                tail.block = 0; tail.sc = sp.sc; tail.dscale = o->dscale.p; tail.n = nl;
                tail.zero_grad = fuse_ii >= fuse_P ? 1 : 0;
            }
            if (log_terms == 2 && rows_log) {           // the printed sums: same extra workgroup (loss_rows is complete before this launch)
                tail.block = 0; tail.lg = LogReduceIn{o->loss_rows.p, losses, log_mask, 1, nl};
                log_in_tail = true;
            }
        }
        int eb = opt_vposer_backward(c, false, st, tail);
        if (eb) return eb;
        if (fuse_ii >= 0) { o->pend.on = true; o->pend.ii = fuse_ii; o->pend.P = fuse_P; }
    }
    // d loss / d scale of this rank = sum of the per-frame partials: formed by the step kernels (fused with Adam /
    // the exchange packing); on logging iterations also here, so a caller can read dscale_d right after the backward
    if (log_terms && !log_in_tail) {
        if (log_terms == 2 && rows_log) {               // the sums ride in the step launch that follows (one launch less per iteration)
            o->log_pending = true; o->log_mask = log_mask; o->log_assign = 1; o->log_dst = losses;
        } else
            hipLaunchKernelGGL(loss_rows_reduce_kernel, dim3(1), dim3(256), 0, st, o->loss_rows.p, 2, nl, log_mask, rows_log ? 1 : 0, losses,
                               o->dscale_row.p, o->dscale.p);
    }
    return (int)hipGetLastError();
}

int fdcap_opt_set_loss_output(fdcap_ctx* c, double* losses_d) {
    if (!c || !c->opt || !losses_d) return FDCAP_E_ARG;
    // a logging backward (log_terms = 2) that no step followed left its reduction pending, aimed at the OLD output: that
    // memory may be gone by now (a caller's history row) -- the pending delivery is dropped, never redirected or kept
    c->opt->log_pending = false;
    c->opt->log_dst = nullptr;
    c->opt->losses.p = losses_d;
    return FDCAP_OK;
}

// The part of iteration ii's forward that depends neither on `scale` nor on the halo rows, for the owned rows: decoder, pose
// state, and (when that iteration has a contact term or logs one) the contact set's pose-blend product.  A sharded run issues
// it between fdcap_opt_step_rows_and_pack and fdcap_opt_unpack_and_step_scale, so that it runs while the all-gather is in
// flight ("overlap C1 with the start of the next forward"); fdcap_opt_backward(ii) then only adds the rest.
int fdcap_opt_forward_ahead(fdcap_ctx* c, int32_t ii, int32_t P, int32_t log_terms, void* stream) {
    if (!c || !c->opt) return FDCAP_E_STATE;
    { int es_ = opt_sync(c, (hipStream_t)stream); if (es_) return es_; }
    OptState* o = c->opt;
    hipStream_t st = (hipStream_t)stream;
    const int nl = o->cfg.n_local, nc = c->nc;
    int e = opt_pose_forward(c, 2, 2 + nl, st);
    if (e) return e;
    const bool contact_fwd = o->contact_on && ((ii < P && o->cfg.phase1_contact != 0.f) || log_terms);
    if (contact_fwd) HIP_TRY(blend_forward(c->contact, o->PF.p + 2 * NPFX, nl, o->Voff.p + (size_t)2 * nc * 3, st));
    o->ahead = true;
    o->ahead_blend = contact_fwd;
    return (int)hipGetLastError();
}

int fdcap_opt_backward(fdcap_ctx* c, int32_t ii, int32_t P, int32_t log_terms, void* stream) {
    if (!c || !c->opt) return FDCAP_E_STATE;
    const fdcap_opt_config& cf = c->opt->cfg;
    const bool phase2 = ii >= P;
    LossWeights lw;
    lw.rec = 1.f;
    lw.smooth = phase2 ? cf.phase2_smooth : cf.phase1_smooth;
    lw.contact = phase2 ? 0.f : cf.phase1_contact;
    lw.world = phase2 ? cf.phase2_world : 0.f;
    lw.dct = 0.f;
    lw.world_on = phase2;
    return opt_backward_impl(c, lw, log_terms, (hipStream_t)stream);
}

// loss.backward() + optimizer.step() of iteration ii (:591-592) in one call and WITHOUT a launch for the step: `scale` is stepped by
// one more workgroup of the backward's last launch; the rows of body_rotation_rec / camera_ext take their Adam update in the first
// two launches of the NEXT forward, where they are read anyway (DeferredStep, csrc/fdc_loss.h) -- or in the ordinary Adam launch as
// soon as anything else needs them (every other entry point; fdcap_opt_sync).  Same arithmetic in the same order: same bits as
// fdcap_opt_backward + fdcap_opt_step, which is also what this call falls back to where the deferral cannot apply (sharded runs:
// the exchange needs the stepped rows; log_terms == 2: the logged sums ride in the step launch).
int fdcap_opt_backward_and_step(fdcap_ctx* c, int32_t ii, int32_t P, int32_t log_terms, void* stream) {
    if (!c || !c->opt) return FDCAP_E_STATE;
    OptState* o = c->opt;
    const fdcap_opt_config& cf = o->cfg;
    const bool fuse = cf.frame0 == 0 && cf.n_local == cf.n_total && o->dctW == 0;
    if (!fuse) {
        const int e = fdcap_opt_backward(c, ii, P, log_terms, stream);
        return e ? e : fdcap_opt_step(c, ii, P, stream);
    }
    const bool phase2 = ii >= P;
    LossWeights lw;
    lw.rec = 1.f;
    lw.smooth = phase2 ? cf.phase2_smooth : cf.phase1_smooth;
    lw.contact = phase2 ? 0.f : cf.phase1_contact;
    lw.world = phase2 ? cf.phase2_world : 0.f;
    lw.dct = 0.f;
    lw.world_on = phase2;
    return opt_backward_impl(c, lw, log_terms, (hipStream_t)stream, ii, P);
}

// ---- mode 'dct' (global_optimization.py:595-630) -----------------------------------------------
int fdcap_opt_set_dct(fdcap_ctx* c, const float* dct_mtx, int32_t T, int32_t C, const float* c_dct_d, void* stream) {
    if (!c || !c->opt || !dct_mtx || !c_dct_d || T <= 0 || T > DCT_MAXT || C <= 0 || C > DCT_MAXC) return FDCAP_E_ARG;
    OptState* o = c->opt;
    const int W = o->cfg.n_total / T;
    if (W <= 0) return FDCAP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)W * 69 * C;
    HIP_TRY(o->dctD.upload(dct_mtx, (size_t)T * C));
    HIP_TRY(o->dctCoef.ensure(n));
    HIP_TRY(o->dctM.ensure(n));
    HIP_TRY(o->dctV.ensure(n));
    HIP_TRY(hipMemcpyAsync(o->dctCoef.p, c_dct_d, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemsetAsync(o->dctM.p, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(o->dctV.p, 0, n * sizeof(float), st));
    o->dctT = T; o->dctC = C; o->dctW = W;
    return FDCAP_OK;
}

int fdcap_opt_dct_fit(fdcap_ctx* c, int32_t iters, int32_t step0, float weight, float* obj_hist, int32_t log_stride,
                      void* stream) {
    if (!c || !c->opt || iters < 0 || step0 < 0 || (obj_hist && log_stride <= 0)) return FDCAP_E_ARG;
    { int es_ = opt_sync(c, (hipStream_t)stream); if (es_) return es_; }
    OptState* o = c->opt;
    if (o->dctW <= 0) return FDCAP_E_STATE;
    hipStream_t st = (hipStream_t)stream;
    const fdcap_opt_config& cf = o->cfg;
    const int T = o->dctT;
    // windows that lie completely inside this rank's frames (the caller shards on window boundaries)
    const int w0 = (cf.frame0 + T - 1) / T;
    const int w1 = std::min((cf.frame0 + cf.n_local) / T, o->dctW);
    if (iters == 0 || w1 <= w0) return FDCAP_OK;
    // weight 0: every gradient is exactly zero whatever the trajectories are (Adam coasts on its moments: the torch < 2
    // zero_grad semantics of a frozen c_dct) -- no forward needed
    if (weight != 0.f) {
        int row_lo, row_hi;
        opt_row_range(o, 1, &row_lo, &row_hi);
        int e = opt_pose_forward(c, row_lo, row_hi, st);
        if (e) return e;
    }
    o->adam_tab_h.resize(iters);
    for (int i = 0; i < iters; ++i) o->adam_tab_h[i] = adam_scalars(cf.lr, step0 + i + 1);
    HIP_TRY(o->adam_tab.ensure(iters));
    HIP_TRY(hipMemcpyAsync(o->adam_tab.p, o->adam_tab_h.data(), (size_t)iters * sizeof(AdamScalars), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(dct_fit_kernel, dim3((w1 - w0) * 69), dim3(64), 0, st, o->Jw.p, 2 + (w0 * T - cf.frame0), T, o->dctC,
                       o->dctD.p, o->dctCoef.p, o->dctM.p, o->dctV.p, w0, o->adam_tab.p, iters,
                       weight / (69.f * (float)o->dctW), obj_hist, log_stride > 0 ? log_stride : 1);
    return (int)hipGetLastError();
}

int fdcap_opt_backward_dct(fdcap_ctx* c, float w_dct, float w_rec, float w_contact, int32_t log_terms, void* stream) {
    if (!c || !c->opt) return FDCAP_E_STATE;
    if (c->opt->dctW <= 0) return FDCAP_E_STATE;
    LossWeights lw;
    lw.rec = w_rec; lw.smooth = 0.f; lw.contact = w_contact; lw.world = 0.f; lw.dct = w_dct; lw.world_on = false;
    return opt_backward_impl(c, lw, log_terms, (hipStream_t)stream);
}

int fdcap_opt_set_dct_coef(fdcap_ctx* c, const float* c_dct_d, void* stream) {
    if (!c || !c->opt || !c_dct_d) return FDCAP_E_ARG;
    OptState* o = c->opt;
    if (o->dctW <= 0) return FDCAP_E_STATE;
    HIP_TRY(hipMemcpyAsync(o->dctCoef.p, c_dct_d, (size_t)o->dctW * 69 * o->dctC * sizeof(float), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
    return FDCAP_OK;
}

int fdcap_opt_get_dct(fdcap_ctx* c, float* c_dct_d, void* stream) {
    if (!c || !c->opt || !c_dct_d) return FDCAP_E_ARG;
    OptState* o = c->opt;
    if (o->dctW <= 0) return FDCAP_E_STATE;
    HIP_TRY(hipMemcpyAsync(c_dct_d, o->dctCoef.p, (size_t)o->dctW * 69 * o->dctC * sizeof(float), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
    return FDCAP_OK;
}
// Adam's moments of c_dct, [W,69,C] each: what a checkpoint of mode 'dct' needs next to fdcap_opt_get_dct / fdcap_opt_export_state
static int dct_state_copy(fdcap_ctx* c, float* m_d, float* v_d, bool out, hipStream_t st) {
    if (!c || !c->opt || !m_d || !v_d) return FDCAP_E_ARG;
    OptState* o = c->opt;
    if (o->dctW <= 0) return FDCAP_E_STATE;
    const size_t bytes = (size_t)o->dctW * 69 * o->dctC * sizeof(float);
    HIP_TRY(hipMemcpyAsync(out ? m_d : o->dctM.p, out ? o->dctM.p : m_d, bytes, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(out ? v_d : o->dctV.p, out ? o->dctV.p : v_d, bytes, hipMemcpyDeviceToDevice, st));
    return FDCAP_OK;
}
int fdcap_opt_get_dct_state(fdcap_ctx* c, float* m_d, float* v_d, void* stream) { return dct_state_copy(c, m_d, v_d, true, (hipStream_t)stream); }
int fdcap_opt_set_dct_state(fdcap_ctx* c, const float* m_d, const float* v_d, void* stream) {
    return dct_state_copy(c, (float*)m_d, (float*)v_d, false, (hipStream_t)stream);
}
int32_t fdcap_opt_dct_windows(fdcap_ctx* c, int32_t* w0, int32_t* w1) {
    if (!c || !c->opt || c->opt->dctW <= 0) return 0;
    const fdcap_opt_config& cf = c->opt->cfg;
    const int T = c->opt->dctT;
    int a = (cf.frame0 + T - 1) / T, b = std::min((cf.frame0 + cf.n_local) / T, c->opt->dctW);
    if (w0) *w0 = a;
    if (w1) *w1 = std::max(a, b);
    return c->opt->dctW;
}

// ---- per-frame inner fit with a 2D reprojection term (outside the reference) ------
int fdcap_opt_set_keypoints(fdcap_ctx* c, const float* kp_d, void* stream) {
    if (!c || !c->opt || !kp_d) return FDCAP_E_ARG;
    OptState* o = c->opt;
    const size_t n = (size_t)o->cfg.n_local * NJW * 3;
    HIP_TRY(o->kp2d.ensure(n));
    HIP_TRY(hipMemcpyAsync(o->kp2d.p, kp_d, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return FDCAP_OK;
}

static int fit2d_eval(fdcap_ctx* c, const fdcap_fit2d_stage* sg, double* losses, float* floss, hipStream_t st, bool fold = true);
int fdcap_opt_backward_fit2d(fdcap_ctx* c, const fdcap_fit2d_stage* sg, int32_t log_terms, void* stream) {
    if (!c || !c->opt || !sg) return FDCAP_E_ARG;
    { int es_ = opt_sync(c, (hipStream_t)stream); if (es_) return es_; }
    OptState* o = c->opt;
    if (!o->kp2d.p) return FDCAP_E_STATE;
    hipStream_t st = (hipStream_t)stream;
    double* const losses = log_terms ? o->losses.p : nullptr;
    if (losses) HIP_TRY(hipMemsetAsync(losses, 0, FDCAP_NUM_LOSSES * sizeof(double), st));
    // (the latent gradient stays in the VPoser backward's four partials: fdcap_opt_step_x / fdcap_opt_get_grads add them)
    return fit2d_eval(c, sg, losses, nullptr, st, false);
}

// ---- batched L-BFGS (csrc/fdc_lbfgs.h) ------------------------------------------------------------------------------
static int lbfgs_cfg_ok(const fdcap_lbfgs_config* cf) {
    return cf && cf->dim > 0 && cf->dim <= LB_DPAD && cf->history > 0 && cf->history <= LB_HMAX && cf->max_iter > 0 && cf->max_steps > 0 &&
           cf->max_ls > 0 && cf->lr > 0.f;
}
int fdcap_lbfgs_create(int32_t n, const fdcap_lbfgs_config* cf, fdcap_lbfgs** out) {
    if (!out || n <= 0 || !lbfgs_cfg_ok(cf)) return FDCAP_E_ARG;
    { int nd = 0; if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return FDCAP_E_NODEVICE; }
    fdcap_lbfgs* L = new (std::nothrow) fdcap_lbfgs();
    if (!L) return FDCAP_E_ARG;
    L->n = n;
    L->cf = {cf->dim, cf->history, cf->max_iter, cf->max_eval > 0 ? cf->max_eval : cf->max_iter * 5 / 4, cf->max_steps, cf->max_ls,
             cf->lr, cf->tolerance_grad, cf->tolerance_change, cf->ftol, cf->gtol};
    hipError_t e = L->S.ensure(n);
    if (e == hipSuccess) e = L->W.ensure((size_t)n * lbfgs_ws_floats(cf->history));
    if (e == hipSuccess) e = L->RO.ensure((size_t)n * LB_HMAX);
    if (e == hipSuccess) e = L->active.ensure(2);
    if (e == hipSuccess) e = hipHostMalloc((void**)&L->active_h, sizeof(int), hipHostMallocDefault);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)lbfgs_advance_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)lbfgs_lds_bytes(LB_HMAX));
    if (e == hipSuccess) { int r = fdcap_lbfgs_reset(L, nullptr); if (r == 0) e = hipDeviceSynchronize(); else e = (hipError_t)r; }
    if (e != hipSuccess) { fdcap_lbfgs_destroy(L); return (int)e; }
    *out = L;
    return FDCAP_OK;
}
void fdcap_lbfgs_destroy(fdcap_lbfgs* L) {
    if (!L) return;
    L->S.release(); L->W.release(); L->RO.release(); L->active.release();
    if (L->active_h) (void)hipHostFree(L->active_h);
    delete L;
}
int fdcap_lbfgs_reset(fdcap_lbfgs* L, void* stream) {
    if (!L) return FDCAP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(L->S.p, 0, (size_t)L->n * sizeof(LbfgsScalars), st));      // phase 0 = LB_INIT
    HIP_TRY(hipMemsetAsync(L->W.p, 0, (size_t)L->n * lbfgs_ws_floats(L->cf.hist) * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(L->RO.p, 0, (size_t)L->n * LB_HMAX * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(L->active.p, 0, 2 * sizeof(int), st));
    L->round = 0;
    return FDCAP_OK;
}
static int lbfgs_advance_impl(fdcap_lbfgs* L, float* x, int32_t x_stride, const float* f, const float* g, int32_t g_stride, int32_t* n_active,
                              LbfgsFold fold, void* stream);
int fdcap_lbfgs_advance(fdcap_lbfgs* L, float* x, int32_t x_stride, const float* f, const float* g, int32_t g_stride, int32_t* n_active,
                        void* stream) {
    return lbfgs_advance_impl(L, x, x_stride, f, g, g_stride, n_active, LbfgsFold(), stream);
}
static int lbfgs_advance_impl(fdcap_lbfgs* L, float* x, int32_t x_stride, const float* f, const float* g, int32_t g_stride, int32_t* n_active,
                              LbfgsFold fold, void* stream) {
    if (!L || !x || !f || !g || x_stride < L->cf.dim || g_stride < L->cf.dim) return FDCAP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    int* const cnt = L->active.p + (L->round & 1);            // this round's counter was zeroed by the previous round's launch
    hipLaunchKernelGGL(lbfgs_advance_kernel, dim3(L->n), dim3(LB_NT), lbfgs_lds_bytes(L->cf.hist), st, L->cf, L->S.p, L->W.p, L->RO.p, x, x_stride,
                       f, g, g_stride, cnt, L->active.p + ((L->round + 1) & 1), fold);
    L->round++;
    if (n_active) HIP_TRY(hipMemcpyAsync(n_active, cnt, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    return (int)hipGetLastError();
}
int fdcap_lbfgs_finalize(fdcap_lbfgs* L, float* x, int32_t x_stride, int32_t* n_unfinished, void* stream) {
    if (!L || !x || x_stride < L->cf.dim) return FDCAP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n_unfinished) HIP_TRY(hipMemsetAsync(n_unfinished, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(lbfgs_finalize_kernel, dim3(L->n), dim3(LB_DPAD), 0, st, L->cf, L->S.p, L->W.p, x, x_stride, n_unfinished);
    return (int)hipGetLastError();
}
namespace {
__global__ void lbfgs_stats_kernel(const LbfgsScalars* __restrict__ S, int n, int* __restrict__ it, int* __restrict__ ev, float* __restrict__ loss) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (it) it[p] = S[p].n_iter_total;
    if (ev) ev[p] = S[p].evals_total;
    if (loss) loss[p] = S[p].loss;
}
}
int fdcap_lbfgs_get_stats(fdcap_lbfgs* L, int32_t* it, int32_t* ev, float* loss, void* stream) {
    if (!L) return FDCAP_E_ARG;
    hipLaunchKernelGGL(lbfgs_stats_kernel, dim3((L->n + 127) / 128), dim3(128), 0, (hipStream_t)stream, L->S.p, L->n, it, ev, loss);
    return (int)hipGetLastError();
}

// one evaluation of the inner fit's objective: forward, loss (+ per-frame value), backward
static int fit2d_eval(fdcap_ctx* c, const fdcap_fit2d_stage* sg, double* losses, float* floss, hipStream_t st, bool fold) {
    OptState* o = c->opt;
    const int nl = o->cfg.n_local;
    Fit2dStage s = {sg->fx, sg->fy, sg->cx, sg->cy, sg->rho, sg->w_data, sg->w_pose, sg->w_shape, sg->w_hand};
    PoseModel pm = c->pose_model();
    int row_lo, row_hi;
    opt_row_range(o, 1, &row_lo, &row_hi);
    int e = opt_pose_forward(c, row_lo, row_hi, st);
    if (e) return e;
    hipLaunchKernelGGL(fit2d_loss_kernel, dim3(nl), dim3(128), 0, st, s, o->X.p, o->Jw.p, o->kp2d.p, 2, o->dX.p, o->dJw.p, losses, floss);
    hipLaunchKernelGGL(pose_bwd_kernel, dim3(nl), dim3(64 * POSE_NW), 0, st, pm, o->X.p, o->O.p, o->CAM.p, o->scale.p, 2, o->Rm.p,
                       o->Jrest.p, o->G.p, (const float*)nullptr, (const float*)nullptr, o->dJw.p, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, 0, (const float*)nullptr, o->dX.p, o->dO.p, o->dCAM.p,
                       o->dscale_row.p, ParamLossIn(), (const float*)nullptr);
    { int eb = opt_vposer_backward(c, fold, st); if (eb) return eb; }
    return (int)hipGetLastError();
}

int fdcap_opt_fit2d_lbfgs(fdcap_ctx* c, const fdcap_fit2d_stage* sg, const fdcap_lbfgs_config* cfg, int32_t max_rounds, int32_t* rounds_out,
                          void* stream) {
    if (!c || !c->opt || !sg || !cfg || max_rounds <= 0) return FDCAP_E_ARG;
    { int es_ = opt_sync(c, (hipStream_t)stream); if (es_) return es_; }
    OptState* o = c->opt;
    if (!o->kp2d.p) return FDCAP_E_STATE;
    hipStream_t st = (hipStream_t)stream;
    const int nl = o->cfg.n_local;
    fdcap_lbfgs_config cf = *cfg;
    cf.dim = XDIM;
    if (!lbfgs_cfg_ok(&cf)) return FDCAP_E_ARG;
    if (o->lbfgs && (o->lbfgs->n != nl || o->lbfgs->cf.hist != cf.history)) { fdcap_lbfgs_destroy(o->lbfgs); o->lbfgs = nullptr; }
    if (!o->lbfgs) { int e = fdcap_lbfgs_create(nl, &cf, &o->lbfgs); if (e) return e; }
    fdcap_lbfgs* L = o->lbfgs;
    L->cf = {cf.dim, cf.history, cf.max_iter, cf.max_eval > 0 ? cf.max_eval : cf.max_iter * 5 / 4, cf.max_steps, cf.max_ls,
             cf.lr, cf.tolerance_grad, cf.tolerance_change, cf.ftol, cf.gtol};
    { int e = fdcap_lbfgs_reset(L, st); if (e) return e; }
    HIP_TRY(o->floss.ensure(nl));
    o->ahead = false; o->log_pending = false; o->log_dst = nullptr;
    int rounds = 0, e = 0;
    const int poll = 8;                                       // rounds between two looks at the number of frames still running
    while (rounds < max_rounds) {
        e = fit2d_eval(c, sg, nullptr, o->floss.p, st, false);   // (the latent gradient's four partials are folded by the advance kernel)
        if (e) break;
        LbfgsFold fold;
        fold.part = o->dZpart.p + (size_t)2 * VP_Z; fold.stride = (size_t)o->R * VP_Z; fold.col0 = X_LATENT; fold.n = VP_Z;
        e = lbfgs_advance_impl(L, o->X.p + 2 * XDIM, XDIM, o->floss.p, o->dX.p + 2 * XDIM, XDIM, nullptr, fold, st);
        if (e) break;
        ++rounds;
        if (rounds % poll == 0 || rounds == max_rounds) {
            HIP_TRY(hipMemcpyAsync(L->active_h, L->active.p + ((L->round - 1) & 1), sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (*L->active_h == 0) break;
        }
    }
    if (rounds_out) *rounds_out = rounds;
    // out of rounds with frames still inside a line search: their rows hold trial points -- roll them back to the accepted ones
    if (!e && rounds >= max_rounds && *L->active_h != 0) e = fdcap_lbfgs_finalize(L, o->X.p + 2 * XDIM, XDIM, nullptr, st);
    if (o->dz_pending) {                                      // leave dX complete, as every other backward of the API does
        hipLaunchKernelGGL(vposer_fold_dz_kernel, dim3((nl * VP_Z + 255) / 256), dim3(256), 0, st, o->dZpart.p, (size_t)o->R * VP_Z, 2, nl, o->dX.p);
        o->dz_pending = false;
    }
    if (e) return e;
    HIP_TRY(hipStreamSynchronize(st));
    return FDCAP_OK;
}

int fdcap_opt_fit2d_lbfgs_stats(fdcap_ctx* c, int32_t* it, int32_t* ev, float* loss, void* stream) {
    if (!c || !c->opt || !c->opt->lbfgs) return FDCAP_E_STATE;
    return fdcap_lbfgs_get_stats(c->opt->lbfgs, it, ev, loss, stream);
}

// zero Adam's moments of body_rotation_rec (SMPLify-X builds a fresh optimiser for every stage of the fit)
int fdcap_opt_reset_adam(fdcap_ctx* c, void* stream) {
    if (!c || !c->opt) return FDCAP_E_STATE;
    { int es_ = opt_sync(c, (hipStream_t)stream); if (es_) return es_; }
    OptState* o = c->opt;
    const size_t n = (size_t)o->R * XDIM * sizeof(float);
    o->log_pending = false; o->log_dst = nullptr;
    HIP_TRY(hipMemsetAsync(o->mX.p, 0, n, (hipStream_t)stream));
    HIP_TRY(hipMemsetAsync(o->vX.p, 0, n, (hipStream_t)stream));
    return FDCAP_OK;
}

// ---- checkpoint / resume of the optimiser state, finite check (the reference has neither) -----------
int32_t fdcap_opt_state_len(fdcap_ctx* c) {
    if (!c || !c->opt) return 0;
    return (int32_t)(2 * ((size_t)c->opt->cfg.n_local * (XDIM + 16)) + 2);
}
static int opt_state_copy(fdcap_ctx* c, float* state, bool to_state, hipStream_t st) {
    OptState* o = c->opt;
    const size_t nl = o->cfg.n_local, nx = nl * XDIM, ncam = nl * 16;
    struct Part { float* lib; size_t n; } parts[] = {{o->mX.p + 2 * XDIM, nx}, {o->vX.p + 2 * XDIM, nx}, {o->mCAM.p + 2 * 16, ncam},
                                                     {o->vCAM.p + 2 * 16, ncam}, {o->mS.p, 1}, {o->vS.p, 1}};
    size_t off = 0;
    for (const Part& p : parts) {
        HIP_TRY(hipMemcpyAsync(to_state ? state + off : p.lib, to_state ? p.lib : state + off, p.n * sizeof(float), hipMemcpyDeviceToDevice, st));
        off += p.n;
    }
    return FDCAP_OK;
}
int fdcap_opt_export_state(fdcap_ctx* c, float* state_d, void* stream) {
    if (!c || !c->opt || !state_d) return FDCAP_E_ARG;
    { int es_ = opt_sync(c, (hipStream_t)stream); if (es_) return es_; }
    return opt_state_copy(c, state_d, true, (hipStream_t)stream);
}
int fdcap_opt_import_state(fdcap_ctx* c, const float* state_d, void* stream) {
    if (!c || !c->opt || !state_d) return FDCAP_E_ARG;
    { int es_ = opt_sync(c, (hipStream_t)stream); if (es_) return es_; }
    c->opt->log_pending = false;
    c->opt->ahead = false;
    return opt_state_copy(c, (float*)state_d, false, (hipStream_t)stream);
}
int fdcap_opt_check_finite(fdcap_ctx* c, int32_t* count_d, void* stream) {
    if (!c || !c->opt || !count_d) return FDCAP_E_ARG;
    { int es_ = opt_sync(c, (hipStream_t)stream); if (es_) return es_; }
    OptState* o = c->opt;
    hipStream_t st = (hipStream_t)stream;
    const size_t nx = (size_t)o->cfg.n_local * XDIM, ncam = (size_t)o->cfg.n_local * 16;
    HIP_TRY(hipMemsetAsync(count_d, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(count_nonfinite_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, st, o->X.p + 2 * XDIM, nx, count_d);
    hipLaunchKernelGGL(count_nonfinite_kernel, dim3((unsigned)((ncam + 255) / 256)), dim3(256), 0, st, o->CAM.p + 2 * 16, ncam, count_d);
    hipLaunchKernelGGL(count_nonfinite_kernel, dim3(1), dim3(64), 0, st, o->scale.p, (size_t)1, count_d);
    return (int)hipGetLastError();
}

// ---- optimization.py: the per-frame smoother (:185-238, :334-348) ------------------------------
int fdcap_frame_smoother(fdcap_ctx* c, const float* data78, int32_t N, int32_t iters, float lr, float w_rec, float w_vposer,
                         float w_prev, float* state, int32_t step0, int32_t has_prev, float* out78, void* stream) {
    if (!data78 || !out78 || N <= 0 || iters <= 0 || step0 < 0 || ((step0 > 0 || has_prev) && !state)) return FDCAP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)N * iters;
    std::vector<AdamScalars> local_h;
    std::vector<AdamScalars>& tab_h = c ? c->ws_adam_h : local_h;
    tab_h.resize(n);
    for (size_t i = 0; i < n; ++i) tab_h[i] = adam_scalars(lr, step0 + (int)(i + 1));
    DevBuf<AdamScalars> local_d;
    DevBuf<AdamScalars>& tab_d = c ? c->ws_adam : local_d;      // ctx == NULL: a temporary, released after a stream sync
    HIP_TRY(tab_d.ensure(n));
    HIP_TRY(hipMemcpyAsync(tab_d.p, tab_h.data(), n * sizeof(AdamScalars), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(frame_smoother_kernel, dim3(1), dim3(128), 0, st, data78, N, iters, tab_d.p,
                       smoother_weights(w_rec, w_vposer, w_prev), state, (step0 > 0 || has_prev) ? 1 : 0, has_prev ? 1 : 0, out78);
    if (!c) {
        hipError_t e_ = hipStreamSynchronize(st);
        local_d.release();
        if (e_ != hipSuccess) return (int)e_;
    }
    return (int)hipGetLastError();
}

namespace {
int opt_step_launch(fdcap_ctx* c, int32_t ii, int32_t P, bool do_rows, bool do_scale, bool reduce_scale, hipStream_t st, float* xch) {
    OptState* o = c->opt;
    const int nl = o->cfg.n_local;
    if (do_rows) o->ahead = false;                         // the rows change: a forward that ran ahead of this step is stale
    const StepPlan sp = opt_step_plan(o, ii, P, do_rows, do_scale);
    const bool tail = sp.step_scale || reduce_scale;                 // the last block: (reduction +) scale (+ message tail)
    if (sp.nb_x + sp.nb_cam + (tail ? 1 : 0) + (o->log_pending ? 1 : 0) == 0) return FDCAP_OK;
    TraceRange tr_("fdcap:adam(K22)");
    const LogReduceIn lg = {o->loss_rows.p, o->log_dst, o->log_mask, o->log_assign, nl};
    hipLaunchKernelGGL(adam_step_kernel, dim3(sp.nb_x + sp.nb_cam + 1 + (o->log_pending ? 1 : 0)), dim3(256), 0, st, sp.x, sp.cam, sp.sc, sp.nb_x, sp.nb_cam,
                       o->dscale_row.p, 2, reduce_scale ? nl : 0, o->dscale.p, (sp.step_scale && ii >= P) ? 1 : 0, xch, nl, o->CAM.p,
                       (do_rows && o->dz_pending) ? (const float*)o->dZpart.p : (const float*)nullptr, (size_t)o->R * VP_Z, lg);
    o->log_pending = false;
    if (do_rows) o->dz_pending = false;
    return (int)hipGetLastError();
}
}  // namespace

static int opt_step_impl(fdcap_ctx* c, int32_t ii, int32_t P, bool do_rows, bool do_scale, bool reduce_scale, void* stream,
                         float* xch = nullptr) {
    if (!c || !c->opt) return FDCAP_E_STATE;
    int e = opt_sync(c, (hipStream_t)stream);              // (a deferred step nobody consumed comes first)
    if (e) return e;
    return opt_step_launch(c, ii, P, do_rows, do_scale, reduce_scale, (hipStream_t)stream, xch);
}

int fdcap_opt_sync(fdcap_ctx* c, void* stream) {
    if (!c || !c->opt) return FDCAP_E_STATE;
    return opt_sync(c, (hipStream_t)stream);
}

// ---- mode 'local' (global_optimization.py:499-556) --------------------------------------------
int fdcap_opt_detect_contact(fdcap_ctx* c, int32_t n_left, float* weight_left, void* stream) {
    if (!c || !c->opt || !weight_left || n_left <= 0 || n_left > c->nc) return FDCAP_E_ARG;
    OptState* o = c->opt;
    if (!o->contact_on) return FDCAP_E_STATE;
    hipStream_t st = (hipStream_t)stream;
    const int nl = o->cfg.n_local, nc = c->nc;
    int row_lo, row_hi;
    opt_row_range(o, 1, &row_lo, &row_hi);
    int e = opt_pose_forward(c, row_lo, row_hi, st);
    if (e) return e;
    e = opt_contact_forward(c, st);
    if (e) return e;
    hipLaunchKernelGGL(detect_contact_kernel, dim3(nl), dim3(256), 0, st, o->dist.p, c->contact_perm.p, nc, n_left, 2, weight_left);
    return (int)hipGetLastError();
}

int fdcap_opt_backward_local2(fdcap_ctx* c, const float* contact_weight, int32_t n_left, void* stream) {
    if (!c || !c->opt || !contact_weight || n_left <= 0 || n_left >= c->nc) return FDCAP_E_ARG;
    { int es_ = opt_sync(c, (hipStream_t)stream); if (es_) return es_; }
    OptState* o = c->opt;
    hipStream_t st = (hipStream_t)stream;
    const fdcap_opt_config& cf = o->cfg;
    const int R = o->R, nl = cf.n_local, nc = c->nc, N = cf.n_total, V = c->V;
    if (!c->full_ready) {
        std::vector<int64_t> all(V);
        for (int i = 0; i < V; ++i) all[i] = i;
        int e = build_skin_set(c, all, &c->full);
        if (e) return e;
        c->full_ready = true;
    }
    const size_t nv3 = (size_t)3 * V;
    HIP_TRY(o->VoffF.ensure((size_t)R * nv3));
    HIP_TRY(o->VwF.ensure((size_t)R * nv3));
    HIP_TRY(o->dVF.ensure((size_t)R * nv3));
    PoseModel pm = c->pose_model();
    HIP_TRY(hipMemsetAsync(o->losses.p, 0, FDCAP_NUM_LOSSES * sizeof(double), st));
    int row_lo, row_hi;
    opt_row_range(o, 2, &row_lo, &row_hi);
    int e = opt_pose_forward(c, row_lo, row_hi, st);
    if (e) return e;
    // full-mesh world vertices of every row (the vertex stencil needs 2 halo frames each side)
    HIP_TRY(blend_forward(c->full, o->PF.p, R, o->VoffF.p, st));
    hipLaunchKernelGGL(skin_fwd_kernel, dim3((V + 255) / 256, R), dim3(256), 0, st, c->full.model(), V, o->X.p, XDIM, X_BETAS,
                       X_TRANSL, o->VoffF.p, o->A.p, o->M.p, o->scale.p, 0, 1, o->VwF.p);
    // losses [0] rec, [1] z^2, [2] local (parameter) smoothing, [5] vertex smoothing, [6] foot skate
    const float w_rec = cf.weight_loss_rec / ((float)N * XDIM);
    const float w_sm = (N >= 3) ? 1.f / ((float)(N - 2) * XDIM) : 0.f;
    hipLaunchKernelGGL(param_loss_kernel, dim3(nl), dim3(128), 0, st, o->X.p, o->X0.p, o->mask.p, o->Jw.p, 2, cf.frame0, N,
                       w_rec, w_sm, 0.f, 0, o->dX.p, o->dJw.p, o->losses.p);
    const float w_vs = (N >= 3) ? 1.f / ((float)(N - 2) * (float)nv3) : 0.f;
    hipLaunchKernelGGL(vert_smooth_kernel, dim3((nv3 + 255) / 256, nl), dim3(256), 0, st, o->VwF.p, nv3, 2, cf.frame0, N, w_vs,
                       o->dVF.p, o->losses.p + 5);
    if (N >= 2)
        hipLaunchKernelGGL(foot_skate_kernel, dim3((nc * 3 + 255) / 256, nl), dim3(256), 0, st, o->VwF.p, nv3, c->contact_vid.p,
                           nc, n_left, contact_weight, 2, cf.frame0, N, o->dVF.p, o->losses.p + 6);
    hipLaunchKernelGGL(skin_bwd_kernel<false>, dim3(nl), dim3(256), (size_t)std::min(V, 1024) * 12 * sizeof(float), st, c->full.model(), V, o->X.p, o->VoffF.p, o->A.p, o->M.p,
                       o->scale.p, 2, o->dVF.p, o->dVF.p, o->dA.p, (float*)nullptr, o->dtransl_v.p, o->dMv.p, o->dsv.p, ContactGradIn());
    HIP_TRY(gemm_f32(true, EPI_STORE, o->dVF.p + 2 * nv3, 3 * V, c->full.posedirs.p, c->full.ldp, o->dPF.p + 2 * NPFX, NPFX, nl, NPFX,
                     3 * V, nullptr, 0, st));
    hipLaunchKernelGGL(pose_bwd_kernel, dim3(nl), dim3(64 * POSE_NW), 0, st, pm, o->X.p, o->O.p, o->CAM.p, o->scale.p, 2, o->Rm.p,
                       o->Jrest.p, o->G.p, o->dA.p, o->dPF.p, (const float*)nullptr, o->dMv.p, o->dsv.p, o->dPF.p + NPF, NPFX,
                       o->dtransl_v.p, o->dX.p, o->dO.p, o->dCAM.p, o->dscale_row.p, ParamLossIn(), (const float*)nullptr);
    { int eb = opt_vposer_backward(c, true, st); if (eb) return eb; }
    return (int)hipGetLastError();
}

int fdcap_opt_step_x(fdcap_ctx* c, int32_t step, void* stream) {
    if (!c || !c->opt || step <= 0) return FDCAP_E_ARG;
    { int es_ = opt_sync(c, (hipStream_t)stream); if (es_) return es_; }
    OptState* o = c->opt;
    o->ahead = false;
    const size_t nx = (size_t)o->cfg.n_local * XDIM;
    hipLaunchKernelGGL(adam_kernel, dim3((nx + 255) / 256), dim3(256), 0, (hipStream_t)stream, o->X.p + 2 * XDIM,
                       o->mX.p + 2 * XDIM, o->vX.p + 2 * XDIM, o->dX.p + 2 * XDIM, nx, adam_scalars(o->cfg.lr, step), 0,
                       o->dz_pending ? (const float*)o->dZpart.p : (const float*)nullptr, (size_t)o->R * VP_Z, 2);
    o->dz_pending = false;                                   // (consumed; dX itself stays without the partials: fdcap_opt_get_grads reads before the step)
    return (int)hipGetLastError();
}

int fdcap_opt_get_results(fdcap_ctx* c, float* body75, float* scale, float* cam, void* stream) {
    if (!c || !c->opt) return FDCAP_E_STATE;
    { int es_ = opt_sync(c, (hipStream_t)stream); if (es_) return es_; }
    OptState* o = c->opt;
    hipStream_t st = (hipStream_t)stream;
    const int nl = o->cfg.n_local;
    if (body75) hipLaunchKernelGGL(p78_to_75_kernel, dim3((nl + 127) / 128), dim3(128), 0, st, o->X.p + 2 * XDIM, nl, body75);
    if (scale) HIP_TRY(hipMemcpyAsync(scale, o->scale.p, sizeof(float), hipMemcpyDeviceToDevice, st));
    if (cam) HIP_TRY(hipMemcpyAsync(cam, o->CAM.p + 2 * 16, (size_t)nl * 16 * sizeof(float), hipMemcpyDeviceToDevice, st));
    return (int)hipGetLastError();
}

int fdcap_opt_forward_world(fdcap_ctx* c, float* verts, float* joints, void* stream) {
    if (!c || !c->opt) return FDCAP_E_STATE;
    OptState* o = c->opt;
    hipStream_t st = (hipStream_t)stream;
    const int nl = o->cfg.n_local, nc = c->nc;
    int row_lo, row_hi;
    opt_row_range(o, 1, &row_lo, &row_hi);
    int e = opt_pose_forward(c, row_lo, row_hi, st);
    if (e) return e;
    if (verts) {
        if (!o->contact_on) return FDCAP_E_STATE;
        e = opt_contact_forward(c, st);
        if (e) return e;
        hipLaunchKernelGGL(unpermute_kernel<float>, dim3(((size_t)nl * nc * 3 + 255) / 256), dim3(256), 0, st,
                           o->Vw.p + (size_t)2 * nc * 3, c->contact_perm.p, nl, nc, 3, verts);
    }
    if (joints)
        HIP_TRY(hipMemcpyAsync(joints, o->Jw.p + 2 * NJW * 3, (size_t)nl * NJW * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
    return (int)hipGetLastError();
}

int fdcap_opt_step(fdcap_ctx* c, int32_t ii, int32_t P, void* stream) { return opt_step_impl(c, ii, P, true, true, true, stream); }

// Multi-GPU iteration tail with ONE collective: Adam on this rank's rows, pack [boundary rows | dscale],
// (caller all-gathers), unpack halos + rank-ordered dscale sum + Adam on scale.
int fdcap_opt_step_rows_and_pack(fdcap_ctx* c, int32_t ii, int32_t P, float* send, void* stream) {
    if (!c || !c->opt || !send) return FDCAP_E_ARG;
    return opt_step_impl(c, ii, P, true, false, true, stream, send);     // Adam on the rows + the message, one launch
}
int fdcap_opt_unpack_and_step_scale(fdcap_ctx* c, int32_t ii, int32_t P, const float* gathered, int32_t rank, int32_t world,
                                    void* stream) {
    if (!c || !c->opt || !gathered || world <= 0 || rank < 0 || rank >= world) return FDCAP_E_ARG;
    { int es_ = opt_sync(c, (hipStream_t)stream); if (es_) return es_; }
    OptState* o = c->opt;
    const fdcap_opt_config& cf = o->cfg;
    // scale: same rule as opt_step_impl (receives a gradient while ii < P, if a term that reaches it exists)
    AdamTensor sc = {};
    const bool step_scale = (o->contact_on || o->dct_grad) && (ii < P || cf.legacy_zero_grad);
    if (step_scale) sc = AdamTensor{o->scale.p, o->mS.p, o->vS.p, o->dscale.p, 1, adam_scalars(cf.lr, ii + 1)};
    hipLaunchKernelGGL(unpack_exchange_kernel, dim3(1), dim3(384), 0, (hipStream_t)stream, gathered, rank, world, cf.n_local,
                       o->X.p, o->CAM.p, o->dscale.p, sc, (step_scale && ii >= P) ? 1 : 0);
    return (int)hipGetLastError();
}
int32_t fdcap_exchange_len(void) { return XCH_LEN; }

// ---- the exchange inside the library (SURVEY 8b "halo_exchange", 8e): RCCL on the compute stream ----------------
namespace {
__global__ void pack_exchange_kernel(const float* __restrict__ X, const float* __restrict__ CAM, int n_local, float* __restrict__ xch) {
    const int t = threadIdx.x;                                          // boundary rows as they are (no step): slots 0,1 first two, 2,3 last two owned rows
    if (t < 4 * XCH_ROW) {
        const int slot = t / XCH_ROW, e = t % XCH_ROW;
        const int row = slot < 2 ? 2 + slot : n_local + slot - 2;
        xch[t] = e < XDIM ? X[(size_t)row * XDIM + e] : CAM[(size_t)row * 16 + e - XDIM];
    } else if (t < XCH_LEN) xch[t] = 0.f;
}
int comm_fail(fdcap_ctx* c, ncclResult_t r, const char* what) {
    c->comm_err = std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
    return FDCAP_E_COMM;
}
int comm_buffers(fdcap_ctx* c) {
    HIP_TRY(c->xch_send.ensure(XCH_LEN));
    HIP_TRY(c->xch_all.ensure((size_t)c->comm.world * XCH_LEN));
    return 0;
}
}  // namespace

int fdcap_comm_unique_id(uint8_t* id128) {
    if (!id128) return FDCAP_E_ARG;
    static_assert(sizeof(ncclUniqueId) == FDCAP_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!rccl().load()) return FDCAP_E_COMM;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return FDCAP_E_COMM;
    memcpy(id128, &id, sizeof(id));
    return FDCAP_OK;
}

int fdcap_comm_create(fdcap_ctx* c, const uint8_t* id128, int32_t rank, int32_t world) {
    if (!c || !id128 || world <= 0 || rank < 0 || rank >= world) return FDCAP_E_ARG;
    if (c->comm.comm) return FDCAP_E_STATE;
    if (!rccl().load()) { c->comm_err = rccl().err; return FDCAP_E_COMM; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = rccl().CommInitRank(&comm, world, id, rank);          // (on the calling thread's current HIP device)
    if (r != ncclSuccess) return comm_fail(c, r, "ncclCommInitRank");
    c->comm.comm = comm; c->comm.rank = rank; c->comm.world = world;
    return FDCAP_OK;
}

int fdcap_comm_destroy(fdcap_ctx* c) {
    if (!c) return FDCAP_E_ARG;
    if (c->comm.comm) { (void)rccl().CommDestroy(c->comm.comm); c->comm = Comm(); }
    return FDCAP_OK;
}

// (a NULL context, or one without a message of its own, reports the loader's: fdcap_comm_unique_id has no context to write to)
const char* fdcap_comm_last_error(fdcap_ctx* c) { return c && !c->comm_err.empty() ? c->comm_err.c_str() : rccl().err.c_str(); }

// Fill the halo rows from the neighbouring ranks (before the first iteration, after fdcap_opt_import_state, after each
// iteration of mode 'local''s second loop): boundary rows as they are -> all-gather -> unpack, three enqueues on `stream`.
int fdcap_opt_halo_exchange(fdcap_ctx* c, void* stream) {
    if (!c || !c->opt) return FDCAP_E_STATE;
    { int es_ = opt_sync(c, (hipStream_t)stream); if (es_) return es_; }
    if (!c->comm.comm) return FDCAP_E_STATE;
    OptState* o = c->opt;
    hipStream_t st = (hipStream_t)stream;
    int e = comm_buffers(c);
    if (e) return e;
    hipLaunchKernelGGL(pack_exchange_kernel, dim3(1), dim3(384), 0, st, o->X.p, o->CAM.p, o->cfg.n_local, c->xch_send.p);
    const ncclResult_t r = rccl().AllGather(c->xch_send.p, c->xch_all.p, XCH_LEN, ncclFloat, c->comm.comm, st);
    if (r != ncclSuccess) return comm_fail(c, r, "ncclAllGather");
    hipLaunchKernelGGL(unpack_exchange_kernel, dim3(1), dim3(384), 0, st, c->xch_all.p, c->comm.rank, c->comm.world, o->cfg.n_local,
                       o->X.p, o->CAM.p, (float*)nullptr, AdamTensor{}, 0);
    return (int)hipGetLastError();
}

// The sharded iteration tail, whole: Adam on this rank's rows + message -> ONE ncclAllGather on `stream` -> halo rows, the
// rank-ordered sum of the scale-gradient partials, Adam on `scale`.  Replaces the caller-side sequence
// fdcap_opt_step_rows_and_pack / all-gather / fdcap_opt_unpack_and_step_scale (same kernels, same bits).
int fdcap_opt_exchange(fdcap_ctx* c, int32_t ii, int32_t P, void* stream) {
    if (!c || !c->opt) return FDCAP_E_STATE;
    if (!c->comm.comm) return FDCAP_E_STATE;
    int e = comm_buffers(c);
    if (e) return e;
    e = fdcap_opt_step_rows_and_pack(c, ii, P, c->xch_send.p, stream);
    if (e) return e;
    const ncclResult_t r = rccl().AllGather(c->xch_send.p, c->xch_all.p, XCH_LEN, ncclFloat, c->comm.comm, (hipStream_t)stream);
    if (r != ncclSuccess) return comm_fail(c, r, "ncclAllGather");
    return fdcap_opt_unpack_and_step_scale(c, ii, P, c->xch_all.p, c->comm.rank, c->comm.world, stream);
}

// The loop :560-593 itself, iterations [ii0, ii1) of a fit of num_iter, in ONE call (r4): what FittingOP.fitting's Python `for` issues --
// every iteration but the fit's last as fdcap_opt_backward_and_step, the last as fdcap_opt_backward + fdcap_opt_step; a sharded
// context (which must hold a communicator) as fdcap_opt_backward + fdcap_opt_exchange.  Logging iterations (log_every > 0:
// ii % log_every == 0, and the fit's last) write their partial sums to consecutive rows of hist_d [hist_rows][FDCAP_NUM_LOSSES]
// (device memory, filled without a host sync; *n_logged rows used).  flags bit 0: every optimiser step as its own launch;
// bit 1: the exchange tail even though the context holds the whole clip (a one-rank group: tests, probes).
// Nothing here waits for the device: the call returns when the launches are enqueued.
int fdcap_opt_run(fdcap_ctx* c, int32_t ii0, int32_t ii1, int32_t num_iter, int32_t P, int32_t log_every, double* hist_d,
                  int32_t hist_rows, int32_t flags, int32_t* n_logged, void* stream) {
    if (n_logged) *n_logged = 0;
    if (!c || !c->opt) return FDCAP_E_STATE;
    if (ii0 < 0 || ii1 < ii0 || ii1 > num_iter || log_every < 0) return FDCAP_E_ARG;
    OptState* o = c->opt;
    const fdcap_opt_config& cf = o->cfg;
    const bool sharded = (flags & 2) != 0 || !(cf.frame0 == 0 && cf.n_local == cf.n_total);
    if (sharded && !c->comm.comm) return FDCAP_E_STATE;
    double* const keep = o->losses.p;
    int k = 0, e = 0;
    for (int ii = ii0; ii < ii1 && !e; ++ii) {
        const bool do_log = log_every > 0 && (ii % log_every == 0 || ii == num_iter - 1);
        if (do_log) {
            if (!hist_d || k >= hist_rows) { e = FDCAP_E_ARG; break; }      // (a stretch without logging iterations needs no history)
            e = fdcap_opt_set_loss_output(c, hist_d + (size_t)k * FDCAP_NUM_LOSSES);
            if (e) break;
            ++k;
        }
        const int lt = do_log ? 2 : 0;
        if (sharded) {
            e = fdcap_opt_backward(c, ii, P, lt, stream);
            if (!e) e = fdcap_opt_exchange(c, ii, P, stream);
        } else if (!(flags & 1) && ii + 1 < num_iter) {
            e = fdcap_opt_backward_and_step(c, ii, P, lt, stream);
        } else {
            e = fdcap_opt_backward(c, ii, P, lt, stream);
            if (!e) e = fdcap_opt_step(c, ii, P, stream);
        }
    }
    if (k) {                                            // (never leave the library pointing into the caller's history)
        const int e2 = fdcap_opt_set_loss_output(c, keep);
        if (!e) e = e2;
    }
    if (n_logged) *n_logged = k;
    return e;
}

// Sum of n doubles over the ranks, in place (the logged loss partial sums; d loss / d scale never travels this way).
int fdcap_comm_allreduce_f64(fdcap_ctx* c, double* buf_d, int32_t n, void* stream) {
    if (!c || !buf_d || n <= 0) return FDCAP_E_ARG;
    if (!c->comm.comm) return FDCAP_E_STATE;
    const ncclResult_t r = rccl().AllReduce(buf_d, buf_d, (size_t)n, ncclDouble, ncclSum, c->comm.comm, (hipStream_t)stream);
    if (r != ncclSuccess) return comm_fail(c, r, "ncclAllReduce");
    return FDCAP_OK;
}

int fdcap_opt_get_contact(fdcap_ctx* c, float* dist, int32_t* idx, void* stream) {
    if (!c || !c->opt) return FDCAP_E_STATE;
    OptState* o = c->opt;
    if (!o->contact_on) return FDCAP_E_STATE;
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)o->cfg.n_local * c->nc;
    const int nl = o->cfg.n_local, nc = c->nc;
    if (dist) hipLaunchKernelGGL(unpermute_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, st, o->dist.p + 2 * nc,
                                 c->contact_perm.p, nl, nc, 1, dist);
    if (idx) hipLaunchKernelGGL(unpermute_kernel<int>, dim3((n + 255) / 256), dim3(256), 0, st, o->idx.p + 2 * nc,
                                c->contact_perm.p, nl, nc, 1, idx);
    return (int)hipGetLastError();
}

int fdcap_opt_get_grads(fdcap_ctx* c, float* dx, float* dcam, void* stream) {
    if (!c || !c->opt) return FDCAP_E_STATE;
    { int es_ = opt_sync(c, (hipStream_t)stream); if (es_) return es_; }
    OptState* o = c->opt;
    hipStream_t st = (hipStream_t)stream;
    const int nl = o->cfg.n_local;
    if (o->dz_pending) {                               // the latent gradient still sits in the four partials: fold it into dX once
        hipLaunchKernelGGL(vposer_fold_dz_kernel, dim3((nl * VP_Z + 255) / 256), dim3(256), 0, st, o->dZpart.p, (size_t)o->R * VP_Z, 2, nl, o->dX.p);
        o->dz_pending = false;
    }
    if (dx) HIP_TRY(hipMemcpyAsync(dx, o->dX.p + 2 * XDIM, (size_t)nl * XDIM * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (dcam) HIP_TRY(hipMemcpyAsync(dcam, o->dCAM.p + 2 * 16, (size_t)nl * 16 * sizeof(float), hipMemcpyDeviceToDevice, st));
    return FDCAP_OK;
}

int fdcap_panel_gemm(const float* A, int32_t lda, int32_t M, int32_t K, const float* B_h, int64_t sk, int64_t sn, int32_t N, float* C,
                     int32_t ldc, void* stream) {
    if (!A || !B_h || !C || M <= 0 || K <= 0 || N <= 0 || lda < K || ldc < N) return FDCAP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    {
        const char* e3 = getenv("FDCAP_GEMM_SPLIT3");                // (read per call here, so a test can run both forms in one process)
        if (!(e3 && e3[0] == '0') && panel_gemm3_fits(K)) {          // the three-way bf16 split form of the same product (the default)
            std::vector<unsigned> p3;
            PanelB3 B3;
            panel_pack3(B_h, (long)sk, (long)sn, K, N, p3, &B3.ntile, &B3.nst);
            DevBuf<unsigned> d3;
            HIP_TRY(d3.upload(p3.data(), p3.size()));
            B3.f = (const uint4*)d3.p;
            hipError_t e = panel_gemm3(A, lda, M, K, B3, C, ldc, N, st);
            hipError_t e2 = hipStreamSynchronize(st);
            d3.release();
            return (int)(e != hipSuccess ? e : e2);
        }
    }
    std::vector<float> pf;
    PanelB B;
    panel_pack(B_h, (long)sk, (long)sn, K, N, pf, &B.ntile, &B.nss);
    DevBuf<float> d;
    HIP_TRY(d.upload(pf.data(), pf.size()));
    B.f = (const float4*)d.p;
    hipError_t e = panel_gemm(A, lda, M, K, B, C, ldc, N, st);
    hipError_t e2 = hipStreamSynchronize(st);
    d.release();
    return (int)(e != hipSuccess ? e : e2);
}

int fdcap_opt_nn_timing(fdcap_ctx* c, int32_t max_launches) {
    if (!c || !c->opt || max_launches < 0) return FDCAP_E_ARG;
    OptState* o = c->opt;
    o->nn_timing = max_launches > 0;
    o->nn_ev_used = 0;
    while ((int)o->nn_ev.size() < 2 * max_launches) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        o->nn_ev.push_back(e);
    }
    return FDCAP_OK;
}
int fdcap_opt_nn_timing_read(fdcap_ctx* c, float* mean_ms, int32_t* launches) {
    if (!c || !c->opt || !mean_ms || !launches) return FDCAP_E_ARG;
    OptState* o = c->opt;
    double sum = 0.0;
    for (int i = 0; i + 1 < o->nn_ev_used; i += 2) {
        HIP_TRY(hipEventSynchronize(o->nn_ev[i + 1]));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, o->nn_ev[i], o->nn_ev[i + 1]));
        sum += t;
    }
    *launches = o->nn_ev_used / 2;
    *mean_ms = *launches ? (float)(sum / *launches) : 0.f;
    return FDCAP_OK;
}

int fdcap_time_blend_gemm(fdcap_ctx* c, int32_t rows, int32_t iters, float* ms, void* stream) {
    if (!c || rows <= 0 || iters <= 0 || !ms) return FDCAP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (!c->full_ready) {
        std::vector<int64_t> all(c->V);
        for (int i = 0; i < c->V; ++i) all[i] = i;
        int e = build_skin_set(c, all, &c->full);
        if (e) return e;
        c->full_ready = true;
    }
    const int V = c->V;
    HIP_TRY(c->ws_f[6].ensure((size_t)rows * NPFX));
    HIP_TRY(c->ws_f[11].ensure((size_t)rows * 3 * V));
    HIP_TRY(hipMemsetAsync(c->ws_f[6].p, 0x3c, (size_t)rows * NPFX * sizeof(float), st));   // arbitrary finite pattern
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(blend_forward(c->full, c->ws_f[6].p, rows, c->ws_f[11].p, st));
    HIP_TRY(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i)
        HIP_TRY(blend_forward(c->full, c->ws_f[6].p, rows, c->ws_f[11].p, st));
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, e0, e1));
    *ms = t / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return FDCAP_OK;
}

int fdcap_opt_time_chamfer(fdcap_ctx* c, int32_t iters, int32_t brute_force, float* ms, void* stream) {
    if (!c || !c->opt || !ms || iters <= 0) return FDCAP_E_ARG;
    { int es_ = opt_sync(c, (hipStream_t)stream); if (es_) return es_; }
    OptState* o = c->opt;
    if (!o->contact_on) return FDCAP_E_STATE;
    hipStream_t st = (hipStream_t)stream;
    const int nl = o->cfg.n_local, nc = c->nc;
    const size_t off = (size_t)2 * nc * 3;
    // brute_force: every (query, scene point) pair is visited (no seed, no chunk bounds);
    // otherwise the launch is exactly what the loop issues in steady state
    const int* seed = (!brute_force && o->use_seed) ? o->idx.p + 2 * nc : nullptr;
    NNTarget T = c->nn_target(!brute_force && o->use_cull);
    if (brute_force) { T.pts = c->scene.p; T.inv_perm = nullptr; T.frags = nullptr; }   // input order (a spatial sort is adversarial for an unseeded running minimum)
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const int nsp = brute_force ? o->nsplit_bf : o->nsplit;
    float4* sp = brute_force ? nullptr : o->seedpt.p + 2 * nc;
    // (warm-up launch; after a brute-force launch rewrote idx it also refreshes the neighbours' coordinates)
    const NNCache cache = o->nn_cache(0);
    const NNCache* cp = !brute_force ? &cache : nullptr;
    NNOrder* const op = !brute_force ? &o->nn_order : nullptr;
    HIP_TRY(nn_search(o->Vw.p + off, nl * nc, T, o->dist.p + 2 * nc, o->idx.p + 2 * nc, o->pd.p, o->pi.p, nsp, st, seed,
                      !brute_force && !o->seeded, sp, nullptr, cp, op));
    if (!brute_force) o->seeded = true;
    HIP_TRY(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i)
        HIP_TRY(nn_search(o->Vw.p + off, nl * nc, T, o->dist.p + 2 * nc, o->idx.p + 2 * nc, o->pd.p, o->pi.p, nsp, st, seed, false, sp, nullptr, cp, op));
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, e0, e1));
    *ms = t / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (brute_force) o->seeded = false;      // idx was rewritten without the neighbours' coordinates: refresh them before the next seeded launch
    return FDCAP_OK;
}

}  // extern "C"
