// Mode 'dct'  and the per-frame smoother of optimization.py.
//
//  * cal_dctloss (/root/reference/global_optimization.py:232-246): one objective per (window k, joint i,
//    axis j) -- the window's 60-frame world-joint trajectory against dct_mtx[60,5] @ c_dct[k,i,j], residual
//    e/(e+1) with e = (trajectory - prediction)^2, summed over the window; mean over objectives.
//  * fitting(mode='dct') first phase (:597-613): body / scale / camera_ext are frozen, so the trajectories
//    are constants and the 0.95*num_iter Adam iterations on c_dct are 23*3*W independent 5-parameter
//    problems: one wavefront per trajectory (lane = frame), coefficients + Adam moments in registers, the
//    whole phase in ONE launch.
//  * optimization.py fitting / fitting_smoothing (:185-238): every element of the 78-d row is an independent
//    scalar problem (L1 data term, L2 on the latent, L1 to the previous frame's result on columns 9:51), but
//    frames are sequential (each needs its predecessor's result and torch's Adam state is carried over
//    from frame to frame): one thread per element, the whole clip in ONE launch.
#pragma once
#include "fdc_loss.h"
#include "fdc_math.h"

namespace fdc {

constexpr int DCT_MAXC = 8;       // coefficients per trajectory (reference: DCT_NUM = 5, :43)
constexpr int DCT_MAXT = 64;      // frames per window = lanes of a wavefront (reference: 60, :41)

// d/d prediction of e/(e+1), e = (traj - pred)^2; *obj = e/(e+1)
FDC_HD float dct_residual(float traj, float pred, float* obj) {
    float r = traj - pred;
    float e = r * r;
    float den = e + 1.f;
    *obj = e / den;
    return -2.f * r / (den * den);
}

// ---- per-frame smoother (optimization.py:155-183) ----------------------------------------------------
struct SmootherWeights { float rec_over_cnt, vposer2_over_cnt, prev_over_cnt; };
// loss = w_rec*mean_78|xdata - x| + w_vp*mean_32(z^2) [+ w_prev*mean_42|prev[9:51] - x[9:51]|]  (:197, :227)
FDC_HD SmootherWeights smoother_weights(float w_rec, float w_vposer, float w_prev) {
    SmootherWeights w;
    w.rec_over_cnt = w_rec / 78.f;
    w.vposer2_over_cnt = 2.f * w_vposer / 32.f;
    w.prev_over_cnt = w_prev / 42.f;
    return w;
}
// gradient of element e of the row
FDC_HD float smoother_grad(int e, float x, float xdata, float xprev, bool has_prev, const SmootherWeights& w) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    float g = -sgn(xdata - x) * w.rec_over_cnt;                         // F.l1_loss(xhr, xhr_rec) (:157)
    if (e >= 19 && e < 51) g += w.vposer2_over_cnt * x;                 // mean(z^2), z = row[19:51] (:161-162)
    if (has_prev && e >= 9 && e < 51) g += -sgn(xprev - x) * w.prev_over_cnt;   // F.l1_loss(prev[:,9:51], rec[:,9:51]) (:182)
    return g;
}

#if defined(__HIPCC__)

__device__ __forceinline__ float dct_wave_sum(float v) { return wave_sum64(v); }

// One wavefront per trajectory.  Jw [rows, 69] world joints (row-major joint, axis); trajectory `traj` of
// local window wl = frames jw_row0 + wl*T .. +T, column ij.  coef/m/v [W_total, 69, C]; this launch covers
// windows w0 .. w0 + gridDim.x/69.  tab[it] = Adam scalars of iteration it.  obj_hist (optional)
// [ceil(iters/log_stride), gridDim.x]: the trajectory's objective BEFORE the update of iterations 0, s, 2s...
__global__ __launch_bounds__(64) void dct_fit_kernel(const float* __restrict__ Jw, int jw_row0, int T, int C,
                                                     const float* __restrict__ D, float* __restrict__ coef,
                                                     float* __restrict__ m, float* __restrict__ v, int w0,
                                                     const AdamScalars* __restrict__ tab, int iters, float w_over_cnt,
                                                     float* __restrict__ obj_hist, int log_stride) {
    const int traj = blockIdx.x, wl = traj / 69, ij = traj % 69, lane = threadIdx.x;
    const bool act = lane < T;
    const float t = act ? Jw[(size_t)(jw_row0 + wl * T + lane) * 69 + ij] : 0.f;
    // Lane c OWNS coefficient c: its value and Adam moments live in that lane only, and one adam_update per iteration serves all C
    // coefficients at once (r5, late: every lane used to carry all C and repeat all C updates -- ~45 instructions each with their
    // exact division and square root -- 1800 instructions per iteration of this one-wave chain, 4.8 us; the coefficient values the
    // prediction needs are read from their lanes).  Same operations on the same operands: bit-identical results.
    static_assert(DCT_MAXC <= 64, "a lane per coefficient");
    float Dl[DCT_MAXC];
    const size_t off = ((size_t)(w0 + wl) * 69 + ij) * C;
#pragma unroll
    for (int c = 0; c < DCT_MAXC; ++c) Dl[c] = (c < C && act) ? D[lane * C + c] : 0.f;
    const bool own = lane < C;
    float cc = own ? coef[off + lane] : 0.f, mm = own ? m[off + lane] : 0.f, vv = own ? v[off + lane] : 0.f;
    AdamScalars a_next = tab[0];
    for (int it = 0; it < iters; ++it) {
        const AdamScalars a = a_next;
        a_next = tab[min(it + 1, iters - 1)];
        float p = 0.f;
#pragma unroll
        for (int c = 0; c < DCT_MAXC; ++c) p += Dl[c] * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cc), c));
        float obj;
        float gp = dct_residual(t, p, &obj) * w_over_cnt;
        if (!act) { gp = 0.f; obj = 0.f; }
        if (obj_hist && it % log_stride == 0) {
            float s = dct_wave_sum(obj);
            if (lane == 0) obj_hist[(size_t)(it / log_stride) * gridDim.x + traj] = s;
        }
        float g = 0.f;
#pragma unroll
        for (int c = 0; c < DCT_MAXC; ++c) {
            if (c < C) {
                const float gc = dct_wave_sum(Dl[c] * gp);
                if (lane == c) g = gc;
            }
        }
        if (own) adam_update(cc, mm, vv, g, a);
    }
    if (own) { coef[off + lane] = cc; m[off + lane] = mm; v[off + lane] = vv; }
}

// d (w * loss_dct) / d Jw for the owned rows (written, not accumulated; rows outside every window get 0)
// and this rank's un-normalised objective sum.  thread per (owned row, column).
__global__ void dct_joint_grad_kernel(const float* __restrict__ Jw, int row0, int frame0, int n_local, int T, int C, int W,
                                      const float* __restrict__ D, const float* __restrict__ coef, float w_over_cnt,
                                      int add, float* __restrict__ dJw, double* __restrict__ obj_sum) {
    __shared__ float sred[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float obj = 0.f;
    if (i < n_local * 69) {
        const int rl = i / 69, ij = i % 69, g = frame0 + rl, k = g / T, f = g % T;
        float grad = 0.f;
        if (k < W) {
            const float* c = coef + ((size_t)k * 69 + ij) * C;
            float p = 0.f;
            for (int q = 0; q < C; ++q) p += D[f * C + q] * c[q];
            grad = -dct_residual(Jw[(size_t)(row0 + rl) * 69 + ij], p, &obj) * w_over_cnt;   // d/d traj = -d/d pred
        }
        float* o = dJw + (size_t)(row0 + rl) * 69 + ij;
        *o = add ? *o + grad : grad;
    }
    obj = dct_wave_sum(obj);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = obj;
    __syncthreads();
    if (threadIdx.x == 0 && obj_sum) atomicAdd(obj_sum, (double)((sred[0] + sred[1]) + (sred[2] + sred[3])));
}

// optimization.py's per-frame loop (:334-348) for a whole clip: block of 128 threads, thread = element.
// data78 [N,78] (6D rows of the SMPLify-X fits), out78 [N,78]; tab [N*iters] Adam scalars (torch's optimiser
// is created once, so the step counter and both moments run on across frames, :126).  state (optional)
// [3,78] = Adam m, v and the previous frame's result: read when `resume`, written at the end, so a caller
// can also go file by file like the reference's driver loop.
__global__ __launch_bounds__(128) void frame_smoother_kernel(const float* __restrict__ data78, int N, int iters,
                                                             const AdamScalars* __restrict__ tab, SmootherWeights w,
                                                             float* __restrict__ state, int resume, int has_prev,
                                                             float* __restrict__ out78) {
    const int e = threadIdx.x;
    if (e >= 78) return;
    float m = 0.f, v = 0.f, prev = 0.f;
    if (state && resume) { m = state[e]; v = state[78 + e]; prev = state[156 + e]; }
    for (int f = 0; f < N; ++f) {
        const float xd = data78[(size_t)f * 78 + e];
        float x = xd;                                                    // self.xhr_rec.data = xhr.clone() (:192, :220)
        const AdamScalars* a = tab + (size_t)f * iters;
        const bool hp = f > 0 || has_prev;
        for (int it = 0; it < iters; ++it) adam_update(x, m, v, smoother_grad(e, x, xd, prev, hp, w), a[it]);
        out78[(size_t)f * 78 + e] = x;
        prev = x;                                                        // xh_prev = xh_rec.detach() (:341, :348)
    }
    if (state) { state[e] = m; state[78 + e] = v; state[156 + e] = prev; }
}

#endif  // __HIPCC__

}  // namespace fdc
