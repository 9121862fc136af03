// Data / temporal losses on the raw 78-d rows and on world joints, with their gradients, and
// the Adam update.  Restates /root/reference/global_optimization.py:255-259 (loss_rec),
// :266-267 (loss_smoothing), :304 (loss_world_smoothing), :262-263 (loss_vposer, logged only)
// and torch.optim.Adam's single-tensor update (:188, :592).
#pragma once
#include "fdc_math.h"

namespace fdc {

FDC_HD float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

// second difference exactly as the reference groups it: (x_i - x_{i+1}) - (x_{i+1} - x_{i+2})
FDC_HD float second_diff(float a, float b, float c) { return (a - b) - (b - c); }

// One element of one frame.  xm2..xp2 are x[g-2..g+2][e] (values outside the clip are ignored
// through the range tests on g).  Returns d loss / d x[g][e] for
//   w_rec * mean(|x0-x|*mask) + w_sm * mean(|second diff|)
// and accumulates this row's share of the two un-weighted sums.
FDC_HD float param_loss_grad(int g, int n_total, float xm2, float xm1, float x0c, float xp1, float xp2,
                             float xdata, float mask, float w_rec_over_cnt, float w_sm_over_cnt,
                             float* rec_abs, float* sm_abs) {
    float diff = xdata - x0c;
    *rec_abs = fabsf(diff) * mask;
    float grad = -sgn(diff) * mask * w_rec_over_cnt;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    *sm_abs = 0.f;
    if (g <= n_total - 3) { float e = second_diff(x0c, xp1, xp2); s0 = sgn(e); *sm_abs = fabsf(e); }
    if (g >= 1 && g <= n_total - 2) s1 = sgn(second_diff(xm1, x0c, xp1));
    if (g >= 2) s2 = sgn(second_diff(xm2, xm1, x0c));
    grad += (s0 - 2.f * s1 + s2) * w_sm_over_cnt;
    return grad;
}

// world-joint first difference (:304): d/dJw[g] of w * mean(|Jw_i - Jw_{i+1}|)
FDC_HD float world_smooth_grad(int g, int n_total, float jm1, float j0, float jp1, float w_over_cnt,
                               float* abs_term) {
    float grad = 0.f;
    *abs_term = 0.f;
    if (g <= n_total - 2) { float d = j0 - jp1; grad += sgn(d); *abs_term = fabsf(d); }
    if (g >= 1) grad -= sgn(jm1 - j0);
    return grad * w_over_cnt;
}

struct AdamScalars { float one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps; };
struct AdamTensor { float* p; float* m; float* v; const float* g; size_t n; AdamScalars a; };

// A DEFERRED optimiser step (fdcap_opt_backward_and_step): the Adam update of iteration ii is not a launch of its own.
//   `scale`            : stepped in place by one extra workgroup of the backward's LAST launch (ScaleTail below: the per-frame
//                        partials of d loss / d scale are complete by then, and no kernel of that launch reads `scale`);
//   body_rotation_rec, : applied by the next iteration's first two launches where they read the parameters anyway --
//   camera_ext           vposer_fwd_*_kernel steps the latent columns of its 16 rows on the way into LDS (nothing written back:
//                        four quarter-workgroups share a row block), pose_fwd_kernel steps its frame's whole row of both
//                        tensors, uses it and writes parameters and moments back (a frame's row has no other reader there).
// Same arithmetic (adam_update, vp_sum_dz, the 256-thread reduction) as adam_step_kernel: same bits.
struct DeferredStep {
    int on = 0;
    AdamTensor x = {}, cam = {};             // as opt_step_plan builds them (first OWNED row; cam.p == nullptr: not stepped)
    int row0 = 2;
    const float* dzpart = nullptr;           // four partial latent gradients (vposer_bwd_*), dz_stride apart; null: dX is complete
    size_t dz_stride = 0;
};
// per-frame partial sums of the printed loss terms (logging iterations): [row][LROW] floats in the slots of losses_d
constexpr int LROW = 8;
// a logging backward's reduction riding in a later launch (the step launch that follows it, or the backward's own extra workgroup)
struct LogReduceIn { const float* rows; double* losses; unsigned mask; int assign, n; };
struct ScaleTail {
    LogReduceIn lg = {nullptr, nullptr, 0u, 0, 0};   // rows != nullptr: this workgroup also sums the logged loss terms (fixed order, double)
    int block = -1;                          // index of the extra workgroup (-1: none)
    AdamTensor sc = {};
    const float* dscale_row = nullptr;       // [rows] per-frame partials of d loss / d scale
    float* dscale = nullptr;                 // this rank's sum (kept for callers that read it)
    int row0 = 2, n = 0, zero_grad = 0;
    // r5: the logged sums are formed by the launch's REGULAR workgroups 0 .. LROW - 1 (one wave, one term each, before their first
    // barrier) instead of the extra workgroup: that one has a CU to itself (these kernels fit once per CU) and a regular workgroup
    // waits for it -- two cold reads and eight double-precision reductions there made every logging launch 1.3-3 us longer
    int lg_spread = 0;
};

// torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8; the bias corrections are evaluated
// in double exactly like torch's python scalars and then applied in fp32.
FDC_HD AdamScalars adam_scalars(double lr, int step) {
    AdamScalars a;
    double b1 = 0.9, b2 = 0.999;
    double bc1 = 1.0 - pow(b1, (double)step);
    double bc2 = 1.0 - pow(b2, (double)step);
    a.one_minus_b1 = (float)(1.0 - b1);
    a.b2 = (float)b2;
    a.one_minus_b2 = (float)(1.0 - b2);
    a.step_size = (float)(lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.eps = 1e-8f;
    return a;
}

// No FMA contraction: torch-CPU's lerp_ / mul_ / addcmul_ / addcdiv_ round every product and sum
// separately, and next to an L1 kink one ulp decides the sign of the next gradient.
FDC_HD void adam_update(float& p, float& m, float& v, float g, const AdamScalars& a) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    m = m + (g - m) * a.one_minus_b1;                 // exp_avg.lerp_(grad, 1-beta1)
    v = v * a.b2 + a.one_minus_b2 * g * g;            // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
    float denom = sqrtf(v) / a.bc2_sqrt + a.eps;      // sqrt(v)/sqrt(bc2) + eps
    p = p - a.step_size * (m / denom);                // addcdiv_(m, denom, -step_size)
}

#if defined(__HIPCC__)
// d loss / d scale = sum of the per-frame partials in a fixed order (256 strided partial sums, wave sums, (s0 + s1) + (s2 + s3)),
// then Adam on `scale` in place.  One workgroup of >= 256 threads; the tail block of adam_step_kernel and the extra workgroup
// of the backward's last launch (ScaleTail) run exactly this.
// (pre != nullptr: the first SCALE_PRE partials of this thread -- rows tid, tid + 256, ... -- were loaded by the caller ahead of other
//  work, scale_grad_prefetch; the sum is formed in the same order either way)
constexpr int SCALE_PRE = 4;
__device__ __forceinline__ void scale_grad_prefetch(const float* __restrict__ dscale_row, int row0, int n, float* pre) {
#pragma unroll
    for (int k = 0; k < SCALE_PRE; ++k) {
        const int i = (int)threadIdx.x + 256 * k;
        pre[k] = (threadIdx.x < 256 && i < n) ? dscale_row[row0 + i] : 0.f;
    }
}
__device__ __forceinline__ float scale_grad_block(const float* __restrict__ dscale_row, int row0, int n, float* sred, const float* pre = nullptr) {
    float a = 0.f;
    if (threadIdx.x < 256) {
        int i = threadIdx.x;
        if (pre) {
#pragma unroll
            for (int k = 0; k < SCALE_PRE; ++k, i += 256) if (i < n) a += pre[k];
        }
        for (; i < n; i += 256) a += dscale_row[row0 + i];
    }
    a = wave_sum64(a);
    if (threadIdx.x < 256 && (threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = a;
    __syncthreads();
    return (sred[0] + sred[1]) + (sred[2] + sred[3]);
}
// the printed loss terms of a logging iteration: per-frame partials -> losses[], in double, in one fixed order whatever the
// workgroup's size (no atomics, no LDS, no barrier): ONE WAVE PER TERM -- lane l adds rows l, l + 64, ... ascending (+0.0 past the end), then a
// butterfly over the lanes.  (r4: the 256-thread LDS tree this replaces held the backward's extra workgroup for ~4 us.)
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
// one term (slot s of the rows), one wave
__device__ __forceinline__ void loss_rows_reduce_slot(const float* __restrict__ rows, int row0, int n, unsigned mask, int assign,
                                                      double* __restrict__ losses, int s, int lane) {
    if (!((mask >> s) & 1u)) {
        if (assign && lane == 0) losses[s] = 0.0;
        return;
    }
    const float* col = rows + (size_t)row0 * LROW + s;
    double a = 0.0;
    for (int i0 = 0; i0 < n; i0 += 1024) {                   // sixteen loads in flight per trip (one trip at the quoted size)
        float r[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = i0 + 64 * k + lane;
            r[k] = col[(size_t)min(i, n - 1) * LROW];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) a += (i0 + 64 * k + lane < n) ? (double)r[k] : 0.0;
    }
    a = wave_sum_f64(a);
    if (lane == 0) losses[s] = assign ? a : losses[s] + a;
}
__device__ __forceinline__ void loss_rows_reduce_block(const float* __restrict__ rows, int row0, int n, unsigned mask, int assign,
                                                       double* __restrict__ losses, const float* __restrict__ dscale_row,
                                                       float* __restrict__ dscale_out) {
    const int tid = threadIdx.x, lane = tid & 63, nw = (int)blockDim.x >> 6;
    const int wave = nw - 1 - (tid >> 6);                    // (the last waves first: the first four also sum d loss / d scale)
    for (int s = wave; s < LROW; s += nw) loss_rows_reduce_slot(rows, row0, n, mask, assign, losses, s, lane);   // (wave-uniform)
    if (dscale_out) {                                        // (the stand-alone launch only: workgroup-uniform)
        __shared__ float sred[4];
        const float g = scale_grad_block(dscale_row, row0, n, sred);
        if (tid == 0) *dscale_out = g;
    }
}
__device__ __forceinline__ void scale_tail_block(const ScaleTail& t) {
    __shared__ float s_tail[4];
    // (a logging iteration: the scale partials are requested BEFORE the printed sums are formed -- both are cold reads of what the
    //  previous launch wrote, one after the other they held this workgroup, and with it the launch, 2.4 us longer)
    float pre[SCALE_PRE];
    const bool prefetched = t.lg.rows && t.n > 0;
    if (prefetched) scale_grad_prefetch(t.dscale_row, t.row0, t.n, pre);
    if (t.lg.rows) loss_rows_reduce_block(t.lg.rows, t.row0, t.lg.n, t.lg.mask, t.lg.assign, t.lg.losses, t.dscale_row, nullptr);
    if (t.n <= 0) return;                                    // (only the logged sums this time: `scale` has no step)
    const float g = scale_grad_block(t.dscale_row, t.row0, t.n, s_tail, prefetched ? pre : nullptr);
    if (threadIdx.x == 0) {
        if (t.dscale) *t.dscale = g;
        if (t.sc.p) {
            float pp = *t.sc.p, mm = *t.sc.m, vv = *t.sc.v;
            adam_update(pp, mm, vv, t.zero_grad ? 0.f : g, t.sc.a);
            *t.sc.p = pp; *t.sc.m = mm; *t.sc.v = vv;
        }
    }
}
#endif

}  // namespace fdc
