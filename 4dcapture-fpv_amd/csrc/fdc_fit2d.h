// Per-frame inner fit with a true 2D-keypoint reprojection residual (BASELINE config 4).
//
// NOT in the reference repository: the per-frame fit is the external SMPLify-X step (/root/reference/README.md:14-17;
// the only 2D projection in-repo is a viewer overlay, local_vis.py:368-378, with the fixed pinhole intrinsics
// fx = fy = 692, cx = 640, cy = 360 of vis.py:358-360).  Restated from the published SMPLify-X objective:
//   data  : w_data^2 * sum_j conf_j^2 * [ GMoF_rho(u_j - kp_j.u) + GMoF_rho(v_j - kp_j.v) ],   GMoF_rho(r) = rho^2 r^2 / (r^2 + rho^2)
//   priors: w_pose^2 * |z|^2 (VPoser latent) + w_shape^2 * |betas|^2 + w_hand^2 * (|lh|^2 + |rh|^2)
// on the camera-frame joints J_j = SMPL-X(VPoser(z)) joint j + transl + camera_translation (the optimiser's
// body2world with camera_ext = I and scale = 1), projected u = fx X / Z + cx, v = fy Y / Z + cy.
// Frames are independent problems (no temporal term), so one launch covers a whole batch of frames.
#pragma once
#include "fdc_frame.h"
#include "fdc_math.h"

namespace fdc {

struct Fit2dStage { float fx, fy, cx, cy, rho, w_data, w_pose, w_shape, w_hand; };

// value and derivative of GMoF_rho at residual r
FDC_HD float gmof(float r, float rho2, float* dg) {
    float r2 = r * r, den = r2 + rho2;
    *dg = 2.f * r * rho2 * rho2 / (den * den);
    return rho2 * r2 / den;
}

// One joint of one frame: camera-frame joint J, keypoint (u, v, conf) -> d loss / d J and the joint's data term.
FDC_HD float fit2d_joint(const Fit2dStage& s, V3 J, float ku, float kv, float conf, V3* dJ) {
    const float iz = 1.f / J.z;
    const float u = s.fx * J.x * iz + s.cx, v = s.fy * J.y * iz + s.cy;
    const float w = s.w_data * s.w_data * conf * conf;
    float dgu, dgv;
    const float gu = gmof(ku - u, s.rho * s.rho, &dgu), gv = gmof(kv - v, s.rho * s.rho, &dgv);
    const float du = -w * dgu, dv = -w * dgv;                  // d loss / d u, d v  (residual = keypoint - projection)
    dJ->x = du * s.fx * iz;
    dJ->y = dv * s.fy * iz;
    dJ->z = -(du * s.fx * J.x + dv * s.fy * J.y) * iz * iz;
    return w * (gu + gv);
}

// gradient of the L2 priors on element e of the 78-d row
FDC_HD float fit2d_prior_grad(const Fit2dStage& s, int e, float x, float* val) {
    float w = 0.f;
    if (e >= X_LATENT && e < X_LATENT + 32) w = s.w_pose * s.w_pose;
    else if (e >= X_BETAS && e < X_BETAS + NBETA) w = s.w_shape * s.w_shape;
    else if (e >= X_LH && e < X_CAMT) w = s.w_hand * s.w_hand;
    *val = w * x * x;
    return 2.f * w * x;
}

#if defined(__HIPCC__)
// block (128 threads) per frame: dX (=) prior gradients, dJw (=) reprojection gradients of the 23 joints;
// losses (optional, logging): [0] += data term, [1] += priors; floss (optional) [frame] = this frame's data term + priors
// (the value a per-frame line search needs: fdcap_opt_fit2d_lbfgs)
__global__ __launch_bounds__(128) void fit2d_loss_kernel(Fit2dStage s, const float* __restrict__ X, const float* __restrict__ Jw,
                                                         const float* __restrict__ kp, int row0, float* __restrict__ dX,
                                                         float* __restrict__ dJw, double* __restrict__ losses,
                                                         float* __restrict__ floss = nullptr) {
    __shared__ float sred[2][2];
    const int tid = threadIdx.x, r = row0 + blockIdx.x;
    float data = 0.f, prior = 0.f;
    if (tid < XDIM) dX[(size_t)r * XDIM + tid] = fit2d_prior_grad(s, tid, X[(size_t)r * XDIM + tid], &prior);
    if (tid < NJW) {
        const float* j = Jw + ((size_t)r * NJW + tid) * 3;
        const float* k = kp + ((size_t)blockIdx.x * NJW + tid) * 3;
        V3 dJ;
        data = fit2d_joint(s, v3(j[0], j[1], j[2]), k[0], k[1], k[2], &dJ);
        float* o = dJw + ((size_t)r * NJW + tid) * 3;
        o[0] = dJ.x; o[1] = dJ.y; o[2] = dJ.z;
    }
    if (!losses && !floss) return;
    float vals[2] = {data, prior};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float v = vals[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((tid & 63) == 0) sred[tid >> 6][i] = v;
    }
    __syncthreads();
    if (floss && tid == 0) floss[blockIdx.x] = (sred[0][0] + sred[1][0]) + (sred[0][1] + sred[1][1]);
    if (losses && tid < 2) atomicAdd(&losses[tid], (double)(sred[0][tid] + sred[1][tid]));
}
#endif

}  // namespace fdc
