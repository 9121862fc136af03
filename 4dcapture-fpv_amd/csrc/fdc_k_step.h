// The small kernels around the loop: mode 'local' (vertex smoothing, foot skate, detect_contact), loss-row reductions, Adam and
// the deferred / exchanged step, 75 <-> 78 conversions, and the operator-level backward helpers.  Part of csrc/fdcap.hip.
#pragma once

namespace {

// mode 'local', cal_loss2 (:404-405): d/dV of mean |second difference over frames| of ALL world vertices.
// V is [rows, nv3] (nv3 = 3 * vertices); owned rows start at row0, global frame = frame0 + blockIdx.y.
// Grid (VS_NB, frames): a workgroup walks its frame's elements blockIdx.x * 256 + tid, + VS_NB * 256, ...  (r5, late: with one
// workgroup per 256 elements the launch was 63 000 double-precision atomics on ONE address -- 762 us for 128 MB of traffic, 60 % of a
// mode-'local' second-loop iteration at 512 frames; tools/trace_outliers.py on tools/modes_trace.py)
#ifndef FDC_VS_NB
#define FDC_VS_NB 4
#endif
constexpr int VS_NB = FDC_VS_NB;                    // (measured at 512 frames: 1 -> 96 us, 2 -> 63, 4 -> 54, 8 -> 63, 16 -> 107, 32 -> 202: the atomics again)
__global__ void vert_smooth_kernel(const float* __restrict__ V, size_t nv3, int row0, int frame0, int n_total,
                                   float w_over_cnt, float* __restrict__ dV, double* __restrict__ loss_sum) {
    __shared__ float sred[4];
    const int r = row0 + blockIdx.y, g = frame0 + blockIdx.y;
    float ab = 0.f;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nv3; e += (size_t)gridDim.x * 256) {
        const float* v = V + (size_t)r * nv3 + e;
        const float x0 = v[0];
        const float xm2 = g >= 2 ? v[-2 * (ptrdiff_t)nv3] : 0.f, xm1 = g >= 1 ? v[-(ptrdiff_t)nv3] : 0.f;
        const float xp1 = g + 1 < n_total ? v[nv3] : 0.f, xp2 = g + 2 < n_total ? v[2 * nv3] : 0.f;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        if (g <= n_total - 3) { float d = second_diff(x0, xp1, xp2); s0 = sgn(d); ab += fabsf(d); }
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
    // (grid (1, frames) since late r5: one atomic per frame instead of one per 256 elements)
    __shared__ float sred[4];
    const int r = row0 + blockIdx.y, g = frame0 + blockIdx.y;
    float ab = 0.f;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < nc * 3; t += gridDim.x * 256) {
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
            ab += fabsf(d) * inv;                            // (the two parts have different denominators)
            grad += sgn(d) * w;
        }
        if (g >= 1) {
            float w = wgt[g];
            w = is_left ? 1.f - w : w;
            w = w < 0.5f ? 0.f : w;
            grad -= sgn((v[-(ptrdiff_t)nv3] - v[0]) * w) * w;
        }
        dV[(size_t)r * nv3 + e] += grad * inv;
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

}  // namespace
