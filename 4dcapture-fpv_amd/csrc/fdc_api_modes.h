// C-ABI, part 3: mode 'dct', the per-frame inner fit (Adam and batched L-BFGS), checkpoint state / finite check, the per-frame
// smoother of optimization.py.  Part of csrc/fdcap.hip.
#pragma once

extern "C" {

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

}  // extern "C"
