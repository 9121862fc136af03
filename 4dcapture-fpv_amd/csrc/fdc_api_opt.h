// C-ABI, part 2: the clip optimiser -- create / inputs / the iteration (forward, losses, backward), fdcap_opt_backward[_and_step].
// Part of csrc/fdcap.hip.
#pragma once

extern "C" {

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
    for (hipEvent_t e : o->lt.ev) (void)hipEventDestroy(e);
    o->lt.ev.clear(); o->lt.what.clear(); o->lt.on = false; o->lt.used = 0;
    delete o;
    c->opt = nullptr;
}

int fdcap_opt_create(fdcap_ctx* c, const fdcap_opt_config* cfg, float* rows_x, float* rows_cam, float* scale_d,
                     float* dscale_d, double* losses_d) {
    if (!c || !cfg || !rows_x || !rows_cam || !scale_d || !dscale_d || !losses_d || cfg->n_local <= 0 || cfg->n_total < cfg->n_local || cfg->frame0 < 0 ||
        cfg->frame0 + cfg->n_local > cfg->n_total)
        return FDCAP_E_ARG;
    SetupTrace tr("fdcap_opt_create");
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
    tr.mark("buffers");
    if (!err && o->contact_on) {
        hipError_t e_ = hipMemset(o->idx.p, 0xFF, nq * sizeof(int));      // -1: no seed yet
        if (e_ != hipSuccess) err = (int)e_;
    }
    if (!err && o->contact_on) {
        if (const char* e = getenv("FDCAP_SKIN_VEC")) o->skin_vec = e[0] != '0';
        if (const char* e = getenv("FDCAP_FUSE_SKIN")) o->fuse_skin = e[0] != '0';
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
    tr.mark("search state");
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
    // clip-sized shares: the blend product and the skinning in one launch (blend_skin_fwd_kernel; FDCAP_FUSE_SKIN=0: two launches)
    const SkinModel smf = c->contact.model();
    const bool fused = !blend_done && o->fuse_skin && gemm_split3_enabled() && c->contact.pn_fwdS.f && smf.vpack && smf.K <= 4 && nl >= clip_forms_min_rows() &&
                       blend_skin_lds_bytes(smf.ja_hi) <= (size_t)150 * 1024;
    if (fused) {
        static std::atomic<uint64_t> attr{0};                  // per DEVICE: the attribute belongs to one device's code object
        if (fdc_attr_needed(attr)) {
            HIP_TRY(hipFuncSetAttribute((const void*)blend_skin_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            fdc_attr_done(attr);
        }
        note_form("blend_skin_fwd_kernel");
        hipLaunchKernelGGL(blend_skin_fwd_kernel, dim3(8 * ((nl + 31) / 32)), dim3(768), blend_skin_lds_bytes(smf.ja_hi), st, o->PF.p, nl,
                           c->contact.pn_fwdS, smf, nc, smf.ja_hi, o->X.p, XDIM, X_TRANSL, o->A.p, o->M.p, o->scale.p, 2, o->Voff.p, o->Vw.p);
    } else {
        if (!blend_done) HIP_TRY(blend_forward(c->contact, o->PF.p + 2 * NPFX, nl, o->Voff.p + off, st));
        note_form("skin_fwd_kernel");
        hipLaunchKernelGGL(skin_fwd_kernel, dim3((nc + 255) / 256, nl), dim3(256), 0, st, smf, nc, o->X.p, XDIM,
                           X_BETAS, X_TRANSL, o->Voff.p, o->A.p, o->M.p, o->scale.p, 2, 1, o->Vw.p);
    }
    lt_mark(o, FDCAP_LT_CONTACT_FWD, st);
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
    lt_mark(o, FDCAP_LT_CHAMFER_NN, st);
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
    o->lt.phase = contact_grad ? 0 : 1;
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
    int dA_rows = NJ;                                       // rows of dA the skinning backward writes (the others are zero and skipped)
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
            dA_rows = c->contact.ja_hi;
            const SkinModel smc = c->contact.model();
            const int G = (smc.K + 3) / 4;                           // weight groups per vertex: the packed layout covers K <= 12
            const size_t ldsv = (size_t)9 * nc * sizeof(float) + (size_t)((nnz + 3) & ~3) * sizeof(float) + (size_t)((nnz + 7) & ~7) * 2;
            if (o->skin_vec && nc <= 512 && G <= 3 && nnz <= 2048 * G && ldsv <= 60000 && (nc & 3) == 0 && smc.vpack && smc.csc_v16 &&
                (((size_t)o->Vw.p | (size_t)o->Voff.p | (size_t)o->dVoff.p | (size_t)o->A.p) & 15) == 0) {
#define FDC_SKV(GG) hipLaunchKernelGGL(skin_bwd_vec_kernel<GG>, dim3(nl), dim3(256), ldsv, st, smc, nc, nnz, o->X.p, o->Voff.p, o->A.p, \
                                   o->M.p, o->scale.p, 2, o->dVoff.p, o->dA.p, o->dtransl_v.p, o->dMv.p, o->dsv.p, cg)
                note_form("skin_bwd_vec_kernel");
                if (G == 1) FDC_SKV(1); else if (G == 2) FDC_SKV(2); else FDC_SKV(3);
#undef FDC_SKV
            } else if (note_form("skin_bwd_small_kernel"), nc <= 512 && nnz <= 2048)
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
        { int es = skin_bwd_any<true>(c->ws_skin, st, nl, c->contact.model(), nc, o->X.p, o->Voff.p, o->A.p, o->M.p, o->scale.p, 2,
                                      (const float*)nullptr, o->dVoff.p, o->dA.p, (float*)nullptr, o->dtransl_v.p, o->dMv.p, o->dsv.p, cg);
          if (es) return es; }
        lt_mark(o, FDCAP_LT_SKIN_BWD, st);
        HIP_TRY(blend_backward(c->contact, o->dVoff.p + (size_t)2 * nc * 3, nl, o->dPF.p + 2 * NPFX, (size_t)o->R * NPFX, c->ws_kpart, st, &dpf_split));
        lt_mark(o, FDCAP_LT_BLEND_BWD, st);
    } else if (contact_fwd && losses) {
        if (fuse_pl && rows_log) { pli.cdist = o->dist.p; pli.cnc = nc; }        // (rides in pose_bwd_kernel's prologue: one launch less)
        else hipLaunchKernelGGL(contact_loss_rows_kernel, dim3(nl), dim3(256), 0, st, o->dist.p, nc, 2, o->loss_rows.p);
    }
    const bool joint_grad = lw.world_on || dct_on;
    hipLaunchKernelGGL(pose_bwd_kernel, dim3(nl), dim3(64 * POSE_NW), 0, st, pm, o->X.p, o->O.p, o->CAM.p, o->scale.p, 2, o->Rm.p,
                       o->Jrest.p, o->G.p, contact_grad ? o->dA.p : nullptr, contact_grad ? o->dPF.p : nullptr,
                       joint_grad ? o->dJw.p : nullptr, contact_grad ? o->dMv.p : nullptr, contact_grad ? o->dsv.p : nullptr,
                       contact_grad ? o->dPF.p + NPF : nullptr, NPFX, contact_grad ? o->dtransl_v.p : nullptr, o->dX.p, o->dO.p,
                       o->dCAM.p, o->dscale_row.p, pli, (contact_grad && dpf_split) ? (const float*)(o->dPF.p + (size_t)o->R * NPFX) : (const float*)nullptr,
                       dA_rows);
    lt_mark(o, FDCAP_LT_POSE_BWD, st);
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

}  // extern "C"
