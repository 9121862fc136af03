// C-ABI, part 1 (include/fdcap.h): context, scene and contact-set registration, the three operators (Chamfer, VPoser decode,
// body model) with their backward passes, parameter conversions.  Part of csrc/fdcap.hip.
#pragma once

extern "C" {

const char* fdcap_version(void) { return "fdcap-hip 0.4 (gfx950)"; }
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

// test / diagnosis: the kernel forms launched since the last call with reset != 0 (fdc_math.h FormLog), "a;b;c" into buf
int fdcap_debug_kernel_forms(char* buf, int32_t len, int32_t reset) {
    if (!buf || len <= 0) return FDCAP_E_ARG;
    FormLog& f = form_log();
    std::string s;
    for (int i = 0; i < f.n; ++i) { if (i) s += ";"; s += f.name[i]; }
    snprintf(buf, (size_t)len, "%s", s.c_str());
    if (reset) f.n = 0;
    return FDCAP_OK;
}

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
    SetupTrace tr("fdcap_ctx_create");
    fdcap_ctx* c = new fdcap_ctx();
    const int V = c->V = md->num_verts;
    c->h_vt.assign(md->v_template, md->v_template + (size_t)V * 3);
    c->h_S10.resize((size_t)V * 30);
    for (size_t i = 0; i < (size_t)V * 3; ++i)
        for (int l = 0; l < NBETA; ++l) c->h_S10[i * NBETA + l] = md->shapedirs[i * md->num_shape + l];
    c->h_lbs.assign(md->lbs_weights, md->lbs_weights + (size_t)V * NJ);
    tr.mark("host copies of the model");
    {   // posedirs (61 MB for SMPL-X) goes straight to the device: vertex sets gather their blend matrix there (build_skin_set)
        hipError_t e_ = c->d_posedirs.upload(md->posedirs, (size_t)NPF * 3 * V);
        if (e_ == hipSuccess) e_ = c->d_S10.upload(c->h_S10.data(), c->h_S10.size());
        if (e_ != hipSuccess) { fdcap_ctx_destroy(c); return (int)e_; }
    }
    tr.mark("posedirs upload");
    HostPoseSetup hs;
    if (!host_pose_setup(V, c->h_vt.data(), c->h_S10.data(), md->J_regressor, md->parents, &hs)) { delete c; return FDCAP_E_ARG; }
    tr.mark("host_pose_setup");
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
    tr.mark("small uploads");
    if (!err) {
        // decoder weights in MFMA fragment order.  torch Linear weights are [out, in]: forward y = x W^T -> B(k, n) = W[n][k];
        // data gradient dx = dy W -> B(k, n) = W[k][n]
        struct { const float* w; long sk, sn; int K, N; PanelB* dst; } pk[6] = {
            {md->vp_fc1_w, 1, VP_Z, VP_Z, VP_H, &c->vp.w1}, {md->vp_fc2_w, 1, VP_H, VP_H, VP_H, &c->vp.w2},
            {md->vp_out_w, 1, VP_H, VP_H, ODIM, &c->vp.w3}, {md->vp_out_w, VP_H, 1, ODIM, VP_H, &c->vp.w3t},
            {md->vp_fc2_w, VP_H, 1, VP_H, VP_H, &c->vp.w2t}, {md->vp_fc1_w, VP_Z, 1, VP_H, VP_Z, &c->vp.w1t}};
        const float* dw[6] = {c->W1.p, c->W2.p, c->W3.p, c->W3.p, c->W2.p, c->W1.p};        // the same weights on the device (uploaded above)
        const size_t dwn[6] = {(size_t)512 * 32, (size_t)512 * 512, (size_t)ODIM * 512, (size_t)ODIM * 512, (size_t)512 * 512, (size_t)512 * 32};
        for (int i = 0; i < 6 && !err; ++i) {
            int nt = 0, ns = 0;
            err = panel_pack_dev(dw[i], dwn[i], pk[i].sk, pk[i].sn, pk[i].K, pk[i].N, c->vp_pn[i], &nt, &ns);
            pk[i].dst->f = (const float4*)c->vp_pn[i].p; pk[i].dst->ntile = nt; pk[i].dst->nss = ns;
        }
        c->vp.b1 = c->b1.p; c->vp.b2 = c->b2.p; c->vp.b3 = c->b3.p;
        PanelB3* dst3[6] = {&c->vp3.w1, &c->vp3.w2, &c->vp3.w3, &c->vp3.w3t, &c->vp3.w2t, &c->vp3.w1t};
        float c1n[6];                                                      // largest column 1-norm of each operand (VPoserPanels3::c1 ...)
        for (int i = 0; i < 6 && !err; ++i) {
            err = pnf_pack_dev(dw[i], dwn[i], pk[i].sk, pk[i].sn, pk[i].K, pk[i].N, c->vp_pn3[i], c->vp_pn3s[i], &dst3[i]->ntile, &dst3[i]->nst);
            dst3[i]->f = (const uint4*)c->vp_pn3[i].p; dst3[i]->isc = c->vp_pn3s[i].p;
            double mx = 0.0;
            for (int n = 0; n < pk[i].N; ++n) {
                double a = 0.0;
                for (int k = 0; k < pk[i].K; ++k) a += fabs((double)pk[i].w[(long)k * pk[i].sk + (long)n * pk[i].sn]);
                mx = std::max(mx, a);
            }
            c1n[i] = (float)(mx * 1.001);
        }
        auto amax = [](const float* b, int n) { float m = 0.f; for (int i = 0; i < n; ++i) m = std::max(m, fabsf(b[i])); return m * 1.001f; };
        c->vp3.c1 = c1n[0]; c->vp3.c2 = c1n[1]; c->vp3.c3t = c1n[3]; c->vp3.c2t = c1n[4];
        c->vp3.b1max = amax(md->vp_fc1_b, 512); c->vp3.b2max = amax(md->vp_fc2_b, 512);
        c->vp3.b1 = c->b1.p; c->vp3.b2 = c->b2.p; c->vp3.b3 = c->b3.p;
    }
    tr.mark("decoder panels");
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
    c->Jt.release(); c->Jd.release(); c->hand_comp.release(); c->hand_mean.release(); c->d_posedirs.release(); c->d_S10.release();
    c->parents.release(); c->order.release(); c->level_start.release(); c->child_start.release(); c->child_list.release(); c->depth.release();
    c->pose_tab.release();
    c->W1.release(); c->b1.release(); c->W2.release(); c->b2.release(); c->W3.release(); c->b3.release();
    for (auto& b : c->vp_pn) b.release();
    for (auto& b : c->vp_pn3) b.release();
    for (auto& b : c->vp_pn3s) b.release();
    c->full.release(); c->contact.release(); c->contact_vid.release(); c->contact_perm.release(); c->scene.release(); c->scene_sorted.release(); c->scene_bounds.release(); c->scene_sbounds.release(); c->scene_qbounds.release(); c->scene_inv.release(); c->scene_frags.release(); c->scene_centers.release(); c->sop.release(); c->ws_skin.release(); c->ws_kpart.release();
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
    static_assert(SC_SUPER == ST4_SUPER, "fdc_scene.h builds the super-cell boxes the search reads");
    // Spatial order by recursive median splits (k-d cells): every MF_CH-point chunk is one cell and every 32-point MFMA tile inside it
    // a sub-cell, so the boxes the NN scan culls with are compact and disjoint (runs of a Morton curve jump across quadrant borders
    // and give long, overlapping boxes).  r6: sorted, boxed and packed ON THE DEVICE (fdc_scene.h); FDCAP_SCENE_BUILD=host takes the
    // order from the host recursion of r1-r5 instead (the same order by specification: tests compare the tables of the two).
    // Results never depend on this order.
    const int64_t nchunk = (ns + MF_CH - 1) / MF_CH, nsuper = (nchunk + ST4_SUPER - 1) / ST4_SUPER;
    HIP_TRY(c->scene.ensure((size_t)ns)); HIP_TRY(c->scene_sorted.ensure((size_t)ns)); HIP_TRY(c->scene_inv.ensure((size_t)ns));
    HIP_TRY(c->scene_bounds.ensure((size_t)nchunk * 2)); HIP_TRY(c->scene_qbounds.ensure((size_t)nchunk * 8));
    HIP_TRY(c->scene_frags.ensure((size_t)nchunk * (MF_CH / 32) * 64)); HIP_TRY(c->scene_centers.ensure((size_t)nchunk));
    {
        std::vector<float4> sb((size_t)std::max<int64_t>(nsuper, 1) * 2, make_float4(0.f, 0.f, 0.f, 0.f));
        HIP_TRY(c->scene_sbounds.upload(sb.data(), sb.size()));
    }
    const char* how = getenv("FDCAP_SCENE_BUILD");
    std::vector<int> order;
    const bool host_order = how && !strcmp(how, "host");
    if (host_order) scene_order_host(xyz, ns, order);
    const SceneTables T{c->scene.p, c->scene_sorted.p, c->scene_inv.p, c->scene_bounds.p, c->scene_qbounds.p, c->scene_sbounds.p,
                        c->scene_frags.p, c->scene_centers.p};
    HIP_TRY(scene_build_device(xyz, ns, T, host_order ? order.data() : nullptr));
    c->ns = ns;
    return FDCAP_OK;
}

// test / diagnosis: FNV-1a hashes of the registered scene's eight device tables (input-order points, sorted points, inverse
// permutation, cell boxes, quarter boxes, super-cell boxes, fragments, centres), so two builds of the same scene can be compared
int fdcap_debug_scene_hash(fdcap_ctx* c, uint64_t* out8) {
    if (!c || !out8) return FDCAP_E_ARG;
    HIP_TRY(hipDeviceSynchronize());
    const int64_t ns = c->ns, nchunk = (ns + MF_CH - 1) / MF_CH, nsuper = std::max<int64_t>((nchunk + ST4_SUPER - 1) / ST4_SUPER, 1);
    struct { const void* p; size_t bytes; } t[8] = {
        {c->scene.p, (size_t)ns * 16}, {c->scene_sorted.p, (size_t)ns * 16}, {c->scene_inv.p, (size_t)ns * 4},
        {c->scene_bounds.p, (size_t)nchunk * 32}, {c->scene_qbounds.p, (size_t)nchunk * 128}, {c->scene_sbounds.p, (size_t)nsuper * 32},
        {c->scene_frags.p, (size_t)nchunk * (MF_CH / 32) * 64 * 16}, {c->scene_centers.p, (size_t)nchunk * 16}};
    std::vector<unsigned char> h;
    for (int i = 0; i < 8; ++i) {
        h.resize(t[i].bytes);
        if (t[i].bytes) HIP_TRY(hipMemcpy(h.data(), t[i].p, t[i].bytes, hipMemcpyDeviceToHost));
        uint64_t v = 1469598103934665603ull;
        for (size_t k = 0; k < t[i].bytes; ++k) { v ^= h[k]; v *= 1099511628211ull; }
        out8[i] = v;
    }
    return FDCAP_OK;
}

int fdcap_set_contact_ids(fdcap_ctx* c, const int64_t* vid, int32_t nc) {
    if (!c || nc < 0 || (nc > 0 && !vid)) return FDCAP_E_ARG;
    if (c->opt) return FDCAP_E_STATE;                  // (see fdcap_set_scene)
    SetupTrace tr("fdcap_set_contact_ids");
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
    tr.mark("slot order");
    int e = build_skin_set(c, ids, &c->contact);
    if (e) return e;
    tr.mark("build_skin_set");
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
    static std::atomic<float> slack{-1.f};
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
        { int es = skin_bwd_any<false>(c->ws_skin, st, B, c->full.model(), V, X, wf[11].p, w[6].p, Mid, one, 0, g_vertices, wf[0].p, w[7].p,
                                       (float*)nullptr, dtv, dMv, dsv, ContactGradIn()); if (es) return es; }
        HIP_TRY(blend_backward(c->full, wf[0].p, B, w[9].p, 0, c->ws_kpart, st));
        dA = w[7].p; dPF = w[9].p; dtr = dtv;
    }
    hipLaunchKernelGGL(pose_bwd_op_kernel, dim3(B), dim3(64), 0, st, c->pose_model(), X, AA, w[3].p, w[4].p, w[5].p, dA, dPF, dtr,
                       g_joints, w[10].p, w[11].p);
    hipLaunchKernelGGL(smplx_bwd_split_kernel, dim3((B + 127) / 128), dim3(128), 0, st, w[10].p, w[11].p, B, g_go, g_bp, g_betas, g_lh,
                       g_rh, g_transl);
    return (int)hipGetLastError();
}

}  // extern "C"
