// Host side: device buffers, the context (model constants, scene, contact set, operator workspaces), the optimiser state and
// the helpers every entry point shares (row ranges, step plans, the decoder / pose forward pair).  Part of csrc/fdcap.hip.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// FDCAP_SETUP_TRACE=1: the set-up entry points (fdcap_ctx_create, fdcap_set_contact_ids, fdcap_opt_create) print the host time
// of their stages to stderr (tools/setup_timing.py reads the totals from outside; this says where they go)
struct SetupTrace {
    bool on;
    const char* who;
    std::chrono::steady_clock::time_point t0, last;
    explicit SetupTrace(const char* w) : who(w) {
        const char* e = getenv("FDCAP_SETUP_TRACE");
        on = e && e[0] == '1';
        if (on) t0 = last = std::chrono::steady_clock::now();
    }
    void mark(const char* what) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[fdcap setup] %s: %-28s %8.3f ms (at %8.3f)\n", who, what, std::chrono::duration<double, std::milli>(now - last).count(),
                std::chrono::duration<double, std::milli>(now - t0).count());
        last = now;
    }
};

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

// skin_bwd_kernel for any vertex set: one workgroup per frame up to a chunk of vertices, the split form + its reduction beyond
// (`part`: workspace, grown as needed).  Same arguments as the kernel.
template <bool CONTACT>
static int skin_bwd_any(DevBuf<float>& part, hipStream_t st, int nrows, SkinModel sm, int nc, const float* X, const float* Voff, const float* A,
                        const float* M, const float* scale, int row0, const float* dVw, float* dVoff, float* dA, float* dbeta_v,
                        float* dtransl_v, float* dMv, float* dsv, ContactGradIn cg) {
    static std::atomic<int> split_on{-1};                              // FDCAP_SKIN_SPLIT=0: the one-workgroup-per-frame form at every size (A/B)
    if (split_on < 0) { const char* e = getenv("FDCAP_SKIN_SPLIT"); split_on = (e && e[0] == '0') ? 0 : 1; }
    // dT rows: 12 floats per vertex for the list form; the matrix form keeps them factored (SKB_ROW = 6 floats) and reuses the space for its
    // four waves' partial tiles (16 KB: more than 512 vertices of rows, which is when the matrix form is built)
    const size_t lds = sm.wf_tab ? std::max((size_t)std::min(nc, SKB_VCH) * SKB_ROW, (size_t)4 * 64 * 16) * sizeof(float)
                                 : (size_t)std::min(nc, SKB_VCH) * 12 * sizeof(float);
    if (nc <= SKB_VCH || !split_on) {
        note_form("skin_bwd_kernel(one workgroup per frame)");
        hipLaunchKernelGGL((skin_bwd_kernel<CONTACT, false>), dim3(nrows), dim3(256), lds, st, sm, nc, X, Voff, A, M, scale, row0, dVw, dVoff, dA,
                           dbeta_v, dtransl_v, dMv, dsv, cg, (float*)nullptr);
        return (int)hipGetLastError();
    }
    const int nch = (nc + SKB_VCH - 1) / SKB_VCH;
    hipError_t e = part.ensure((size_t)nrows * nch * SKP_STRIDE);
    if (e != hipSuccess) return (int)e;
    note_form(sm.wf_tab ? "skin_bwd_kernel(chunks, MFMA dA)" : "skin_bwd_kernel(chunks, list dA)");
    hipLaunchKernelGGL((skin_bwd_kernel<CONTACT, true>), dim3(nrows, nch), dim3(256), lds, st, sm, nc, X, Voff, A, M, scale, row0, dVw, dVoff, dA,
                       dbeta_v, dtransl_v, dMv, dsv, cg, part.p);
    hipLaunchKernelGGL(skin_bwd_reduce_kernel, dim3(nrows), dim3(256), 0, st, part.p, nch, row0, dA, dbeta_v, dtransl_v, dMv, dsv,
                       CONTACT ? cg.loss_rows : (float*)nullptr);
    return (int)hipGetLastError();
}

struct SkinSet {          // skinning constants for a vertex set (all V, or the contact subset)
    int nv = 0, K = 0, nnz = 0;
    int ldp = 0;                                // row stride of posedirs: 3*nv rounded up to a multiple of 4 (16-byte rows)
    DevBuf<float> vt, ww, posedirs, csc_w;      // posedirs [496, ldp] = [posedirs ; shapedirs^T], zero padding
    DevBuf<int> wj, csc_start, csc_v, csc_chunk;
    int nch = 0, ja_hi = NJ;
    DevBuf<int> wf_tab; DevBuf<uint2> wf_step; DevBuf<float4> wf_frag;              // SkinModel::wf_*
    DevBuf<float4> vpack;                       // SkinModel::vpack / csc_v16 (K <= 4 and nv <= 65535 only)
    DevBuf<unsigned short> csc_v16;
    // the blend matrix in MFMA fragment order (fdc_panel.h), built for vertex sets whose K = 3 nv image fits the LDS slabs:
    // pn_fwd: B(k, n) = posedirs[k, n] (offsets = [pose feature | betas] x B), pn_bwd: B(k, n) = posedirs[n, k] (data gradient)
    DevBuf<float> pn_fwd_f, pn_bwd_f;
    PanelB pn_fwd, pn_bwd;
    DevBuf<unsigned> pn_fwd3_f, pn_bwd3_f;      // the same two operands as split planes (format PnF: panel_gemm3_* kernels)
    DevBuf<float> pn_fwd3_s, pn_bwd3_s;         // their column tiles' inverse scales (PnH2)
    // the forward operand once more with its columns permuted for blend_skin_fwd_kernel (fdc_k_skin.h): sets of <= 512 vertices, K <= 4
    DevBuf<unsigned> pn_fwdS_f; DevBuf<float> pn_fwdS_s; PanelB3 pn_fwdS;
    PanelB3 pn_fwd3, pn_bwd3;
    SkinModel model() const {
        SkinModel m; m.vt = vt.p; m.S = nullptr; m.wj = wj.p; m.ww = ww.p; m.K = K;
        m.csc_start = csc_start.p; m.csc_v = csc_v.p; m.csc_w = csc_w.p;
        m.vpack = (const float*)vpack.p; m.csc_v16 = csc_v16.p;
        m.csc_chunk = csc_chunk.p; m.nch = nch; m.ja_hi = ja_hi;
        m.wf_tab = wf_tab.p; m.wf_step = (const unsigned*)wf_step.p; m.wf_frag = (const float*)wf_frag.p;
        return m;
    }
    void release() { vt.release(); ww.release(); posedirs.release(); wj.release(); csc_w.release(); csc_start.release(); csc_v.release();
                     vpack.release(); csc_v16.release(); csc_chunk.release(); wf_tab.release(); wf_step.release(); wf_frag.release();
                     pn_fwd_f.release(); pn_bwd_f.release(); pn_fwd = PanelB(); pn_bwd = PanelB();
                     pn_fwd3_f.release(); pn_bwd3_f.release(); pn_fwd3_s.release(); pn_bwd3_s.release(); pn_fwdS_f.release(); pn_fwdS_s.release(); pn_fwdS = PanelB3(); pn_fwd3 = PanelB3(); pn_bwd3 = PanelB3(); }
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
    bool fuse_skin = true;         // FDCAP_FUSE_SKIN=0 (read by fdcap_opt_create; A/B): blend product and skinning forward as two launches
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
    // fdcap_opt_launch_timing (r6): a HIP event on the launch stream at every boundary between two launches of an iteration; the time
    // from one event to the next is booked on the launch in between (its ~1 us dependent-launch gap included), per phase of the fit
    struct LaunchTimes {
        bool on = false;
        int used = 0, phase = 0;
        std::vector<hipEvent_t> ev;
        std::vector<signed char> what;         // what[k]: the stage that ended at event k (FDCAP_LT_*; -1: an iteration's first event), + 16 in phase 2
    } lt;
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

// (fdcap_opt_launch_timing) the stage `what` ended here; -1: an iteration begins
inline void lt_mark(OptState* o, int what, hipStream_t st) {
    OptState::LaunchTimes& t = o->lt;
    if (!t.on || t.used >= (int)t.ev.size()) return;
    if (hipEventRecord(t.ev[t.used], st) != hipSuccess) return;
    t.what[t.used++] = (signed char)(what < 0 ? -1 : what + 16 * t.phase);
}

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
    std::vector<float> h_vt, h_S10, h_lbs;
    DevBuf<float> d_posedirs, d_S10;    // posedirs [486, 3V] and shapedirs' first ten [3V, 10]: what a vertex set's blend matrix is gathered from (r6: on the device)
    // device constants
    DevBuf<float> Jt, Jd, hand_comp, hand_mean;
    DevBuf<int> parents, order, level_start, child_start, child_list, depth;
    DevBuf<float> pose_tab;        // all of the above as ONE image in PoseStage's layout (what the staged pose kernels copy)
    int nlevels = 0;
    DevBuf<float> W1, b1, W2, b2, W3, b3;
    DevBuf<float> vp_pn[6];            // decoder weights in MFMA fragment order: forward w1 w2 w3, backward w3t w2t w1t
    VPoserPanels vp;
    DevBuf<unsigned> vp_pn3[6];        // ... and as split planes each (format VpF; the default form of the products)
    DevBuf<float> vp_pn3s[6];          // their output columns' inverse scales (PnH2)
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
    DevBuf<float> ws_skin;          // chunk partials of the split skinning backward (skin_bwd_any)
    DevBuf<float> ws_kpart;         // K-part partial products of the full-mesh data gradient (blend_backward)
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

// [posedirs ; shapedirs^T] of a vertex set, rows of ldp floats: pd[r, 3 i + k] = component k of vertex ids[i], zero padding
__global__ __launch_bounds__(256) void blend_rows_gather_kernel(const float* __restrict__ posedirs, const float* __restrict__ S10, int V,
                                                                const int* __restrict__ ids, int nv, int ldp, float* __restrict__ pd) {
    const int col = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (col >= ldp) return;
    float v = 0.f;
    if (col < 3 * nv) {
        const int src = 3 * ids[col / 3] + col % 3;
        v = r < NPF ? posedirs[(size_t)r * 3 * V + src] : S10[(size_t)src * NBETA + (r - NPF)];
    }
    pd[(size_t)r * ldp + col] = v;
}
// its columns permuted for blend_skin_fwd_kernel: block b = vertices 64 b .. + 63, tile 3 g + c = component c of vertices 64 b + 16 g .. + 15
__global__ __launch_bounds__(256) void blend_rows_permute_kernel(const float* __restrict__ pd, int ldp, int nv, int ncs, float* __restrict__ ps) {
    const int dst = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    if (dst >= ncs) return;
    const int b = dst / 192, t = (dst % 192) / 16, jj = dst % 16;
    const int v = 64 * b + 16 * (t / 3) + jj;
    ps[(size_t)k * ncs + dst] = v < nv ? pd[(size_t)k * ldp + 3 * v + (t % 3)] : 0.f;
}

// FDCAP_PANEL_PACK=host: the static operands are packed by the host loops of fdc_panel.h (r1-r5) instead of the device kernels --
// the same arithmetic, kept as the specification (tests compare whole fits of the two, bit for bit).  Read at every call.
inline bool panel_pack_on_host() { const char* e = getenv("FDCAP_PANEL_PACK"); return e && !strcmp(e, "host"); }

// B(k, n) = src[k * sk + n * sn] (DEVICE pointer; sk == 1 or sn == 1) -> fp32 fragment order
int panel_pack_dev(const float* src, size_t src_floats, long sk, long sn, int K, int N, DevBuf<float>& out, int* ntile, int* nss) {
    const int nt = (N + 15) / 16, ns = (K + 15) / 16;
    *ntile = nt; *nss = ns;
    if (panel_pack_on_host()) {
        std::vector<float> h(src_floats), pf;
        HIP_TRY(hipMemcpy(h.data(), src, src_floats * sizeof(float), hipMemcpyDeviceToHost));
        panel_pack(h.data(), sk, sn, K, N, pf, ntile, nss);
        return (int)out.upload(pf.data(), pf.size());
    }
    const size_t lanes = (size_t)nt * ns * 64;
    HIP_TRY(out.ensure(lanes * 4));
    hipLaunchKernelGGL(panel_pack_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, 0, src, sk, sn, K, N, nt, ns, (float4*)out.p);
    return (int)hipGetLastError();
}
// ... -> the split format PnF (planes + the column tiles' inverse scales)
int pnf_pack_dev(const float* src, size_t src_floats, long sk, long sn, int K, int N, DevBuf<unsigned>& out, DevBuf<float>& isc, int* ntile,
                 int* nst) {
    const int nt = (N + 15) / 16, ns = (K + 31) / 32;
    if (PnF::NP != 2 || panel_pack_on_host() || (sk != 1 && sn != 1)) {
        std::vector<float> h(src_floats), sc;
        std::vector<unsigned> p3;
        HIP_TRY(hipMemcpy(h.data(), src, src_floats * sizeof(float), hipMemcpyDeviceToHost));
        PnF::pack(h.data(), sk, sn, K, N, p3, sc, ntile, nst);
        HIP_TRY(out.upload(p3.data(), p3.size()));
        return (int)isc.upload(sc.data(), sc.size());
    }
    *ntile = nt; *nst = ns;
    const size_t lanes = (size_t)nt * ns * 64;
    HIP_TRY(out.ensure(lanes * 2 * 4)); HIP_TRY(isc.ensure((size_t)nt * 16));
    DevBuf<float> scl;                                            // the columns' scales: only the packing launch needs them
    hipError_t e = scl.ensure((size_t)nt * 16);
    if (e == hipSuccess) {
        if (sn == 1) hipLaunchKernelGGL(pnh2_colscale_rowmajor_kernel, dim3((nt * 16 + 255) / 256), dim3(256), 0, 0, src, sk, K, N, nt * 16, scl.p, isc.p);
        else hipLaunchKernelGGL(pnh2_colscale_colmajor_kernel, dim3(nt * 16), dim3(256), 0, 0, src, sn, K, N, nt * 16, scl.p, isc.p);
        hipLaunchKernelGGL(pnh2_pack_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, 0, src, sk, sn, K, N, nt, ns, scl.p, (uint4*)out.p);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();          // (scl is freed below: the launch must be done with it)
    }
    scl.release();
    return (int)e;
}

int build_skin_set(fdcap_ctx* c, const std::vector<int64_t>& ids, SkinSet* out) {
    SetupTrace tr("build_skin_set");
    const int V = c->V;
    const int nv = (int)ids.size();
    int K = 1;
    for (int v : ids) {
        int k = 0;
        for (int j = 0; j < NJ; ++j) k += c->h_lbs[(size_t)v * NJ + j] != 0.f;
        K = std::max(K, k);
    }
    const int ldp = (3 * nv + 3) & ~3;
    std::vector<float> vt((size_t)nv * 3), ww((size_t)nv * K, 0.f);
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
    // the set's blend matrix [posedirs ; shapedirs^T] (rows 486..495: the betas part, so [pose feature | betas] x this matrix =
    // pose offsets + shape offsets), gathered on the device from the context's copies
    const size_t pd_floats = (size_t)NPFX * ldp;
    HIP_TRY(out->posedirs.ensure(pd_floats));
    {
        std::vector<int> id32((size_t)std::max(nv, 1), 0);
        for (int i = 0; i < nv; ++i) id32[i] = (int)ids[i];
        DevBuf<int> d_ids;
        { hipError_t eu = d_ids.upload(id32.data(), id32.size()); if (eu != hipSuccess) { d_ids.release(); return (int)eu; } }
        if (ldp > 0) hipLaunchKernelGGL(blend_rows_gather_kernel, dim3((ldp + 255) / 256, NPFX), dim3(256), 0, 0, c->d_posedirs.p, c->d_S10.p, V, d_ids.p, nv, ldp, out->posedirs.p);
        hipError_t e_ = hipGetLastError();
        if (e_ == hipSuccess) e_ = hipDeviceSynchronize();
        d_ids.release();
        HIP_TRY(e_);
    }
    const float* pd = out->posedirs.p;
    tr.mark("gather vt / weights / posedirs");
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
    {   // chunk entry points of every joint's list (skin_bwd_kernel's chunks of SKB_VCH vertices)
        const int nch = std::max(1, (nv + SKB_VCH - 1) / SKB_VCH);
        std::vector<int> cc((size_t)NJ * (nch + 1), 0);
        for (int j = 0; j < NJ; ++j) {
            int i = csc_start[j];
            for (int ch = 0; ch <= nch; ++ch) {
                while (i < csc_start[j + 1] && csc_v[i] < ch * SKB_VCH) ++i;
                cc[(size_t)j * (nch + 1) + ch] = ch == nch ? csc_start[j + 1] : i;
            }
        }
        out->nch = nch;
        out->ja_hi = 1;
        for (int j = 0; j < NJ; ++j) if (csc_start[j + 1] > csc_start[j]) out->ja_hi = j + 1;
        HIP_TRY(out->csc_chunk.upload(cc.data(), cc.size()));
    }
    tr.mark("lists, vpack, chunk table");
    out->wf_tab.release(); out->wf_step.release(); out->wf_frag.release();
    {   // the weights as MFMA fragments for skin_bwd_kernel's dA (SkinModel::wf_*): vertex sets beyond the contact-set kernels' reach;
        // FDCAP_SKIN_DA_MFMA=0: the ordered list form
        const char* e = getenv("FDCAP_SKIN_DA_MFMA");
        if (!(e && e[0] == '0') && nv > 512) {
            const int nchm = (nv + SKB_VCH - 1) / SKB_VCH;
            std::vector<int> tab((size_t)nchm * 16 + 1, 0);
            std::vector<uint2> steps;
            std::vector<float4> frag;
            for (int ch = 0; ch < nchm; ++ch)
                for (int q = 0; q < 4; ++q)
                    for (int jt = 0; jt < 4; ++jt) {
                        tab[((size_t)ch * 4 + q) * 4 + jt] = (int)steps.size();
                        std::vector<int> st;                               // the steps of this quarter that touch the tile
                        std::vector<float> fr;                             // their fragments [step][64]
                        for (int s = 64 * q; s < 64 * q + 64; ++s) {
                            const int v0 = ch * SKB_VCH + 4 * s;
                            if (v0 >= nv) break;
                            float f[64];
                            bool any = false;
                            for (int l = 0; l < 64; ++l) f[l] = 0.f;
                            for (int kk = 0; kk < 4 && v0 + kk < nv; ++kk)
                                for (int k = 0; k < K; ++k) {
                                    const int j = wj[(size_t)(v0 + kk) * K + k];
                                    const float w = ww[(size_t)(v0 + kk) * K + k];
                                    if (w != 0.f && (j >> 4) == jt) { f[16 * kk + (j & 15)] += w; any = true; }
                                }
                            if (any) { st.push_back(s); fr.insert(fr.end(), f, f + 64); }
                        }
                        while (st.size() & 3) { st.push_back(st.empty() ? 0 : st.back()); fr.insert(fr.end(), 64, 0.f); }   // zero weights: no contribution
                        for (size_t g = 0; g < st.size(); g += 4) {
                            steps.push_back(make_uint2((unsigned)st[g] | ((unsigned)st[g + 1] << 16), (unsigned)st[g + 2] | ((unsigned)st[g + 3] << 16)));
                            for (int l = 0; l < 64; ++l)
                                frag.push_back(make_float4(fr[(g + 0) * 64 + l], fr[(g + 1) * 64 + l], fr[(g + 2) * 64 + l], fr[(g + 3) * 64 + l]));
                        }
                    }
            tab.back() = (int)steps.size();
            if (steps.empty()) { steps.push_back(make_uint2(0u, 0u)); frag.resize(64, make_float4(0.f, 0.f, 0.f, 0.f)); }
            HIP_TRY(out->wf_tab.upload(tab.data(), tab.size()));
            HIP_TRY(out->wf_step.upload(steps.data(), steps.size()));
            HIP_TRY(out->wf_frag.upload(frag.data(), frag.size()));
        }
    }
    tr.mark("weight fragments");
    HIP_TRY(out->csc_start.upload(csc_start.data(), csc_start.size()));
    HIP_TRY(out->csc_v.upload(csc_v.data(), csc_v.size()));
    HIP_TRY(out->csc_w.upload(csc_w.data(), csc_w.size()));
    HIP_TRY(out->vt.upload(vt.data(), vt.size()));
    HIP_TRY(out->ww.upload(ww.data(), ww.size()));
    HIP_TRY(out->wj.upload(wj.data(), wj.size()));
    tr.mark("uploads");
    out->pn_fwd = PanelB(); out->pn_bwd = PanelB();
    if (nv > 0) {                                  // forward panel for every set (the full mesh takes the wide form of the kernel)
        int nt = 0, ns = 0;
        int e = panel_pack_dev(pd, pd_floats, ldp, 1, NPFX, 3 * nv, out->pn_fwd_f, &nt, &ns);
        if (e) return e;
        out->pn_fwd.f = (const float4*)out->pn_fwd_f.p; out->pn_fwd.ntile = nt; out->pn_fwd.nss = ns;
    }
    if (nv > 0 && 3 * nv <= PANEL_MAX_K) {         // data-gradient panel while a 16-row block of K = 3 nv columns fits the LDS slabs
        int nt = 0, ns = 0;
        int e = panel_pack_dev(pd, pd_floats, 1, ldp, 3 * nv, NPFX, out->pn_bwd_f, &nt, &ns);
        if (e) return e;
        out->pn_bwd.f = (const float4*)out->pn_bwd_f.p; out->pn_bwd.ntile = nt; out->pn_bwd.nss = ns;
    }
    tr.mark("fp32 panels");
    out->pn_fwd3 = PanelB3(); out->pn_bwd3 = PanelB3();
    if (nv > 0) {                                 // the forward operand of every set also as split planes (format PnF)
        int e = pnf_pack_dev(pd, pd_floats, ldp, 1, NPFX, 3 * nv, out->pn_fwd3_f, out->pn_fwd3_s, &out->pn_fwd3.ntile, &out->pn_fwd3.nst);
        if (e) return e;
        out->pn_fwd3.f = (const uint4*)out->pn_fwd3_f.p; out->pn_fwd3.isc = out->pn_fwd3_s.p;
    }
    tr.mark("split forward panel");
    out->pn_fwdS = PanelB3();
    if (nv > 0 && nv <= 512 && K <= 4) {          // ... and with permuted columns (blend_rows_permute_kernel)
        const int nb = (nv + 63) / 64, ncs = nb * 192;
        DevBuf<float> ps;
        { hipError_t ep = ps.ensure((size_t)NPFX * ncs); if (ep != hipSuccess) return (int)ep; }
        hipLaunchKernelGGL(blend_rows_permute_kernel, dim3((ncs + 255) / 256, NPFX), dim3(256), 0, 0, pd, ldp, nv, ncs, ps.p);
        int e = (int)hipGetLastError();
        if (!e) e = pnf_pack_dev(ps.p, (size_t)NPFX * ncs, ncs, 1, NPFX, ncs, out->pn_fwdS_f, out->pn_fwdS_s, &out->pn_fwdS.ntile, &out->pn_fwdS.nst);
        if (!e) e = (int)hipDeviceSynchronize();
        ps.release();
        if (e) return e;
        out->pn_fwdS.f = (const uint4*)out->pn_fwdS_f.p; out->pn_fwdS.isc = out->pn_fwdS_s.p;
    }
    tr.mark("permuted forward panel");
    if (nv > 0) {                                 // ... and the data-gradient operand (one LDS image up to K = 1696: panel_gemm3 / _rb2k; beyond: panel_gemm3_kloop)
        int e = pnf_pack_dev(pd, pd_floats, 1, ldp, 3 * nv, NPFX, out->pn_bwd3_f, out->pn_bwd3_s, &out->pn_bwd3.ntile, &out->pn_bwd3.nst);
        if (e) return e;
        out->pn_bwd3.f = (const uint4*)out->pn_bwd3_f.p; out->pn_bwd3.isc = out->pn_bwd3_s.p;
    }
    HIP_TRY(hipDeviceSynchronize());
    tr.mark("split gradient panel");
    return 0;
}

// dense products on the split formats of fdc_panel.h (FDCAP_GEMM_SPLIT3=0: exact-fp32 MFMA chains instead)
inline bool gemm_split3_enabled() {
    static std::atomic<int> v{-1};
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

// its data gradient: dPF[M, 496] = dVoff[M, 3 nv] x [posedirs ; shapedirs^T]^T.  *split (optional): the product was left as TWO partial
// sums, dPF and dPF + part2_stride (the consumer adds them: pose_bwd_kernel's dPF2); without it the sum is formed here.
hipError_t blend_backward(const SkinSet& ss, const float* dV, int M, float* dPF, size_t part2_stride, DevBuf<float>& ws, hipStream_t st,
                          bool* split = nullptr) {
    TraceRange tr_("fdcap:blend_bwd(K8)");
    if (split) *split = false;
    const int K = 3 * ss.nv;
    if (gemm_split3_enabled() && ss.pn_bwd3.f) {
        if (split && part2_stride && panel_gemm3_rb2k_ok(M, K, ss.pn_bwd3)) {
            *split = true;
            return panel_gemm3_rb2k(dV, K, M, K, ss.pn_bwd3, dPF, part2_stride, NPFX, NPFX, st);
        }
        // r6: the K-loop form also where one LDS image would still fit, from K = 1664 at clip sizes and K = 1904 from 192 rows -- contact
        // sets of 560-840 vertices ran the one-image forms at 17-35 us where the K-loop form takes 14-21 (tools/launch_times.py
        // --per-leg 280 / 320 / 375 / 420 at 1024 / 512 / 256 / 128 rows; at 128 rows the one-image forms stay ahead).  FDCAP_KLOOP_MIN_K: A/B.
        static std::atomic<int> kmin{-1};
        if (kmin < 0) { const char* e = getenv("FDCAP_KLOOP_MIN_K"); kmin = e ? atoi(e) : 1664; }
        const bool big_k = (M >= 384 && K >= kmin) || (M >= 192 && K >= kmin + 240);
        if (!big_k && panel_gemm3_ksw_ok(M, K, ss.pn_bwd3)) return panel_gemm3_ksw(dV, K, M, K, ss.pn_bwd3, dPF, NPFX, NPFX, st);
        if (!big_k && panel_gemm3_fits(K)) return panel_gemm3(dV, K, M, K, ss.pn_bwd3, dPF, NPFX, NPFX, st);
        hipError_t e = ws.ensure((size_t)panel_gemm3_kloop_parts(M, ss.pn_bwd3) * M * NPFX);
        if (e != hipSuccess) return e;
        return panel_gemm3_kloop(dV, K, M, K, ss.pn_bwd3, ws.p, dPF, NPFX, NPFX, st);
    }
    if (ss.pn_bwd.f) return panel_gemm(dV, K, M, K, ss.pn_bwd, dPF, NPFX, NPFX, st);
    return gemm_f32(true, EPI_STORE, dV, K, ss.posedirs.p, ss.ldp, dPF, NPFX, M, NPFX, K, nullptr, 0, st);
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
    lt_mark(o, -1, st);
    int e = vposer_forward(c, o->X.p, XDIM, X_LATENT, lo, hi, o->H1.p, o->H2.p, o->Opart.p, ps, nullptr, st, 0, 0, ds);
    if (e) return e;
    lt_mark(o, FDCAP_LT_VPOSER_FWD, st);
    hipLaunchKernelGGL(pose_fwd_kernel<true>, dim3(hi - lo), dim3(64 * POSE_NW), 0, st, c->pose_model(), o->X.p, o->O.p, o->CAM.p, o->scale.p, lo,
#ifdef FDC_DEBUG_BUFFERS
                       o->Rm.p,                                // (the per-joint rotations: nobody reads them back but fdcap_debug_rows)
#else
                       (float*)nullptr,
#endif
                       contact_state ? o->PF.p : (float*)nullptr, o->Jrest.p, o->G.p, contact_state ? o->A.p : (float*)nullptr, o->M.p, o->Jw.p,
                       (const float*)nullptr, (const float*)o->Opart.p, ps, 0, 0, ds);
    lt_mark(o, FDCAP_LT_POSE_FWD, st);
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
    if (gemm_split3_enabled() && tail.lg.rows && nb >= LROW) {
        tail.lg_spread = 1;                                // the logged sums: one regular workgroup per term (ScaleTail::lg_spread)
        if (tail.n <= 0) tail.block = -1;                  // ... and with no `scale` step there is nothing left for an extra workgroup
    }
    if (gemm_split3_enabled())
        hipLaunchKernelGGL(vposer_bwd_split3_kernel, dim3(nb + (tail.block >= 0 ? 1 : 0)), dim3(512), 0, st, c->vp3, o->dO.p, 2, 2 + nl, o->H1.p, o->H2.p,
                           o->dZpart.p, ps, tail);
    else
        hipLaunchKernelGGL(vposer_bwd_fused_kernel, dim3(nb + (tail.block >= 0 ? 1 : 0)), dim3(512), 0, st, c->vp, o->dO.p, 2, 2 + nl, o->H1.p, o->H2.p,
                           o->dZpart.p, ps, tail);
    lt_mark(o, FDCAP_LT_VPOSER_BWD, st);
    if (fold) {
        hipLaunchKernelGGL(vposer_fold_dz_kernel, dim3((nl * VP_Z + 255) / 256), dim3(256), 0, st, o->dZpart.p, ps, 2, nl, o->dX.p);
        o->dz_pending = false;
    } else {
        o->dz_pending = true;
    }
    return (int)hipGetLastError();
}

}  // namespace
