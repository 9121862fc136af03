// C-ABI, part 4: mode 'local', the optimiser step and results, the exchange inside the library (RCCL), the loop in one call
// (fdcap_opt_run), read-backs and the timing entry points bench.py uses.  Part of csrc/fdcap.hip.
#pragma once

extern "C" {

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
    hipLaunchKernelGGL(vert_smooth_kernel, dim3(VS_NB, nl), dim3(256), 0, st, o->VwF.p, nv3, 2, cf.frame0, N, w_vs,
                       o->dVF.p, o->losses.p + 5);
    if (N >= 2)
        hipLaunchKernelGGL(foot_skate_kernel, dim3(1, nl), dim3(256), 0, st, o->VwF.p, nv3, c->contact_vid.p,
                           nc, n_left, contact_weight, 2, cf.frame0, N, o->dVF.p, o->losses.p + 6);
    { int es = skin_bwd_any<false>(c->ws_skin, st, nl, c->full.model(), V, o->X.p, o->VoffF.p, o->A.p, o->M.p, o->scale.p, 2, o->dVF.p, o->dVF.p,
                                   o->dA.p, (float*)nullptr, o->dtransl_v.p, o->dMv.p, o->dsv.p, ContactGradIn()); if (es) return es; }
    HIP_TRY(blend_backward(c->full, o->dVF.p + 2 * nv3, nl, o->dPF.p + 2 * NPFX, 0, c->ws_kpart, st));
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

// What the exchange costs on THIS group, measured (DESIGN 6 assumed 8 / 10 / 12 us for 2 / 4 / 8 ranks until a multi-GPU box shows up):
// `iters` whole tails (fdcap_opt_exchange: Adam on the rows + message, all-gather, unpack + `scale` step) and `iters` bare all-gathers of
// the message, each train between two HIP events on `stream`; mean microseconds of each.  EVERY rank of the communicator must make the
// same call.  The optimiser's parameters and moments are stepped `iters` times with whatever gradients are there: call it after the fit.
int fdcap_opt_time_exchange(fdcap_ctx* c, int32_t iters, float* us_exchange, float* us_allgather, void* stream) {
    if (!c || !c->opt || iters <= 0 || !us_exchange || !us_allgather) return FDCAP_E_ARG;
    if (!c->comm.comm) return FDCAP_E_STATE;
    hipStream_t st = (hipStream_t)stream;
    int e = comm_buffers(c);
    if (e) return e;
    hipEvent_t ev[4];
    for (auto& x : ev) HIP_TRY(hipEventCreate(&x));
    const int big = 1 << 30;                                  // (P: every iteration is a phase-1 iteration -- `scale` steps, camera_ext rests)
    for (int k = 0; k < 8 && !e; ++k) e = fdcap_opt_exchange(c, k, big, stream);              // warm-up
    if (!e) e = (int)hipEventRecord(ev[0], st);
    for (int k = 0; k < iters && !e; ++k) e = fdcap_opt_exchange(c, 8 + k, big, stream);
    if (!e) e = (int)hipEventRecord(ev[1], st);
    if (!e) e = (int)hipEventRecord(ev[2], st);
    for (int k = 0; k < iters && !e; ++k) {
        const ncclResult_t r = rccl().AllGather(c->xch_send.p, c->xch_all.p, XCH_LEN, ncclFloat, c->comm.comm, st);
        if (r != ncclSuccess) e = comm_fail(c, r, "ncclAllGather");
    }
    if (!e) e = (int)hipEventRecord(ev[3], st);
    if (!e) e = (int)hipEventSynchronize(ev[3]);
    float a = 0.f, b = 0.f;
    if (!e) e = (int)hipEventElapsedTime(&a, ev[0], ev[1]);
    if (!e) e = (int)hipEventElapsedTime(&b, ev[2], ev[3]);
    for (auto& x : ev) (void)hipEventDestroy(x);
    *us_exchange = a * 1e3f / iters;
    *us_allgather = b * 1e3f / iters;
    return e;
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
        if (!(e3 && e3[0] == '0') && panel_gemm3_fits(K)) {          // the split form of the same product on the 16-bit matrix cores (format PnF; the default)
            std::vector<unsigned> p3;
            std::vector<float> sc;
            PanelB3 B3;
            PnF::pack(B_h, (long)sk, (long)sn, K, N, p3, sc, &B3.ntile, &B3.nst);
            DevBuf<unsigned> d3;
            DevBuf<float> ds;
            HIP_TRY(d3.upload(p3.data(), p3.size()));
            HIP_TRY(ds.upload(sc.data(), sc.size()));
            B3.f = (const uint4*)d3.p; B3.isc = ds.p;
            hipError_t e = panel_gemm3(A, lda, M, K, B3, C, ldc, N, st);
            hipError_t e2 = hipStreamSynchronize(st);
            d3.release(); ds.release();
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

int fdcap_opt_launch_timing(fdcap_ctx* c, int32_t max_events) {
    if (!c || !c->opt || max_events < 0) return FDCAP_E_ARG;
    OptState::LaunchTimes& t = c->opt->lt;
    t.on = max_events > 0;
    t.used = 0;
    while ((int)t.ev.size() < max_events) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        t.ev.push_back(e);
    }
    t.what.resize(t.ev.size(), (signed char)-1);
    return FDCAP_OK;
}
int fdcap_opt_launch_timing_read(fdcap_ctx* c, float* mean_us, int32_t* counts) {
    if (!c || !c->opt || !mean_us || !counts) return FDCAP_E_ARG;
    OptState::LaunchTimes& t = c->opt->lt;
    double sum[2 * FDCAP_LT_NUM] = {0.0};
    for (int i = 0; i < 2 * FDCAP_LT_NUM; ++i) counts[i] = 0;
    if (t.used > 0) HIP_TRY(hipEventSynchronize(t.ev[t.used - 1]));
    for (int k = 1; k < t.used; ++k) {
        const int w = t.what[k];
        if (w < 0) continue;                                 // (from the previous iteration's last launch to this one's first event: host time)
        const int slot = (w & 15) + FDCAP_LT_NUM * (w >> 4);
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, t.ev[k - 1], t.ev[k]));
        sum[slot] += ms * 1e3;
        counts[slot] += 1;
    }
    for (int i = 0; i < 2 * FDCAP_LT_NUM; ++i) mean_us[i] = counts[i] ? (float)(sum[i] / counts[i]) : 0.f;
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
