/* fdcap.h -- C-ABI of libfdcap_hip.so: the MI355X (gfx950) hot path of
 * aptx4869lm/4DCapture-FPV `global_optimization.py` FittingOP.fitting(mode='global').
 *
 * The reference has no FFI layer; its boundary is three third-party Python operators plus the
 * optimiser loop around them.  Each entry point below names the reference
 * call site (file:line in /root/reference) it replaces.  Binding stub: INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no torch types; every `*_d` / "device" pointer is a caller-owned HIP device
 *     pointer (a torch tensor's storage), never freed or retained beyond the call unless the
 *     function says "registered";
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued on it, nothing synchronises unless stated;
 *   - return value: 0 ok, negative FDCAP_E_* bad argument / state, positive = hipError_t;
 *   - one context per device per clip; calls on one context are serialised by the caller
 *     (the reference is single-threaded, single-stream);
 *   - fp32 everywhere (reference dtype: global_optimization.py:175,:185,:224,:707), int32 /
 *     int64 only for indices.
 */
#ifndef FDCAP_H
#define FDCAP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDCAP_OK 0
#define FDCAP_E_ARG (-1)      /* null pointer / bad size */
#define FDCAP_E_STATE (-2)    /* call order (e.g. no scene registered) */
#define FDCAP_E_NODEVICE (-3) /* no HIP device visible */
#define FDCAP_E_COMM (-4)     /* RCCL not loadable, or an RCCL call failed: fdcap_comm_last_error() */
#define FDCAP_UNIQUE_ID_BYTES 128

#define FDCAP_NUM_JOINTS 55
#define FDCAP_XDIM 78         /* optimised row: transl3 6D6 betas10 latent32 lh12 rh12 camt3 */
#define FDCAP_PDIM 75         /* file row:      transl3 aa3 betas10 latent32 lh12 rh12 camt3 */
#define FDCAP_MAX_SCENE_POINTS 67000000   /* fdcap_set_scene refuses more (FDCAP_E_ARG): 32 B of MFMA fragments per point, 32-bit offsets */
#define FDCAP_NUM_LOSSES 8    /* rec, vposer, smoothing, contact, world_smoothing, total, 2 spare */

typedef struct fdcap_ctx fdcap_ctx;

/* Host arrays (copied to the device by fdcap_ctx_create).  Names follow the SMPL-X npz keys
 * read by smplx.create (global_optimization.py:154-168) and the VPoser v1 state-dict keys read
 * by load_vposer (:153). */
typedef struct fdcap_model_desc {
    int32_t num_verts;            /* V (10475 for SMPL-X) */
    const float* v_template;      /* [V,3] */
    const float* shapedirs;       /* [V,3,num_shape] first 10 = betas (num_betas=10, :156-159) */
    int32_t num_shape;            /* >= 10; columns >= 10 are expression (zero in the reference) */
    const float* posedirs;        /* [486, V*3] (library layout) */
    const float* J_regressor;     /* [55,V] */
    const int32_t* parents;       /* [55], root -1 */
    const float* lbs_weights;     /* [V,55] */
    const float* hands_componentsl; /* [12,45] first num_pca_comps=12 rows */
    const float* hands_componentsr; /* [12,45] */
    const float* hands_meanl;     /* [45] (flat_hand_mean=False) */
    const float* hands_meanr;     /* [45] */
    const float* vp_fc1_w;        /* bodyprior_dec_fc1.weight [512,32] */
    const float* vp_fc1_b;        /* [512] */
    const float* vp_fc2_w;        /* bodyprior_dec_fc2.weight [512,512] */
    const float* vp_fc2_b;        /* [512] */
    const float* vp_out_w;        /* bodyprior_dec_out.weight [126,512] */
    const float* vp_out_b;        /* [126] */
} fdcap_model_desc;

/* Optimiser configuration = fittingconfig / lossconfig (global_optimization.py:663-686) plus
 * the constants buried in fitting() (:564, :570, :582). */
typedef struct fdcap_opt_config {
    int32_t n_total;        /* frames in the whole clip (mean denominators) */
    int32_t n_local;        /* frames owned by this rank */
    int32_t frame0;         /* global index of the first owned frame */
    float lr;               /* init_lr_h = 0.005 (:671) */
    float weight_loss_rec;  /* 1 (:682) */
    float weight_loss_vposer; /* 0.001 (:683), logged only */
    float weight_contact;   /* 0.1 (:684) */
    float phase1_contact;   /* 0.1  (:570) */
    float phase1_smooth;    /* 1.0  (:570) */
    float phase2_world;     /* 1.0  (:582) */
    float phase2_smooth;    /* 0.5  (:582) */
    float scale_init;       /* 1.8  (:179) */
    int32_t legacy_zero_grad; /* 0: torch>=2 zero_grad(set_to_none=True) semantics */
} fdcap_opt_config;

/* ---- context ------------------------------------------------------------------------------ */
/* Replaces FittingOP.__init__'s model construction (global_optimization.py:153-171). */
int fdcap_ctx_create(const fdcap_model_desc* model, fdcap_ctx** out);
void fdcap_ctx_destroy(fdcap_ctx* ctx);
const char* fdcap_version(void);
/* "packed_fp32=off" for a product build (compiled without v_pk_*_f32, see csrc/fdcap.hip's build requirement); the Python
 * binding refuses a library that says otherwise unless FDCAP_ALLOW_PK_F32=1 (instrumentation variants). */
const char* fdcap_build_info(void);

/* Scene vertices, stored ONCE (the reference repeats them per frame, :175-176).  `scene_xyz`
 * is a HOST pointer [ns,3]; registered (copied, sorted into k-d cells and packed for the NN kernel -- on the device, one
 * synchronous call); ns <= FDCAP_MAX_SCENE_POINTS.
 * Both setters return FDCAP_E_STATE while an optimiser exists on the context (fdcap_opt_create .. fdcap_opt_destroy): its
 * buffers are sized for the registered sets and its pruning state (seeds, kept work lists) is only valid for them. */
int fdcap_set_scene(fdcap_ctx* ctx, const float* scene_xyz, int64_t ns);
/* Diagnosis: out8[0..7] = FNV-1a hashes of the registered scene's device tables (input-order points, cell-ordered points, inverse
 * permutation, cell / quarter-cell / super-cell boxes, MFMA fragments, cell centres).  The tables are built on the device since r6
 * (csrc/fdc_scene.h); with FDCAP_SCENE_BUILD=host in the environment fdcap_set_scene takes the cell order from the host recursion of
 * r1-r5 instead -- the same order by specification, which tests check by comparing these hashes.  Synchronises the device. */
int fdcap_debug_scene_hash(fdcap_ctx* ctx, uint64_t* out8);
/* Diagnosis: names of the kernel FORMS launched by this process since the last reset, "a;b;c" (several stages pick among forms by
 * row count and set size: blend products, contact forward, skinning backward, the Chamfer search).  Tests that mean to cover a form
 * check that it ran.  FDCAP_CLIP_FORMS_MIN_ROWS (environment, read once per process) lowers the row count from which the
 * clip-sized forms are selected (default 336): the reference's 300-frame fixtures then run through them. */
int fdcap_debug_kernel_forms(char* buf, int32_t len, int32_t reset);
/* Contact vertex ids = get_contact_id(...) (global_optimization.py:79-94, :288); HOST pointer. */
int fdcap_set_contact_ids(fdcap_ctx* ctx, const int64_t* vid, int32_t nc);

/* ---- Op 1: Chamfer (ext.chamferDist()(xyz1, xyz2), global_optimization.py:292-294) -------- */
/* xyz1_d [B,n,3] queries, xyz2_d [B,m,3] targets with batch stride `stride2` elements
 * (0 = one scene shared by all batches).  Writes dist1_d [B,n] (squared L2 to the nearest
 * target, direct-difference form) and idx1_d [B,n] (lowest index among ties).  If dist2_d is
 * non-null also the reverse direction dist2_d [B,m], idx2_d [B,m] (the reference discards it). */
int fdcap_chamfer_fwd(fdcap_ctx* ctx, const float* xyz1_d, const float* xyz2_d, int32_t B,
                      int32_t n, int32_t m, int64_t stride2, float* dist1_d, int32_t* idx1_d,
                      float* dist2_d, int32_t* idx2_d, void* stream);
/* The same operator when xyz2 IS the scene registered with fdcap_set_scene -- the call site :292-294 passes the one scene in every
 * iteration of the caller's loop.  dist1_d / idx1_d [B,n] as fdcap_chamfer_fwd writes them, bit for bit (idx: original scene
 * indices, lowest among ties), found by the optimiser loop's search instead of visiting every pair: k-d-sorted scene, cell boxes,
 * seeds from the previous call's neighbours while B * n stays the same (first call / forget != 0: cheap fresh seeds), kept work
 * lists.  Library-owned state, one per context: calls on one context are serialised by the caller as everywhere.  FDCAP_E_STATE
 * without a registered scene.  fdcap_chamfer_bwd_scene: the matching gradient wrt the queries (fdcap_chamfer_bwd's formula, the
 * scene points read from the library's copy).  fdcap_chamfer_fwd itself accepts dist1_d == NULL (only the reverse direction). */
int fdcap_chamfer_fwd_scene(fdcap_ctx* ctx, const float* xyz1_d, int32_t B, int32_t n, float* dist1_d, int32_t* idx1_d, int32_t forget,
                            void* stream);
int fdcap_chamfer_bwd_scene(fdcap_ctx* ctx, const float* xyz1_d, int32_t B, int32_t n, const float* gdist1_d, const int32_t* idx1_d,
                            float* gxyz1_d, void* stream);
/* grad wrt the queries only (the scene needs none): gxyz1_d[b,i] = 2 g1[b,i] (x1[b,i]-x2[b,idx]) */
int fdcap_chamfer_bwd(fdcap_ctx* ctx, const float* xyz1_d, const float* xyz2_d, int32_t B,
                      int32_t n, int32_t m, int64_t stride2, const float* gdist1_d,
                      const int32_t* idx1_d, float* gxyz1_d, void* stream);

/* Nearest-neighbour kernel selection for A/B measurement: 0 = by size (default), 1 = plain VALU
 * scan (nn_direct_kernel), 2 = MFMA-filtered exact scan (nn_mfma_kernel).  Both give bit-identical
 * results.  Process-wide. */
int fdcap_set_nn_kernel(int32_t mode);

/* ---- Op 3: VPoser decode (self.vposer.decode(z,'aa'), global_optimization.py:270-271) ----- */
/* z_d [B,32] with row stride ldz -> rot_d [B,21,9] rotation matrices; aa_d (optional) [B,63]. */
int fdcap_vposer_decode(fdcap_ctx* ctx, const float* z_d, int32_t ldz, int32_t B, float* rot_d,
                        float* aa_d, void* stream);
/* Its backward -- what loss.backward() (:591) runs through the reference's vposer.decode: gradients of the rotation
 * matrices g_rot_d [B,21,9] and / or of the angle-axis output g_aa_d [B,63] (either may be NULL, not both; the aa path
 * goes through torchgeometry's rotation_matrix_to_angle_axis, cvae.py:83) -> g_z_d [B,32].  The operator keeps no
 * state: the decoder's activations are recomputed from z_d. */
int fdcap_vposer_decode_bwd(fdcap_ctx* ctx, const float* z_d, int32_t ldz, int32_t B, const float* g_rot_d,
                            const float* g_aa_d, float* g_z_d, void* stream);

/* ---- Op 2: body model (self.body_mesh_model(...), global_optimization.py:280-283) ---------- */
/* params_d [B,75] file layout rows (transl, global_orient aa, betas, latent, lh, rh, cam_t);
 * runs VPoser + SMPL-X and writes vertices_d [B,V,3] and joints_d [B,55,3] (either optional),
 * body frame, `+transl` applied, no scale / world transform. */
int fdcap_body_forward(fdcap_ctx* ctx, const float* params_d, int32_t B, float* vertices_d,
                       float* joints_d, void* stream);
/* World-space mesh of saved results, as global_vis.py:126-152 builds it per frame on the CPU:
 * vertices_d [B,V,3] = camera_ext_b @ T(cam_t_b * scale) applied to scale * (SMPL-X(VPoser(row_b)) + transl).
 * params_d [B,75] rows as saved (:633), cam_ext_d [B,16], scale_d [1] device scalar. */
int fdcap_world_mesh(fdcap_ctx* ctx, const float* params_d, int32_t B, const float* cam_ext_d,
                     const float* scale_d, float* vertices_d, void* stream);
/* The same with the operator's own argument list (:280-283): global_orient_d [B,3] and
 * body_pose_d [B,63] axis-angle (Rodrigues as smplx.lbs.batch_rodrigues), betas_d [B,10],
 * left/right_hand_pose_d [B,12] PCA coefficients, transl_d [B,3]. */
int fdcap_smplx_forward(fdcap_ctx* ctx, const float* global_orient_d, const float* body_pose_d,
                        const float* betas_d, const float* left_hand_pose_d,
                        const float* right_hand_pose_d, const float* transl_d, int32_t B,
                        float* vertices_d, float* joints_d, void* stream);
/* Its backward (loss.backward() through self.body_mesh_model(...), :280-283 / :591): gradients of the outputs
 * g_vertices_d [B,V,3] and / or g_joints_d [B,55,3] (either may be NULL, not both) -> gradients of the six inputs (any
 * may be NULL).  Same inputs as the forward; its state is recomputed.  LBS, the K = 3V blend-shape data gradient, the
 * reverse kinematic chain and Rodrigues' backward are the kernels of mode 'local''s full-mesh backward. */
int fdcap_smplx_backward(fdcap_ctx* ctx, const float* global_orient_d, const float* body_pose_d,
                         const float* betas_d, const float* left_hand_pose_d,
                         const float* right_hand_pose_d, const float* transl_d, int32_t B,
                         const float* g_vertices_d, const float* g_joints_d, float* g_global_orient_d,
                         float* g_body_pose_d, float* g_betas_d, float* g_left_hand_pose_d,
                         float* g_right_hand_pose_d, float* g_transl_d, void* stream);

/* ---- parameter-vector conversions (global_optimization.py:96-115, cvae.py:62-93) ----------- */
int fdcap_params_75_to_78(const float* p75_d, int32_t B, float* x78_d, void* stream);
int fdcap_params_78_to_75(const float* x78_d, int32_t B, float* p75_d, void* stream);

/* ---- the optimiser loop (FittingOP.init + fitting('global'), :450-489, :558-593) ----------- */
/* Allocates scratch for n_local (+2 halo rows each side) frames and REGISTERS the caller-owned
 * optimiser state (kept alive by the caller until fdcap_opt_destroy; a multi-GPU caller hands the
 * same tensors to RCCL):
 *   rows_x_d   [n_local+4,78] body_rotation_rec (:180) with 2 halo rows each side; owned rows start at 2
 *   rows_cam_d [n_local+4,16] camera_ext (:182), same row layout
 *   scale_d    [1]  scale (:179); set to cfg->scale_init here
 *   dscale_d   [1]  this rank's d loss / d scale (sum over owned frames): written by fdcap_opt_step /
 *              fdcap_opt_step_rows_and_pack (fused with the update) and, when log_terms != 0, by the backward
 *   losses_d   [FDCAP_NUM_LOSSES] double: this rank's un-normalised partial sums of the last
 *              backward that ran with log_terms != 0 (other iterations do not form them):
 *              [0] sum|x0-x|*mask  [1] sum z^2  [2] sum|2nd diff|  [3] sum r/(r+1)
 *              [4] sum|Jw_i-Jw_{i+1}| */
int fdcap_opt_create(fdcap_ctx* ctx, const fdcap_opt_config* cfg, float* rows_x_d, float* rows_cam_d,
                     float* scale_d, float* dscale_d, double* losses_d);
/* data78_d [n_local,78]: the 6D-converted SMPLify-X rows (loss_rec target);
 * init78_d [n_local,78]: initial value of body_rotation_rec (= data with outlier rows replaced, :487);
 * mask_d   [n_local]   : 0 for outlier rows (idx1), 1 otherwise (:255-257);
 * cam_ext_d[n_local,16]: extract_ext() (:455).  All copied. */
int fdcap_opt_set_inputs(fdcap_ctx* ctx, const float* data78_d, const float* init78_d,
                         const float* mask_d, const float* cam_ext_d, void* stream);
/* One pass of the loop body :562-592 for iteration `ii` of `num_iter`, split in two so a
 * multi-GPU caller can all-reduce the scalar `scale` gradient in between:
 *   backward: zero_grad + cal_loss + loss.backward()  -> gradients + loss partial sums
 *   step    : optimizer.step() (fused Adam over body_rotation_rec, scale, camera_ext)
 * `phase2` = (ii >= 0.8*num_iter) decides the loss total; the requires_grad toggling of
 * :564-568/:577-580 (effective one forward late) is reproduced from ii and first_phase2_iter.
 * log_terms != 0: also evaluate what the reference only prints (:573-575, :587-589) -- the loss partial
 * sums in losses_d, the contact term in phase 2 -- and leave dscale_d valid after the backward.
 * log_terms == 2: the same terms, but losses_d is complete only after the fdcap_opt_step / fdcap_opt_step_rows_and_pack call
 * that follows (the fixed-order reduction of the per-frame partial sums rides in that launch: one launch less per logged
 * iteration; dscale_d as on a non-logging iteration). */
int fdcap_opt_backward(fdcap_ctx* ctx, int32_t ii, int32_t first_phase2_iter, int32_t log_terms,
                       void* stream);
int fdcap_opt_step(fdcap_ctx* ctx, int32_t ii, int32_t first_phase2_iter, void* stream);
/* loss.backward() + optimizer.step() of iteration ii (:591-592) in ONE call and without a launch for the step (r4): `scale` is
 * stepped by one more workgroup of the backward's last launch, the rows of body_rotation_rec / camera_ext take their update in
 * the first two launches of the next backward, where they are read anyway -- same arithmetic, same bits as fdcap_opt_backward
 * followed by fdcap_opt_step.  Until then rows_x_d / rows_cam_d and their Adam moments hold the PRE-step values: every other entry
 * point of this group applies a still-pending update first (an ordinary Adam launch), and a caller that reads the registered
 * buffers itself calls fdcap_opt_sync before it does.  log_terms == 2 here: the printed sums are reduced by that same extra
 * workgroup (losses_d complete when this call's work is).  Sharded runs and mode 'dct' take the two-call path, silently. */
int fdcap_opt_backward_and_step(fdcap_ctx* ctx, int32_t ii, int32_t first_phase2_iter, int32_t log_terms, void* stream);
int fdcap_opt_sync(fdcap_ctx* ctx, void* stream);
/* The loop :560-593 itself -- iterations [ii0, ii1) of a fit of num_iter iterations -- in ONE call (r4): the sequence of the calls
 * above that FittingOP.fitting issues (fdcap_opt_backward_and_step; the fit's last iteration as fdcap_opt_backward + fdcap_opt_step;
 * on a sharded context, which must hold a communicator, fdcap_opt_backward + fdcap_opt_exchange), so the same bits.  Logging
 * iterations (log_every > 0: ii % log_every == 0, and ii == num_iter - 1; the reference prints every iteration) leave their partial
 * sums in consecutive rows of hist_d [hist_rows][FDCAP_NUM_LOSSES] on the device, *n_logged of them, with no host sync; the output
 * registered before the call is registered again after it.  flags bit 0: every optimiser step as its own launch; bit 1: the
 * exchange tail although the context holds the whole clip (a communicator of one rank).  Returns when the
 * work is enqueued.  A caller with something to do between iterations (snapshots, checkpoints, a finite check) calls it per stretch. */
int fdcap_opt_run(fdcap_ctx* ctx, int32_t ii0, int32_t ii1, int32_t num_iter, int32_t first_phase2_iter, int32_t log_every,
                  double* hist_d, int32_t hist_rows, int32_t flags, int32_t* n_logged, void* stream);
/* Checkpoint / resume (the reference only ever writes its final result, :637-653).  The parameters live in the
 * caller's registered tensors; these move the rest of the optimiser state -- Adam's moments of the owned rows:
 * state_d [fdcap_opt_state_len()] floats = [m_x n_local*78 | v_x | m_cam n_local*16 | v_cam | m_scale | v_scale].
 * The step counters are functions of the iteration index the caller passes to fdcap_opt_step; seeds and kept work lists of
 * the Chamfer search are pruning state only (results never depend on them) and are not part of a checkpoint.  Mode 'dct''s
 * c_dct moments are not covered. */
int32_t fdcap_opt_state_len(fdcap_ctx* ctx);
int fdcap_opt_export_state(fdcap_ctx* ctx, float* state_d, void* stream);
int fdcap_opt_import_state(fdcap_ctx* ctx, const float* state_d, void* stream);
/* count_d [1] int32 <- number of non-finite values among the owned rows of body_rotation_rec / camera_ext and scale
 * (the reference wraps every iteration in torch.autograd.set_detect_anomaly(True), :561, at ~4x the host cost; this is
 * the opt-in equivalent: FittingOP.fitting(check_finite_every=k)). */
int fdcap_opt_check_finite(fdcap_ctx* ctx, int32_t* count_d, void* stream);
/* Redirect the loss partial sums of the following logging backwards to another [FDCAP_NUM_LOSSES] double buffer (e.g. the
 * next row of a device-side history, so that a caller logging every iteration -- as the reference prints every
 * iteration, :573-575 -- needs no copy and no host sync inside the loop).  Host-side only; nothing is launched. */
int fdcap_opt_set_loss_output(fdcap_ctx* ctx, double* losses_d);
/* ---- mode 'local' (:499-556).  Its first loop is fdcap_opt_backward / _step with
 * phase1_contact = 0.2 and phase2_world = 0 in the config (:511, :523); then: */
/* detect_contact (:315-365): weight_left_d [n_local] = left / (left + left) per frame, `left` = mean
 * squared NN distance of the first n_left contact ids (the L_Leg part) -- the reference's own formula,
 * identically 0.5 (NaN where the distance is 0). */
int fdcap_opt_detect_contact(fdcap_ctx* ctx, int32_t n_left, float* weight_left_d, void* stream);
/* zero_grad + cal_loss2 + backward (:368-447, :538-554): loss = vertex-space second-difference smoothing
 * over ALL mesh vertices + parameter smoothing + loss_rec + foot-skate term; only body_rotation_rec gets
 * a gradient.  contact_weight_d [n_total]: detect_contact's output for the WHOLE clip (all ranks).
 * losses_d afterwards: [0] rec, [2] parameter smoothing, [5] vertex smoothing, [6] foot-skate (already
 * normalised per part). */
int fdcap_opt_backward_local2(fdcap_ctx* ctx, const float* contact_weight_d, int32_t n_left, void* stream);
/* Adam on body_rotation_rec only, with its running step count `step` (continues after the first loop). */
int fdcap_opt_step_x(fdcap_ctx* ctx, int32_t step, void* stream);

/* ---- mode 'dct' (global_optimization.py:595-630) ---------------------------------
 * cal_dctloss (:232-246) over W = n_total / T windows of T frames (reference: 5 x 60, :41-42) and the 23
 * world joints x 3 axes; c_dct [W,23,3,C] and its Adam moments are library-owned.
 * fdcap_opt_set_dct: dct_mtx HOST [T,C] (= load_dct_base(), :131-136; T <= 64, C <= 8), c_dct_d DEVICE
 * [W,23,3,C] initial coefficients (the reference draws them with torch.randn, :186).  After fdcap_opt_create. */
int fdcap_opt_set_dct(fdcap_ctx* ctx, const float* dct_mtx, int32_t T, int32_t C, const float* c_dct_d, void* stream);
/* The first phase of the loop (:601-613): body_rotation_rec / scale / camera_ext are frozen, loss =
 * weight * loss_dct (weight = 10, :607), so `iters` Adam iterations on c_dct (step counters step0+1 ...)
 * run in one launch against the fixed world-joint trajectories of the current state.  Only windows that
 * lie inside this rank's frames are fitted.  obj_hist_d (optional) [ceil(iters/log_stride), 69*(w1-w0)]:
 * each fitted trajectory's objective (sum over the window of e/(e+1)) before the update of iterations
 * 0, log_stride, 2*log_stride ...
 * weight = 0: every gradient is exactly zero, so c_dct coasts on its Adam moments -- what torch < 2's
 * zero_grad() (grads zeroed, not None) does to the frozen c_dct from iteration ceil(0.95 num_iter) on
 * (legacy_zero_grad = 1 callers issue one such iteration per loop iteration); no forward runs. */
int fdcap_opt_dct_fit(fdcap_ctx* ctx, int32_t iters, int32_t step0, float weight, float* obj_hist_d,
                      int32_t log_stride, void* stream);
/* The second phase (:614-626): zero_grad + cal_loss + backward of
 *   loss = w_dct * loss_dct + w_rec * loss_rec + w_contact * loss_contact   (1e-4, 0.5, 0.1 at :620);
 * body_rotation_rec and scale get gradients, c_dct / camera_ext do not.  Step with fdcap_opt_step(ctx, k,
 * INT32_MAX, ...) (k = 0, 1, ...: both Adam step counters start at 1 here) or the multi-GPU pair below.
 * losses_d afterwards: [0] [1] [3] as fdcap_opt_backward, [7] = un-normalised sum of e/(e+1) over this
 * rank's frames (loss_dct = sum / (69 W)). */
int fdcap_opt_backward_dct(fdcap_ctx* ctx, float w_dct, float w_rec, float w_contact, int32_t log_terms, void* stream);
/* c_dct_d [W,23,3,C] <- current coefficients (a sharded caller merges the windows each rank fitted);
 * fdcap_opt_set_dct_coef writes merged coefficients back without touching the Adam moments. */
int fdcap_opt_set_dct_coef(fdcap_ctx* ctx, const float* c_dct_d, void* stream);
int fdcap_opt_get_dct(fdcap_ctx* ctx, float* c_dct_d, void* stream);
/* Adam's two moments of c_dct, DEVICE [W,69,C] each (zero after fdcap_opt_set_dct): with fdcap_opt_get_dct and
 * fdcap_opt_export_state the whole optimiser state of mode 'dct' -- a fit continued from them (fdcap_opt_dct_fit with step0 = the
 * iterations already made) ends on the uninterrupted fit's bits. */
int fdcap_opt_get_dct_state(fdcap_ctx* ctx, float* m_d, float* v_d, void* stream);
int fdcap_opt_set_dct_state(fdcap_ctx* ctx, const float* m_d, const float* v_d, void* stream);
/* Returns W; *w0 / *w1 = first / one-past-last window this rank fits (0 when fdcap_opt_set_dct has not run). */
int32_t fdcap_opt_dct_windows(fdcap_ctx* ctx, int32_t* w0, int32_t* w1);

/* ---- optimization.py: the per-frame smoother (:185-238, driver loop :334-348) -------
 * data78_d [N,78] = convert_to_6D_rot of the SMPLify-X rows in file order; for every frame `iters` (50, :313)
 * Adam iterations (lr 0.1, :312) on one 78-d row started at the data row, with
 *   loss = w_rec * L1(data, x) + w_vposer * mean(x[19:51]^2) [+ w_prev * L1(previous result[9:51], x[9:51])]
 * (1, 0.001, 5: :197, :227, :323-324); frame 0 has no previous-frame term (:185-208).  torch's optimiser
 * object is created once (:126), so Adam's moments and step counter carry over from frame to frame -- kept.
 * out78_d [N,78]: the optimised rows (fdcap_params_78_to_75 gives the saved layout, :206, :236).
 * state_d (optional) DEVICE [3,78] = Adam m | v | previous frame's result: read when step0 > 0 or has_prev,
 * always written, so a caller can go file by file like the reference's driver; step0 = Adam steps taken so
 * far (frames done x iters), has_prev = the first row of this call has a predecessor.  `ctx` only lends a
 * workspace and may be NULL (the body model plays no part, although the reference loads it, :106-123); the
 * call then synchronises `stream` before it returns. */
int fdcap_frame_smoother(fdcap_ctx* ctx, const float* data78_d, int32_t N, int32_t iters, float lr, float w_rec,
                         float w_vposer, float w_prev, float* state_d, int32_t step0, int32_t has_prev,
                         float* out78_d, void* stream);

/* ---- per-frame inner fit with a true 2D-keypoint reprojection residual (BASELINE config 4) ----
 * NOT part of the reference repository: there the per-frame fit is the external SMPLify-X step (README.md:14-17) and
 * the only in-repo projection is a viewer overlay (local_vis.py:368-378; intrinsics fx = fy = 692, cx = 640, cy = 360,
 * vis.py:358-360).  Objective restated from the published SMPLify-X data term and L2 priors (csrc/fdc_fit2d.h);
 * frames are independent.  Use: fdcap_opt_create (scale_init = 1), fdcap_opt_set_inputs (camera_ext rows = identity, so
 * the "world" joints are camera-frame joints + camera_translation), fdcap_opt_set_keypoints, then per iteration
 * fdcap_opt_backward_fit2d + fdcap_opt_step_x; fdcap_opt_reset_adam between stages; fdcap_opt_get_results. */
typedef struct fdcap_fit2d_stage {
    float fx, fy, cx, cy;   /* pinhole intrinsics */
    float rho;              /* GMoF scale in pixels (SMPLify-X: 100) */
    float w_data, w_pose, w_shape, w_hand;   /* stage weights (enter squared) */
} fdcap_fit2d_stage;
/* kp_d DEVICE [n_local,23,3]: (u, v, confidence) of SMPL-X joints 0..22 (the joints[:, 0:23] the reference reads, :298). */
int fdcap_opt_set_keypoints(fdcap_ctx* ctx, const float* kp_d, void* stream);
/* zero_grad + loss + backward; only body_rotation_rec gets a gradient.  log_terms != 0: losses_d [0] = data term,
 * [1] = priors (both already weighted, summed over this rank's frames).  (The latent columns' gradient is left in the VPoser
 * backward's four partial arrays; fdcap_opt_step_x and fdcap_opt_get_grads add them -- one launch less per iteration.) */
int fdcap_opt_backward_fit2d(fdcap_ctx* ctx, const fdcap_fit2d_stage* stage, int32_t log_terms, void* stream);
int fdcap_opt_reset_adam(fdcap_ctx* ctx, void* stream);

/* ---- batched L-BFGS with a strong-Wolfe line search (csrc/fdc_lbfgs.h) ----
 * SMPLify-X fits every frame with L-BFGS (its `optimizer.step(closure)` loop with line_search_fn = "strong_wolfe"), not Adam;
 * neither is in the reference repository (README.md:14-17 delegates the per-frame fit).  The algorithm is the published one
 * of torch.optim.LBFGS, restated as a resumable state machine per problem: n_problems INDEPENDENT problems of `dim` <= 128
 * unknowns advance together, one launch between two evaluations of the caller's objective.
 *   max_iter / max_eval / history / lr / tolerance_grad / tolerance_change: torch.optim.LBFGS's arguments (max_eval <= 0:
 *       max_iter * 5 / 4; history <= 128);  max_ls: evaluations per line search (torch: 25);
 *   max_steps, ftol, gtol: SMPLify-X's loop around optimizer.step(): at most max_steps calls, stop when the loss at the start
 *       of two successive calls differs by <= ftol relative to max(|a|, |b|, 1) or every gradient entry is below gtol
 *       (max_steps = 1, ftol = gtol = 0: exactly one optimizer.step()). */
typedef struct fdcap_lbfgs_config {
    int32_t dim, history, max_iter, max_eval, max_steps, max_ls;
    float lr, tolerance_grad, tolerance_change, ftol, gtol;
} fdcap_lbfgs_config;
typedef struct fdcap_lbfgs fdcap_lbfgs;
int fdcap_lbfgs_create(int32_t n_problems, const fdcap_lbfgs_config* cfg, fdcap_lbfgs** out);
void fdcap_lbfgs_destroy(fdcap_lbfgs* opt);
/* forget everything (a fresh optimiser, as SMPLify-X builds for every stage) */
int fdcap_lbfgs_reset(fdcap_lbfgs* opt, void* stream);
/* One round.  In: f_d DEVICE [n_problems], g_d DEVICE [n_problems rows of g_stride floats] = objective and gradient at the
 * points in x_d DEVICE [n_problems rows of x_stride floats] (the first call after create / reset: at the starting points).
 * Out: x_d = the next points to evaluate (for a finished problem: its result, no longer touched); n_active_d DEVICE int32 (may
 * be NULL) = problems that want another round.  Call until n_active is 0. */
int fdcap_lbfgs_advance(fdcap_lbfgs* opt, float* x_d, int32_t x_stride, const float* f_d, const float* g_d, int32_t g_stride,
                        int32_t* n_active_d, void* stream);
/* A caller that stops before n_active is 0 (a round budget): every unfinished problem's x row holds a line-search TRIAL point
 * (bracketing extrapolates up to 10x per step), not an accepted one.  This puts the last accepted point back (the one `loss` of
 * fdcap_lbfgs_get_stats belongs to), marks those problems finished and counts them in n_unfinished_d (DEVICE int32, may be NULL). */
int fdcap_lbfgs_finalize(fdcap_lbfgs* opt, float* x_d, int32_t x_stride, int32_t* n_unfinished_d, void* stream);
/* per problem, DEVICE, any may be NULL: iterations [n] (L-BFGS directions taken), evaluations [n] (objective calls consumed),
 * loss [n] (objective at the accepted point) */
int fdcap_lbfgs_get_stats(fdcap_lbfgs* opt, int32_t* iterations_d, int32_t* evaluations_d, float* loss_d, void* stream);

/* The inner fit of one stage with L-BFGS instead of Adam: rounds of (forward, fit2d loss, backward, fdcap_lbfgs_advance on the
 * n_local rows of body_rotation_rec) until every frame has stopped or max_rounds evaluations were made; a fresh optimiser per
 * call; cfg->dim is ignored (78).  *rounds_out (may be NULL) = evaluations made.  Synchronises `stream` every few rounds (to
 * read the number of frames still running) and before it returns.  Frames still inside a line search when max_rounds is reached
 * are rolled back to their last accepted point (fdcap_lbfgs_finalize): the rows returned are never trial points.
 * Outer stop, as built: relative change of the loss between two step() calls <= ftol, or max|g| < gtol with g the gradient at the
 * ACCEPTED point.  (SMPLify-X tests param.grad as the last line-search evaluation left it; with strong Wolfe that is the accepted
 * point's gradient except when the search ends on its evaluation cap.  oracle/innerfit.py makes the same choice.) */
int fdcap_opt_fit2d_lbfgs(fdcap_ctx* ctx, const fdcap_fit2d_stage* stage, const fdcap_lbfgs_config* cfg, int32_t max_rounds,
                          int32_t* rounds_out, void* stream);
/* fdcap_lbfgs_get_stats of the last fdcap_opt_fit2d_lbfgs call, per frame [n_local] (FDCAP_E_STATE before the first) */
int fdcap_opt_fit2d_lbfgs_stats(fdcap_ctx* ctx, int32_t* iterations_d, int32_t* evaluations_d, float* loss_d, void* stream);

/* Multi-GPU iteration tail with ONE collective per iteration (instead of an all-reduce before the
 * step and point-to-point halo messages after it):
 *   fdcap_opt_step_rows_and_pack : Adam on body_rotation_rec / camera_ext of the owned rows, then writes
 *       send_d [fdcap_exchange_len()] = this rank's first two + last two owned rows (x | camera_ext)
 *       and its d loss / d scale partial;
 *   (caller: all-gather send_d over the ranks into gathered_d [world, fdcap_exchange_len()]);
 *   fdcap_opt_unpack_and_step_scale : fills the halo rows from ranks rank-1 / rank+1, sums the scale
 *       gradient over ranks in rank order (bit-identical on every rank) and applies Adam to scale. */
int fdcap_opt_step_rows_and_pack(fdcap_ctx* ctx, int32_t ii, int32_t first_phase2_iter, float* send_d, void* stream);
int fdcap_opt_unpack_and_step_scale(fdcap_ctx* ctx, int32_t ii, int32_t first_phase2_iter, const float* gathered_d,
                                    int32_t rank, int32_t world, void* stream);
int32_t fdcap_exchange_len(void);

/* ---- the exchange INSIDE the library (SURVEY 8b "halo_exchange", 8e) -----------------------------------------------------
 * The reference has no distributed code; this is the communicator of the frame-sharded optimiser for callers that are not
 * Python (and for Python: two calls per iteration instead of five, no stream hand-over).  RCCL is bound at run time
 * (dlopen librccl.so.1; FDCAP_RCCL_LIB names another file, tried first; with FDCAP_RCCL_LIB_ONLY=1 nothing else is tried):
 * FDCAP_E_COMM when it is not there, the loader's message in fdcap_comm_last_error (a NULL context, or a context without a
 * message of its own, returns the process-wide loader message).  One communicator per context, created
 * on the calling thread's current HIP device; every rank of the job calls fdcap_comm_create with the SAME id.
 *   fdcap_comm_unique_id : `id128` [FDCAP_UNIQUE_ID_BYTES] host bytes; rank 0's go to the other ranks out of band (any rank may
 *                          call it: it is also the cheap test that librccl can be bound in this process)
 *   fdcap_comm_create    : ncclCommInitRank (collective: blocks until all `world` ranks arrive)
 *   fdcap_opt_halo_exchange : halo rows <- the neighbours' boundary rows as they are (before the first iteration, after
 *                          fdcap_opt_import_state, after each fdcap_opt_step_x of mode 'local')
 *   fdcap_opt_exchange   : the sharded iteration tail, whole -- fdcap_opt_step_rows_and_pack, ONE ncclAllGather of
 *                          fdcap_exchange_len() floats per rank on `stream`, fdcap_opt_unpack_and_step_scale -- in place of
 *                          fdcap_opt_step; same kernels, same bits as the caller-side sequence above
 *   fdcap_comm_allreduce_f64 : in-place sum over the ranks (the logged loss partial sums) */
int fdcap_comm_unique_id(uint8_t* id128);
int fdcap_comm_create(fdcap_ctx* ctx, const uint8_t* id128, int32_t rank, int32_t world);
int fdcap_comm_destroy(fdcap_ctx* ctx);
const char* fdcap_comm_last_error(fdcap_ctx* ctx);
int fdcap_opt_halo_exchange(fdcap_ctx* ctx, void* stream);
int fdcap_opt_exchange(fdcap_ctx* ctx, int32_t ii, int32_t first_phase2_iter, void* stream);
/* The exchange measured on the communicator at hand: mean microseconds of `iters` whole tails (fdcap_opt_exchange) and of `iters` bare
 * ncclAllGather calls of the iteration's message, each train between two HIP events on `stream`.  Every rank must make the same
 * call; the optimiser's state is stepped `iters` times with the gradients it holds (call it after the fit).  Synchronises. */
int fdcap_opt_time_exchange(fdcap_ctx* ctx, int32_t iters, float* us_exchange, float* us_allgather, void* stream);
int fdcap_comm_allreduce_f64(fdcap_ctx* ctx, double* buf_d, int32_t n, void* stream);
/* Overlap of the exchange with the next forward.  Issued between the two calls above, i.e. while the all-gather is in
 * flight: the part of iteration ii's forward (ii = the NEXT iteration; log_terms as its fdcap_opt_backward will get) that needs
 * neither `scale` nor the halo rows -- decoder, pose state and the contact set's pose-blend product of the owned rows
 * (global_optimization.py:270-283 up to `verts * scale`, :284).  The next fdcap_opt_backward(ii) then only decodes the halo
 * rows and refreshes the scale-dependent outputs (two nearly empty launches) before it goes on; parameters are bit-identical
 * to the schedule without this call.  Anything that changes the owned rows in between (a step, fdcap_opt_import_state) drops
 * what ran ahead. */
int fdcap_opt_forward_ahead(fdcap_ctx* ctx, int32_t ii, int32_t first_phase2_iter, int32_t log_terms, void* stream);

/* Results: body_rec75_d [n_local,75] (= convert_to_3D_rot, :633), scale_d [1], cam_ext_d [n_local,16]. */
int fdcap_opt_get_results(fdcap_ctx* ctx, float* body_rec75_d, float* scale_d, float* cam_ext_d,
                          void* stream);
void fdcap_opt_destroy(fdcap_ctx* ctx);

/* World-space contact vertices / joints of the current state (testing + viewers):
 * verts_d [n_local,nc,3] (optional), joints_d [n_local,23,3] (optional). */
int fdcap_opt_forward_world(fdcap_ctx* ctx, float* verts_d, float* joints_d, void* stream);
/* Nearest-neighbour result of the last contact forward (testing): dist_d [n_local,nc] squared
 * distances, idx_d [n_local,nc] scene indices.  The optimiser seeds each search with the previous
 * call's idx (an exact upper bound that only prunes; FDCAP_NN_SEED=0 disables). */
int fdcap_opt_get_contact(fdcap_ctx* ctx, float* dist_d, int32_t* idx_d, void* stream);
/* Gradients of the last fdcap_opt_backward (testing): dx_d [n_local,78], dcam_d [n_local,16]. */
int fdcap_opt_get_grads(fdcap_ctx* ctx, float* dx_d, float* dcam_d, void* stream);

/* Kernel-level timing of the full-mesh pose-blendshape GEMM ([rows,486] x posedirs
 * [486, 3V] on the fp32 matrix cores), HIP events on `stream`, mean milliseconds per launch. */
int fdcap_time_blend_gemm(fdcap_ctx* ctx, int32_t rows, int32_t iters, float* ms, void* stream);

/* The loop's dense contraction on its own (testing): C_d[M,N] (row-major, ldc) = A_d[M,K] (row-major, lda) x B,
 * B a HOST array, element (k, n) at B_h[k * sk + n * sn] -- re-laid out in MFMA fragment order exactly as the
 * context does for the VPoser weights and the blend directions, then run by the kernel the loop uses
 * (v_mfma_f32_16x16x4_f32: every output is the k-ordered fp32 fmaf chain).  Synchronises `stream`. */
int fdcap_panel_gemm(const float* A_d, int32_t lda, int32_t M, int32_t K, const float* B_h, int64_t sk, int64_t sn, int32_t N,
                     float* C_d, int32_t ldc, void* stream);

/* In-loop timing of the Chamfer NN launch: while enabled (max_launches > 0), every contact forward of the optimiser is
 * bracketed by two HIP events on its launch stream (up to max_launches of them; 0 disables and resets).
 * fdcap_opt_nn_timing_read waits for the recorded events and returns the mean milliseconds per launch and their number:
 * the launch as the loop really issues it, iteration by iteration (`roofline.ms_per_launch` of bench.py). */
int fdcap_opt_nn_timing(fdcap_ctx* ctx, int32_t max_launches);
int fdcap_opt_nn_timing_read(fdcap_ctx* ctx, float* mean_ms, int32_t* launches);

/* In-loop timing of EVERY launch of an iteration (r6; bench.py's roofline.per_kernel[].us_live): while enabled (max_events > 0) the
 * optimiser records a HIP event on its launch stream at every boundary between two launches of an iteration (up to max_events
 * events in all; 0 disables and resets) -- one fit of 500 iterations records ~4100.  fdcap_opt_launch_timing_read waits for them and
 * returns, per stage and per phase of the fit, the mean microseconds from the event before the stage's launch(es) to the event
 * after them (so ~1 us of dependent-launch gap is inside every figure) and the number of samples:
 *   mean_us[FDCAP_LT_NUM * phase + stage], counts[...]   phase 0: iterations whose loss has the contact term (ii < P), 1: the others.
 * The events themselves cost a few microseconds per iteration: a fit timed this way is for the per-kernel table, never `value`. */
#define FDCAP_LT_VPOSER_FWD 0    /* VPoser decode (+ the deferred Adam step of the rows) */
#define FDCAP_LT_POSE_FWD 1      /* rotations, joint regression, kinematic chain, world joints */
#define FDCAP_LT_CONTACT_FWD 2   /* contact set: blend product + skinning + world transform (one launch at clip size, else two) */
#define FDCAP_LT_CHAMFER_NN 3    /* Chamfer nearest-neighbour search (+ its seeding launch in a fit's first iteration) */
#define FDCAP_LT_SKIN_BWD 4      /* contact robustifier + skinning backward */
#define FDCAP_LT_BLEND_BWD 5     /* data gradient of the blend product */
#define FDCAP_LT_POSE_BWD 6      /* chain / rotation backward + parameter-space losses */
#define FDCAP_LT_VPOSER_BWD 7    /* VPoser data gradient (+ the `scale` step) */
#define FDCAP_LT_NUM 8
int fdcap_opt_launch_timing(fdcap_ctx* ctx, int32_t max_events);
int fdcap_opt_launch_timing_read(fdcap_ctx* ctx, float* mean_us /* [2 * FDCAP_LT_NUM] */, int32_t* counts /* [2 * FDCAP_LT_NUM] */);

/* Kernel-level timing of the Chamfer NN launch for the roofline line: runs `iters` launches of
 * the optimiser's Chamfer forward on `stream` between two HIP events and returns the mean
 * milliseconds per launch in *ms.  brute_force = 1: every (query, scene point) pair is visited
 * (no seed, no chunk bounds) -- the launch the algorithmic byte count describes; 0: the launch
 * exactly as the loop issues it in steady state (seeded, chunk-culled). */
int fdcap_opt_time_chamfer(fdcap_ctx* ctx, int32_t iters, int32_t brute_force, float* ms, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FDCAP_H */
