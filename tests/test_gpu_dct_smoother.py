"""GPU parity tests of SURVEY.md §8f F2 through the C-ABI: mode 'dct' (global_optimization.py:595-630) and
optimization.py's per-frame smoother, against the oracle and the reference-generated goldens."""
import os

import numpy as np
import pytest
import torch

import fdcap_amd  # noqa: F401
from fdcap_amd import capi, synth
from fdcap_amd.fitting import FittingOP
from fdcap_amd.io import load_dct_base, read_camerapose
from fdcap_amd.smoother import FittingOP as SmootherOP
from oracle.fitting import FittingOracle
from oracle.smoother import SmootherOracle
from oracle.smplx import SMPLXOracle
from oracle.vposer import VPoserDecoder

pytestmark = pytest.mark.gpu


# ---- per-frame smoother -------------------------------------------------------------------------
def test_smoother_matches_the_reference_run(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_smoother.npz"))
    out = SmootherOP().fitting_clip(g["body_in"]).cpu().numpy()
    d = np.abs(out - g["body_out"])
    print("smoother GPU-vs-reference: max", d.max())
    assert d.max() < 1e-5            # host build of the same math: 1.2e-6


def test_smoother_file_by_file_equals_one_launch_and_the_oracle(tmp_path):
    import pickle
    n = 40
    clip = synth.make_clip(n, seed=41, num_outliers=2)
    keys = ("transl", "global_orient", "betas", "body_pose", "left_hand_pose", "right_hand_pose", "camera_translation")
    dims = (3, 3, 10, 32, 12, 12, 3)
    files = []
    for i in range(n):
        d, o = {}, 0
        for k, w in zip(keys, dims):
            d[k] = clip.body_params[i:i + 1, o:o + w].astype(np.float32)
            o += w
        fn = str(tmp_path / ("%06d.pkl" % i))
        with open(fn, "wb") as f:
            pickle.dump(d, f)
        files.append(fn)
    one = SmootherOP().fitting_clip(clip.body_params).cpu().numpy()
    fop = SmootherOP()
    rows, prev = [], None
    for ii, fn in enumerate(files):                                        # optimization.py:334-348
        prev = fop.fitting(fn) if ii == 0 else fop.fitting_smoothing(fn, prev)
        rows.append(prev.cpu().numpy())
    np.testing.assert_array_equal(np.concatenate(rows, 0), one)            # same arithmetic, same order
    ref = SmootherOracle().fitting_clip(clip.body_params).numpy()
    assert np.abs(one - ref).max() < 1e-5
    fop.save_result(prev, str(tmp_path / "out" / "smoothed_body" / "000039.pkl"))
    with open(tmp_path / "out" / "smoothed_body" / "000039.pkl", "rb") as f:
        saved = pickle.load(f)
    assert list(saved) == list(keys) and saved["body_pose"].shape == (1, 32)
    np.testing.assert_array_equal(saved["camera_translation"], one[-1:, 72:75])


def test_smoother_rejects_a_first_call_without_predecessor():
    with pytest.raises(capi.FdcapError):
        SmootherOP().fitting_smoothing("/nonexistent.pkl", None)


# ---- mode 'dct' ---------------------------------------------------------------------------------
def _dct_case(n, num_verts, ns, seed, per_part=8):
    bm = synth.make_body_model(num_verts, seed=seed)
    vp = synth.make_vposer(seed=seed + 1)
    clip = synth.make_clip(n, seed=seed + 2, num_outliers=2)
    scene = synth.make_scene(ns, seed=seed + 3)
    left, right = synth.make_contact_ids(bm.v_template, per_part=per_part, seed=seed + 4)
    vid = np.concatenate([left, right])
    rng = np.random.Generator(np.random.PCG64(seed + 5))
    c0 = rng.standard_normal((n // 60, 23, 3, 5)).astype(np.float32)
    return bm, vp, clip, scene, vid, c0


@pytest.mark.parametrize("n,num_iter", [(120, 200), (150, 400)])
def test_dct_mode_matches_oracle(n, num_iter):
    """Short 'dct' runs (the 10000 of :596 shortened) across the 95 % switch; n = 150 leaves 30 frames
    outside every 60-frame window."""
    bm, vp, clip, scene, vid, c0 = _dct_case(n, 200, 1500, seed=50 + n)
    D = load_dct_base(None)
    fop = FittingOP({}, {}, n, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid,
                    camera_ext=read_camerapose(clip.camerapose_lines), dct_mtx=D, c_dct_init=c0, dct_num_iter=num_iter)
    body, scale, cam = fop.fitting(torch.tensor(clip.body_params).cuda(), "dct", log_every=1)
    orc = FittingOracle(SMPLXOracle(bm), VPoserDecoder.from_data(vp), scene, vid, clip.camerapose_lines, n,
                        dct_mtx=D, c_dct_init=c0)
    obody, oscale, ocam = orc.fitting_dct(torch.tensor(clip.body_params), num_iter=num_iter)
    olog = np.array(orc.loss_log)
    P = int(np.ceil(num_iter * 0.95 - 1e-9))
    # phase 1: smooth objective, c_dct only -- trajectories agree to fp32 rounding
    np.testing.assert_allclose(fop.c_dct.cpu().numpy(), orc.c_dct.detach().numpy(), rtol=0, atol=3e-5)
    ldct = np.array(fop.log_dct)
    np.testing.assert_allclose(ldct[:, 1], olog[:P, 4], rtol=3e-5, atol=1e-6)
    # phase 2 (L1 data term + Adam: +-lr per sign flip, bounded by 2*lr*k)
    k = num_iter - P - 1
    err = np.abs(body.cpu().numpy() - obody.numpy())
    q50, q90, q99 = np.quantile(err, [0.5, 0.9, 0.99])
    print("dct phase 2: k", k, "max", err.max(), "q50/q90/q99", q50, q90, q99)
    assert err.max() <= 2 * 0.005 * k and q50 < 1e-6 and q90 < 1e-4 and q99 < 3e-3
    np.testing.assert_allclose(float(scale), float(oscale), atol=1e-4)
    np.testing.assert_array_equal(cam.cpu().numpy(), ocam.numpy())          # camera_ext is never stepped in this mode
    log2 = np.array(fop.log2)
    assert log2.shape[0] == k and int(log2[0, 0]) == P + 1
    tol = 3e-6 + 2e-6 * np.arange(k)
    for col, ocol in ((1, 0), (2, 1), (4, 3), (5, 4), (6, 5)):               # rec, vposer, contact, dct, total
        # loss_dct is O(10) here and moves with every +-lr step of the body (one sign flip at an L1 kink shifts it by
        # ~1e-5 relative per step): absolute bar for the small terms, relative 5e-5 on top for the large one
        bar = 2 * tol + 5e-5 * np.abs(olog[P + 1:, ocol])
        assert np.all(np.abs(log2[:, col] - olog[P + 1:, ocol]) <= bar), (col, np.abs(log2[:, col] - olog[P + 1:, ocol]).max())
    fop.close()


def test_dct_mode_legacy_zero_grad_lets_c_dct_coast():
    """torch < 2 semantics in mode 'dct': from the 95 % switch on the frozen c_dct keeps a ZERO gradient (not
    None), so Adam keeps moving it on its decaying moments while body / scale are optimised."""
    n, num_iter = 120, 200
    bm, vp, clip, scene, vid, c0 = _dct_case(n, 200, 1500, seed=50 + n)
    D = load_dct_base(None)
    out = {}
    for legacy in (False, True):
        fop = FittingOP({}, {}, n, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid, legacy_zero_grad=legacy,
                        camera_ext=read_camerapose(clip.camerapose_lines), dct_mtx=D, c_dct_init=c0, dct_num_iter=num_iter)
        body, scale, cam = fop.fitting(torch.tensor(clip.body_params).cuda(), "dct", log_every=1)
        out[legacy] = (body.cpu().numpy(), float(scale), fop.c_dct.cpu().numpy(), np.array(fop.log2))
        fop.close()
    orc = FittingOracle(SMPLXOracle(bm), VPoserDecoder.from_data(vp), scene, vid, clip.camerapose_lines, n,
                        dct_mtx=D, c_dct_init=c0, legacy_zero_grad=True)
    obody, oscale, _ = orc.fitting_dct(torch.tensor(clip.body_params), num_iter=num_iter)
    body, scale, cd, log2 = out[True]
    np.testing.assert_allclose(cd, orc.c_dct.detach().numpy(), rtol=0, atol=3e-5)
    moved = np.abs(cd - out[False][2]).max()
    assert moved > 1e-3, moved                                   # ten coasting steps at lr 0.005 really moved the coefficients
    P = int(np.ceil(num_iter * 0.95 - 1e-9))
    k = num_iter - P - 1
    err = np.abs(body - obody.numpy())
    assert err.max() <= 2 * 0.005 * k and np.quantile(err, 0.5) < 1e-6 and np.quantile(err, 0.9) < 1e-4
    np.testing.assert_allclose(scale, float(oscale), atol=1e-4)
    olog = np.array(orc.loss_log)
    np.testing.assert_allclose(log2[:, 5], olog[P + 1:, 4], rtol=5e-5, atol=1e-5)     # loss_dct follows the coasting c_dct


def test_dct_gradient_matches_autograd():
    """d(1e-4 dct + 0.5 rec + 0.1 contact)/d(body_rotation_rec, scale) vs fp64 autograd."""
    n = 120
    bm, vp, clip, scene, vid, c0 = _dct_case(n, 160, 900, seed=77)
    D = load_dct_base(None)
    fop = FittingOP({}, {}, n, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid,
                    camera_ext=read_camerapose(clip.camerapose_lines), dct_mtx=D, c_dct_init=c0, dct_num_iter=40)
    body_in = torch.tensor(clip.body_params).cuda()
    fop.fitting(body_in, "dct")                                             # 38 c_dct steps, then one body / scale step
    lib, h = fop.ctx.lib, fop.ctx.handle
    # the loss_rec target as the GPU holds it (fp32 6D rows): the L1 sign gradient is 0 exactly where a row
    # element still equals its target, so the oracle must see the same target bits
    x78_gpu = torch.empty(n, 78, device="cuda")
    capi.check(lib.fdcap_params_75_to_78(capi.dptr(body_in), n, capi.dptr(x78_gpu), capi.current_stream()), "75_to_78")
    # weights large enough for the DCT term to dominate the check
    capi.check(lib.fdcap_opt_backward_dct(h, 0.7, 0.5, 0.1, 1, capi.current_stream()), "backward_dct")
    dx = torch.empty(n, 78, device="cuda")
    capi.check(lib.fdcap_opt_get_grads(h, capi.dptr(dx), None, capi.current_stream()), "get_grads")
    dscale = float(fop._dscale.cpu())
    orc = FittingOracle(SMPLXOracle(bm, dtype=torch.float64), VPoserDecoder.from_data(vp, dtype=torch.float64), scene, vid,
                        clip.camerapose_lines, n, dtype=torch.float64, dct_mtx=D, c_dct_init=fop.c_dct.cpu().numpy())
    x78 = x78_gpu.cpu().double()
    idx1 = orc.init(x78)
    orc.body_rotation_rec.data = fop.body_rotation_rec.detach().cpu().double()
    orc.scale.data = fop._scale.detach().cpu().double().reshape(())
    l_rec, l_vp, l_con, l_sm, l_ws = orc.cal_loss(x78.detach(), idx1)
    _, _, joints = orc.forward_world()
    loss = 0.7 * orc.cal_dctloss(joints) + 0.5 * l_rec + 0.1 * l_con
    loss.backward()
    g = orc.body_rotation_rec.grad.numpy()
    np.testing.assert_allclose(dx.cpu().numpy(), g, rtol=2e-3, atol=2e-4 * np.abs(g).max())
    np.testing.assert_allclose(dscale, float(orc.scale.grad), rtol=2e-3)
    fop.close()


def test_dct_mode_matches_reference_golden(golden_dir):
    """The reference's own 10000-iteration 'dct' run (tests/golden/make_golden.py --dct)."""
    path = os.path.join(golden_dir, "ref_dct_10000it.npz")
    g = np.load(path)
    bm = synth.make_body_model(int(g["num_verts"]), seed=int(g["model_seed"]))
    vp = synth.make_vposer(seed=int(g["vposer_seed"]))
    fop = FittingOP({}, {}, 300, body_model=bm, vposer=vp, scene_verts=g["scene"], contact_ids=g["vid"],
                    camera_ext=read_camerapose(list(g["camerapose"])), dct_mtx=g["dct_mtx"], c_dct_init=g["c_dct0"])
    body, scale, cam = fop.fitting(torch.tensor(g["body_in"]).cuda(), "dct", log_every=1)
    np.testing.assert_array_equal(fop.idx1, g["idx1"])
    # c_dct after the 9500-iteration first phase (frozen afterwards, so this is the reference's final c_dct).
    # Yardstick = the oracle (same torch ops as the reference) against this golden: q90 5e-7, q99 3e-4, max 7e-4 --
    # the Geman-McClure objective has flat directions along which 9500 Adam steps amplify rounding.
    dc = np.abs(fop.c_dct.cpu().numpy() - g["c_dct"])
    print("c_dct |d| q90/q99/max", np.quantile(dc, 0.9), np.quantile(dc, 0.99), dc.max())
    assert np.quantile(dc, 0.9) < 1e-5 and dc.max() < 3e-3
    logd = g["logd"]
    ref1 = logd[logd[:, 0] < 9400]
    mine = np.array(fop.log_dct)
    sel = np.isin(mine[:, 0], ref1[:, 0])
    np.testing.assert_allclose(mine[sel, 1], ref1[:, 5], rtol=0, atol=8e-6)          # printed with 6 decimals; values 18.6 -> 0.167
    # the no-op iteration 9500 and the first steps of the second phase, log line for log line
    log2 = np.array(fop.log2)
    ref2 = logd[logd[:, 0] >= 9501]
    assert log2.shape[0] == ref2.shape[0] == 499 and int(log2[0, 0]) == 9501
    np.testing.assert_allclose(log2[:4, 1:], ref2[:4, 1:], rtol=0, atol=5e-6)
    err = np.abs(body.cpu().numpy() - g["body_rec"])
    q50, q90, q99 = np.quantile(err, [0.5, 0.9, 0.99])
    print("dct golden: max", err.max(), "q50/q90/q99", q50, q90, q99, "scale", float(scale), float(g["scale"]))
    # 499 Adam steps over an L1 data term.  Yardstick (oracle vs this golden): q50 2.5e-4, q90 1.5e-3, q99 3e-3, max 1.3e-2
    assert q50 < 1e-3 and q90 < 5e-3 and q99 < 1e-2 and err.max() <= 2 * 0.005 * 499
    np.testing.assert_allclose(float(scale), float(g["scale"]), atol=2e-3)
    fop.close()


# ---- mode 'dct' sharded over 2 ranks on the test box's one GPU (gloo; the shard boundary is a window boundary) ----
def _dct_fit(group):
    n, num_iter = 120, 200
    bm, vp, clip, scene, vid, c0 = _dct_case(n, 200, 1500, seed=91)
    fop = FittingOP({}, {}, n, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid,
                    camera_ext=read_camerapose(clip.camerapose_lines), dct_mtx=load_dct_base(None), c_dct_init=c0,
                    dct_num_iter=num_iter, group=group)
    body, scale, cam = fop.fitting(torch.tensor(clip.body_params).cuda(), "dct", log_every=5)
    out = (fop.shard.frame0, body.cpu().numpy(), float(scale), fop.c_dct.cpu().numpy(), np.array(fop.log_dct), np.array(fop.log2))
    fop.close()
    return out


def _dct_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _dct_fit(dist.group.WORLD))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_dct_mode_sharded_matches_single_rank():
    import socket
    import torch.multiprocessing as mp
    from tests.shared_gpu import retry_on_shared_gpu_glitch
    ref = _dct_fit(None)

    def check():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_dct_worker, args=(r, 2, port, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=600) for _ in range(2))
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        assert [r[1] for r in res] == [0, 60]
        for r in res:
            np.testing.assert_array_equal(r[4], ref[3])              # every window is fitted by exactly one rank, same arithmetic
            np.testing.assert_allclose(r[5][:, 1], ref[4][:, 1], rtol=1e-6)
            assert abs(r[3] - ref[2]) < 2e-6
            np.testing.assert_allclose(r[6], ref[5], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(np.concatenate([r[2] for r in res]), ref[1], rtol=0, atol=5e-6)

    retry_on_shared_gpu_glitch(check)        # (two processes on one GPU: tests/shared_gpu.py)


def test_dct_mode_refuses_shards_off_the_window_grid():
    """150 frames over 2 ranks would split at frame 75, inside a window."""
    from fdcap_amd.dist import FrameShard
    sh = FrameShard(150, None, rank=1, world=2)
    assert sh.frame0 % 60 != 0


def test_dct_mode_needs_a_full_window_and_smoother_handles_one_frame():
    bm, vp, clip, scene, vid, _ = _dct_case(60, 120, 500, seed=5)
    short = synth.make_clip(40, seed=9)
    fop = FittingOP({}, {}, 40, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid,
                    camera_ext=read_camerapose(short.camerapose_lines), dct_num_iter=20)
    with pytest.raises(capi.FdcapError):                                  # the reference's windows are 60 frames (:41)
        fop.fitting(torch.tensor(short.body_params).cuda(), "dct")
    fop.close()
    one = SmootherOP().fitting_clip(clip.body_params[:1]).cpu().numpy()
    ref = SmootherOracle().fitting_clip(clip.body_params[:1]).numpy()
    assert one.shape == (1, 75) and np.abs(one - ref).max() < 1e-5


@pytest.mark.parametrize("legacy", [False, True])
def test_dct_mode_checkpoint_resume_is_bit_identical(tmp_path, legacy):
    """VERDICT r4 (missing 5): mode 'dct' is the 10000-iteration mode (:596), the one where a resume matters.  A 200-iteration fit
    (phase switch at 190) checkpointing every 40 iterations -- which also cuts the one-launch first phase into five launches --
    equals the plain fit bit for bit; new optimisers resumed from the checkpoint of iteration 120 (inside the first phase) and of
    iteration 196 (inside the second; written by a run with checkpoint_every=14) end on the same bits, c_dct included."""
    n, num_iter = 120, 200
    bm, vp, clip, scene, vid, c0 = _dct_case(n, 200, 1500, seed=91)
    D = load_dct_base(None)
    body = torch.tensor(clip.body_params).cuda()

    def make():
        return FittingOP({}, {}, n, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid, camera_ext=read_camerapose(clip.camerapose_lines),
                         dct_mtx=D, c_dct_init=c0, dct_num_iter=num_iter, legacy_zero_grad=legacy)

    def run(**kw):
        fop = make()
        b, s, c = fop.fitting(body, "dct", **kw)
        out = (b.clone(), float(s), fop.c_dct.clone(), np.array(fop.log_dct), np.array(fop.log2))
        fop.close()
        return out

    plain = run(log_every=10)
    ck1, ck2 = str(tmp_path / "a.npz"), str(tmp_path / "b.npz")
    cut = run(log_every=10, checkpoint_every=40, checkpoint_path=ck1, check_finite_every=3)
    for a, b in zip(plain, cut):
        assert (torch.equal(a, b) if torch.is_tensor(a) else np.array_equal(a, b))
    assert int(np.load(ck1)["next_iter"]) == 160 and str(np.load(ck1)["mode"]) == "dct"
    # resume inside phase 1: stop a fresh run's file at iteration 120 by giving it a budget-compatible checkpoint_every
    fop = make()
    fop.fitting(body, "dct", checkpoint_every=120, checkpoint_path=ck1)
    fop.close()
    assert int(np.load(ck1)["next_iter"]) == 120
    res = run(log_every=10, resume=ck1)
    assert torch.equal(res[0], plain[0]) and res[1] == plain[1] and torch.equal(res[2], plain[2])
    np.testing.assert_array_equal(res[3], plain[3][12:])                # the resumed run's history starts at iteration 120
    np.testing.assert_array_equal(res[4], plain[4])
    # resume inside phase 2
    fop = make()
    fop.fitting(body, "dct", checkpoint_every=14, checkpoint_path=ck2)
    fop.close()
    assert int(np.load(ck2)["next_iter"]) == 196
    res = run(resume=ck2)
    assert torch.equal(res[0], plain[0]) and res[1] == plain[1] and torch.equal(res[2], plain[2])
    # a global-mode fit refuses the file
    fop = FittingOP({"num_iter": num_iter}, {}, n, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid,
                    camera_ext=read_camerapose(clip.camerapose_lines))
    with pytest.raises(capi.FdcapError, match="mode 'dct'"):
        fop.fitting(body, "global", resume=ck2)
    fop.close()
