"""Parity at the REAL budget: the reference's fixed 500 iterations (/root/reference/global_optimization.py:672,
loop :560-593, result :633-635) on tests/golden/ref_global_500it.npz -- the reference's own loop, run in the build container by
tests/golden/make_golden.py --g500 (N = 300, 640 vertices, 3000 scene points, 48 contact vertices).

Two statements, because 500 Adam steps through three L1 terms are chaotic (a rounding-level sign flip moves a parameter by up to
2 lr per step; from iteration 401 on camera_ext takes sign-normalised +-lr steps and 0.01 in a camera rotation entry is 30 mm at 3 m):
  (a) END TO END: distance of the HIP run from the reference's run in mm of world-space vertex / joint position (both parameter
      sets decoded through the oracle's body model in fp64), next to the distance of two YARDSTICK runs of the oracle itself from
      the same reference run (tests/golden/oracle_global_500it_{f64,f32t1}.npz: the oracle in fp64, and the oracle in fp32 on one
      thread instead of four -- the same code as the golden run, another summation order).  Measured:
      iteration 400 (end of phase 1): HIP 5.6 mm mean / 21 mm max; fp64 oracle 6.4 / 23; fp32 oracle on one thread 6.5 / 21.
      iteration 500:                  HIP 18 mm mean / 80 mm max;  fp64 oracle 19 / 85; fp32 oracle on one thread 19 / 89.
      Every implementation, the reference's own arithmetic on another thread count included, lands equally far from the
      reference's run; the HIP path is not further than the yardsticks.  Losses: total within 1.4 % along the whole curve.
  (b) RE-SYNCHRONISED WINDOWS: the fixture holds the reference's parameters AND torch.optim.Adam's state after iterations 100,
      300, 400, 450, 495; the HIP optimiser is started from each (fitting(resume=...)) and compared with the reference five
      steps later -- per-step fidelity along the real trajectory (late phase 1, across the phase switch, phase 2 with camera_ext
      moving), free of the accumulated divergence.  Bars = the 5-iteration golden test's."""
import os

import numpy as np
import pytest
import torch

import fdcap_amd  # noqa: F401
from fdcap_amd import synth
from fdcap_amd.fitting import FittingOP, first_phase2_iter
from fdcap_amd.io import read_camerapose
from tests.parity500 import distance_report

pytestmark = pytest.mark.gpu
LR = 0.005


# Two fixtures (r5): "" = r4's (seeds 40-44, 640 vertices, 3000 scene points, 2 x 24 contact vertices); "_b" = a second run of the
# reference's loop with other seeds throughout, 800 vertices, 4000 scene points and a DENSER contact set, 2 x 110 vertices
# (tests/golden/make_golden.py --g500b) -- is the distance of r4's fixture a property of the problem class or an accident of one seed?
FIXTURES = ["", "_b"]


def _setup(golden_dir, sfx=""):
    g = np.load(os.path.join(golden_dir, f"ref_global_500it{sfx}.npz"))
    bm = synth.make_body_model(int(g["num_verts"]), seed=int(g["model_seed"]))
    vp = synth.make_vposer(seed=int(g["vposer_seed"]))
    fop = FittingOP({"num_iter": 500}, {}, 300, body_model=bm, vposer=vp, scene_verts=g["scene"], contact_ids=g["vid"],
                    camera_ext=read_camerapose(list(g["camerapose"])))
    return g, bm, vp, fop


def _snap(d, k, pre=""):
    i = [int(v) for v in d[pre + "snap_iters"]].index(k)
    return d[pre + "snap_x78"][i], d[pre + "snap_scale"][i], d[pre + "snap_cam"][i]


@pytest.mark.parametrize("sfx", FIXTURES)
def test_fixed_budget_distance_from_the_reference_run(golden_dir, sfx):
    g, bm, vp, fop = _setup(golden_dir, sfx)
    lines = list(g["camerapose"])
    body, scale, cam = fop.fitting(torch.tensor(g["body_in"]).cuda(), "global", log_every=1, snapshot_at=[100, 400, 500])
    np.testing.assert_array_equal(fop.idx1, g["idx1"])
    yard = {k: np.load(os.path.join(golden_dir, f"oracle_global_500it{sfx}_{k}.npz")) for k in ("f64", "f32t1")}
    rep = {}
    for k in (100, 400, 500):
        s = fop.snapshots[k]
        hip = (s[0].cpu().numpy(), float(s[1].cpu()), s[2].cpu().numpy())
        rep[k] = distance_report(bm, vp, lines, hip, _snap(g, k))
        ys = [distance_report(bm, vp, lines, _snap(y, k), _snap(g, k)) for y in yard.values()]
        # not further from the reference's run than the yardsticks are (1.5 x the larger yardstick + 1 mm of slack)
        for key in ("vert_mm_mean", "vert_mm_q99", "joint_mm_mean"):
            assert rep[k][key] <= 1.5 * max(y[key] for y in ys) + 1.0, (k, key, rep[k][key], [y[key] for y in ys])
        print(k, {m: round(v, 4) for m, v in rep[k].items()}, "yardsticks", [round(y["vert_mm_mean"], 2) for y in ys])
    # absolute bars (measured values in the module docstring): end of phase 1, then the end of the budget
    assert rep[400]["vert_mm_mean"] < 10 and rep[400]["vert_mm_q99"] < 25 and rep[400]["joint_mm_max"] < 40, rep[400]
    assert rep[500]["vert_mm_mean"] < 35 and rep[500]["vert_mm_q99"] < 90 and rep[500]["vert_mm_max"] < 250, rep[500]
    assert rep[400]["cam_max"] == 0.0                                        # camera_ext does not move before iteration 401 (:564-568)
    assert rep[500]["x78_q50"] < 2e-3 and rep[500]["x78_q90"] < 8e-3 and rep[500]["x78_q99"] < 2e-2, rep[500]
    assert rep[500]["hands_max"] < 5e-2 and rep[500]["scale_abs"] < 3e-3, rep[500]      # (measured 0.014 / 6e-4; yardsticks 0.02-0.04 / 1.1e-3)
    # the returned triple is the last snapshot (:633-635)
    np.testing.assert_allclose(float(scale), float(fop.snapshots[500][1].cpu()), rtol=0, atol=0)
    np.testing.assert_array_equal(cam.cpu().numpy().reshape(300, 16), fop.snapshots[500][2].cpu().numpy())
    err75 = np.abs(body.cpu().numpy() - g["body_rec"])
    assert np.quantile(err75, 0.5) < 2e-3 and np.quantile(err75, 0.99) < 3e-2
    # per-iteration loss curves, relative to the reference's printed values (:573-575, :587-589)
    lg = fop.log
    ref = g["log"]
    rel = lambda a, col: np.abs(np.array(a) - ref[:, col]) / np.maximum(np.abs(ref[:, col]), 1e-12)
    assert rel(lg.l_rec, 1).max() < 3e-3 and rel(lg.loss_smoothing, 3).max() < 1e-2 and rel(lg.loss_contact, 4).max() < 6e-3
    assert rel(lg.total, 6).max() < 4e-2 and rel(lg.total, 6)[:400].max() < 5e-3 and rel(lg.total, 6)[-1] < 1e-2
    assert rel(lg.loss_world_smoothing, 5)[400:].max() < 0.15
    assert rel(lg.total, 6)[:20].max() < 2e-5                                 # before any divergence: rounding level
    fop.close()


@pytest.mark.parametrize("sfx", FIXTURES)
@pytest.mark.parametrize("k0", [100, 300, 400, 450, 495])
def test_resynchronised_five_step_windows_along_the_real_budget(golden_dir, tmp_path, k0, sfx):
    g, bm, vp, fop = _setup(golden_dir, sfx)
    P = first_phase2_iter(500)
    assert int(g[f"adam{k0}_x_step"]) == k0 and int(g[f"adam{k0}_c_step"]) == max(k0 - P - 1, 0) and int(g[f"adam{k0}_s_step"]) == min(k0, P)
    x, s, c = _snap(g, k0)
    state = np.concatenate([g[f"adam{k0}_x_m"].ravel(), g[f"adam{k0}_x_v"].ravel(), g[f"adam{k0}_c_m"].ravel(), g[f"adam{k0}_c_v"].ravel(),
                            g[f"adam{k0}_s_m"].ravel(), g[f"adam{k0}_s_v"].ravel()]).astype(np.float32)
    ck = str(tmp_path / "resync.npz")
    np.savez(ck, next_iter=np.int64(k0), num_iter=np.int64(500), n_total=np.int64(300), frame0=np.int64(0), n_local=np.int64(300),
             rows_x=x.astype(np.float32), rows_cam=c.reshape(300, 16).astype(np.float32), scale=np.array([s], np.float32), state=state)
    fop.fitting(torch.tensor(g["body_in"]).cuda(), "global", log_every=1, resume=ck, snapshot_at=[k0 + 5])
    sn = fop.snapshots[k0 + 5]
    gx, gs, gc = _snap(g, k0 + 5)
    err = np.abs(sn[0].cpu().numpy() - gx)
    q50, q90, q99 = np.quantile(err, [0.5, 0.9, 0.99])
    frac = float((err <= 2e-5).mean())
    print(k0, "x78 q50 %.2e q90 %.2e q99 %.2e max %.2e within 2e-5: %.5f" % (q50, q90, q99, err.max(), frac),
          "scale", abs(float(sn[1].cpu()) - float(gs)), "cam", np.abs(sn[2].cpu().numpy().reshape(300, 4, 4) - gc).max())
    assert err.max() <= 2 * LR * 5 + 1e-6                  # five steps: nothing can be further
    assert q50 < 1e-6 and q90 < 2e-5 and frac > 0.97, (q50, q90, q99, frac)
    assert err[:, 51:75].max() <= 2e-5                     # hands
    assert abs(float(sn[1].cpu()) - float(gs)) < 2e-5
    ecam = np.abs(sn[2].cpu().numpy().reshape(300, 4, 4) - gc)
    assert float((ecam <= 2e-5).mean()) > 0.97 and ecam.max() <= 2 * LR * 5 + 1e-6
    # the losses the reference printed inside the window (log rows k0 .. k0+4 are evaluated BEFORE steps k0+1 .. k0+5)
    it = np.array(fop.log.iters)
    sel = (it >= k0) & (it < k0 + 5)
    ref = g["log"][k0:k0 + 5]
    np.testing.assert_allclose(np.array(fop.log.l_rec)[sel], ref[:, 1], atol=2e-5)
    np.testing.assert_allclose(np.array(fop.log.loss_smoothing)[sel], ref[:, 3], atol=2e-5)
    np.testing.assert_allclose(np.array(fop.log.loss_contact)[sel], ref[:, 4], atol=5e-5)
    np.testing.assert_allclose(np.array(fop.log.total)[sel][:1], ref[:1, 6], atol=1e-5)     # first row: identical state
    fop.close()
