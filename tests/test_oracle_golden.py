"""The oracle against vectors produced by the reference's OWN code (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

import fdcap_amd  # noqa: F401
from fdcap_amd import synth
from oracle import rotrepr
from oracle.fitting import FittingOracle, qvec2rotmat, verts_transform
from oracle.smplx import SMPLXOracle
from oracle.vposer import VPoserDecoder
from tests.golden.make_golden import sha


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_units_against_reference_functions(golden_dir):
    u = _load(golden_dir, "ref_units.npz")
    for q, R in zip(u["qvec"], u["qvec_R"]):
        np.testing.assert_allclose(qvec2rotmat(q), R, rtol=0, atol=1e-15)
    # (1/2,1/2,1/2,1/2) is the cyclic permutation
    np.testing.assert_allclose(u["qvec_R"][0], [[0, 0, 1], [1, 0, 0], [0, 1, 0]], atol=1e-15)
    x78 = rotrepr.convert_to_6D_rot(torch.tensor(u["x75"]))
    np.testing.assert_allclose(x78.numpy(), u["x78"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(rotrepr.convert_to_3D_rot(x78).numpy(), u["x75_back"], rtol=0, atol=1e-6)
    R = rotrepr.decode_6d(torch.tensor(u["six"])).numpy()
    np.testing.assert_allclose(R, u["six_R"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(np.einsum("nij,nkj->nik", R, R), np.tile(np.eye(3), (10, 1, 1)), atol=1e-6)
    np.testing.assert_allclose(np.linalg.det(R), 1.0, atol=1e-6)
    out = verts_transform(torch.tensor(u["vt_v"]), torch.tensor(u["vt_M"])).numpy()
    np.testing.assert_allclose(out, u["vt_out"], rtol=0, atol=1e-6)
    from fdcap_amd.io import body_params_parse
    d = {k[len("parse_"):]: u[k] for k in u.files if k.startswith("parse_") and k != "parse_out"}
    np.testing.assert_array_equal(body_params_parse(d), u["parse_out"])


_SLOW = pytest.mark.skipif(os.environ.get("FDCAP_SLOW_TESTS") != "1", reason="10 / 63 min of CPU (V = 10 475): FDCAP_SLOW_TESTS=1; "
                           "run once in the build container when the fixture was made (DESIGN section 7)")


@pytest.mark.parametrize("name", ["ref_global_5it.npz", "ref_global_20it.npz",
                                  pytest.param("ref_global_5it_full.npz", marks=_SLOW), pytest.param("ref_global_5it_allverts.npz", marks=_SLOW)])
def test_global_trajectory_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    if "scene" not in g.files:            # (regenerated from its seed; checked by hash)
        scene = synth.make_scene(int(g["ns"]), seed=int(g["scene_seed"]))
        assert sha(scene) == str(g["sha_scene"])
        g = dict(g.items(), scene=scene)
    bm = synth.make_body_model(int(g["num_verts"]), seed=int(g["model_seed"]))
    vp = synth.make_vposer(seed=int(g["vposer_seed"]))
    # the synthetic generators must reproduce the arrays the golden run used
    assert sha(bm.posedirs) == str(g["sha_posedirs"])
    assert sha(bm.v_template) == str(g["sha_vtemplate"])
    assert sha(vp.fc2_w) == str(g["sha_fc2"])
    f = FittingOracle(SMPLXOracle(bm), VPoserDecoder.from_data(vp), g["scene"], g["vid"],
                      list(g["camerapose"]), 300, num_iter=int(g["num_iter"]),
                      one_direction_chamfer=False)
    body_rec, scale, cam = f.fitting(torch.tensor(g["body_in"]))
    np.testing.assert_array_equal(f.idx1, g["idx1"])
    np.testing.assert_array_equal(g["idx1"], g["planted_outliers"])
    # same torch ops on the same CPU: agreement is at rounding level (measured 3e-7)
    np.testing.assert_allclose(body_rec.numpy(), g["body_rec"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(float(scale), float(g["scale"]), rtol=0, atol=1e-6)
    np.testing.assert_allclose(cam.numpy(), g["camera_ext"], rtol=0, atol=1e-6)
    log = np.array(f.loss_log)
    ref = g["log"]
    # the reference prints 6 decimals
    np.testing.assert_allclose(log[:, 0], ref[:, 1], atol=2e-6)   # l_rec
    np.testing.assert_allclose(log[:, 1], ref[:, 2], atol=2e-6)   # l_vposer
    np.testing.assert_allclose(log[:, 2], ref[:, 3], atol=2e-6)   # loss_smoothing
    np.testing.assert_allclose(log[:, 3], ref[:, 4], atol=2e-6)   # loss_contact
    np.testing.assert_allclose(log[:, 5], ref[:, 6], atol=2e-6)   # total
    n1 = int(np.ceil(int(g["num_iter"]) * 0.8))
    np.testing.assert_allclose(log[n1:, 4], ref[n1:, 5], atol=2e-6)  # world smoothing (phase 2)
    # phase semantics: scale frozen from the phase switch on, camera_ext
    # first moves one iteration after it
    assert np.isnan(ref[:n1, 5]).all() and not np.isnan(ref[n1:, 5]).any()


def test_one_direction_chamfer_is_equivalent(golden_dir):
    """Only dist1 is consumed (:293) so dropping the scene->body half changes nothing."""
    g = _load(golden_dir, "ref_global_5it.npz")
    bm = synth.make_body_model(int(g["num_verts"]), seed=int(g["model_seed"]))
    vp = synth.make_vposer(seed=int(g["vposer_seed"]))
    f = FittingOracle(SMPLXOracle(bm), VPoserDecoder.from_data(vp), g["scene"], g["vid"],
                      list(g["camerapose"]), 300, num_iter=5, one_direction_chamfer=True)
    body_rec, scale, cam = f.fitting(torch.tensor(g["body_in"]))
    np.testing.assert_allclose(body_rec.numpy(), g["body_rec"], rtol=0, atol=5e-6)


def test_local_mode_trajectory_matches_reference(golden_dir):
    """mode='local' (:499-556): phase A, detect_contact (weight_left = left/(left+left) == 0.5),
    then 0.4*num_iter iterations of cal_loss2 (vertex-space smoothing + foot-skate term)."""
    g = _load(golden_dir, "ref_local_10it.npz")
    bm = synth.make_body_model(int(g["num_verts"]), seed=int(g["model_seed"]))
    vp = synth.make_vposer(seed=int(g["vposer_seed"]))
    f = FittingOracle(SMPLXOracle(bm), VPoserDecoder.from_data(vp), g["scene"], g["vid"], list(g["camerapose"]), 300,
                      num_iter=int(g["num_iter"]), one_direction_chamfer=False)
    body_rec, scale, cam = f.fitting_local(torch.tensor(g["body_in"]), int(g["n_left"]))
    assert bool((f.contact_weight == 0.5).all())
    # same torch ops, but summation orders differ slightly from the reference's cal_loss2 graph and the L1
    # kinks amplify that: 99.8 % of the entries agree to 5e-6, the rest to 2.3e-5
    err = np.abs(body_rec.numpy() - g["body_rec"])
    assert np.mean(err < 5e-6) > 0.995 and err.max() < 1e-4, (np.mean(err < 5e-6), err.max())
    np.testing.assert_allclose(float(scale), float(g["scale"]), rtol=0, atol=1e-6)
    np.testing.assert_allclose(cam.numpy(), g["camera_ext"], rtol=0, atol=1e-6)     # never stepped in this mode
    log, ref = np.array(f.loss_log), g["log"]
    np.testing.assert_allclose(log[:, 0], ref[:, 1], atol=2e-6)
    np.testing.assert_allclose(log[:, 3], ref[:, 4], atol=2e-6)
    np.testing.assert_allclose(log[:, 5], ref[:, 6], atol=2e-6)
    log2, ref2 = np.array(f.loss_log2), g["log2"]
    for k in range(5):
        np.testing.assert_allclose(log2[:, k], ref2[:, k + 1], atol=2e-6)


@pytest.mark.parametrize("sfx", ["", "_b"])
def test_500_iteration_fixture_is_the_oracles_trajectory_and_the_yardsticks_say_what_design_says(golden_dir, sfx):
    """ref_global_500it.npz (the reference's own loop at its real budget, global_optimization.py:672) against the oracle: the
    first 20 iterations are re-run here and must land on the fixture's snapshot to rounding (the whole 500 take 8 minutes:
    tests/golden/make_golden.py --yardstick500 ran them and committed oracle_global_500it_*.npz).  The yardstick fixtures then
    carry the statement DESIGN.md section 7 makes: the SAME oracle code in fp32 on one thread instead of four agrees with the
    reference's run to rounding for 20 iterations and is centimetres away after 500 -- as far as the fp64 run is."""
    from tests.parity500 import distance_report
    g = _load(golden_dir, f"ref_global_500it{sfx}.npz")      # ("_b", r5: a second run of the reference's loop -- other seeds, 220 contact vertices)
    bm = synth.make_body_model(int(g["num_verts"]), seed=int(g["model_seed"]))
    vp = synth.make_vposer(seed=int(g["vposer_seed"]))
    assert sha(bm.posedirs) == str(g["sha_posedirs"]) and sha(vp.fc2_w) == str(g["sha_fc2"])
    assert int(g["num_iter"]) == 500 and g["log"].shape == (500, 7)
    f = FittingOracle(SMPLXOracle(bm), VPoserDecoder.from_data(vp), g["scene"], g["vid"], list(g["camerapose"]), 300, num_iter=500,
                      one_direction_chamfer=False)
    x78 = rotrepr.convert_to_6D_rot(torch.tensor(g["body_in"]))
    idx1 = f.init(x78)
    np.testing.assert_array_equal(idx1, g["idx1"])
    x78 = x78.detach()
    for ii in range(20):
        f.step(ii, x78, idx1)
    its = [int(k) for k in g["snap_iters"]]
    np.testing.assert_allclose(f.body_rotation_rec.detach().numpy(), g["snap_x78"][its.index(20)], rtol=0, atol=5e-6)
    np.testing.assert_allclose(np.array(f.loss_log)[:, 5], g["log"][:20, 6], rtol=0, atol=2e-6)
    lines = list(g["camerapose"])
    snap = lambda d, k: (d["snap_x78"][[int(v) for v in d["snap_iters"]].index(k)], d["snap_scale"][[int(v) for v in d["snap_iters"]].index(k)],
                         d["snap_cam"][[int(v) for v in d["snap_iters"]].index(k)])
    for name in ("f64", "f32t1"):
        y = _load(golden_dir, f"oracle_global_500it{sfx}_{name}.npz")
        r20, r400, r500 = (distance_report(bm, vp, lines, snap(y, k), snap(g, k)) for k in (20, 400, 500))
        assert r20["vert_mm_mean"] < (0.05 if sfx == "" else 0.2), (name, r20)   # (fp64 vs fp32: 0.009 mm; one thread: 0; fixture b: 0.10 / 0)
        assert 2.0 < r400["vert_mm_mean"] < 12.0 and r400["cam_max"] < 1e-6, (name, r400)
        assert 8.0 < r500["vert_mm_mean"] < 40.0 and r500["vert_mm_max"] > 30.0, (name, r500)
        rel = np.abs(y["log"][:, 5] - g["log"][:, 6]) / g["log"][:, 6]
        assert rel.max() < 4e-2 and rel[:400].max() < 5e-3, (name, rel.max())
