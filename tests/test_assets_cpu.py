"""assets.py: loaders for the real (licensed, absent) model files, checked on synthetic files written in the REAL key layout --
SMPLX_NEUTRAL.npz as smplx.create(model_path, model_type='smplx', gender='neutral', ext='npz', num_pca_comps=12) reads it
(/root/reference/global_optimization.py:154-168) and a VPoser v1.0 snapshot as load_vposer(dir, vp_model='snapshot') does (:153).
SURVEY.md Appendix A.2 / A.3 hold the key names and shapes (upstream recall: the files themselves are not in this image)."""
import os

import numpy as np
import pytest
import torch

import fdcap_amd  # noqa: F401
from fdcap_amd import assets, synth

V = 96


def _write_npz(path, shape_cols, rng, lbs_nnz=9):
    bm = synth.make_body_model(V, seed=3, lbs_nnz=lbs_nnz)
    shapedirs = (rng.standard_normal((V, 3, shape_cols)) * 0.01).astype(np.float64)          # the real file is float64
    posedirs = (rng.standard_normal((V, 3, 486)) * 0.003).astype(np.float64)                  # [V,3,486]: vertex-major, basis last
    kintree = np.stack([bm.parents.astype(np.int64), np.arange(55, dtype=np.int64)])
    kintree[0, 0] = 4294967295                                                                # the real root marker (uint32 -1)
    comps_l = rng.standard_normal((45, 45))
    comps_r = rng.standard_normal((45, 45))
    np.savez(path, v_template=bm.v_template.astype(np.float64), shapedirs=shapedirs, posedirs=posedirs,
             J_regressor=bm.J_regressor.astype(np.float64), kintree_table=kintree, weights=bm.lbs_weights.astype(np.float64),
             hands_componentsl=comps_l, hands_componentsr=comps_r, hands_meanl=rng.standard_normal(45),
             hands_meanr=rng.standard_normal(45),
             # keys the path never reads (faces, landmarks, dynamic landmarks): must be ignored
             f=np.zeros((10, 3), np.uint32), lmk_faces_idx=np.zeros(51, np.int64), lmk_bary_coords=np.zeros((51, 3)),
             joint2num=np.array({"Pelvis": 0}, dtype=object))
    return bm, shapedirs, posedirs, comps_l, comps_r


@pytest.mark.parametrize("shape_cols", [20, 400])
def test_load_smplx_npz_both_shapedirs_layouts(tmp_path, shape_cols):
    rng = np.random.Generator(np.random.PCG64(50 + shape_cols))
    folder = tmp_path / "models"
    (folder / "smplx").mkdir(parents=True)
    bm, shapedirs, posedirs, cl, cr = _write_npz(str(folder / "smplx" / "SMPLX_NEUTRAL.npz"), shape_cols, rng)
    got = assets.load_smplx_npz(str(folder))                     # the reference passes the FOLDER (:154, './models' :669)
    assert got.num_verts == V and got.v_template.dtype == np.float32
    # 10 betas + 10 expression coefficients: columns 0:10 and (legacy 20-column files) 10:20 / (400-column files) 300:310
    e0 = 10 if shape_cols == 20 else 300
    np.testing.assert_array_equal(got.shapedirs[:, :, :10], shapedirs[:, :, :10].astype(np.float32))
    np.testing.assert_array_equal(got.shapedirs[:, :, 10:20], shapedirs[:, :, e0:e0 + 10].astype(np.float32))
    # posedirs as smplx keeps it: reshape(-1, 486).T -> [486, 3V], column 3 v + c
    assert got.posedirs.shape == (486, 3 * V) and got.posedirs.flags["C_CONTIGUOUS"]
    for (v, c, k) in ((0, 0, 0), (5, 2, 17), (V - 1, 1, 485)):
        assert got.posedirs[k, 3 * v + c] == np.float32(posedirs[v, c, k])
    np.testing.assert_array_equal(got.parents[1:], bm.parents[1:])
    assert got.parents[0] == -1 and got.parents.dtype == np.int32
    np.testing.assert_array_equal(got.lbs_weights, bm.lbs_weights)
    assert int((got.lbs_weights != 0).sum(1).max()) == 9        # not 4-sparse: what a real file may hold
    np.testing.assert_array_equal(got.J_regressor, bm.J_regressor)
    # num_pca_comps=12 (:158): the first 12 ROWS of the 45 x 45 component matrices
    np.testing.assert_array_equal(got.hands_componentsl, cl[:12].astype(np.float32))
    np.testing.assert_array_equal(got.hands_componentsr, cr[:12].astype(np.float32))
    assert got.hands_meanl.shape == (45,) and got.hands_meanr.shape == (45,)
    # the file itself, or a folder that holds it directly, are accepted too
    direct = assets.load_smplx_npz(str(folder / "smplx" / "SMPLX_NEUTRAL.npz"))
    np.testing.assert_array_equal(direct.posedirs, got.posedirs)
    with pytest.raises(FileNotFoundError):
        assets.load_smplx_npz(str(tmp_path / "nowhere"))


def test_loaded_model_feeds_the_context_descriptor(tmp_path):
    """What capi.Context hands to fdcap_ctx_create: every array float32 / int32, C-contiguous, of the shapes include/fdcap.h states."""
    rng = np.random.Generator(np.random.PCG64(7))
    p = str(tmp_path / "SMPLX_NEUTRAL.npz")
    _write_npz(p, 400, rng)
    bm = assets.load_smplx_npz(p)
    for name, shape in (("v_template", (V, 3)), ("shapedirs", (V, 3, 20)), ("posedirs", (486, 3 * V)), ("J_regressor", (55, V)),
                        ("lbs_weights", (V, 55)), ("hands_componentsl", (12, 45)), ("hands_meanl", (45,))):
        a = getattr(bm, name)
        assert a.shape == shape and a.dtype == np.float32, name
    # the oracle's body model runs on it (it is what the parity tests would compare a real-file run with)
    from oracle.smplx import SMPLXOracle
    out = SMPLXOracle(bm)(return_verts=True, body_pose=torch.zeros(2, 63), transl=torch.zeros(2, 3), global_orient=torch.zeros(2, 3),
                          betas=torch.zeros(2, 10), left_hand_pose=torch.zeros(2, 12), right_hand_pose=torch.zeros(2, 12))
    assert out.vertices.shape == (2, V, 3) and torch.isfinite(out.vertices).all()


def test_load_vposer_snapshot_picks_the_latest_and_only_the_decoder(tmp_path):
    vp = synth.make_vposer(seed=9)
    snaps = tmp_path / "vposer" / "snapshots"
    snaps.mkdir(parents=True)
    t = lambda a: torch.tensor(a)
    sd = {"bodyprior_dec_fc1.weight": t(vp.fc1_w), "bodyprior_dec_fc1.bias": t(vp.fc1_b), "bodyprior_dec_fc2.weight": t(vp.fc2_w),
          "bodyprior_dec_fc2.bias": t(vp.fc2_b), "bodyprior_dec_out.weight": t(vp.out_w), "bodyprior_dec_out.bias": t(vp.out_b),
          # encoder half of the checkpoint: present in the real file, never read by decode() (:270)
          "bodyprior_enc_fc1.weight": torch.zeros(512, 189), "bodyprior_enc_mu.weight": torch.zeros(32, 512)}
    old = dict(sd)
    old["bodyprior_dec_out.bias"] = torch.zeros(126)
    torch.save(old, str(snaps / "TR00_E010.pt"))
    torch.save(sd, str(snaps / "TR00_E096.pt"))
    got = assets.load_vposer_snapshot(str(tmp_path / "vposer"))
    for k in ("fc1_w", "fc1_b", "fc2_w", "fc2_b", "out_w", "out_b"):
        np.testing.assert_array_equal(getattr(got, k), getattr(vp, k))
        assert getattr(got, k).dtype == np.float32
    assert got.fc1_w.shape == (512, 32) and got.fc2_w.shape == (512, 512) and got.out_w.shape == (126, 512)
    with pytest.raises(FileNotFoundError):
        assets.load_vposer_snapshot(str(tmp_path / "empty"))
