"""Loaders for the real (licensed, not shipped) assets the reference reads: SMPLX_NEUTRAL.npz
(global_optimization.py:154-155, :669) and the VPoser v1 snapshot (:153, :670).  Output objects
carry the same attribute names as synth.BodyModelData / synth.VPoserData.  SURVEY.md A.2 / A.3."""
from __future__ import annotations

import glob
import os

import numpy as np

from .synth import BodyModelData, VPoserData


def load_smplx_npz(model_folder: str, gender: str = "neutral", num_pca_comps: int = 12) -> BodyModelData:
    """`smplx.create(model_folder, model_type='smplx', gender='neutral', ext='npz')` looks for
    <folder>/smplx/SMPLX_NEUTRAL.npz (or the file itself)."""
    cands = [model_folder, os.path.join(model_folder, "smplx", f"SMPLX_{gender.upper()}.npz"),
             os.path.join(model_folder, f"SMPLX_{gender.upper()}.npz")]
    path = next((c for c in cands if os.path.isfile(c)), None)
    if path is None:
        raise FileNotFoundError(f"SMPL-X model not found under {model_folder} (tried {cands[1:]})")
    d = np.load(path, allow_pickle=True, encoding="latin1")
    shapedirs = np.asarray(d["shapedirs"], dtype=np.float32)
    # old files: 10 shape + 10 expression; new files: 300 shape + 100 expression
    n_shape = 10
    if shapedirs.shape[2] > 20:
        expr0 = 300
        shapedirs = np.concatenate([shapedirs[:, :, :n_shape], shapedirs[:, :, expr0:expr0 + 10]], 2)
    V = shapedirs.shape[0]
    posedirs = np.asarray(d["posedirs"], dtype=np.float32).reshape(V * 3, -1).T       # [486, 3V]
    parents = np.asarray(d["kintree_table"][0], dtype=np.int64).astype(np.int32)
    parents[0] = -1
    return BodyModelData(
        v_template=np.asarray(d["v_template"], dtype=np.float32), shapedirs=shapedirs,
        posedirs=np.ascontiguousarray(posedirs), J_regressor=np.asarray(d["J_regressor"], dtype=np.float32),
        parents=parents, lbs_weights=np.asarray(d["weights"], dtype=np.float32),
        hands_componentsl=np.asarray(d["hands_componentsl"], dtype=np.float32)[:num_pca_comps],
        hands_componentsr=np.asarray(d["hands_componentsr"], dtype=np.float32)[:num_pca_comps],
        hands_meanl=np.asarray(d["hands_meanl"], dtype=np.float32),
        hands_meanr=np.asarray(d["hands_meanr"], dtype=np.float32))


def load_vposer_snapshot(ckpt_dir: str) -> VPoserData:
    """Decoder weights from <ckpt_dir>/snapshots/*.pt (load_vposer picks the latest)."""
    import torch
    files = sorted(glob.glob(os.path.join(ckpt_dir, "snapshots", "*.pt")))
    if not files:
        raise FileNotFoundError(f"no VPoser snapshot under {ckpt_dir}/snapshots/*.pt")
    sd = torch.load(files[-1], map_location="cpu")
    g = lambda k: sd[k].detach().cpu().numpy().astype(np.float32)
    return VPoserData(fc1_w=g("bodyprior_dec_fc1.weight"), fc1_b=g("bodyprior_dec_fc1.bias"),
                      fc2_w=g("bodyprior_dec_fc2.weight"), fc2_b=g("bodyprior_dec_fc2.bias"),
                      out_w=g("bodyprior_dec_out.weight"), out_b=g("bodyprior_dec_out.bias"))
