"""Command line of the hot path, argument-compatible with the reference:

    python3 global_optimization.py <body_path> <fit_path> <mode>      (global_optimization.py:658-660)

`<body_path>/results/*/*.pkl` in, `<fit_path>/body_gen_%06d.pkl` out.  The reference hard-codes
`/home/miao/<sample>/meshed-poisson.ply` and `camerapose.txt` (:667-668); the root is configurable
here (`--scene-root`, or explicit `--scene` / `--camera`).  `./models`, `./vposer/`,
`./body_segments` default to the reference's CWD-relative paths (:669-675)."""
from __future__ import annotations

import argparse
import os
import sys


def main(argv=None):
    ap = argparse.ArgumentParser(prog="global_optimization (fdcap_amd / MI355X)")
    ap.add_argument("body_path")
    ap.add_argument("fit_path")
    ap.add_argument("mode", nargs="?", default="global", choices=["global", "local", "dct"])
    ap.add_argument("--scene-root", default="/home/miao/")
    ap.add_argument("--scene", default=None, help="scene vertices (.ply/.xyz/.npy); default <root>/<sample>/meshed-poisson.ply")
    ap.add_argument("--camera", default=None, help="camerapose.txt; default <root>/<sample>/camerapose.txt")
    ap.add_argument("--models", default="./models")
    ap.add_argument("--vposer", default="./vposer/")
    ap.add_argument("--body-segments", default="./body_segments")
    ap.add_argument("--num-iter", type=int, default=500)
    ap.add_argument("--lr", type=float, default=0.005)
    ap.add_argument("--log-every", type=int, default=0)
    ap.add_argument("--dct-mat", default="../Data/DCT_Basis/60.mat", help="DCT basis .mat (:45); generated if absent")
    ap.add_argument("--dct-num-iter", type=int, default=10000, help="iterations of mode 'dct' (:596)")
    a = ap.parse_args(argv)

    import torch
    from . import io
    from .fitting import FittingOP

    sample_name = a.body_path.split("/")[-2]                                    # :662
    scene = a.scene or os.path.join(a.scene_root, sample_name, "meshed-poisson.ply")
    camera = a.camera or os.path.join(a.scene_root, sample_name, "camerapose.txt")
    fittingconfig = {"scene_verts_path": scene, "camera_path": camera, "human_model_path": a.models,
                     "vposer_ckpt_path": a.vposer, "init_lr_h": a.lr, "num_iter": a.num_iter,
                     "contact_id_folder": a.body_segments, "contact_part": ["L_Leg", "R_Leg"],
                     "verbose": bool(a.log_every), "dct_mat_path": a.dct_mat}
    lossconfig = {"weight_loss_rec": 1, "weight_loss_vposer": 0.001, "weight_contact": 0.1, "weight_collision": 0.5}
    data = io.load_body_gen(a.body_path)                                         # :688-707
    fop = FittingOP(fittingconfig, lossconfig, data.shape[0], dct_num_iter=a.dct_num_iter)
    body_rec, scale, camera_ext = fop.fitting(torch.tensor(data).cuda(), a.mode, log_every=a.log_every)
    fop.save_result(body_rec, scale, camera_ext, a.fit_path)                     # :714
    print("[INFO][fitting] fitting finish, returning optimal value")
    return 0


if __name__ == "__main__":
    sys.exit(main())
