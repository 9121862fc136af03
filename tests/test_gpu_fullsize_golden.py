"""The reference's OWN loop at BASELINE body size (missing 2): tests/golden/make_golden.py --g5full / --g5allverts ran
/root/reference/global_optimization.py's FittingOP.fitting (:491-635, unmodified, through the stubs) with V = 10 475 vertices on
300 frames, five iterations across the phase switch:
  ref_global_5it_full.npz      500 contact vertices vs a 100 k-point scene  (the shapes of BASELINE configs 2 / 3)
  ref_global_5it_allverts.npz  all 10 475 vertices as contacts vs 20 k points (config 5: chunked skinning backward, K = 31 425 products)
so that "reference loop -> oracle -> HIP" closes on the kernel forms the bench times, not only on toy bodies.

The reference hard-codes 300 frames (:41-42, :465, :472) and the clip-sized forms (two row blocks per fragment stream, the fused
contact forward) are selected from 384 rows: each fixture is therefore run twice -- as the library would run a 300-frame clip, and
in a child process with FDCAP_CLIP_FORMS_MIN_ROWS=256 (read once per process), where fdcap_debug_kernel_forms must list the forms
configs 3 / 5 select.  Bars: those of tests/test_gpu_parity.py::test_trajectory_matches_reference_golden (Adam through L1 kinks)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

_CHILD = r"""
import ctypes, hashlib, json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import fdcap_amd
from fdcap_amd import capi, synth
from fdcap_amd.fitting import FittingOP, first_phase2_iter
from fdcap_amd.io import read_camerapose

g = np.load(os.path.join(%(golden)r, %(name)r))
bm = synth.make_body_model(int(g["num_verts"]), seed=int(g["model_seed"]))
vp = synth.make_vposer(seed=int(g["vposer_seed"]))
if "scene" in g.files:
    scene = g["scene"]
else:                                   # regenerated from its seed (deterministic generator), checked by hash
    scene = synth.make_scene(int(g["ns"]), seed=int(g["scene_seed"]))
assert hashlib.sha256(np.ascontiguousarray(scene).tobytes()).hexdigest()[:16] == str(g["sha_scene"])
assert hashlib.sha256(np.ascontiguousarray(bm.posedirs).tobytes()).hexdigest()[:16] == str(g["sha_posedirs"])
num_iter = int(g["num_iter"])
lib = capi.load_library()
buf = ctypes.create_string_buffer(4096)
lib.fdcap_debug_kernel_forms(buf, 4096, 1)
fop = FittingOP({"num_iter": num_iter}, {}, 300, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=g["vid"],
                camera_ext=read_camerapose(list(g["camerapose"])))
body, scale, cam = fop.fitting(torch.tensor(g["body_in"]).cuda(), "global", log_every=1, snapshot_at=tuple(int(k) for k in g["snap_iters"]))
lib.fdcap_debug_kernel_forms(buf, 4096, 0)
err = np.abs(body.cpu().numpy() - g["body_rec"])
snap = []
for k, x in zip(g["snap_iters"], g["snap_x78"]):
    e = np.abs(fop.snapshots[int(k)][0].cpu().numpy() - x)
    snap.append([float(np.quantile(e, 0.5)), float(np.quantile(e, 0.9)), float(np.quantile(e, 0.99)), float(e.max())])
P = first_phase2_iter(num_iter)
log = fop.log
lg = g["log"]
out = {"forms": buf.value.decode(), "idx1_equal": bool(np.array_equal(fop.idx1, g["idx1"])),
       "q": [float(v) for v in np.quantile(err, [0.5, 0.9, 0.99])], "max": float(err.max()), "hand_max": float(err[:, 48:72].max()),
       "scale_err": abs(float(scale) - float(g["scale"])), "cam_err": float(np.abs(cam.cpu().numpy() - g["camera_ext"]).max()),
       "snap": snap, "P": P, "num_iter": num_iter,
       "d_rec": np.abs(np.array(log.l_rec) - lg[:, 1]).tolist(), "d_vp": np.abs(np.array(log.l_vposer) - lg[:, 2]).tolist(),
       "d_sm": np.abs(np.array(log.loss_smoothing) - lg[:, 3]).tolist(), "d_con": np.abs(np.array(log.loss_contact) - lg[:, 4]).tolist(),
       "d_tot": np.abs(np.array(log.total) - lg[:, 6]).tolist(),
       "d_ws": np.abs(np.array(log.loss_world_smoothing)[P:] - lg[P:, 5]).tolist(), "finite": bool(np.isfinite(body.cpu().numpy()).all())}
fop.close()
print("RESULT " + json.dumps(out))
"""


def _run(name, clip_forms):
    env = dict(os.environ)
    env.pop("FDCAP_CLIP_FORMS_MIN_ROWS", None)
    if clip_forms:
        env["FDCAP_CLIP_FORMS_MIN_ROWS"] = "256"
    p = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT, "golden": GOLDEN, "name": name}], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    assert p.returncode == 0 and line, (p.stdout[-1500:], p.stderr[-3000:])
    return json.loads(line[-1][7:])


def _check(r, name):
    num_iter, P = r["num_iter"], r["P"]
    print(name, "forms:", r["forms"], "| body_rec |err| q50 q90 q99 max", r["q"], r["max"], "| per snapshot (q50 q90 q99 max)", r["snap"],
          "| scale", r["scale_err"], "cam", r["cam_err"])
    assert r["finite"] and r["idx1_equal"]
    q50, q90, q99 = r["q"]
    assert r["max"] <= 6 * 0.005
    assert q50 < 1e-6 and q90 < 1e-4 and q99 < 3e-3, r["q"]
    # (the small fixtures' bar hand_max <= 2e-6 does not carry over: at this body size the synthetic contact vertices are also skinned
    #  to hand joints, so the hand coefficients receive the contact gradient like every other column; they stay under the common bars)
    assert r["hand_max"] <= 6 * 0.005
    assert r["scale_err"] < 1e-4
    assert r["cam_err"] <= 2 * 0.005 * max(num_iter - P - 1, 0) + 1e-6
    tol = 3e-6 + 2e-6 * np.arange(num_iter)
    it = np.arange(num_iter)
    assert np.all(np.array(r["d_rec"]) <= tol) and np.all(np.array(r["d_vp"]) <= tol) and np.all(np.array(r["d_sm"]) <= tol)
    assert np.all(np.array(r["d_con"]) <= np.where(it > P, 4 * tol, tol))
    assert np.all(np.array(r["d_tot"]) <= np.where(it > P, 5 * tol, 2 * tol))
    assert np.all(np.array(r["d_ws"]) <= 4 * tol[P:])
    # the first step is one Adam step from identical inputs: every entry within the sign-flip bound, nearly all at rounding level
    assert r["snap"][0][3] <= 2 * 0.005 + 1e-6 and r["snap"][0][1] < 1e-5


@pytest.mark.parametrize("clip_forms", [False, True], ids=["as_a_300_frame_clip", "clip_sized_forms"])
def test_reference_loop_at_baseline_body_size_500_contacts(clip_forms):
    r = _run("ref_global_5it_full.npz", clip_forms)
    _check(r, "ref_global_5it_full")
    forms = r["forms"].split(";")
    assert "skin_bwd_vec_kernel" in forms and "nn_stream4_kernel<1,1,1>" in forms
    if clip_forms:       # what BASELINE config 3 selects (bench.py's per-kernel table)
        assert "blend_skin_fwd_kernel" in forms and "panel_gemm3_rb2k_kernel" in forms, forms
    else:
        assert "skin_fwd_kernel" in forms and "panel_gemm3_kernel" in forms, forms


@pytest.mark.parametrize("clip_forms", [False, True], ids=["as_a_300_frame_clip", "clip_sized_forms"])
def test_reference_loop_at_baseline_body_size_all_vertices_as_contacts(clip_forms):
    if not os.path.exists(os.path.join(GOLDEN, "ref_global_5it_allverts.npz")):
        pytest.skip("fixture not generated (tests/golden/make_golden.py --g5allverts)")
    r = _run("ref_global_5it_allverts.npz", clip_forms)
    _check(r, "ref_global_5it_allverts")
    forms = r["forms"].split(";")
    # BASELINE config 5's forms: wide forward product, K-loop data gradient (K = 31 425), chunked skinning backward
    assert "panel_gemm3_wide_kernel" in forms and "panel_gemm3_kloop_kernel" in forms, forms
    assert any(f.startswith("skin_bwd_kernel(chunks") for f in forms), forms
