"""The drop-in boundary is a C-ABI (include/fdcap.h): a plain C program -- tests/c_abi_fit.c, compiled here with gcc as C11 against the
header and linked to libfdcap_hip.so + the HIP runtime, no Python and no torch in its process -- runs the whole optimisation through
it (fdcap_ctx_create ... fdcap_opt_run ... fdcap_opt_get_results) on files this test writes, and must land on FittingOP's results
for the same inputs BIT FOR BIT, parameters and the logged loss sums alike."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

import fdcap_amd  # noqa: F401
from fdcap_amd import capi, synth
from fdcap_amd.fitting import (FittingOP, PHASE1_CONTACT, PHASE1_SMOOTH, PHASE2_SMOOTH, PHASE2_WORLD, SCALE_INIT, find_outliers,
                               first_phase2_iter)
from fdcap_amd.io import read_camerapose

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "c_abi_fit")
    pkg = os.path.dirname(capi.LIB_PATH)
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "tests", "c_abi_fit.c"), "-o", exe, "-L", pkg, "-lfdcap_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
           f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.mark.parametrize("n,num_iter,log_every", [(23, 30, 1), (130, 12, 5)])
def test_a_c_host_lands_on_the_python_hosts_bits(tmp_path, n, num_iter, log_every):
    exe = _build(tmp_path)
    bm = synth.make_body_model(300, seed=41)
    vp = synth.make_vposer(seed=42)
    clip = synth.make_clip(n, seed=43)
    scene = synth.make_scene(7000, seed=44)
    left, right = synth.make_contact_ids(bm.v_template, per_part=20, seed=45)
    vid = np.concatenate([left, right]).astype(np.int64)
    cam0 = read_camerapose(clip.camerapose_lines)
    # ---- the Python host
    fop = FittingOP({"num_iter": num_iter}, {}, n, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid, camera_ext=cam0)
    body, scale, cam = fop.fitting(torch.tensor(clip.body_params).cuda(), "global", log_every=log_every)
    want_body, want_scale, want_cam = body.cpu().numpy(), np.float32(scale), cam.cpu().numpy().reshape(n, 16)
    want_log = fop.log
    # ---- the same inputs as files: what FittingOP.init() hands to fdcap_opt_set_inputs
    x75 = torch.tensor(clip.body_params).cuda()
    x78 = torch.empty(n, capi.XDIM, device="cuda")
    capi.check(fop.ctx.lib.fdcap_params_75_to_78(capi.dptr(x75), n, capi.dptr(x78), capi.current_stream()), "75->78")
    x78 = x78.cpu().numpy()
    idx1, pos = find_outliers(x78)
    init78 = x78.copy()
    if idx1.size and pos.size:
        init78[idx1] = x78[pos]
    mask = np.ones(n, np.float32)
    mask[idx1] = 0.0
    d = tmp_path

    def put(name, a, dt=np.float32):
        np.ascontiguousarray(a, dtype=dt).tofile(d / name)

    V = bm.v_template.shape[0]
    shapedirs = np.asarray(bm.shapedirs, np.float32)
    put("v_template.bin", bm.v_template); put("shapedirs.bin", shapedirs); put("posedirs.bin", bm.posedirs)
    put("J_regressor.bin", bm.J_regressor); put("parents.bin", bm.parents, np.int32); put("lbs_weights.bin", bm.lbs_weights)
    put("hcl.bin", bm.hands_componentsl[:12]); put("hcr.bin", bm.hands_componentsr[:12]); put("hml.bin", bm.hands_meanl); put("hmr.bin", bm.hands_meanr)
    put("w1.bin", vp.fc1_w); put("b1.bin", vp.fc1_b); put("w2.bin", vp.fc2_w); put("b2.bin", vp.fc2_b); put("w3.bin", vp.out_w); put("b3.bin", vp.out_b)
    put("scene.bin", scene); put("vid.bin", vid, np.int64)
    put("data78.bin", x78); put("init78.bin", init78); put("mask.bin", mask); put("cam.bin", np.asarray(cam0, np.float32).reshape(n, 16))
    oc = capi.OptConfig(n, n, 0, float(fop.init_lr_h), float(fop.weight_loss_rec), float(fop.weight_loss_vposer), float(fop.weight_contact),
                        PHASE1_CONTACT, PHASE1_SMOOTH, PHASE2_WORLD, PHASE2_SMOOTH, SCALE_INIT, 0)
    (d / "cfg.bin").write_bytes(bytes(oc))
    P = first_phase2_iter(num_iter)
    (d / "dims.txt").write_text(f"{V} {shapedirs.shape[2]} {len(scene)} {len(vid)} {n} {num_iter} {P} {log_every}\n")
    fop.close()
    torch.cuda.synchronize()
    # ---- the C host
    r = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "packed_fp32=off" in r.stdout
    got_body = np.fromfile(d / "out_body.bin", np.float32).reshape(n, capi.PDIM)
    got_scale = np.fromfile(d / "out_scale.bin", np.float32)[0]
    got_cam = np.fromfile(d / "out_cam.bin", np.float32).reshape(n, 16)
    np.testing.assert_array_equal(got_body, want_body)
    assert got_scale == want_scale
    np.testing.assert_array_equal(got_cam, want_cam)
    # the history: raw partial sums; FittingOP normalises them as the reference's means (fitting._append_log)
    hist = np.fromfile(d / "out_hist.bin", np.float64).reshape(-1, capi.NUM_LOSSES)
    assert len(hist) == len(want_log.iters)
    np.testing.assert_array_equal(fop.weight_loss_rec * hist[:, 0] / (n * capi.XDIM), np.asarray(want_log.l_rec))
    np.testing.assert_array_equal(hist[:, 2] / ((n - 2) * capi.XDIM), np.asarray(want_log.loss_smoothing))
    np.testing.assert_array_equal(fop.weight_contact * hist[:, 3] / (n * len(vid)), np.asarray(want_log.loss_contact))


def test_the_batched_lbfgs_from_a_c_host(tmp_path):
    """fdcap_lbfgs_* from plain C (tests/c_abi_lbfgs.c): nine problems whose objective the C host evaluates itself."""
    exe = str(tmp_path / "c_abi_lbfgs")
    pkg = os.path.dirname(capi.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                        os.path.join(ROOT, "tests", "c_abi_lbfgs.c"), "-o", exe, "-L", pkg, "-lfdcap_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
                        "-lm", f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout.strip())
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
