"""Host-side file boundary (body_gen -> smoothed_body pickles, camerapose.txt, scene readers) and
the C-ABI surface: the library loads and exports every symbol include/fdcap.h declares (no compute
calls -- this file runs without a GPU)."""
import ctypes
import os
import pickle
import re

import numpy as np
import torch

import fdcap_amd  # noqa: F401
from fdcap_amd import capi, io, synth
from oracle.fitting import extract_ext

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "fdcap.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int32_t|void|const char\*)\s+(fdcap_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 20
    lib = capi.load_library()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fdcap.h but not exported"
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    assert lib.fdcap_version().startswith(b"fdcap-hip")


def test_library_is_built_without_packed_fp32_instructions():
    """The build requirement at the top of csrc/fdcap.hip (NOTES.md section 6), enforced on the artefact that ships: the
    library says so (fdcap_build_info, which capi.load_library also insists on) AND its gfx950 code object holds no
    v_pk_{fma,mul,add}_f32 -- a build script that passed the define without the target feature would fail here."""
    lib = capi.load_library()
    assert b"packed_fp32=off" in lib.fdcap_build_info()
    assert capi.count_packed_fp32() == 0


def test_scene_size_limit_is_refused_not_truncated():
    hdr = open(os.path.join(ROOT, "include", "fdcap.h")).read()
    lim = int(re.search(r"#define FDCAP_MAX_SCENE_POINTS (\d+)", hdr).group(1))
    assert lim * 32 < 2 ** 31            # the fragment stream (32 B per point) stays addressable with 32-bit offsets
    csrc = os.path.join(ROOT, "4dcapture-fpv_amd", "csrc")        # (the translation unit: fdcap.hip + the parts it includes)
    src = "".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)))
    assert "ns > FDCAP_MAX_SCENE_POINTS) return FDCAP_E_ARG" in src


def test_struct_layouts_match_header():
    # 18 pointers/ints of fdcap_model_desc and 13 fields of fdcap_opt_config, natural alignment
    assert ctypes.sizeof(capi.OptConfig) == 13 * 4
    assert ctypes.sizeof(capi.ModelDesc) == 8 + 2 * 8 + 8 + 14 * 8   # int32+pad, 2 ptr, int32+pad, 14 ptr


def test_ctypes_structs_have_the_c_compilers_layout(tmp_path):
    """Every struct that crosses the boundary by value or by pointer: field offsets and size as gcc lays out include/fdcap.h,
    against the ctypes mirrors in capi.py (a field added on one side only shows here, not as a wrong number on the GPU)."""
    import subprocess
    pairs = {"fdcap_opt_config": capi.OptConfig, "fdcap_fit2d_stage": capi.Fit2dStage, "fdcap_lbfgs_config": capi.LbfgsConfig,
             "fdcap_model_desc": capi.ModelDesc}
    lines = ["#include <stddef.h>", "#include <stdio.h>", '#include "fdcap.h"', "int main(void) {"]
    for cname, ct in pairs.items():
        lines.append(f'    printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'    printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["    return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = str(tmp_path / "layout")
    r = subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = {}
    for ln in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines():
        c, f, v = ln.split()
        got[(c, f)] = int(v)
    for cname, ct in pairs.items():
        assert got[(cname, "size")] == ctypes.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert got[(cname, fname)] == getattr(ct, fname).offset, (cname, fname)


def test_body_gen_roundtrip_and_output_schema(tmp_path):
    clip = synth.make_clip(6, seed=1)
    io.write_body_gen(clip.body_params, str(tmp_path / "sample" / "body_gen"))
    got = io.load_body_gen(str(tmp_path / "sample" / "body_gen"))
    np.testing.assert_array_equal(got, clip.body_params)
    cam = np.tile(np.eye(4, dtype=np.float32), (6, 1, 1))
    files = io.save_result(clip.body_params, np.float32(1.7), cam, str(tmp_path / "smoothed_body"))
    assert [os.path.basename(f) for f in files][:2] == ["body_gen_000000.pkl", "body_gen_000001.pkl"]
    d = pickle.load(open(files[3], "rb"))
    # what global_vis.py:116-129 / local_vis.py:309-313 read
    assert set(d) == {"transl", "global_orient", "betas", "body_pose", "left_hand_pose", "right_hand_pose",
                      "camera_translation", "scale", "camera_ext"}
    assert d["body_pose"].shape == (1, 32) and d["betas"].shape == (1, 10) and d["camera_ext"].shape == (4, 4)
    np.testing.assert_array_equal(d["camera_translation"], clip.body_params[3:4, 72:75])
    assert float(d["scale"]) == np.float32(1.7)


def test_camerapose_parsing_matches_reference_restatement(tmp_path):
    clip = synth.make_clip(9, seed=2)
    p = tmp_path / "camerapose.txt"
    p.write_text("\n".join(clip.camerapose_lines) + "\n")
    got = io.read_camerapose(str(p))
    want = extract_ext(clip.camerapose_lines).numpy()
    np.testing.assert_array_equal(got, want)
    np.testing.assert_allclose(got, clip.cam_ext, atol=1e-5)       # the lines invert to the generator's poses
    assert clip.camerapose_lines[0].startswith(" ")                # utils/camerapose_helper.py:27 format


def test_scene_readers(tmp_path):
    pts = synth.make_scene(1234, seed=5)
    for binary in (True, False):
        f = str(tmp_path / f"s{int(binary)}.ply")
        io.write_ply_points(f, pts, binary=binary)
        np.testing.assert_allclose(io.read_scene_points(f), pts, rtol=0, atol=0 if binary else 1e-7)
    np.savetxt(tmp_path / "s.xyz", pts)
    np.testing.assert_allclose(io.read_scene_points(str(tmp_path / "s.xyz")), pts, atol=1e-6)


def test_contact_ids_follow_reference_set_semantics(tmp_path):
    import json
    (tmp_path / "L_Leg.json").write_text(json.dumps({"verts_ind": [5, 3, 3, 9], "faces_ind": [1]}))
    (tmp_path / "R_Leg.json").write_text(json.dumps({"verts_ind": [7, 7, 2], "faces_ind": [1]}))
    vid = io.read_contact_ids(str(tmp_path))
    assert sorted(vid[:3].tolist()) == [3, 5, 9] and sorted(vid[3:].tolist()) == [2, 7]   # :87 list(set(...)), :90 concat


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "4dcapture-fpv_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in text and "from oracle" not in text, fn


def test_colmap_helpers_follow_reference_scripts(tmp_path):
    img = tmp_path / "images.txt"
    img.write_text("# Image list\n# IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n# Number of images: 2\n"
                   "1 0.9 0.1 0.2 0.3 1.0 2.0 3.0 1 frame_000001.jpg\n"
                   "10.5 20.5 -1 30.5 40.5 -1\n"
                   "2 0.8 -0.1 0.25 0.35 1.5 2.5 3.5 1 frame_000002.jpg\n"
                   "11.5 21.5 -1\n")
    lines = io.colmap_images_to_camerapose(str(img), str(tmp_path / "camerapose.txt"))
    assert lines == [" 0.9 0.1 0.2 0.3 1.0 2.0 3.0", " 0.8 -0.1 0.25 0.35 1.5 2.5 3.5"]
    ext = io.read_camerapose(str(tmp_path / "camerapose.txt"))
    assert ext.shape == (2, 4, 4)
    pts = tmp_path / "points3D.txt"
    pts.write_text("# 3D point list\n# POINT3D_ID, X, Y, Z, R, G, B, ERROR, TRACK[]\n# Number of points: 2\n"
                   "1 0.5 1.5 2.5 255 0 0 0.1 1 2\n2 -0.5 -1.5 -2.5 0 255 0 0.2 2 3\n")
    xyz = io.colmap_points_to_xyz(str(pts), str(tmp_path / "xyz.xyz"))
    np.testing.assert_allclose(xyz, [[0.5, 1.5, 2.5], [-0.5, -1.5, -2.5]])
    np.testing.assert_allclose(io.read_scene_points(str(tmp_path / "xyz.xyz")), xyz)


def test_the_header_is_plain_c_and_the_c_host_program_compiles_against_it(tmp_path):
    """include/fdcap.h is the boundary a non-Python caller binds: tests/c_abi_fit.c (the whole optimisation from a C host; run on
    the GPU by tests/test_gpu_c_host.py) must compile as C11 with -Wall -Wextra -Werror and link against the built library --
    no C++-isms in the header, every entry point it uses exported."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from fdcap_amd import capi
    pkg = os.path.dirname(capi.LIB_PATH)
    exe = str(tmp_path / "c_abi_fit")
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), "-I", "/opt/rocm/include",
                        os.path.join(root, "tests", "c_abi_fit.c"), "-o", exe, "-L", pkg, "-lfdcap_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
                        f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr
    # ... and the batched L-BFGS's C host (tests/c_abi_lbfgs.c; compile and link only: it needs a GPU to run)
    exe2 = str(tmp_path / "c_abi_lbfgs")
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), "-I", "/opt/rocm/include",
                        os.path.join(root, "tests", "c_abi_lbfgs.c"), "-o", exe2, "-L", pkg, "-lfdcap_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
                        "-lm", f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_missing_rccl_is_an_error_code_not_a_crash():
    """on a host without librccl the loader built its message from two dlerror() calls (the second returns NULL ->
    std::string + NULL -> crash).  A fresh process whose only candidate is a file that does not exist must get FDCAP_E_COMM from
    fdcap_comm_unique_id and a readable message from fdcap_comm_last_error(NULL) -- the code FittingOP._init_c_comm falls back on."""
    import subprocess
    import sys
    code = ("import ctypes, sys; sys.path.insert(0, %r); import fdcap_amd; from fdcap_amd import capi; lib = capi.load_library(); "
            "idb = (ctypes.c_uint8 * 128)(); rc = lib.fdcap_comm_unique_id(idb); msg = lib.fdcap_comm_last_error(None); "
            "print(rc, msg.decode())") % ROOT
    env = dict(os.environ, FDCAP_RCCL_LIB="/nonexistent/librccl-not-here.so", FDCAP_RCCL_LIB_ONLY="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-800:]
    rc, msg = p.stdout.strip().split(" ", 1)
    assert int(rc) == -4 and "librccl not found" in msg and "librccl-not-here" in msg, p.stdout
