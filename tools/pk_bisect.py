#!/usr/bin/env python3
"""Where does the packed-fp32 glitch of DESIGN.md section 7 live?  (run on the GPU box.)
  A. the library built WITH packed fp32 (FDC_PK=+ tools/build_variant.sh pk): body-model forwards on stream 1 next to the
     full-mesh blend product on stream 2 -- does r3's finding still reproduce here, and WHICH joints / elements differ?
  B. the stand-alone victim of tools/pk_f32_mfma_repro.hip (explicit v_pk_* vs scalar instructions, compared in place; no
     library code) on stream 1 next to the SAME neighbour -- a hardware interplay would show here too;
  C. variants of the library in which only ONE kernel is compiled without packed fp32 (FDC_NOPK_<KERNEL> -> per-function
     target attribute): which kernel's packed code is it?
usage: python tools/pk_bisect.py [reps]"""
import ctypes
import os
import subprocess
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
VARIANT = sys.argv[2] if len(sys.argv) > 2 else "pk"
os.environ["FDCAP_LIB"] = os.path.join(ROOT, "4dcapture-fpv_amd", f"libfdcap_hip_{VARIANT}.so")
os.environ["FDCAP_ALLOW_PK_F32"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402
import fdcap_amd  # noqa: E402,F401
from fdcap_amd import capi, ops  # noqa: E402
from fdcap_amd.fitting import FittingOP  # noqa: E402
from fdcap_amd.io import read_camerapose  # noqa: E402
from tests.test_gpu_sharded import _inputs  # noqa: E402

print("library:", os.environ["FDCAP_LIB"], capi.load_library().fdcap_build_info().decode(), "packed instructions:", capi.count_packed_fp32(os.environ["FDCAP_LIB"]), flush=True)
N = 100
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def make(stream):
    with torch.cuda.stream(stream):
        bm, vp, clip, scene, vid = _inputs(N)
        fop = FittingOP({"num_iter": 40}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid, camera_ext=read_camerapose(clip.camerapose_lines))
        stream.synchronize()
    return fop


f1, f2 = make(s1), make(s2)
rng = np.random.default_rng(3)
B = 256
kw = dict(body_pose=rng.standard_normal((B, 63)) * 0.3, transl=rng.standard_normal((B, 3)), global_orient=rng.standard_normal((B, 3)),
          betas=rng.standard_normal((B, 10)), left_hand_pose=rng.standard_normal((B, 12)), right_hand_pose=rng.standard_normal((B, 12)))
with torch.cuda.stream(s1):
    kw = {k: torch.tensor(v, dtype=torch.float32).cuda() for k, v in kw.items()}
    model = ops.BodyModel(f1.ctx)
    out = model(return_verts=True, **kw)
    s1.synchronize()
    ref_v, ref_j = out.vertices.clone(), out.joints.clone()
stop = threading.Event()


def background():
    ms = ctypes.c_float()
    with torch.cuda.stream(s2):
        while not stop.is_set():
            capi.check(f2.ctx.lib.fdcap_time_blend_gemm(f2.ctx.handle, 1024, 10, ctypes.byref(ms), capi.current_stream()), "blend")


# ---- B first (it does not depend on the variant's victim code): stand-alone packed chains next to the library's neighbour
so = "/tmp/pk_victim.so"
if VARIANT == "pk":
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-DPK_VICTIM_LIB", "-o", so,
                           os.path.join(ROOT, "tools", "pk_f32_mfma_repro.hip")], stderr=subprocess.DEVNULL)
    vic = ctypes.CDLL(so)
    vic.pk_chain_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.c_void_p]
    vic.pk_war_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]
    rec = torch.zeros(128, dtype=torch.int32, device="cuda")
    vic.pk_tree_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]
    vic.pk_swz_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.c_void_p]
    for lds in (0, 3, 9, 16):
        for nb in (False, True):
            rec.zero_()
            stop.clear()
            bg = threading.Thread(target=background) if nb else None
            if bg:
                bg.start()
            with torch.cuda.stream(s1):
                for k in range(REPS * 4):
                    if lds == 16:
                        vic.pk_swz_launch(ctypes.c_void_p(rec.data_ptr()), 600, 17 * k, 55, capi.current_stream())
                    elif lds == 3:
                        vic.pk_tree_launch(ctypes.c_void_p(rec.data_ptr()), 40, 17 * k, capi.current_stream())
                    elif lds == 2:
                        vic.pk_war_launch(ctypes.c_void_p(rec.data_ptr()), 600, 17 * k, capi.current_stream())
                    else:
                        vic.pk_chain_launch(ctypes.c_void_p(rec.data_ptr()), 600, 17 * k, lds, capi.current_stream())
                    if k % 16 == 15:
                        s1.synchronize()
                s1.synchronize()
            stop.set()
            if bg:
                bg.join()
            r = rec.cpu().numpy().view(np.uint32)
            kind = {0: 'registers only', 1: 'LDS hand-overs', 2: 'C: sources reloaded behind the packed ops', 3: 'D: kinematic chain handed through LDS',
                    5: 'E: LDS hand-overs, lanes 55-63 masked off', 9: 'F: LDS hand-overs, lanes 25, 28, ..., 52 only', 16: 'H: op_sel-swizzled packed sequence, lanes 0-54'}[lds]
            print(f"B. stand-alone victim ({kind}) {'next to the blend product' if nb else 'alone'}: "
                  f"{REPS * 4} launches, {int(r[0])} mismatches", [tuple(int(v) for v in r[8 * (k + 1):8 * (k + 1) + 7]) for k in range(min(int(r[0]), 4))], flush=True)

# ---- A: the library's own forward
stop.clear()
bg = threading.Thread(target=background)
bg.start()
bad, where = 0, {}
try:
    with torch.cuda.stream(s1):
        for _ in range(REPS):
            out = model(return_verts=True, **kw)
            if not (torch.equal(out.vertices, ref_v) and torch.equal(out.joints, ref_j)):
                bad += 1
                dj = (out.joints != ref_j).nonzero().cpu().numpy()
                for b, j, e in dj[:50]:
                    where[(int(j), int(e))] = where.get((int(j), int(e)), 0) + 1
        s1.synchronize()
finally:
    stop.set()
    bg.join()
print(f"A. [{VARIANT}] {bad} of {REPS} body-model forwards next to the blend product differ; (joint, xyz element) -> count:", dict(sorted(where.items())[:30]), flush=True)
