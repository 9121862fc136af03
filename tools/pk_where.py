#!/usr/bin/env python3
"""Which of pose_fwd_kernel's outputs carries the wrong word of NOTES.md section 6?  Library built with packed fp32 AND the debug
buffer taps (FDC_PK=+ tools/build_variant.sh pkdbg -DFDC_DEBUG_BUFFERS): the optimiser's forward on stream 1 next to the full-mesh
blend product on stream 2, every repetition's Rm (local rotations), Jrest (rest joints), G (world transforms), A (skinning
transforms), PF (pose features) compared with the same forward run alone."""
import ctypes
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
VARIANT = sys.argv[2] if len(sys.argv) > 2 else "pkdbg"
os.environ["FDCAP_LIB"] = os.path.join(ROOT, "4dcapture-fpv_amd", f"libfdcap_hip_{VARIANT}.so")
os.environ["FDCAP_ALLOW_PK_F32"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402
import fdcap_amd  # noqa: E402,F401
from fdcap_amd import capi  # noqa: E402
from fdcap_amd.fitting import FittingOP  # noqa: E402
from fdcap_amd.io import read_camerapose  # noqa: E402
from tests.test_gpu_sharded import _inputs  # noqa: E402

raw = ctypes.CDLL(os.environ["FDCAP_LIB"])
raw.fdcap_debug_rows.restype = ctypes.c_int
raw.fdcap_debug_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
N = 256
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def make(stream):
    with torch.cuda.stream(stream):
        bm, vp, clip, scene, vid = _inputs(N)
        fop = FittingOP({"num_iter": 40}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid, camera_ext=read_camerapose(clip.camerapose_lines))
        x78 = torch.empty(N, capi.XDIM, device="cuda")
        capi.check(fop.ctx.lib.fdcap_params_75_to_78(capi.dptr(torch.tensor(clip.body_params).cuda()), N, capi.dptr(x78), capi.current_stream()), "75->78")
        fop._mode = "global"
        fop.init(x78)
        stream.synchronize()
    return fop


f1, f2 = make(s1), make(s2)
TAPS = {"Rm": (10, 55 * 9), "Jrest": (11, 55 * 3), "G": (6, 55 * 12), "A": (2, 55 * 12), "PF": (1, 496), "O": (0, 126)}


def forward_and_tap():
    out = {}
    with torch.cuda.stream(s1):
        jw = torch.empty(N, 23, 3, device="cuda")
        capi.check(f1.ctx.lib.fdcap_opt_forward_world(f1.ctx.handle, None, capi.dptr(jw), capi.current_stream()), "fw")
        for name, (which, w) in TAPS.items():
            t = torch.empty(N, w, device="cuda")
            raw.fdcap_debug_rows(f1.ctx.handle, which, ctypes.c_void_p(t.data_ptr()), capi.current_stream())
            out[name] = t
        s1.synchronize()
    return out


ref = forward_and_tap()
stop = threading.Event()


def background():
    ms = ctypes.c_float()
    with torch.cuda.stream(s2):
        while not stop.is_set():
            capi.check(f2.ctx.lib.fdcap_time_blend_gemm(f2.ctx.handle, 1024, 10, ctypes.byref(ms), capi.current_stream()), "blend")


bg = threading.Thread(target=background)
bg.start()
hits = {k: {} for k in TAPS}
nbad = 0
try:
    for _ in range(REPS):
        got = forward_and_tap()
        bad = False
        for name in TAPS:
            d = (got[name] != ref[name]).nonzero().cpu().numpy()
            if len(d):
                bad = True
                w = TAPS[name][1] // 55 if name not in ("PF", "O") else 1
                for fr, col in d[:200]:
                    key = (int(col) // w, int(col) % w) if w > 1 else int(col)
                    hits[name][key] = hits[name].get(key, 0) + 1
        nbad += bad
finally:
    stop.set()
    bg.join()
print(f"[{VARIANT}] {nbad} of {REPS} forwards next to the blend product differ from the forward alone")
for name in TAPS:
    print(f"  {name}: (joint, element) -> count", dict(sorted(hits[name].items())[:40]))
