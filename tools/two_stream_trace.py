#!/usr/bin/env python3
"""Development tool: ONE process, two fits at the same time on two HIP streams / host threads; are the joint transforms (A) after
every backward the same in every repetition?  (The one-word glitch seen with two PROCESSES on one GPU: tests/shared_gpu.py.)
Needs the -DFDC_DEBUG_BUFFERS build via FDCAP_LIB."""
import os, sys, ctypes, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fdcap_amd  # noqa
from fdcap_amd import capi
from fdcap_amd.fitting import FittingOP, first_phase2_iter
from fdcap_amd.io import read_camerapose
from tests.test_gpu_sharded import _inputs

N, ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 100, 10
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
raw = ctypes.CDLL(capi.LIB_PATH)
raw.fdcap_debug_rows.restype = ctypes.c_int
raw.fdcap_debug_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]


ITERS = 100


def make(stream):
    with torch.cuda.stream(stream):
        bm, vp, clip, scene, vid = _inputs(N)
        fop = FittingOP({"num_iter": ITERS}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid,
                        camera_ext=read_camerapose(clip.camerapose_lines))
        x78 = torch.empty(N, capi.XDIM, device="cuda")
        capi.check(fop.ctx.lib.fdcap_params_75_to_78(capi.dptr(torch.tensor(clip.body_params).cuda()), N, capi.dptr(x78), capi.current_stream()), "75->78")
        stream.synchronize()
    return fop, x78


def one_fit(stream, fop, x78, out):
    with torch.cuda.stream(stream):
        lib, h = fop.ctx.lib, fop.ctx.handle
        fop._mode = "global"; fop.init(x78)
        P = first_phase2_iter(ITERS)
        tr = torch.empty(ITERS, N * 660, device="cuda")
        for ii in range(ITERS):
            st = capi.current_stream()
            capi.check(lib.fdcap_opt_backward(h, ii, P, 0, st), "b")
            w = raw.fdcap_debug_rows(h, 2, ctypes.c_void_p(tr[ii].data_ptr()), st)
            capi.check(lib.fdcap_opt_step(h, ii, P, st), "s")
        stream.synchronize()
        out.append(tr.view(ITERS, N, 660).cpu().numpy())


def background(kind, stream, fop, x78, stop):
    """keeps stream 2 busy with ONE kind of work of the second context until `stop` is set"""
    with torch.cuda.stream(stream):
        lib, h = fop.ctx.lib, fop.ctx.handle
        ms = ctypes.c_float()
        nc = fop.ctx.num_contact
        verts = torch.empty(N, nc, 3, device="cuda")
        z = torch.randn(N, 32, device="cuda"); rot = torch.empty(N, 21, 9, device="cuda"); aa = torch.empty(N, 63, device="cuda")
        P = first_phase2_iter(ITERS)
        ii = 0
        while not stop.is_set():
            st = capi.current_stream()
            if kind == "nn":
                capi.check(lib.fdcap_opt_time_chamfer(h, 50, 0, ctypes.byref(ms), st), "nn")
            elif kind == "fwd":
                for _ in range(50): capi.check(lib.fdcap_opt_forward_world(h, capi.dptr(verts), None, st), "fw")
                stream.synchronize()
            elif kind == "blend":
                capi.check(lib.fdcap_time_blend_gemm(h, 1024, 20, ctypes.byref(ms), st), "blend")
            elif kind == "vposer":
                for _ in range(50): capi.check(lib.fdcap_vposer_decode(h, capi.dptr(z), 32, N, capi.dptr(rot), capi.dptr(aa), st), "vp")
                stream.synchronize()
            elif kind == "full":
                for _ in range(20):
                    capi.check(lib.fdcap_opt_backward(h, ii % P, P, 0, st), "b"); capi.check(lib.fdcap_opt_step(h, ii % P, P, st), "s"); ii += 1
                stream.synchronize()
            elif kind == "bwdonly":
                for _ in range(20): capi.check(lib.fdcap_opt_backward(h, 0, P, 0, st), "b")
                stream.synchronize()


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
f1, x1 = make(s1); f2, x2 = make(s2)
with torch.cuda.stream(s2):
    f2._mode = "global"; f2.init(x2)
    capi.check(f2.ctx.lib.fdcap_opt_backward(f2.ctx.handle, 0, 80, 0, capi.current_stream()), "b"); s2.synchronize()
if os.environ.get("MODE") == "canary":
    raw.fdcap_debug_lds_canary.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    for kind in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["none", "nn", "full", "fwd", "vposer", "blend"]):
        rep = torch.zeros(64, device="cuda", dtype=torch.int32)
        stop = threading.Event()
        bg = None
        if kind != "none":
            bg = threading.Thread(target=background, args=(kind, s2, f2, x2, stop)); bg.start()
        with torch.cuda.stream(s1):
            for _ in range(REPS):
                raw.fdcap_debug_lds_canary(ctypes.c_void_p(rep.data_ptr()), 2048, 6, 10, capi.current_stream())
                s1.synchronize()
        stop.set()
        if bg is not None: bg.join()
        r = rep.cpu().numpy().view(np.uint32)
        recs = [(int(r[4 * (k + 1)]), hex(int(r[4 * (k + 1) + 1])), int(r[4 * (k + 1) + 2]), int(r[4 * (k + 1) + 3])) for k in range(min(int(r[0]), 15))]
        print(f"canary next to [{kind}]: {int(r[0])} corrupted LDS words; first (word index, value, block, canary LDS bytes): {recs[:8]}", flush=True)
    sys.exit(0)
ref = []
one_fit(s1, f1, x1, ref)
for kind in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["full", "nn", "fwd", "vposer", "blend"]):
    bad = 0
    for k in range(REPS):
        o1 = []
        stop = threading.Event()
        bg = threading.Thread(target=background, args=(kind, s2, f2, x2, stop)); bg.start()
        one_fit(s1, f1, x1, o1)
        stop.set(); bg.join()
        if not np.array_equal(o1[0], ref[0]):
            bad += 1
            it, fr, col = np.nonzero(o1[0] != ref[0])
            first = it.min(); m = it == first
            if bad <= 3: print(f"  [{kind}] rep {k}: A first differs at iteration {first} frames {sorted(set(fr[m].tolist()))[:4]} cols {sorted(set(col[m].tolist()))[:10]}", flush=True)
    print(f"background [{kind}]: {bad} of {REPS} traced fits differ from a fit run alone", flush=True)
