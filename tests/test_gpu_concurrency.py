"""Two kernels of this library resident on the GPU at once must not change anybody's results.

Round 3 (NOTES.md section 6): built WITH packed fp32 instructions, work that ran next to the full-mesh blend product on a
second stream came back with a wrong word in a finger joint's transform now and then on the boxes of this pool (30 of 30
100-iteration fits had at least one; the parameters moved whenever the joint's own rotation was hit); built without them,
never.  This test is the tripwire for that build flag (__graft_entry__.build) and for anything else that makes co-resident
kernels interact: the body-model operator -- all 55 joints and the whole mesh, hands included -- and a fit, repeated next to the
library's heaviest matrix kernel on a second stream, against the same calls alone."""
import ctypes
import threading

import numpy as np
import pytest
import torch

import fdcap_amd  # noqa: F401
from fdcap_amd import capi, ops
from fdcap_amd.fitting import FittingOP
from fdcap_amd.io import read_camerapose
from tests.test_gpu_sharded import _inputs

pytestmark = pytest.mark.gpu
N, ITERS = 100, 40


def _make(stream):
    with torch.cuda.stream(stream):
        bm, vp, clip, scene, vid = _inputs(N)
        fop = FittingOP({"num_iter": ITERS}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid,
                        camera_ext=read_camerapose(clip.camerapose_lines))
        body = torch.tensor(clip.body_params).cuda()
        stream.synchronize()
    return fop, body


def _fit(stream, fop, body):
    with torch.cuda.stream(stream):
        rec, scale, cam = fop.fitting(body, "global", log_every=1)
        stream.synchronize()
        return rec.cpu().numpy(), float(scale), cam.cpu().numpy(), np.array(fop.log.total)


def test_work_next_to_matrix_kernels_on_a_second_stream_keeps_its_bits():
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    f1, b1 = _make(s1)
    f2, _ = _make(s2)
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
    ref_fit = _fit(s1, f1, b1)
    stop = threading.Event()

    def background():
        ms = ctypes.c_float()
        with torch.cuda.stream(s2):
            while not stop.is_set():                      # the full-mesh pose-blend product: the library's heaviest MFMA kernel
                capi.check(f2.ctx.lib.fdcap_time_blend_gemm(f2.ctx.handle, 1024, 10, ctypes.byref(ms), capi.current_stream()), "blend")

    bg = threading.Thread(target=background)
    bg.start()
    try:
        bad = 0
        with torch.cuda.stream(s1):
            for _ in range(300):
                out = model(return_verts=True, **kw)
                bad += int(not (torch.equal(out.vertices, ref_v) and torch.equal(out.joints, ref_j)))
            s1.synchronize()
        fits = [_fit(s1, f1, b1) for _ in range(3)]
    finally:
        stop.set()
        bg.join()
    assert bad == 0, f"{bad} of 300 body-model forwards next to the blend product differ from the forward that ran alone"
    for k, r in enumerate(fits):
        for a, b, what in zip(ref_fit, r, ("parameters", "scale", "camera_ext", "logged totals")):
            assert np.array_equal(np.asarray(a), np.asarray(b)), f"fit {k}: {what} differ from the fit that ran alone"
    f1.close(); f2.close()
