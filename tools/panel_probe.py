#!/usr/bin/env python3
"""Kernel-time probe for csrc/fdc_panel.h (run under rocprofv3 --kernel-trace --stats; FDCAP_LIB selects an ablation build):
fused VPoser forward at 1024 rows and the two blend products of the contact set."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fdcap_amd  # noqa: F401,E402
from fdcap_amd import capi, ops, synth  # noqa: E402

rows = int(os.environ.get("ROWS", "1024"))
bm = synth.make_body_model(300, seed=0)
vp = synth.make_vposer(seed=1)
ctx = capi.Context(bm, vp)
z = torch.randn(rows, 32, device="cuda")
v = ops.VPoser(ctx)
for _ in range(20):
    v.decode(z, "matrot")
lib = ctx.lib
rng = np.random.default_rng(0)
for (M, K, N) in ((rows, 496, 1500), (rows, 1500, 496)):
    A = torch.randn(M, K, device="cuda")
    B = rng.standard_normal((K, N)).astype(np.float32)
    C = torch.empty(M, N, device="cuda")
    for _ in range(5):
        capi.check(lib.fdcap_panel_gemm(capi.dptr(A), K, M, K, B.ctypes.data_as(ctypes.c_void_p), N, 1, N, capi.dptr(C), N,
                                        capi.current_stream()), "panel_gemm")
torch.cuda.synchronize()
