"""Measured error of the split product forms against fp64, as a fraction of sum |a||b| (per output element, worst and rms):
well-scaled normal data and badly scaled rows / columns.  FDCAP_LIB selects the build (PnH2 default, -DFDC_PN_H2=0: three bf16 planes)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdcap_amd  # noqa: F401,E402
from fdcap_amd import capi  # noqa: E402


def run(A, Bm):
    lib = capi.load_library()
    M, K = A.shape
    N = Bm.shape[1]
    Ad = torch.tensor(A).cuda()
    Cd = torch.zeros((M, N), device="cuda")
    capi.check(lib.fdcap_panel_gemm(capi.dptr(Ad), K, M, K, Bm.ctypes.data_as(ctypes.c_void_p), N, 1, N, capi.dptr(Cd), N,
                                    capi.current_stream()), "fdcap_panel_gemm")
    want = A.astype(np.float64) @ Bm.astype(np.float64)
    den = np.abs(A).astype(np.float64) @ np.abs(Bm).astype(np.float64) + 1e-300
    r = np.abs(Cd.cpu().numpy() - want) / den
    return r.max(), np.sqrt((r ** 2).mean())


def main():
    rng = np.random.default_rng(5)
    cases = {}
    M, K, N = 1024, 496, 1500
    cases["normal 1024x496x1500"] = (rng.standard_normal((M, K)).astype(np.float32), rng.standard_normal((K, N)).astype(np.float32))
    A = (rng.standard_normal((M, K)) * 10.0 ** rng.uniform(-5, 0, (M, K)) * 10.0 ** rng.uniform(-12, 6, (M, 1))).astype(np.float32)
    B = (rng.standard_normal((K, N)) * 10.0 ** rng.uniform(-5, 0, (K, N)) * 10.0 ** rng.uniform(-8, 4, (1, N))).astype(np.float32)
    cases["rows 1e-12..1e6, columns 1e-8..1e4, elements over 5 decades"] = (A, B)
    A = rng.standard_normal((130, 1500)).astype(np.float32) * 1e-9
    A[:, 7] = 3.0e4                                                    # one huge element per row: everything else 13 decades below it
    cases["one element 13 decades above its row (130x1500x496)"] = (A, rng.standard_normal((1500, 496)).astype(np.float32))
    for form in ("1", "0"):
        os.environ["FDCAP_GEMM_SPLIT3"] = form
        for name, (A, B) in cases.items():
            mx, rms = run(A, B)
            print("%-10s %-66s max %.2e  rms %.2e  (of sum |a||b|)" % ("split" if form == "1" else "fp32 mfma", name, mx, rms))


if __name__ == "__main__":
    main()
