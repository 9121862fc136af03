import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import fdcap_amd
from fdcap_amd import capi
lib = capi.load_library()
for (M, K, N) in ((16, 64, 16), (16, 80, 16), (33, 496, 1500), (130, 1500, 496), (1024, 512, 512)):
    rng = np.random.default_rng(1)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    Ad = torch.tensor(A).cuda(); Cd = torch.zeros(M, N, device="cuda")
    capi.check(lib.fdcap_panel_gemm(capi.dptr(Ad), K, M, K, B.ctypes.data_as(ctypes.c_void_p), N, 1, N, capi.dptr(Cd), N, capi.current_stream()), "g")
    C = Cd.cpu().numpy(); want = A.astype(np.float64) @ B.astype(np.float64)
    mag = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    err = np.abs(C - want) / mag
    print(M, K, N, "max err / sum|a||b| %.3e  rms %.3e" % (err.max(), np.sqrt((err ** 2).mean())))
