import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import fdcap_amd
from fdcap_amd import capi
lib = capi.load_library()
for (M, K, N) in ((16, 64, 16), (16, 80, 16), (16, 128, 16), (33, 496, 1500)):
    rng = np.random.default_rng(1)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    Ad = torch.tensor(A).cuda(); Cd = torch.zeros(M, N, device="cuda")
    capi.check(lib.fdcap_panel_gemm(capi.dptr(Ad), K, M, K, B.ctypes.data_as(ctypes.c_void_p), N, 1, N, capi.dptr(Cd), N, capi.current_stream()), "g")
    C = Cd.cpu().numpy(); want = A.astype(np.float64) @ B.astype(np.float64)
    err = np.abs(C - want)
    print(M, K, N, "max err", err.max(), "bad frac", (err > 1e-3).mean())
    if err.max() > 1e-3:
        # which k-blocks are wrong?  compare against partial sums dropping / duplicating 16-k super-steps
        nss = (K + 15) // 16
        r, c = np.unravel_index(err.argmax(), err.shape)
        parts = np.array([A[r, 16*s:16*s+16].astype(np.float64) @ B[16*s:16*s+16, c].astype(np.float64) for s in range(nss)])
        print("  at", r, c, "got", C[r, c], "want", want[r, c], "diff", C[r, c] - want[r, c])
        d = C[r, c] - want[r, c]
        for s in range(nss):
            for s2 in range(nss):
                if abs(d - (parts[s2] - parts[s])) < 1e-3: print("   == step", s, "replaced by", s2)
