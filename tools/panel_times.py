#!/usr/bin/env python3
"""Per-workgroup phase times (s_memtime) of the panel kernels; needs the -DFDC_PN_TIMING build (FDCAP_LIB).
The stamps live in the exact-fp32 kernels (panel_gemm_kernel, panel_gemm_wide_kernel, vposer_*_fused_kernel), the FDCAP_GEMM_SPLIT3=0
twins of the default three-way-split ones: this tool selects them."""
import ctypes, os, sys
os.environ.setdefault("FDCAP_GEMM_SPLIT3", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fdcap_amd  # noqa
from fdcap_amd import capi, ops, synth
lib = capi.load_library()
raw = ctypes.CDLL(capi.LIB_PATH)
def stamps(nblk, nst):
    buf = (ctypes.c_ulonglong * (8192 * 8))()
    assert raw.fdcap_debug_panel_times(buf, 8192 * 8) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8)[:nblk, :nst].astype(np.int64)
    return a
def report(name, a):
    if len(a) == 0:
        print(f"{name}: no stamps (this shape runs a kernel without them)"); return
    t0 = a[:, 0].min()
    span = a[:, -1].max() - t0
    print(f"{name}: {len(a)} WGs, span {span} ticks; start skew q50/q90/max {np.quantile(a[:,0]-t0,0.5):.0f}/{np.quantile(a[:,0]-t0,0.9):.0f}/{(a[:,0]-t0).max()}")
    for i in range(a.shape[1] - 1):
        d = a[:, i + 1] - a[:, i]
        print(f"   phase {i}->{i+1}: q10 {np.quantile(d,0.1):.0f} q50 {np.quantile(d,0.5):.0f} q90 {np.quantile(d,0.9):.0f} max {d.max()}")
    print(f"   WG lifetime q50 {np.quantile(a[:,-1]-a[:,0],0.5):.0f} max {(a[:,-1]-a[:,0]).max()}")
    # the counter is per XCD: block b runs on XCD b % 8

rows = 1024
bm = synth.make_body_model(300, seed=0); vp = synth.make_vposer(seed=1)
ctx = capi.Context(bm, vp)
z = torch.randn(rows, 32, device="cuda"); v = ops.VPoser(ctx)
for _ in range(5): v.decode(z, "matrot")
torch.cuda.synchronize()
report("vposer_fwd (stage | L1 | L2 | L3)", stamps(256, 5))
raw.fdcap_debug_panel_reset()
rng = np.random.default_rng(0)
for (M, K, N) in ((rows, 496, 1500), (rows, 1500, 496), (rows, 496, 31425)):
    A = torch.randn(M, K, device="cuda"); B = rng.standard_normal((K, N)).astype(np.float32); C = torch.empty(M, N, device="cuda")
    for _ in range(3):
        capi.check(lib.fdcap_panel_gemm(capi.dptr(A), K, M, K, B.ctypes.data_as(ctypes.c_void_p), N, 1, N, capi.dptr(C), N, capi.current_stream()), "g")
    a = stamps(8192, 4)
    raw.fdcap_debug_panel_reset()
    report(f"panel_gemm {M}x{K}x{N} (stage | mma | store)", a[a[:, 3] != 0])
