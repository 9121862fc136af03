"""GPU tests of the fragment-ordered ("panel") fp32 MFMA kernels of the loop (csrc/fdc_panel.h): the generic skinny-M product,
the fused VPoser decoder forward and its fused data-gradient chain -- through the C-ABI, against fp64 numpy / the oracle."""
import ctypes

import numpy as np
import pytest
import torch

import fdcap_amd  # noqa: F401
from fdcap_amd import capi, ops, synth
from fdcap_amd.fitting import find_outliers
from oracle import rotrepr
from oracle.fitting import FittingOracle
from oracle.smplx import SMPLXOracle
from oracle.vposer import VPoserDecoder
from tests.test_gpu_parity import _make_fop

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,N,transposed,pad", [
    (1, 1, 1, False, 0), (5, 3, 7, False, 2), (16, 16, 16, True, 0), (17, 33, 31, True, 3),
    (33, 496, 1500, False, 0),          # the loop's pose/shape blend product (contact set)
    (130, 1500, 496, True, 0),          # its data gradient (transposed operand), shard-sized M
    (70, 2100, 40, False, 5),           # K beyond one LDS slab
    (1028, 32, 512, True, 0), (257, 126, 512, False, 2),
    (130, 496, 13001, False, 0),        # wide output: four row blocks per fragment stream, column-block-major workgroup order
    (1024, 496, 1500, False, 0),        # clip-sized M: two row blocks per fragment stream (panel_gemm3_rb2_kernel), the bench's product
    (531, 700, 1000, True, 3),          # ... ragged rows / tiles, unaligned operand and output rows
    (130, 1690, 496, True, 0),          # longest K whose three-plane LDS image fits a CU's 160 KB (kpad 1696)
    (130, 2000, 496, True, 0),          # (beyond the three-plane image; the two-plane fp16 image reaches kpad 2528)
    (130, 2600, 496, True, 0),          # beyond it: the split form must hand over to the K-slabbed fp32 kernel, not fail the launch
    (400, 1700, 496, True, 0),          # clip-sized M above the K-split kernel's range (K <= 1536)
])
@pytest.mark.parametrize("form", ["split3", "fp32"])
def test_panel_gemm_matches_fp64(M, K, N, transposed, pad, form, monkeypatch):
    """Both forms of the loop's dense product against fp64: the three-way bf16 split on the bf16 matrix cores (default) and the
    exact fp32 MFMA chain (FDCAP_GEMM_SPLIT3=0) are held to the SAME bar."""
    monkeypatch.setenv("FDCAP_GEMM_SPLIT3", "1" if form == "split3" else "0")
    lib = capi.load_library()
    rng = np.random.default_rng(M * 7919 + K * 31 + N)
    A = rng.standard_normal((M, K + pad)).astype(np.float32)
    Bm = rng.standard_normal((K, N)).astype(np.float32)                 # the mathematical operand
    if transposed:
        store, sk, sn = np.ascontiguousarray(Bm.T), 1, K                # stored [N, K]
    else:
        store, sk, sn = Bm, N, 1
    Ad = torch.tensor(A).cuda()
    Cd = torch.full((M, N + pad), -7.0, device="cuda")
    capi.check(lib.fdcap_panel_gemm(capi.dptr(Ad), K + pad, M, K, store.ctypes.data_as(ctypes.c_void_p), sk, sn, N,
                                    capi.dptr(Cd), N + pad, capi.current_stream()), "fdcap_panel_gemm")
    C = Cd.cpu().numpy()
    want = A[:, :K].astype(np.float64) @ Bm.astype(np.float64)
    # fp32 accumulation of K products: the error random-walks at ~6e-8 of the partial sums' magnitude (measured max 3e-7 of
    # sum |a||b| for either form); bar = 1e-6 of sum |a||b|
    tol = 1e-6 * (np.abs(A[:, :K]).astype(np.float64) @ np.abs(Bm).astype(np.float64)) + 1e-6
    assert (np.abs(C[:, :N] - want) <= tol).all()
    if pad:
        assert (C[:, N:] == -7.0).all()                                  # nothing written past the N columns


@pytest.mark.parametrize("case", ["rows_and_columns_over_18_decades", "one_element_13_decades_above_its_row", "zero_rows_and_columns"])
def test_split_product_on_badly_scaled_operands(case, monkeypatch):
    """The two-plane fp16 form (csrc/fdc_panel.h, PnH2) scales every frame row of the dynamic operand (in the kernel) and every output
    column of the static one (host) by a power of two before the split; fp16's narrow exponent must not show in the result: rows from
    1e-12 to 1e6, columns from 1e-8 to 1e4, elements spread over five decades inside them -- same bar as the well-scaled product
    (measured 5.5e-7 of sum |a||b|; the exact fp32 MFMA chain: 1.4e-6)."""
    monkeypatch.setenv("FDCAP_GEMM_SPLIT3", "1")
    lib = capi.load_library()
    rng = np.random.default_rng(5)
    if case == "rows_and_columns_over_18_decades":
        M, K, N = 1024, 496, 1500
        A = (rng.standard_normal((M, K)) * 10.0 ** rng.uniform(-5, 0, (M, K)) * 10.0 ** rng.uniform(-12, 6, (M, 1))).astype(np.float32)
        Bm = (rng.standard_normal((K, N)) * 10.0 ** rng.uniform(-5, 0, (K, N)) * 10.0 ** rng.uniform(-8, 4, (1, N))).astype(np.float32)
    elif case == "one_element_13_decades_above_its_row":
        M, K, N = 130, 1500, 496
        A = rng.standard_normal((M, K)).astype(np.float32) * np.float32(1e-9)
        A[:, 7] = 3.0e4
        Bm = rng.standard_normal((K, N)).astype(np.float32)
    else:
        M, K, N = 70, 496, 100
        A = rng.standard_normal((M, K)).astype(np.float32)
        Bm = rng.standard_normal((K, N)).astype(np.float32)
        A[3] = 0.0; A[40:60] = 0.0; Bm[:, 17] = 0.0; Bm[:, 32:48] = 0.0
    Ad = torch.tensor(A).cuda()
    Cd = torch.full((M, N), -7.0, device="cuda")
    capi.check(lib.fdcap_panel_gemm(capi.dptr(Ad), K, M, K, Bm.ctypes.data_as(ctypes.c_void_p), N, 1, N, capi.dptr(Cd), N,
                                    capi.current_stream()), "fdcap_panel_gemm")
    C = Cd.cpu().numpy()
    want = A.astype(np.float64) @ Bm.astype(np.float64)
    den = np.abs(A).astype(np.float64) @ np.abs(Bm).astype(np.float64)
    assert np.isfinite(C).all()
    assert (np.abs(C - want) <= 1e-6 * den).all(), float((np.abs(C - want) / (den + 1e-300)).max())
    if case == "zero_rows_and_columns":
        assert (C[3] == 0).all() and (C[40:60] == 0).all() and (C[:, 17] == 0).all() and (C[:, 32:48] == 0).all()


@pytest.mark.parametrize("M,K,N", [(70, 496, 100), (1024, 496, 1500), (130, 1500, 496)])
def test_split_product_propagates_nan_and_inf(M, K, N, monkeypatch):
    """A non-finite entry of the dynamic operand must reach every output of its row (FittingOP's check_finite relies on it) and no
    other row: the row scales of the fp16-plane form come from the rows' largest FINITE magnitudes."""
    monkeypatch.setenv("FDCAP_GEMM_SPLIT3", "1")
    lib = capi.load_library()
    rng = np.random.default_rng(9)
    A = rng.standard_normal((M, K)).astype(np.float32)
    Bm = rng.standard_normal((K, N)).astype(np.float32)
    A[3, 17] = np.nan
    A[40, K - 1] = np.inf
    Ad = torch.tensor(A).cuda()
    Cd = torch.zeros((M, N), device="cuda")
    capi.check(lib.fdcap_panel_gemm(capi.dptr(Ad), K, M, K, Bm.ctypes.data_as(ctypes.c_void_p), N, 1, N, capi.dptr(Cd), N,
                                    capi.current_stream()), "fdcap_panel_gemm")
    C = Cd.cpu().numpy()
    assert not np.isfinite(C[3]).any() and not np.isfinite(C[40]).any()
    ok = np.ones(M, bool); ok[[3, 40]] = False
    want = A[ok].astype(np.float64) @ Bm.astype(np.float64)
    den = np.abs(A[ok]).astype(np.float64) @ np.abs(Bm).astype(np.float64)
    assert (np.abs(C[ok] - want) <= 1e-6 * den).all()


@pytest.mark.parametrize("B", [1, 15, 16, 17, 63, 1028])
def test_fused_vposer_forward_ragged_rows(B):
    vp = synth.make_vposer(seed=11)
    bm = synth.make_body_model(300, seed=0)
    ctx = capi.Context(bm, vp)
    rng = np.random.default_rng(B)
    z = rng.standard_normal((B, 32)).astype(np.float32)
    want = VPoserDecoder.from_data(vp).decode(torch.tensor(z), output_type="matrot").numpy()
    got = ops.VPoser(ctx).decode(torch.tensor(z).cuda(), "matrot").cpu().numpy()
    np.testing.assert_allclose(got, want, atol=4e-5)
    # rows are independent: the same latent decoded alone gives the same bits as inside a batch
    if B > 1:
        one = ops.VPoser(ctx).decode(torch.tensor(z[B // 2:B // 2 + 1]).cuda(), "matrot").cpu().numpy()
        assert (one == got[B // 2:B // 2 + 1]).all()
    ctx.close()


@pytest.mark.parametrize("n,V,per_part", [(37, 300, 20), (70, 300, 20), (21, 2000, 90), (530, 300, 20), (523, 2000, 90),
                                          (40, 2000, 300), (390, 2000, 283)])
def test_fused_vposer_backward_in_the_optimiser_gradient(n, V, per_part):
    """Phase-1 gradient of a clip that spans several 16-row blocks (ragged last block): the latent columns come out of
    vposer_bwd_fused_kernel's four partial sums, folded by fdcap_opt_get_grads; compared with fp64 autograd.
    The third case has a 180-vertex contact set on a 2000-vertex body (blend products with K, N = 540); the last two are
    clip-sized (>= 512 frames): the data-gradient product runs as two K halves on two row blocks per fragment stream
    (panel_gemm3_rb2k_kernel: 2 + 2 and 9 + 8 steps), the partial products added by pose_bwd_kernel.  The 600- / 566-vertex
    contact sets (K = 1800 / 1698) are past the single-image three-plane form (kpad <= 1696: 160 KB of LDS) and past the
    K-split kernel: their data gradient runs on the K-slabbed fp32 panel kernel."""
    fop, bm, vp, clip, scene, vid = _make_fop(n, V, 800, per_part, 500)
    dt = torch.float64
    f = FittingOracle(SMPLXOracle(bm, dt), VPoserDecoder.from_data(vp, dt), scene, vid, clip.camerapose_lines, n, dtype=dt)
    x78 = rotrepr.convert_to_6D_rot(torch.tensor(clip.body_params, dtype=dt)).detach()
    f.init(x78)
    g = torch.Generator().manual_seed(5)
    pert = 0.01 * torch.randn(f.body_rotation_rec.shape, generator=g, dtype=dt)
    f.body_rotation_rec.data += pert
    idx1, _ = find_outliers(x78.numpy().astype(np.float32))
    l_rec, l_vp, l_con, l_sm, l_ws = f.cal_loss(x78, idx1)
    (0.1 * l_con + l_sm + l_rec).backward()
    fop.init(torch.tensor(x78.numpy(), dtype=torch.float32).cuda())
    fop._rows_x[2:2 + n] += pert.float().cuda()
    lib, h = fop.ctx.lib, fop.ctx.handle
    capi.check(lib.fdcap_opt_backward(h, 5, 10 ** 6, 0, capi.current_stream()), "backward")
    dx = torch.empty(n, 78, device="cuda")
    dx2 = torch.empty(n, 78, device="cuda")
    capi.check(lib.fdcap_opt_get_grads(h, capi.dptr(dx), None, capi.current_stream()), "grads")
    capi.check(lib.fdcap_opt_get_grads(h, capi.dptr(dx2), None, capi.current_stream()), "grads")   # folding happens once
    torch.cuda.synchronize()
    gx = f.body_rotation_rec.grad.numpy()
    np.testing.assert_allclose(dx.cpu().numpy(), gx, rtol=2e-3, atol=2e-4 * np.abs(gx).max())
    assert torch.equal(dx, dx2)
    lat = gx[:, 19:51]
    assert np.abs(lat).max() > 0                                          # the latent block is really exercised
    np.testing.assert_allclose(dx.cpu().numpy()[:, 19:51], lat, rtol=2e-3, atol=2e-4 * np.abs(lat).max())
    fop.close()
