"""The set-up entry points run on the device since r6 (their host loops cost more than the fits they served).

fdcap_set_scene builds the search's scene tables on the device (csrc/fdc_scene.h: three radix-sorted index lists, one stable
partition per k-d level, boxes / fragments per cell).  The order is a specification (longest axis, (coordinate, index) rank,
512 / 32-point units, input order inside a tile) that the host recursion of r1-r5 also follows, so every table must come out
the same bit for bit from both -- ragged sizes, duplicate points, coordinate ties that straddle a cut, signed zeros, and the
BASELINE scenes.  What the search RETURNS never depends on the order at all (tests/test_gpu_parity.py, test_gpu_fullsize.py);
this file pins the order itself, and that an arbitrary scene's neighbours match the oracle's scan after the device build.

fdcap_ctx_create / fdcap_set_contact_ids pack the static operands of the dense products (fp32 fragment order, two scaled fp16
planes + column scales) with device kernels; FDCAP_PANEL_PACK=host keeps the host loops of csrc/fdc_panel.h as the specification:
whole fits on the two must agree bit for bit -- contact sets that take the fused contact forward, the generic path, and the
full mesh (K-loop data gradient, wide forward)."""
import ctypes
import os

import numpy as np
import pytest
import torch

import fdcap_amd  # noqa: F401
from fdcap_amd import capi, synth
from oracle.chamfer import nn_direct

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(synth.make_body_model(400, seed=0), synth.make_vposer(seed=1))
    yield c
    c.close()


def _hashes(ctx, scene, how):
    old = os.environ.get("FDCAP_SCENE_BUILD")
    try:
        if how == "host":
            os.environ["FDCAP_SCENE_BUILD"] = "host"
        else:
            os.environ.pop("FDCAP_SCENE_BUILD", None)
        ctx.set_scene(scene)
    finally:
        if old is None:
            os.environ.pop("FDCAP_SCENE_BUILD", None)
        else:
            os.environ["FDCAP_SCENE_BUILD"] = old
    out = (ctypes.c_uint64 * 8)()
    capi.check(ctx.lib.fdcap_debug_scene_hash(ctx.handle, out), "fdcap_debug_scene_hash")
    return list(out)


def _scenes():
    rng = np.random.default_rng(11)
    for n in (1, 31, 32, 33, 64, 65, 511, 512, 513, 1000, 1025, 4097, 16385, 70001):
        yield f"uniform{n}", rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    # ties: coordinates on a coarse grid (hundreds of equal coordinates across every cut), duplicates of whole points
    g = (rng.integers(0, 12, (20000, 3)) * 0.25 - 1.5).astype(np.float32)
    yield "grid_ties", g
    yield "duplicates", np.repeat(rng.uniform(-1, 1, (700, 3)).astype(np.float32), 9, axis=0)
    z = rng.uniform(-1, 1, (5000, 3)).astype(np.float32)
    z[::3, 0] = 0.0
    z[1::3, 0] = -0.0                                       # -0 == +0 for the order; the point keeps its bits
    z[:, 2] = 0.5                                           # a flat axis (extent 0: never the longest)
    yield "signed_zeros_flat", z
    yield "floor_like", synth.make_scene(30000, seed=5)


@pytest.mark.parametrize("name,scene", list(_scenes()), ids=[n for n, _ in _scenes()])
def test_device_build_gives_the_host_recursions_tables(ctx, name, scene):
    dev = _hashes(ctx, scene, "device")
    host = _hashes(ctx, scene, "host")
    names = ["scene", "sorted", "inv", "bounds", "qbounds", "sbounds", "frags", "centers"]
    assert dev == host, [n for n, a, b in zip(names, dev, host) if a != b]
    assert _hashes(ctx, scene, "device") == dev             # and run to run


@pytest.mark.parametrize("ns", [100_000, 500_000, 2_000_000])
def test_baseline_scenes_build_identically_and_fast(ctx, ns):
    import time
    scene = synth.make_scene(ns, seed=2)
    dev = _hashes(ctx, scene, "device")
    assert dev == _hashes(ctx, scene, "host")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.set_scene(scene)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"fdcap_set_scene({ns}) {dt * 1e3:.1f} ms")
    assert dt < (0.08 if ns >= 2_000_000 else 0.02) * 3, dt   # <= 20 ms at 500 k, <= 80 ms at 2 M (x3: shared hosts)


def test_search_on_a_device_built_scene_matches_the_oracles_scan(ctx):
    rng = np.random.default_rng(3)
    scene = np.concatenate([synth.make_scene(40000, seed=9), rng.uniform(-2, 2, (3000, 3)).astype(np.float32)])
    scene = scene[rng.permutation(len(scene))]
    ctx.set_scene(scene)
    q = torch.tensor(rng.uniform(-1.5, 1.5, (4, 300, 3)).astype(np.float32)).cuda()
    d = torch.empty(4, 300, device="cuda")
    i = torch.empty(4, 300, device="cuda", dtype=torch.int32)
    for forget in (1, 0):                                   # seeded by nn_seed_kernel, then by its own neighbours
        capi.check(ctx.lib.fdcap_chamfer_fwd_scene(ctx.handle, capi.dptr(q), 4, 300, capi.dptr(d), capi.dptr(i), forget,
                                                   capi.current_stream()), "fdcap_chamfer_fwd_scene")
        torch.cuda.synchronize()
        od, oi = nn_direct(q.cpu().reshape(-1, 3), torch.tensor(scene))
        np.testing.assert_allclose(d.cpu().numpy().ravel(), od.numpy(), rtol=2e-6, atol=1e-12)
        same = i.cpu().numpy().ravel() == oi.numpy()
        assert same.mean() > 0.999                          # (indices: equal up to rounding ties between the two distance forms)


def _fit(pack, n, V, per_part, all_contacts, ns=20000, iters=8):
    from fdcap_amd.fitting import FittingOP
    from fdcap_amd.io import read_camerapose
    old = os.environ.get("FDCAP_PANEL_PACK")
    try:
        if pack == "host":
            os.environ["FDCAP_PANEL_PACK"] = "host"
        else:
            os.environ.pop("FDCAP_PANEL_PACK", None)
        bm = synth.make_body_model(V, seed=0)
        vp = synth.make_vposer(seed=1)
        clip = synth.make_clip(n, seed=3)
        scene = synth.make_scene(ns, seed=2)
        left, right = synth.make_contact_ids(bm.v_template, per_part=per_part, seed=4)
        vid = np.arange(V) if all_contacts else np.concatenate([left, right])
        fop = FittingOP({"num_iter": iters}, {}, n, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid,
                        camera_ext=read_camerapose(clip.camerapose_lines))
        body, scale, cam = fop.fitting(torch.tensor(clip.body_params).cuda(), "global", log_every=1)
        out = body.cpu().numpy().copy(), float(scale), cam.cpu().numpy().copy(), np.array(fop.log.total)
        fop.close()
        return out
    finally:
        if old is None:
            os.environ.pop("FDCAP_PANEL_PACK", None)
        else:
            os.environ["FDCAP_PANEL_PACK"] = old


@pytest.mark.parametrize("n,V,per_part,all_contacts", [(40, 700, 30, False), (400, 10475, 250, False), (12, 10475, 0, True), (20, 1500, 0, True)],
                         ids=["small", "config3_like_fused_forward", "full_mesh_contacts", "mid_mesh_contacts"])
def test_device_packed_operands_give_the_host_packed_fit_bit_for_bit(n, V, per_part, all_contacts):
    a = _fit("device", n, V, per_part, all_contacts)
    b = _fit("host", n, V, per_part, all_contacts)
    assert np.isfinite(a[0]).all()
    np.testing.assert_array_equal(a[0], b[0])
    assert a[1] == b[1]
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(a[3], b[3])


def test_launch_timing_books_every_launch_of_every_iteration():
    """fdcap_opt_launch_timing (bench.py's roofline.per_kernel[].us_live): one interval per launch and iteration, booked on the right
    stage and phase -- a 20-iteration fit with the phase switch at 16 has 16 samples of all eight stages in phase 1 and 4 samples of the
    four pose / VPoser stages in phase 2 -- and leaves the fit's results untouched."""
    from fdcap_amd.fitting import FittingOP
    from fdcap_amd.io import read_camerapose
    n, iters = 48, 20
    bm = synth.make_body_model(600, seed=0)
    vp = synth.make_vposer(seed=1)
    clip = synth.make_clip(n, seed=3)
    scene = synth.make_scene(8000, seed=2)
    left, right = synth.make_contact_ids(bm.v_template, per_part=20, seed=4)
    fop = FittingOP({"num_iter": iters}, {}, n, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.concatenate([left, right]),
                    camera_ext=read_camerapose(clip.camerapose_lines))
    body = torch.tensor(clip.body_params).cuda()
    ref = fop.fitting(body, "global")[0].cpu().numpy().copy()
    lib, h = fop.ctx.lib, fop.ctx.handle
    capi.check(lib.fdcap_opt_launch_timing(h, 10 * iters + 16), "fdcap_opt_launch_timing")
    out = fop.fitting(body, "global")[0].cpu().numpy()
    us = (ctypes.c_float * 16)()
    cnt = (ctypes.c_int32 * 16)()
    capi.check(lib.fdcap_opt_launch_timing_read(h, us, cnt), "fdcap_opt_launch_timing_read")
    capi.check(lib.fdcap_opt_launch_timing(h, 0), "fdcap_opt_launch_timing")
    np.testing.assert_array_equal(out, ref)
    assert list(cnt[:8]) == [16] * 8, list(cnt)
    assert list(cnt[8:]) == [4, 4, 0, 0, 0, 0, 4, 4], list(cnt)
    live = [us[i] for i in range(16) if cnt[i]]
    assert all(0.5 < v < 2000.0 for v in live), live
    fop.close()
