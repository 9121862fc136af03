#!/usr/bin/env python3
"""Workgroup timeline of the in-loop NN launch (needs the -DFDC_NN_TIMELINE build via FDCAP_LIB): runs the bench fit, then one
timed launch, and prints how many workgroups are resident over time and per-XCD finish times."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fdcap_amd  # noqa
from fdcap_amd import capi, synth
from fdcap_amd.fitting import FittingOP
from fdcap_amd.io import read_camerapose
N, ns = int(os.environ.get("FRAMES", "1024")), 500000
bm = synth.make_body_model(10475, seed=0); vp = synth.make_vposer(seed=1); clip = synth.make_clip(N, seed=3)
scene = synth.make_scene(ns, seed=2); l, r = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
fop = FittingOP({"num_iter": int(os.environ.get("ITERS", "500"))}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.concatenate([l, r]),
                camera_ext=read_camerapose(clip.camerapose_lines))
fop.fitting(torch.tensor(clip.body_params).cuda(), "global")
lib = fop.ctx.lib
ms = ctypes.c_float()
if os.environ.get("TIMED", "1") == "1":       # TIMED=0: look at the last launch the loop itself issued
    capi.check(lib.fdcap_opt_time_chamfer(fop.ctx.handle, 1, 0, ctypes.byref(ms), capi.current_stream()), "time")
raw = ctypes.CDLL(capi.LIB_PATH)
CAP = 16384
nwg = (N * 500 + 31) // 32                    # one-wave workgroups (32 queries)
nb = min(CAP, (nwg + 7) // 8 * 8)
buf = (ctypes.c_ulonglong * (CAP * 8))()
assert raw.fdcap_debug_nn_timeline(buf, CAP * 8) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(CAP, 8)[:nb].astype(np.int64)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
st, en, xcc = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, a[:, 2]     # microseconds
print(f"launch {ms.value*1e3:.1f} us (HIP events); {len(a)} workgroups; span {en.max():.1f} us; lifetime q10/q50/q90/max {np.quantile(en-st,0.1):.1f}/{np.quantile(en-st,0.5):.1f}/{np.quantile(en-st,0.9):.1f}/{(en-st).max():.1f} us")
print("resident workgroups at t (us):", " ".join(f"{t}:{int(((st <= t) & (en > t)).sum())}" for t in range(0, int(en.max()) + 1, 5)))
for x in range(8):
    m = xcc == x
    if m.any(): print(f"  xcc {x}: {m.sum()} WGs, first start {st[m].min():.1f}, last start {st[m].max():.1f}, last end {en[m].max():.1f}, sum of lifetimes {(en[m]-st[m]).sum():.0f} us")

o = np.argsort(-(en - st))[:12]
idxs = np.nonzero(np.frombuffer(buf, dtype=np.uint64).reshape(CAP, 8)[:nb, 1] > 0)[0]
print("longest workgroups: (blockIdx, xcc, start, end, lifetime us)")
for k in o: print("  ", int(idxs[k]), int(xcc[k]), f"{st[k]:.1f} {en[k]:.1f} {en[k]-st[k]:.1f}")
late = np.argsort(-en)[:12]
print("last to finish:")
for k in late: print("  ", int(idxs[k]), int(xcc[k]), f"{st[k]:.1f} {en[k]:.1f} {en[k]-st[k]:.1f}")

# phases of a wave's life (us): set-up loads | list (kept-list check or build) | filter | main loop | merge + stores
ph = np.stack([a[:, 3] - a[:, 0], a[:, 4] - a[:, 3], a[:, 5] - a[:, 4], a[:, 6] - a[:, 5], a[:, 1] - a[:, 6]], axis=1) / 100.0
names = ["set-up", "list", "filter", "main loop", "tail"]
for label, sel in (("started in the first 30 us (full machine)", st < 30), ("started after 28 us (drain)", st > 28)):
    if sel.sum() == 0: continue
    print(f"{label}: {sel.sum()} waves, lifetime median {np.median((en - st)[sel]):.1f} us, work items median {np.median(a[sel, 7]):.0f}")
    for k, nm in enumerate(names):
        print(f"   {nm:10s} q10 {np.quantile(ph[sel, k], 0.1):5.2f}  q50 {np.quantile(ph[sel, k], 0.5):5.2f}  q90 {np.quantile(ph[sel, k], 0.9):5.2f}")
