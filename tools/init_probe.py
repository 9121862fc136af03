import os, sys, time, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import fdcap_amd
from fdcap_amd import capi, synth
from fdcap_amd.fitting import FittingOP, find_outliers
from fdcap_amd.io import read_camerapose
N = 1024
bm = synth.make_body_model(10475, seed=0); vp = synth.make_vposer(seed=1); clip = synth.make_clip(N, seed=3)
scene = synth.make_scene(500000, seed=2); l, r = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
fop = FittingOP({"num_iter": 500}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.concatenate([l, r]),
                camera_ext=read_camerapose(clip.camerapose_lines))
body = torch.tensor(clip.body_params).cuda()
fop.fitting(body, "global")
x78 = torch.empty(N, capi.XDIM, device="cuda")
lib, h = fop.ctx.lib, fop.ctx.handle
def T(label, f, n=5):
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print("%-50s %.3f ms (min of %d: %.3f)" % (label, sorted(ts)[len(ts)//2], n, min(ts)))
    return r
T("75->78 kernel", lambda: capi.check(lib.fdcap_params_75_to_78(capi.dptr(body), N, capi.dptr(x78), capi.current_stream()), "x"))
xh = T("x78.cpu().numpy()", lambda: x78.detach().cpu().numpy())
T("find_outliers (host)", lambda: find_outliers(xh))
T("7 x torch.zeros", lambda: [torch.zeros(N + 4, 78, device="cuda"), torch.zeros(N + 4, 16, device="cuda"), torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda"), torch.zeros(8, device="cuda", dtype=torch.float64), torch.zeros(400, device="cuda"), torch.zeros(1, 400, device="cuda")])
T("4 x H2D upload", lambda: [torch.from_numpy(xh).to("cuda"), torch.from_numpy(xh).to("cuda"), torch.from_numpy(np.ones(N, np.float32)).to("cuda"), torch.from_numpy(np.zeros((N, 16), np.float32)).to("cuda")])
fop._mode = "global"
T("fop.init (all of it)", lambda: fop.init(x78))
