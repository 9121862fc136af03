"""TEST INFRASTRUCTURE: drives tests/cpu_harness (the kernels' math headers compiled for the
host) through the same sequence fdcap_opt_backward enqueues on the GPU, with numpy standing in
for the MFMA GEMMs and the oracle's nn_direct for the Chamfer kernel.  Lets the hand-derived
backward be checked against the oracle's autograd without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpu_harness", "harness.cpp")
BUILD = os.path.join(ROOT, "tests", "_build")
LIB = os.path.join(BUILD, "libfdcap_host_harness.so")
CSRC = os.path.join(ROOT, "4dcapture-fpv_amd", "csrc")

NJ, NJW, XDIM, ODIM, NPF = 55, 23, 78, 126, 486
fp = ctypes.POINTER(ctypes.c_float)
ip = ctypes.POINTER(ctypes.c_int)
dp = ctypes.POINTER(ctypes.c_double)


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build():
    os.makedirs(BUILD, exist_ok=True)
    deps = [SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if any(_newer(d, LIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-o", LIB, SRC])
    return ctypes.CDLL(LIB)


def P(a, t=fp):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(t)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- the sharded iteration's one message (csrc/fdcap.hip adam_step_kernel / unpack_exchange_kernel) ---------------
XCH_ROW = XDIM + 16            # one boundary row: body_rotation_rec [78] | camera_ext [16]
XCH_LEN = 4 * XCH_ROW + 8      # [first two | last two owned rows] + this rank's d loss / d scale (+ padding to 32 B)


def xch_pack(rows_x, rows_cam, n_local, dscale):
    """Host mirror of the message fdcap_opt_step_rows_and_pack writes (rows_* include the 2 halo rows each side)."""
    msg = np.zeros(XCH_LEN, np.float32)
    for slot, row in enumerate((2, 3, n_local, n_local + 1)):
        msg[slot * XCH_ROW:slot * XCH_ROW + XDIM] = rows_x[row]
        msg[slot * XCH_ROW + XDIM:(slot + 1) * XCH_ROW] = rows_cam[row]
    msg[4 * XCH_ROW] = dscale
    return msg


def xch_unpack(gathered, rank, world, n_local, rows_x, rows_cam):
    """Host mirror of fdcap_opt_unpack_and_step_scale's data movement: halo rows <- the neighbours' boundary rows (clip ends
    keep what they hold); returns the scale gradient = the partials summed IN RANK ORDER in fp32 (same bits on every rank)."""
    g = np.asarray(gathered, dtype=np.float32).reshape(world, XCH_LEN)
    for k in range(4):
        src = rank - 1 if k < 2 else rank + 1
        if 0 <= src < world:
            slot = 2 + k if k < 2 else k - 2
            row = k if k < 2 else n_local + k
            rows_x[row] = g[src, slot * XCH_ROW:slot * XCH_ROW + XDIM]
            rows_cam[row] = g[src, slot * XCH_ROW + XDIM:(slot + 1) * XCH_ROW]
    s = np.float32(0)
    for r in range(world):
        s = np.float32(s + g[r, 4 * XCH_ROW])
    return float(s)


class HostPipeline:
    def __init__(self, bm, vp, scene, vid):
        self.lib = build()
        self.lib.h_pose_setup.restype = ctypes.c_void_p
        V = bm.num_verts
        self.V = V
        self.S10 = f32(bm.shapedirs[:, :, :10])
        hc = f32(np.stack([bm.hands_componentsl, bm.hands_componentsr]))
        hm = f32(np.concatenate([bm.hands_meanl, bm.hands_meanr]))
        self.h = ctypes.c_void_p(self.lib.h_pose_setup(
            V, P(f32(bm.v_template)), P(self.S10), P(f32(bm.J_regressor)),
            P(np.ascontiguousarray(bm.parents, dtype=np.int32), ip), P(hc), P(hm)))
        assert self.h.value
        self.vp = vp
        self.scene = f32(scene)
        self.scene4 = f32(np.concatenate([self.scene, (self.scene ** 2).sum(1, keepdims=True)], 1))
        vid = np.asarray(vid, dtype=np.int64)
        self.vid = vid
        self.nc = len(vid)
        lbs = bm.lbs_weights[vid]
        self.K = int(max(1, (lbs != 0).sum(1).max()))
        self.wj = np.zeros((self.nc, self.K), np.int32)
        self.ww = np.zeros((self.nc, self.K), np.float32)
        for i in range(self.nc):
            nz = np.nonzero(lbs[i])[0]
            self.wj[i, :len(nz)] = nz
            self.ww[i, :len(nz)] = lbs[i, nz]
        self.vt = f32(bm.v_template[vid])
        self.S = f32(self.S10[vid])
        cols = (3 * vid[:, None] + np.arange(3)[None]).reshape(-1)
        self.Pc = f32(bm.posedirs[:, cols])                     # [486, 3nc]

    # ---- VPoser MLP (numpy fp32) -----------------------------------------------------------
    def mlp_forward(self, z):
        vp = self.vp
        h1 = z @ vp.fc1_w.T + vp.fc1_b
        h1 = np.where(h1 > 0, h1, 0.2 * h1).astype(np.float32)
        h2 = h1 @ vp.fc2_w.T + vp.fc2_b
        h2 = np.where(h2 > 0, h2, 0.2 * h2).astype(np.float32)
        o = (h2 @ vp.out_w.T + vp.out_b).astype(np.float32)
        return h1, h2, o

    def mlp_backward(self, dO, h1, h2):
        vp = self.vp
        dh2 = (dO @ vp.out_w) * np.where(h2 > 0, 1.0, 0.2)
        dh1 = (dh2.astype(np.float32) @ vp.fc2_w) * np.where(h1 > 0, 1.0, 0.2)
        return (dh1.astype(np.float32) @ vp.fc1_w).astype(np.float32)

    def pose_forward(self, X, CAM, scale):
        n = X.shape[0]
        h1, h2, O = self.mlp_forward(X[:, 19:51])
        out = dict(h1=h1, h2=h2, O=f32(O))
        for k, w in (("Rm", NJ * 9), ("PF", NPF), ("Jrest", NJ * 3), ("G", NJ * 12), ("A", NJ * 12),
                     ("M", 12), ("Jw", NJW * 3)):
            out[k] = np.zeros((n, w), np.float32)
        self.lib.h_pose_forward(self.h, n, P(X), P(out["O"]), P(CAM), ctypes.c_float(scale), P(out["Rm"]),
                                P(out["PF"]), P(out["Jrest"]), P(out["G"]), P(out["A"]), P(out["M"]),
                                P(out["Jw"]))
        return out

    def contact_forward(self, X, fw, scale):
        n = X.shape[0]
        Voff = f32(fw["PF"] @ self.Pc)
        Vw = np.zeros((n, self.nc, 3), np.float32)
        self.lib.h_skin_forward(self.nc, self.K, P(self.vt), P(self.S), P(self.wj, ip), P(self.ww), n, P(X),
                                P(Voff), P(fw["A"]), P(fw["M"]), ctypes.c_float(scale), 1, P(Vw))
        return Voff, Vw

    def backward(self, X, X0, mask, CAM, scale, n_total, frame0, row0, n_own, phase2, cfg):
        """Mirror of fdcap_opt_backward.  X/X0/mask/CAM cover rows [0, rows); owned rows are
        [row0, row0+n_own).  Returns dict with dX, dCAM, dscale, losses."""
        from oracle.chamfer import nn_direct
        n = X.shape[0]
        X, X0, mask, CAM = f32(X), f32(X0), f32(mask), f32(CAM.reshape(n, 16))
        fw = self.pose_forward(X, CAM, scale)
        losses = np.zeros(8, np.float64)
        contact_on = self.scene.shape[0] > 0 and self.nc > 0 and cfg["weight_contact"] != 0
        contact_grad = contact_on and not phase2
        N = n_total
        w_rec = np.float32(cfg["weight_loss_rec"] / (np.float32(N) * XDIM))
        w_sm = np.float32((cfg["phase2_smooth"] if phase2 else cfg["phase1_smooth"]) / (np.float32(N - 2) * XDIM)) if N >= 3 else np.float32(0)
        w_ws = np.float32(cfg["phase2_world"] / (np.float32(N - 1) * NJW * 3)) if (phase2 and N >= 2) else np.float32(0)
        dX = np.zeros((n, XDIM), np.float32)
        dJw = np.zeros((n, NJW * 3), np.float32)
        self.lib.h_param_loss(P(X), P(X0), P(mask), P(fw["Jw"]), row0, n_own, frame0, N, ctypes.c_float(w_rec),
                              ctypes.c_float(w_sm), ctypes.c_float(w_ws), 1 if phase2 else 0, P(dX), P(dJw),
                              P(losses, dp))
        # everything below touches owned rows only (halo rows exist for the temporal stencils)
        o = slice(row0, row0 + n_own)
        Xo, CAMo = f32(X[o]), f32(CAM[o])
        fwo = {k: f32(v[o]) for k, v in fw.items()}
        dA = dPF = dMv = dsv = dbeta_v = dtransl_v = None
        if contact_on:
            Voff, Vw = self.contact_forward(Xo, fwo, scale)
            q = torch.from_numpy(Vw.reshape(-1, 3))
            d, idx = nn_direct(q, torch.from_numpy(self.scene))
            dist = f32(d.numpy())
            idx = np.ascontiguousarray(idx.numpy().astype(np.int32))
            fw["Vw"], fw["dist"], fw["idx"] = Vw, dist, idx
        if contact_grad:
            coef = np.float32(cfg["phase1_contact"] * cfg["weight_contact"] / (np.float32(N) * self.nc))
            dVoff = np.zeros((n_own, 3 * self.nc), np.float32)
            dA = np.zeros((n_own, NJ * 12), np.float32)
            dbeta_v = np.zeros((n_own, 10), np.float32)
            dtransl_v = np.zeros((n_own, 3), np.float32)
            dMv = np.zeros((n_own, 12), np.float32)
            dsv = np.zeros(n_own, np.float32)
            lc = np.zeros(1, np.float64)
            self.lib.h_skin_backward(self.nc, self.K, P(self.vt), P(self.S), P(self.wj, ip), P(self.ww), n_own, P(Xo),
                                     P(Voff), P(fwo["A"]), P(fwo["M"]), ctypes.c_float(scale), P(Vw), P(dist),
                                     P(idx, ip), P(self.scene4), ctypes.c_float(coef), P(dVoff), P(dA), P(dbeta_v),
                                     P(dtransl_v), P(dMv), P(dsv), P(lc, dp))
            losses[3] = lc[0]
            dPF = f32(dVoff @ self.Pc.T)
        dXo = f32(dX[o])
        dJwo = f32(dJw[o])
        dO = np.zeros((n_own, ODIM), np.float32)
        dCAM = np.zeros((n_own, 16), np.float32)
        dscale_row = np.zeros(n_own, np.float32)
        self.lib.h_pose_backward(self.h, n_own, P(Xo), P(fwo["O"]), P(CAMo), ctypes.c_float(scale), P(fwo["Rm"]),
                                 P(fwo["Jrest"]), P(fwo["G"]), P(dA), P(dPF), P(dJwo) if phase2 else None, P(dMv),
                                 P(dsv), P(dbeta_v), P(dtransl_v), P(dXo), P(dO), P(dCAM), P(dscale_row))
        dXo[:, 19:51] += self.mlp_backward(dO, fwo["h1"], fwo["h2"])
        return dict(dX=dXo, dCAM=dCAM.reshape(n_own, 4, 4), dscale=float(dscale_row.sum()), losses=losses, fw=fw)

    def adam(self, p, m, v, g, lr, step, zero_grad=False):
        self.lib.h_adam(P(p), P(m), P(v), P(g), ctypes.c_int64(p.size), ctypes.c_double(lr), step,
                        1 if zero_grad else 0)
