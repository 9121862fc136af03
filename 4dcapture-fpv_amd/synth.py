"""Seeded synthetic stand-ins for the licensed assets the reference needs.

The reference (global_optimization.py:153-176, :669-676) loads SMPLX_NEUTRAL.npz, a VPoser v1
snapshot, body_segments/{L_Leg,R_Leg}.json, a COLMAP meshed-poisson.ply and camerapose.txt.
None of them ship with the reference or this image, so every test / bench input is generated
here with the real shapes (V=10475, J=55, 486 pose-blend rows, 32-d latent, ...).
Spec: SURVEY.md §8d.  Everything is numpy (PCG64) so the same arrays feed the HIP path, the
oracle and the golden-vector generator.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

NUM_JOINTS = 55
NUM_VERTS_SMPLX = 10475
NUM_BETAS = 10
NUM_POSE_BASIS = (NUM_JOINTS - 1) * 9  # 486
NUM_HAND_PCA = 12

# Public SMPL-X kinematic tree.
SMPLX_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
     20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
     21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53], dtype=np.int32)


def _rest_skeleton() -> np.ndarray:
    """Approximate T-pose joint positions (metres, y up, pelvis at origin)."""
    J = np.zeros((NUM_JOINTS, 3), dtype=np.float64)
    J[0] = (0.0, 0.0, 0.0)
    J[1] = (0.09, -0.08, 0.0); J[2] = (-0.09, -0.08, 0.0)
    J[3] = (0.0, 0.12, -0.01)
    J[4] = (0.10, -0.48, 0.0); J[5] = (-0.10, -0.48, 0.0)
    J[6] = (0.0, 0.26, 0.0)
    J[7] = (0.10, -0.88, -0.02); J[8] = (-0.10, -0.88, -0.02)
    J[9] = (0.0, 0.33, 0.01)
    J[10] = (0.11, -0.94, 0.10); J[11] = (-0.11, -0.94, 0.10)
    J[12] = (0.0, 0.52, -0.01)
    J[13] = (0.07, 0.44, 0.0); J[14] = (-0.07, 0.44, 0.0)
    J[15] = (0.0, 0.62, 0.02)
    J[16] = (0.18, 0.46, 0.0); J[17] = (-0.18, 0.46, 0.0)
    J[18] = (0.44, 0.46, 0.0); J[19] = (-0.44, 0.46, 0.0)
    J[20] = (0.69, 0.46, 0.0); J[21] = (-0.69, 0.46, 0.0)
    J[22] = (0.0, 0.60, 0.06)
    J[23] = (0.03, 0.68, 0.08); J[24] = (-0.03, 0.68, 0.08)
    # fingers: 5 chains of 3 per hand, fanned out from the wrist
    for side, wrist, base in ((1.0, 20, 25), (-1.0, 21, 40)):
        for f in range(5):
            spread = (f - 2) * 0.018
            for k in range(3):
                J[base + 3 * f + k] = J[wrist] + np.array(
                    [side * (0.09 + 0.028 * k), -0.005 * k, spread])
    return J


@dataclass
class BodyModelData:
    """Arrays with the SMPL-X npz key names the reference's smplx.create() reads (A.3)."""
    v_template: np.ndarray      # [V,3]
    shapedirs: np.ndarray       # [V,3,20]  (10 shape + 10 expression)
    posedirs: np.ndarray        # [486, V*3]  (library layout: reshape(-1,486).T)
    J_regressor: np.ndarray     # [55,V]
    parents: np.ndarray         # [55] int32, root = -1
    lbs_weights: np.ndarray     # [V,55]
    hands_componentsl: np.ndarray  # [12,45]
    hands_componentsr: np.ndarray  # [12,45]
    hands_meanl: np.ndarray     # [45]
    hands_meanr: np.ndarray     # [45]

    @property
    def num_verts(self) -> int:
        return int(self.v_template.shape[0])


def make_body_model(num_verts: int = NUM_VERTS_SMPLX, seed: int = 0, lbs_nnz: int = 4) -> BodyModelData:
    """`lbs_nnz`: non-zero skinning weights per vertex (nearest joints).  4 is the SURVEY §8d spec; the real SMPLX_NEUTRAL.npz
    is not promised to be 4-sparse (global_optimization.py:154-168 loads whatever the file holds), so 8 / 12 exercise the
    general-K skinning paths.  Every other array is identical for every lbs_nnz."""
    rng = np.random.Generator(np.random.PCG64(seed))
    J = _rest_skeleton()
    parents = SMPLX_PARENTS
    # vertices: points on capsules around each bone (parent -> child), radius by body part
    bones = [(int(parents[j]), j) for j in range(1, NUM_JOINTS)]
    lens = np.array([np.linalg.norm(J[c] - J[p]) + 0.03 for p, c in bones])
    radius = np.full(len(bones), 0.05)
    for bi, (p, c) in enumerate(bones):
        if c >= 25:
            radius[bi] = 0.008
        elif c in (3, 6, 9):
            radius[bi] = 0.13
        elif c in (1, 2, 4, 5):
            radius[bi] = 0.075
        elif c in (15, 22, 23, 24):
            radius[bi] = 0.09
    prob = lens * radius
    prob = prob / prob.sum()
    which = rng.choice(len(bones), size=num_verts, p=prob)
    t = rng.random(num_verts)
    ang = rng.random(num_verts) * 2.0 * math.pi
    v = np.zeros((num_verts, 3))
    for bi, (p, c) in enumerate(bones):
        sel = np.nonzero(which == bi)[0]
        if sel.size == 0:
            continue
        a, b = J[p], J[c]
        d = b - a
        n = np.linalg.norm(d)
        d = d / n if n > 1e-9 else np.array([0.0, 1.0, 0.0])
        e1 = np.cross(d, [0.0, 0.0, 1.0])
        if np.linalg.norm(e1) < 1e-6:
            e1 = np.cross(d, [1.0, 0.0, 0.0])
        e1 /= np.linalg.norm(e1)
        e2 = np.cross(d, e1)
        tt = (t[sel] * 1.2 - 0.1)[:, None]
        v[sel] = a + tt * (b - a) + radius[bi] * (np.cos(ang[sel])[:, None] * e1 +
                                                  np.sin(ang[sel])[:, None] * e2)
    # guarantee each joint has vertices close by (regressor rows stay local)
    v_template = v.astype(np.float32)

    shapedirs = (rng.standard_normal((num_verts, 3, 20)) * 0.01).astype(np.float32)
    posedirs = (rng.standard_normal((NUM_POSE_BASIS, num_verts * 3)) * 0.003).astype(np.float32)

    d2 = ((v[None, :, :] - J[:, None, :]) ** 2).sum(-1)         # [55,V]
    nnz = min(30, num_verts)
    J_regressor = np.zeros((NUM_JOINTS, num_verts), dtype=np.float64)
    for j in range(NUM_JOINTS):
        idx = np.argpartition(d2[j], nnz - 1)[:nnz]
        w = rng.random(nnz) + 0.1
        J_regressor[j, idx] = w / w.sum()
    lbs = np.zeros((num_verts, NUM_JOINTS), dtype=np.float64)
    near = np.argpartition(d2.T, lbs_nnz - 1, axis=1)[:, :lbs_nnz]   # [V,lbs_nnz]
    dn = np.take_along_axis(d2.T, near, axis=1)
    w = np.exp(-dn / (2 * 0.08 ** 2)) + 1e-6
    w /= w.sum(1, keepdims=True)
    np.put_along_axis(lbs, near, w, axis=1)

    return BodyModelData(
        v_template=v_template,
        shapedirs=shapedirs,
        posedirs=posedirs,
        J_regressor=J_regressor.astype(np.float32),
        parents=parents.copy(),
        lbs_weights=lbs.astype(np.float32),
        hands_componentsl=(rng.standard_normal((NUM_HAND_PCA, 45)) * 0.1).astype(np.float32),
        hands_componentsr=(rng.standard_normal((NUM_HAND_PCA, 45)) * 0.1).astype(np.float32),
        hands_meanl=(rng.standard_normal(45) * 0.1).astype(np.float32),
        hands_meanr=(rng.standard_normal(45) * 0.1).astype(np.float32),
    )


@dataclass
class VPoserData:
    """VPoser v1.0 decoder weights under the checkpoint's state-dict key names (A.2)."""
    fc1_w: np.ndarray  # bodyprior_dec_fc1.weight [512,32]
    fc1_b: np.ndarray  # [512]
    fc2_w: np.ndarray  # bodyprior_dec_fc2.weight [512,512]
    fc2_b: np.ndarray  # [512]
    out_w: np.ndarray  # bodyprior_dec_out.weight [126,512]
    out_b: np.ndarray  # [126]


def make_vposer(seed: int = 1, pose_gain: float = 0.25) -> VPoserData:
    """Kaiming-uniform hidden layers; the output layer is scaled by `pose_gain` and biased to
    the 6D code of the identity so decoded joints stay within a plausible range of motion."""
    rng = np.random.Generator(np.random.PCG64(seed))

    def ku(out_f, in_f):
        bound = math.sqrt(6.0 / in_f)
        return ((rng.random((out_f, in_f)) * 2 - 1) * bound).astype(np.float32)

    out_b = np.tile(np.array([1, 0, 0, 1, 0, 0], dtype=np.float32), 21)
    return VPoserData(
        fc1_w=ku(512, 32), fc1_b=np.zeros(512, np.float32),
        fc2_w=ku(512, 512), fc2_b=np.zeros(512, np.float32),
        out_w=(ku(126, 512) * pose_gain).astype(np.float32), out_b=out_b)


def _smooth_walk(rng, n, dim, sigma, window):
    x = rng.standard_normal((n + 2 * window, dim))
    k = np.hanning(2 * window + 1)
    k /= k.sum()
    y = np.stack([np.convolve(x[:, d], k, mode="valid") for d in range(dim)], 1)
    y = y[:n]
    y = y / (y.std() + 1e-12) * sigma
    return y


def _aa_to_R(aa):
    th = np.linalg.norm(aa)
    if th < 1e-12:
        return np.eye(3)
    k = aa / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K


def _R_to_quat_wxyz(R):
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s,
                      (R[1, 0] - R[0, 1]) / s])
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s,
                      (R[0, 2] + R[2, 0]) / s])
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = np.array([(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s,
                      (R[1, 2] + R[2, 1]) / s])
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = np.array([(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s,
                      (R[1, 2] + R[2, 1]) / s, 0.25 * s])
    return q / np.linalg.norm(q)


@dataclass
class ClipData:
    body_params: np.ndarray     # [N,75] fp32, SMPLify-X layout (global_optimization.py:64-76)
    camerapose_lines: list      # N strings " qw qx qy qz tx ty tz" (utils/camerapose_helper.py:27)
    cam_ext: np.ndarray         # [N,4,4] fp64 camera-to-world the lines invert to
    outlier_frames: np.ndarray  # planted outlier indices


def make_clip(num_frames: int, seed: int = 3, scale_init: float = 1.8,
              num_outliers: int | None = None) -> ClipData:
    rng = np.random.Generator(np.random.PCG64(seed))
    n = num_frames
    win = max(2, min(15, n // 4))
    transl = np.zeros((n, 3))
    go = _smooth_walk(rng, n, 3, 0.25, win)
    betas = rng.standard_normal((1, 10)) * 0.5 + rng.standard_normal((n, 10)) * 0.02
    z = np.zeros((n, 32))
    z[0] = rng.standard_normal(32) * 0.7
    rho = 0.95
    for i in range(1, n):
        z[i] = rho * z[i - 1] + math.sqrt(1 - rho * rho) * 0.7 * rng.standard_normal(32)
    lh = rng.standard_normal((n, 12)) * 0.2
    rh = rng.standard_normal((n, 12)) * 0.2
    cam_t = np.array([0.0, 0.0, 3.0]) + _smooth_walk(rng, n, 3, 0.05, win)
    if num_outliers is None:
        num_outliers = max(1, -(-n // 64))
    if n >= 4:
        out = np.sort(rng.choice(np.arange(1, n - 1), size=min(num_outliers, n - 2),
                                 replace=False))
        z[out] *= 4.0
    else:
        out = np.zeros(0, dtype=np.int64)
    body = np.concatenate([transl, go, betas, z, lh, rh, cam_t], 1).astype(np.float32)

    # camera-to-world: body y-up -> world z-up plus a smooth wobble; feet (y ~ -0.95) land
    # a few cm above the z=0 floor at the initial scale
    base = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float64)
    wob = _smooth_walk(rng, n, 3, 0.03, win)
    pos = _smooth_walk(rng, n, 3, 0.15, win)
    cam_ext = np.zeros((n, 4, 4))
    lines = []
    for i in range(n):
        R = _aa_to_R(wob[i]) @ base
        t = np.array([pos[i, 0], pos[i, 1], 0.0])
        foot_cam = scale_init * (np.array([0.0, -0.96, 0.0]) + cam_t[i])
        t[2] = 0.04 + 0.01 * pos[i, 2] - (R @ foot_cam)[2]
        t[:2] -= (R @ (scale_init * cam_t[i]))[:2]
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = t
        cam_ext[i] = E
        W = np.linalg.inv(E)                      # world -> camera (COLMAP convention)
        q = _R_to_quat_wxyz(W[:3, :3])
        lines.append(" " + " ".join(repr(float(x)) for x in (*q, *W[:3, 3])))
    return ClipData(body_params=body, camerapose_lines=lines, cam_ext=cam_ext, outlier_frames=out)


def make_scene(num_points: int, seed: int = 2) -> np.ndarray:
    """Floor plane z=0 (60 %) + walls/boxes (40 %), 10 m x 10 m x 3 m, 5 mm noise, shuffled."""
    rng = np.random.Generator(np.random.PCG64(seed))
    if num_points == 0:
        return np.zeros((0, 3), np.float32)
    nf = int(round(num_points * 0.6))
    nw = num_points - nf
    floor = np.stack([rng.random(nf) * 10 - 5, rng.random(nf) * 10 - 5, np.zeros(nf)], 1)
    parts = [floor]
    per = [nw // 4 + (1 if k < nw % 4 else 0) for k in range(4)]
    specs = [(0, -5.0), (0, 5.0), (1, -5.0), (1, 5.0)]
    for cnt, (axis, val) in zip(per, specs):
        w = np.zeros((cnt, 3))
        w[:, axis] = val
        w[:, 1 - axis] = rng.random(cnt) * 10 - 5
        w[:, 2] = rng.random(cnt) * 3
        parts.append(w)
    pts = np.concatenate(parts, 0)
    pts += rng.standard_normal(pts.shape) * 0.005
    rng.shuffle(pts, axis=0)
    return pts.astype(np.float32)


def make_contact_ids(v_template: np.ndarray, per_part: int = 250, seed: int = 4):
    """Two disjoint index sets from the lowest 15 % (body-frame y) of the template: the
    stand-in for body_segments/L_Leg.json and R_Leg.json (global_optimization.py:675-676)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    V = v_template.shape[0]
    order = np.argsort(v_template[:, 1])
    low = order[:max(2 * per_part, int(0.15 * V))]
    per_part = min(per_part, low.size // 2)
    pick = rng.permutation(low)[:2 * per_part]
    left = np.sort(pick[:per_part]).astype(np.int64)
    right = np.sort(pick[per_part:]).astype(np.int64)
    return left, right


def dct_basis(num_frames: int = 60, num_coef: int = 5) -> np.ndarray:
    """Orthonormal DCT-II basis [num_frames, num_coef]: stand-in for ../Data/DCT_Basis/60.mat
    (global_optimization.py:131-136 returns D[:5].T)."""
    n = np.arange(num_frames)
    D = np.zeros((num_coef, num_frames))
    for k in range(num_coef):
        a = math.sqrt(1.0 / num_frames) if k == 0 else math.sqrt(2.0 / num_frames)
        D[k] = a * np.cos(math.pi * (n + 0.5) * k / num_frames)
    return D.T.astype(np.float32)
