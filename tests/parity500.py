"""TEST INFRASTRUCTURE: distances between two results of the fixed-budget optimisation, in the units the north star asks for --
millimetres of world-space vertex / joint position (both parameter sets decoded through the ORACLE's body model in fp64), next
to the parameter-space quantiles, `scale` and camera_ext."""
import numpy as np
import torch

from oracle.fitting import FittingOracle
from oracle.smplx import SMPLXOracle
from oracle.vposer import VPoserDecoder


def world_mesh(bm, vp, camerapose_lines, x78, scale, cam):
    """x78 [N,78], scale scalar, cam [N,4,4] (or [N,16]) -> world vertices [N,V,3] and the 23 world joints [N,23,3], float64
    (the reference's forward: global_optimization.py:253, :261-285, :298-299)."""
    dt = torch.float64
    n = x78.shape[0]
    f = FittingOracle(SMPLXOracle(bm, dt), VPoserDecoder.from_data(vp, dt), np.zeros((0, 3)), np.zeros(0, np.int64), camerapose_lines, n, dtype=dt)
    f.body_rotation_rec.data = torch.as_tensor(np.asarray(x78, dtype=np.float64))
    f.scale.data = torch.tensor(float(scale), dtype=dt)
    f.camera_ext.data = torch.as_tensor(np.asarray(cam, dtype=np.float64).reshape(n, 4, 4))
    with torch.no_grad():
        _, verts, joints = f.forward_world()
    return verts.numpy(), joints.numpy()


def distance_report(bm, vp, camerapose_lines, a, b):
    """a, b: (x78, scale, cam).  Returns a dict of plain floats."""
    va, ja = world_mesh(bm, vp, camerapose_lines, *a)
    vb, jb = world_mesh(bm, vp, camerapose_lines, *b)
    dv = np.linalg.norm(va - vb, axis=-1) * 1e3          # mm, [N,V]
    dj = np.linalg.norm(ja - jb, axis=-1) * 1e3
    ex = np.abs(np.asarray(a[0], np.float64) - np.asarray(b[0], np.float64))
    ec = np.abs(np.asarray(a[2], np.float64).reshape(-1, 16) - np.asarray(b[2], np.float64).reshape(-1, 16))
    q = lambda x, p: float(np.quantile(x, p))
    return {"vert_mm_mean": float(dv.mean()), "vert_mm_q50": q(dv, 0.5), "vert_mm_q99": q(dv, 0.99), "vert_mm_max": float(dv.max()),
            "vert_mm_worst_frame_mean": float(dv.mean(1).max()),
            "joint_mm_mean": float(dj.mean()), "joint_mm_q99": q(dj, 0.99), "joint_mm_max": float(dj.max()),
            "x78_q50": q(ex, 0.5), "x78_q90": q(ex, 0.9), "x78_q99": q(ex, 0.99), "x78_max": float(ex.max()),
            "hands_max": float(ex[:, 51:75].max()), "scale_abs": abs(float(a[1]) - float(b[1])),
            "cam_q99": q(ec, 0.99), "cam_max": float(ec.max())}
