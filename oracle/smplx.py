"""Restatement of smplx.SMPLX.forward + smplx.lbs.lbs as configured by the reference.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The package is absent and unpinned; call sites
/root/reference/global_optimization.py:154-168 (create: model_type='smplx', gender='neutral',
num_pca_comps=12, use_pca=True, flat_hand_mean=False) and :280-283 (forward).  Algorithm:
SURVEY.md Appendix A.3.  Parity unpinned against the real package.
"""
from types import SimpleNamespace

import torch
import torch.nn.functional as F


def batch_rodrigues(rot_vecs: torch.Tensor) -> torch.Tensor:
    """smplx.lbs.batch_rodrigues: angle = ||r + 1e-8|| (added per component)."""
    n = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.cos(angle).unsqueeze(1)
    sin = torch.sin(angle).unsqueeze(1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((n, 1), dtype=rot_vecs.dtype, device=rot_vecs.device)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(n, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device).unsqueeze(0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def batch_rigid_transform(rot_mats, joints, parents):
    """smplx.lbs.batch_rigid_transform."""
    joints = joints.unsqueeze(-1)
    rel_joints = joints.clone()
    rel_joints[:, 1:] = rel_joints[:, 1:] - joints[:, parents[1:]]
    b, nj = rot_mats.shape[0], joints.shape[1]
    tm = torch.cat([F.pad(rot_mats.reshape(-1, 3, 3), [0, 0, 0, 1]),
                    F.pad(rel_joints.reshape(-1, 3, 1), [0, 0, 0, 1], value=1)], dim=2)
    tm = tm.reshape(b, nj, 4, 4)
    chain = [tm[:, 0]]
    for i in range(1, nj):
        chain.append(torch.matmul(chain[int(parents[i])], tm[:, i]))
    transforms = torch.stack(chain, dim=1)
    posed_joints = transforms[:, :, :3, 3]
    joints_homogen = F.pad(joints, [0, 0, 0, 1])
    rel_transforms = transforms - F.pad(torch.matmul(transforms, joints_homogen),
                                        [3, 0, 0, 0, 0, 0, 0, 0])
    return posed_joints, rel_transforms


class SMPLXOracle(torch.nn.Module):
    """`model(return_verts=True, body_pose=[B,63], transl, global_orient, betas,
    left_hand_pose=[B,12], right_hand_pose=[B,12])` -> object with .vertices [B,V,3] and
    .joints [B,55,3] (the reference reads joints[:, 0:23] only; landmark/extra joints that the
    real model appends after index 54 are not restated)."""

    def __init__(self, data, dtype=torch.float32):
        super().__init__()
        t = lambda a: torch.as_tensor(a).to(dtype).clone()
        self.register_buffer("v_template", t(data.v_template))
        self.register_buffer("shapedirs", t(data.shapedirs))          # [V,3,20]
        self.register_buffer("posedirs", t(data.posedirs))            # [486,3V]
        self.register_buffer("J_regressor", t(data.J_regressor))
        self.register_buffer("lbs_weights", t(data.lbs_weights))
        self.register_buffer("lh_comp", t(data.hands_componentsl))    # [12,45]
        self.register_buffer("rh_comp", t(data.hands_componentsr))
        pose_mean = torch.zeros(165, dtype=dtype)
        pose_mean[75:120] = t(data.hands_meanl)
        pose_mean[120:165] = t(data.hands_meanr)
        self.register_buffer("pose_mean", pose_mean)
        self.parents = torch.as_tensor(data.parents).long()
        self.dtype = dtype

    def forward(self, return_verts=True, body_pose=None, transl=None, global_orient=None,
                betas=None, left_hand_pose=None, right_hand_pose=None, **unused):
        b = body_pose.shape[0]
        z3 = torch.zeros(b, 3, dtype=self.dtype)
        expression = torch.zeros(b, 10, dtype=self.dtype)
        lh = torch.einsum("bi,ij->bj", left_hand_pose, self.lh_comp)
        rh = torch.einsum("bi,ij->bj", right_hand_pose, self.rh_comp)
        full_pose = torch.cat([global_orient, body_pose, z3, z3, z3, lh, rh], dim=1)
        full_pose = full_pose + self.pose_mean
        shape_components = torch.cat([betas, expression], dim=-1)

        v_shaped = self.v_template + torch.einsum("bl,mkl->bmk", shape_components, self.shapedirs)
        J = torch.einsum("bik,ji->bjk", v_shaped, self.J_regressor)
        rot_mats = batch_rodrigues(full_pose.reshape(-1, 3)).view(b, -1, 3, 3)
        ident = torch.eye(3, dtype=self.dtype)
        pose_feature = (rot_mats[:, 1:] - ident).reshape(b, -1)
        v_posed = v_shaped + torch.matmul(pose_feature, self.posedirs).view(b, -1, 3)
        J_transformed, A = batch_rigid_transform(rot_mats, J, self.parents)
        W = self.lbs_weights.unsqueeze(0).expand(b, -1, -1)
        nj = self.J_regressor.shape[0]
        T = torch.matmul(W, A.view(b, nj, 16)).view(b, -1, 4, 4)
        homo = torch.cat([v_posed, torch.ones(b, v_posed.shape[1], 1, dtype=self.dtype)], dim=2)
        verts = torch.matmul(T, homo.unsqueeze(-1))[:, :, :3, 0]
        joints = J_transformed
        if transl is not None:
            joints = joints + transl.unsqueeze(1)
            verts = verts + transl.unsqueeze(1)
        return SimpleNamespace(vertices=verts, joints=joints, full_pose=full_pose)
