"""Rotation-representation and camera glue of the PSI path, same names and argument meaning as the reference.

* ``ContinousRotReprDecoder``, ``GeometryTransformer``, ``BodyParamParser``  <- source/cvae.py:36-89, 97-199, 217-334
* ``angle_axis_to_rotation_matrix`` / ``rotation_matrix_to_angle_axis``      <- ``torchgeometry==0.1.2`` (third party, not
  in the reference tree; behaviour restated per SURVEY.md Appendix D, call sites cvae.py:79,88, vposer_smpl.py:160,170)

These are [B,<=75]-sized elementwise ops; they stay in PyTorch on whatever device the tensors live on (the
fused fitting engine in libpsi_hip.so carries its own copies of the same formulas, csrc/fit.hip).
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


# ------------------------------------------------------------------------------------------
# torchgeometry 0.1.2 conversions
# ------------------------------------------------------------------------------------------
def angle_axis_to_rotation_matrix(angle_axis: torch.Tensor) -> torch.Tensor:
    """[N,3] -> [N,4,4] (Rodrigues; first-order Taylor branch when theta^2 <= 1e-6)."""
    aa = angle_axis
    theta2 = (aa * aa).sum(dim=1, keepdim=True)
    theta = torch.sqrt(theta2)
    wxyz = aa / (theta + 1e-6)
    wx, wy, wz = wxyz[:, 0:1], wxyz[:, 1:2], wxyz[:, 2:3]
    c, s = torch.cos(theta), torch.sin(theta)
    k = 1.0 - c
    normal = torch.cat([c + wx * wx * k, wx * wy * k - wz * s, wy * s + wx * wz * k,
                        wz * s + wx * wy * k, c + wy * wy * k, -wx * s + wy * wz * k,
                        -wy * s + wx * wz * k, wx * s + wy * wz * k, c + wz * wz * k], dim=1).view(-1, 3, 3)
    rx, ry, rz = aa[:, 0:1], aa[:, 1:2], aa[:, 2:3]
    one = torch.ones_like(rx)
    taylor = torch.cat([one, -rz, ry, rz, one, -rx, -ry, rx, one], dim=1).view(-1, 3, 3)
    mask = (theta2 > 1e-6).view(-1, 1, 1).to(aa.dtype)
    out = torch.eye(4, dtype=aa.dtype, device=aa.device).view(1, 4, 4).repeat(aa.shape[0], 1, 1)
    out[:, :3, :3] = mask * normal + (1.0 - mask) * taylor
    return out


def rotation_matrix_to_quaternion(rotation_matrix: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """[N,3,4] -> [N,4] (w,x,y,z); four-branch formula evaluated on the transpose."""
    m = rotation_matrix.transpose(1, 2)
    m00, m11, m22 = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
    neg_d2 = m22 < eps
    d0_gt_d1 = m00 > m11
    d0_lt_nd1 = m00 < -m11
    t0 = 1 + m00 - m11 - m22
    t1 = 1 - m00 + m11 - m22
    t2 = 1 - m00 - m11 + m22
    t3 = 1 + m00 + m11 + m22
    q0 = torch.stack([m[:, 1, 2] - m[:, 2, 1], t0, m[:, 0, 1] + m[:, 1, 0], m[:, 2, 0] + m[:, 0, 2]], -1)
    q1 = torch.stack([m[:, 2, 0] - m[:, 0, 2], m[:, 0, 1] + m[:, 1, 0], t1, m[:, 1, 2] + m[:, 2, 1]], -1)
    q2 = torch.stack([m[:, 0, 1] - m[:, 1, 0], m[:, 2, 0] + m[:, 0, 2], m[:, 1, 2] + m[:, 2, 1], t2], -1)
    q3 = torch.stack([t3, m[:, 1, 2] - m[:, 2, 1], m[:, 2, 0] - m[:, 0, 2], m[:, 0, 1] - m[:, 1, 0]], -1)
    c0 = (neg_d2 & d0_gt_d1).unsqueeze(1).to(m.dtype)
    c1 = (neg_d2 & ~d0_gt_d1).unsqueeze(1).to(m.dtype)
    c2 = (~neg_d2 & d0_lt_nd1).unsqueeze(1).to(m.dtype)
    c3 = (~neg_d2 & ~d0_lt_nd1).unsqueeze(1).to(m.dtype)
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    q = q / torch.sqrt(t0.unsqueeze(1) * c0 + t1.unsqueeze(1) * c1 + t2.unsqueeze(1) * c2 + t3.unsqueeze(1) * c3)
    return q * 0.5


def quaternion_to_angle_axis(quaternion: torch.Tensor) -> torch.Tensor:
    q1, q2, q3 = quaternion[..., 1], quaternion[..., 2], quaternion[..., 3]
    sin_sq = q1 * q1 + q2 * q2 + q3 * q3
    sin_t = torch.sqrt(sin_sq)
    cos_t = quaternion[..., 0]
    two_theta = 2.0 * torch.where(cos_t < 0.0, torch.atan2(-sin_t, -cos_t), torch.atan2(sin_t, cos_t))
    k = torch.where(sin_sq > 0.0, two_theta / sin_t, 2.0 * torch.ones_like(sin_t))
    return torch.stack([q1 * k, q2 * k, q3 * k], dim=-1)


def rotation_matrix_to_angle_axis(rotation_matrix: torch.Tensor) -> torch.Tensor:
    """[N,3,4] -> [N,3]."""
    return quaternion_to_angle_axis(rotation_matrix_to_quaternion(rotation_matrix))


# ------------------------------------------------------------------------------------------
# source/cvae.py classes
# ------------------------------------------------------------------------------------------
class ContinousRotReprDecoder(nn.Module):
    """6-D continuous rotation representation (Zhou et al.), cvae.py:36-89 / vposer_smpl.py:49-62."""

    def forward(self, module_input):
        return self.decode(module_input)

    @staticmethod
    def decode(module_input):
        x = module_input.view(-1, 3, 2)
        b1 = F.normalize(x[:, :, 0], dim=1)
        dot = torch.sum(b1 * x[:, :, 1], dim=1, keepdim=True)
        b2 = F.normalize(x[:, :, 1] - dot * b1, dim=-1)
        b3 = torch.cross(b1, b2, dim=1)
        return torch.stack([b1, b2, b3], dim=-1)

    @staticmethod
    def matrot2aa(pose_matrot):
        homogen = F.pad(pose_matrot.view(-1, 3, 3), [0, 1])
        return rotation_matrix_to_angle_axis(homogen).view(-1, 3).contiguous()

    @staticmethod
    def aa2matrot(pose):
        return angle_axis_to_rotation_matrix(pose.reshape(-1, 3))[:, :3, :3].contiguous()


_CONTACT_CACHE: dict = {}


class GeometryTransformer:
    @staticmethod
    def get_contact_id(body_segments_folder, contact_body_parts=['L_Hand', 'R_Hand']):
        """cvae.py:99-115.  The reference re-reads the JSON files on every loss evaluation; here the result is
        cached per (folder, parts) — same expression (list(set(.)) per part, concatenated), evaluated once."""
        key = (os.path.abspath(body_segments_folder), tuple(contact_body_parts))
        hit = _CONTACT_CACHE.get(key)
        if hit is None:
            verts, faces = [], []
            for part in contact_body_parts:
                with open(os.path.join(body_segments_folder, part + '.json'), 'r') as f:
                    data = json.load(f)
                verts.append(list(set(data['verts_ind'])))
                faces.append(list(set(data['faces_ind'])))
            hit = (np.concatenate(verts), np.concatenate(faces))
            _CONTACT_CACHE[key] = hit
        return hit

    @staticmethod
    def convert_to_6D_rot(x_batch):
        xt, xr, xb = x_batch[:, :3], x_batch[:, 3:6], x_batch[:, 6:]
        xr_mat = ContinousRotReprDecoder.aa2matrot(xr)
        return torch.cat([xt, xr_mat[:, :, :-1].reshape([-1, 6]), xb], dim=-1)

    @staticmethod
    def convert_to_3D_rot(x_batch):
        xt, xr, xb = x_batch[:, :3], x_batch[:, 3:9], x_batch[:, 9:]
        xr_aa = ContinousRotReprDecoder.matrot2aa(ContinousRotReprDecoder.decode(xr))
        return torch.cat([xt, xr_aa, xb], dim=-1)

    @staticmethod
    def verts_transform(verts_batch, cam_ext_batch):
        homo = F.pad(verts_batch, (0, 1), mode='constant', value=1)
        return torch.matmul(homo, cam_ext_batch.permute(0, 2, 1))[:, :, :-1]

    @staticmethod
    def recover_global_T(x_batch, cam_intrisic, max_depth):
        xt, xr = x_batch[:, :3], x_batch[:, 3:]
        fx, fy = cam_intrisic[:, 0, 0], cam_intrisic[:, 1, 1]
        px, py = cam_intrisic[:, 0, 2], cam_intrisic[:, 1, 2]
        s_ = 1.0 / torch.max(px, py)
        z = (xt[:, 2] + 1.0) / 2.0 * max_depth
        x = xt[:, 0] * z / s_ / fx
        y = xt[:, 1] * z / s_ / fy
        return torch.cat([torch.stack([x, y, z], dim=-1), xr], dim=-1)

    @staticmethod
    def normalize_global_T(x_batch, cam_intrisic, max_depth):
        xt, xr = x_batch[:, :3], x_batch[:, 3:]
        fx, fy = cam_intrisic[:, 0, 0], cam_intrisic[:, 1, 1]
        px, py = cam_intrisic[:, 0, 2], cam_intrisic[:, 1, 2]
        s_ = 1.0 / torch.max(px, py)
        x = s_ * xt[:, 0] * fx / (xt[:, 2] + 1e-6)
        y = s_ * xt[:, 1] * fy / (xt[:, 2] + 1e-6)
        z = 2.0 * xt[:, 2] / max_depth - 1.0
        return torch.cat([torch.stack([x, y, z], dim=-1), xr], dim=-1)


_BODY_KEYS = (('transl', 0, 3), ('global_orient', 3, 6), ('betas', 6, 16), ('body_pose', 16, 48),
              ('left_hand_pose', 48, 60), ('right_hand_pose', 60, 72))


class BodyParamParser:
    """72-D body vector <-> the pkl dict of generated bodies (cvae.py:217-334)."""

    device = None   # set to a torch.device to place parsed tensors there (the reference hard-codes .cuda())

    @staticmethod
    def _dev():
        if BodyParamParser.device is not None:
            return BodyParamParser.device
        return torch.device('cuda' if torch.cuda.is_available() else 'cpu')

    @staticmethod
    def body_params_encapsulate(x_body_rec):
        x = x_body_rec.detach().cpu().numpy()
        return [{k: x[b:b + 1, lo:hi] for k, lo, hi in _BODY_KEYS} for b in range(x.shape[0])]

    @staticmethod
    def body_params_encapsulate_batch(x_body_rec):
        out = {k: x_body_rec[:, lo:hi] for k, lo, hi in _BODY_KEYS if k != 'body_pose'}
        out['body_pose_vp'] = x_body_rec[:, 16:48]
        return out

    @staticmethod
    def body_params_encapsulate_latent(x_body_rec, eps=None):
        recs = BodyParamParser.body_params_encapsulate(x_body_rec)
        e = eps.detach().cpu().numpy()
        for b, r in enumerate(recs):
            r['z'] = e[b:b + 1, :]
        return recs

    @staticmethod
    def _vector(body_params_batch):
        return np.concatenate([body_params_batch[k] for k, _, _ in _BODY_KEYS], axis=-1)

    @staticmethod
    def body_params_parse(body_params_batch):
        return torch.tensor(BodyParamParser._vector(body_params_batch), dtype=torch.float32, device=BodyParamParser._dev())

    @staticmethod
    def body_params_parse_fitting(body_params_batch):
        dev = BodyParamParser._dev()
        cam_ext = torch.tensor(body_params_batch['cam_ext'], dtype=torch.float32, device=dev)
        cam_int = torch.tensor(body_params_batch['cam_int'], dtype=torch.float32, device=dev)
        x = torch.tensor(BodyParamParser._vector(body_params_batch), dtype=torch.float32, device=dev)
        return x, cam_ext, cam_int
