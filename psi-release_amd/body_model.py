"""SMPL-X body model on the HIP LBS kernels, with the call surface PSI uses from ``smplx``.

Reference call sites: ``smplx.create(path, model_type='smplx', gender='neutral', ext='npz', num_pca_comps=12,
create_*=True, batch_size=B)`` and ``model(return_verts=True, body_pose=[B,63], transl=[B,3], global_orient=[B,3],
betas=[B,10], left_hand_pose=[B,12], right_hand_pose=[B,12]).vertices`` — fitting_proxe.py:55-69,125-128,
train_s1.py:66-81,150-153, test_proxe_s1.py:56-71.  ``smplx==0.1.13`` itself is third party and not in the
reference tree; its forward is restated per SURVEY.md Appendix D (parity unpinned at that boundary), the
LBS arithmetic follows human_body_prior/body_model/lbs.py:34-118.

The hand PCA (12->45) and the pose_mean add are three tiny torch ops on the GPU; everything V-sized runs
in libpsi_hip.so (psi_lbs_forward / psi_lbs_backward).
"""
from __future__ import annotations

import ctypes
import os
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn
from torch.autograd import Function

from . import hip


class LbsModel:
    """Owns a ``psi_lbs_model`` handle (device copy of the model tensors in the kernels' layout)."""

    def __init__(self, v_template, shapedirs, posedirs_PxN, J_regressor, weights, parents, device):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        vt, sd, pd, jr, w = f(v_template), f(shapedirs), f(posedirs_PxN), f(J_regressor), f(weights)
        par = np.ascontiguousarray(parents, dtype=np.int32)
        self.V, self.J, self.NB = vt.shape[0], jr.shape[0], sd.shape[2]
        assert sd.shape == (self.V, 3, self.NB) and pd.shape == ((self.J - 1) * 9, 3 * self.V)
        assert jr.shape == (self.J, self.V) and w.shape == (self.V, self.J) and par.shape == (self.J,)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise hip.PsiHipError('LbsModel needs a GPU device (no CPU implementation exists in this package)')
        h = ctypes.c_void_p()
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        with torch.cuda.device(self.device):
            hip.check(hip.lib().psi_lbs_create(ctypes.byref(h), p(vt), p(sd), p(pd), p(jr), p(w), p(par),
                                               self.V, self.J, self.NB), 'psi_lbs_create')
        self.handle = h
        self._ws = {}

    def workspace(self, B):
        ws = self._ws.get(B)
        if ws is None:
            n = hip.lib().psi_lbs_workspace_floats(self.handle, B)
            ws = torch.zeros(n, dtype=torch.float32, device=self.device)
            self._ws = {B: ws}          # keep one size resident
        return ws

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                hip.lib().psi_lbs_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class _LbsFn(Function):
    @staticmethod
    def forward(ctx, model: LbsModel, betas, pose, transl, cam_ext, want_joints):
        betas = betas.contiguous().float()
        pose = pose.contiguous().float()
        B = betas.shape[0]
        if tuple(betas.shape) != (B, model.NB) or tuple(pose.shape) != (B, model.J * 3):
            raise ValueError('lbs: betas must be [B,%d] and pose [B,%d], got %s / %s' % (model.NB, model.J * 3, tuple(betas.shape), tuple(pose.shape)))
        if transl is not None and tuple(transl.shape) != (B, 3):
            raise ValueError('lbs: transl must be [%d,3], got %s' % (B, tuple(transl.shape)))
        if cam_ext is not None:
            if cam_ext.dim() != 3 or tuple(cam_ext.shape[1:]) != (4, 4) or cam_ext.shape[0] not in (1, B):
                raise ValueError('lbs: cam_ext must be [%d,4,4] or [1,4,4], got %s' % (B, tuple(cam_ext.shape)))
            cam_ext = cam_ext.expand(B, 4, 4)
        transl_c = transl.contiguous().float() if transl is not None else None
        cam_c = cam_ext.contiguous().float() if cam_ext is not None else None
        verts = torch.empty(B, model.V, 3, device=betas.device)
        joints = torch.empty(B, model.J, 3, device=betas.device) if want_joints else None
        # a private workspace per call keeps forward/backward pairs independent (autograd may interleave them)
        ws = torch.empty(hip.lib().psi_lbs_workspace_floats(model.handle, B), dtype=torch.float32, device=betas.device)
        hip.check(hip.lib().psi_lbs_forward(model.handle, hip.ptr(betas), hip.ptr(pose), hip.ptr(transl_c), hip.ptr(cam_c),
                                            B, hip.ptr(verts), hip.ptr(joints), hip.ptr(ws), hip.stream()), 'psi_lbs_forward')
        ctx.model = model
        ctx.has_transl = transl is not None
        ctx.save_for_backward(betas, pose, cam_c if cam_c is not None else torch.empty(0, device=betas.device), ws)
        if want_joints:
            ctx.mark_non_differentiable(joints)
            return verts, joints
        return verts, torch.empty(0, device=betas.device)

    @staticmethod
    def backward(ctx, gverts, _gj):
        betas, pose, cam, ws = ctx.saved_tensors
        model = ctx.model
        B = betas.shape[0]
        gb = torch.empty_like(betas)
        gp = torch.empty_like(pose)
        gt = torch.empty(B, 3, device=betas.device) if ctx.has_transl else None
        hip.check(hip.lib().psi_lbs_backward(model.handle, hip.ptr(gverts.contiguous().float()), hip.ptr(betas), hip.ptr(pose),
                                             hip.ptr(cam) if cam.numel() else None, B, hip.ptr(ws), hip.ptr(gb), hip.ptr(gp),
                                             hip.ptr(gt), hip.stream()), 'psi_lbs_backward')
        return None, gb, gp, gt, None, None


def lbs(model: LbsModel, betas, pose, transl=None, cam_ext=None, return_joints=False):
    """verts [B,V,3] (and joints [B,J,3]) from betas [B,NB] and full axis-angle pose [B,J*3]."""
    v, j = _LbsFn.apply(model, betas, pose, transl, cam_ext, return_joints)
    return (v, j) if return_joints else v


class SMPLXLayer(nn.Module):
    """``smplx.create(..., model_type='smplx', num_pca_comps=12, batch_size=B)`` stand-in for PSI's call pattern."""

    NUM_BODY_JOINTS = 21

    def __init__(self, data, num_pca_comps: int = 12, num_betas: int = 10, num_expression_coeffs: int = 10,
                 flat_hand_mean: bool = False, batch_size: int = 1, device='cuda', **unused):
        super().__init__()
        g = (lambda k: np.asarray(data[k])) if not hasattr(data, 'v_template') else (lambda k: np.asarray(getattr(data, k)))
        sd = g('shapedirs')
        expr0 = 300 if sd.shape[-1] > 300 else 10                                   # body_model.py:105-106
        shapedirs = np.concatenate([sd[:, :, :num_betas], sd[:, :, expr0:expr0 + num_expression_coeffs]], -1)
        pdirs = g('posedirs')
        posedirs = pdirs.reshape(-1, pdirs.shape[-1]).T.copy()                      # [P, 3V], body_model.py:123-125
        parents = np.asarray(g('kintree_table'))[0].astype(np.int64).copy()
        parents[0] = -1
        J = parents.shape[0]
        self.batch_size = batch_size
        self.num_betas, self.num_expr, self.num_pca_comps = num_betas, num_expression_coeffs, num_pca_comps
        self.lbs_model = LbsModel(g('v_template'), shapedirs, posedirs, g('J_regressor'), g('weights'), parents, device)
        dev = self.lbs_model.device
        t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
        self.register_buffer('left_hand_components', t(g('hands_componentsl')[:num_pca_comps]))
        self.register_buffer('right_hand_components', t(g('hands_componentsr')[:num_pca_comps]))
        pm = np.zeros(J * 3, np.float32)
        if not flat_hand_mean:
            pm[(J - 30) * 3:(J - 15) * 3] = g('hands_meanl')
            pm[(J - 15) * 3:] = g('hands_meanr')
        self.register_buffer('pose_mean', t(pm))
        self.register_buffer('faces_tensor', torch.tensor(np.asarray(g('f')).astype(np.int64).reshape(-1, 3), device=dev))
        self.J = J
        # smplx exposes the per-batch defaults as zero Parameters (create_*=True); PSI always passes values
        z = lambda n: nn.Parameter(torch.zeros(batch_size, n, device=dev), requires_grad=True)
        self.expression, self.jaw_pose, self.leye_pose, self.reye_pose = z(num_expression_coeffs), z(3), z(3), z(3)

    def forward(self, betas=None, global_orient=None, body_pose=None, left_hand_pose=None, right_hand_pose=None,
                transl=None, expression=None, jaw_pose=None, leye_pose=None, reye_pose=None, return_verts=True,
                return_full_pose=False, cam_ext=None, **unused):
        B = betas.shape[0]
        dev = betas.device
        zeros3 = torch.zeros(B, 3, device=dev)
        jaw = zeros3 if jaw_pose is None else jaw_pose
        le = zeros3 if leye_pose is None else leye_pose
        re = zeros3 if reye_pose is None else reye_pose
        lh = left_hand_pose @ self.left_hand_components
        rh = right_hand_pose @ self.right_hand_components
        full_pose = torch.cat([global_orient, body_pose, jaw, le, re, lh, rh], dim=1) + self.pose_mean
        expr = torch.zeros(B, self.num_expr, device=dev) if expression is None else expression
        shape = torch.cat([betas, expr], dim=-1)
        verts, joints = lbs(self.lbs_model, shape, full_pose, transl, cam_ext, return_joints=True)
        return SimpleNamespace(vertices=verts, joints=joints, full_pose=full_pose if return_full_pose else None,
                               betas=betas, global_orient=global_orient, body_pose=body_pose)


def load_smplx_npz(path):
    """Read ``{human_model_path}/smplx/SMPLX_NEUTRAL.npz`` (train_s1.py:83-85 reads 'f' from the same file)."""
    d = np.load(path, allow_pickle=True)
    return {k: d[k] for k in d.files}


def create(model_path, model_type='smplx', gender='neutral', ext='npz', **kwargs):
    """``smplx.create`` (fitting_proxe.py:55).  ``model_path`` may be the folder that holds ``smplx/SMPLX_<GENDER>.npz``,
    the file itself, or an in-memory dict / synth.SMPLXData (tests, bench)."""
    if model_type != 'smplx':
        raise ValueError('only model_type="smplx" is on the PSI path')
    if isinstance(model_path, (dict,)) or hasattr(model_path, 'v_template'):
        data = model_path
    else:
        p = model_path
        if os.path.isdir(p):
            cand = os.path.join(p, 'smplx', 'SMPLX_%s.%s' % (gender.upper(), ext))
            p = cand if os.path.exists(cand) else os.path.join(p, 'SMPLX_%s.%s' % (gender.upper(), ext))
        data = load_smplx_npz(p)
    kwargs.pop('create_global_orient', None)
    for k in list(kwargs):
        if k.startswith('create_'):
            kwargs.pop(k)
    return SMPLXLayer(data, **kwargs)
