"""TEST INFRASTRUCTURE — generates tests/golden/*.npz by IMPORTING the reference (this container only).

Run:  python oracle/make_golden.py            (needs /root/reference; never runs on the GPU box)

The reference is Python, so it cannot travel; what travels are the vectors written here: seeded
inputs (regenerated on the other side from ``psi-release_amd/synth.py``) and the reference's
outputs on them.  Import recipe:

* ``torch.Tensor.cuda`` -> identity (the reference hard-codes ``.cuda()``: cvae.py:299,326-332,462);
* ``sys.modules`` stubs for packages that are not installed and not under /root/reference:
  ``torchgeometry`` (two functions, this build's restatement — parity unpinned there), ``open3d``
  (``io.read_triangle_mesh``), ``torchvision.models.resnet18`` (standard BasicBlock ResNet-18,
  restated here), ``configer``;
* ``smplx`` stub = SMPL-X forward restated per SURVEY Appendix D **over the reference's own vendored
  lbs** (human_body_prior/body_model/lbs.py — imported, not restated);
* ``chamfer_pytorch.dist_chamfer`` stub = autograd.Function over oracle/chamfer_oracle.c (the CUDA
  extension cannot be built here);
* ``load_vposer`` -> the reference's own ``VPoser`` class holding synth weights.
"""
from __future__ import annotations

import json
import os
import pickle
import sys
import tempfile
import types
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import psi_oracle as O  # noqa: E402
from psi_release_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


# ------------------------------------------------------------------------------------------
def install_stubs(scene_holder: dict, smplx_data):
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, 'source'))
    torch.Tensor.cuda = lambda self, *a, **k: self

    tgm = types.ModuleType('torchgeometry')
    tgm.angle_axis_to_rotation_matrix = O.tgm_angle_axis_to_rotation_matrix
    tgm.rotation_matrix_to_angle_axis = O.tgm_rotation_matrix_to_angle_axis
    sys.modules['torchgeometry'] = tgm

    o3d = types.ModuleType('open3d')
    o3d.io = types.SimpleNamespace(read_triangle_mesh=lambda p: types.SimpleNamespace(
        vertices=scene_holder['verts'], triangles=np.zeros((1, 3), int)))
    sys.modules['open3d'] = o3d

    tv = types.ModuleType('torchvision')
    tv.models = types.SimpleNamespace(resnet18=resnet18)
    sys.modules['torchvision'] = tv
    cf = types.ModuleType('configer')
    cf.Configer = object
    sys.modules['configer'] = cf

    from human_body_prior.body_model import lbs as L
    _v2j = L.vertices2joints
    L.vertices2joints = lambda Jr, v: _v2j(Jr, v).contiguous()  # torch>=2 einsum view; no arithmetic change
    smplx = types.ModuleType('smplx')
    smplx.lbs = L
    sys.modules['smplx'] = smplx
    sys.modules['smplx.lbs'] = L

    ref_model = O.SMPLXOracle(smplx_data)   # only its tensors are used below; the lbs call is the reference's

    class SMPLXOverRefLBS(nn.Module):
        def __init__(s, batch_size):
            super().__init__()
            s.B = batch_size

        def forward(s, return_verts=True, body_pose=None, transl=None, global_orient=None, betas=None,
                    left_hand_pose=None, right_hand_pose=None, **kw):
            m = ref_model
            B = betas.shape[0]
            z3 = torch.zeros(B, 3)
            lh = torch.einsum('bi,ij->bj', [left_hand_pose, m.lh_comp])
            rh = torch.einsum('bi,ij->bj', [right_hand_pose, m.rh_comp])
            full = torch.cat([global_orient, body_pose, z3, z3, z3, lh, rh], 1) + m.pose_mean
            sc = torch.cat([betas, torch.zeros(B, 10)], -1)
            v, j = L.lbs(sc, full, m.v_template.unsqueeze(0).repeat(B, 1, 1), m.shapedirs, m.posedirs,
                         m.J_regressor, m.parents, m.lbs_weights, num_joints=m.J_regressor.shape[0])
            return types.SimpleNamespace(vertices=v + transl.unsqueeze(1), joints=j + transl.unsqueeze(1))

    smplx.create = lambda *a, batch_size=1, **k: SMPLXOverRefLBS(batch_size)

    cm = types.ModuleType('chamfer_pytorch.dist_chamfer')

    class chamferDist(nn.Module):
        def forward(s, a, b):
            return O.ChamferOracleFn.apply(a, b)

    cm.chamferDist = chamferDist
    import chamfer_pytorch
    sys.modules['chamfer_pytorch.dist_chamfer'] = cm
    chamfer_pytorch.dist_chamfer = cm
    return L


def resnet18():
    """torchvision==0.4.0 resnet18 architecture (BasicBlock [2,2,2,2]); only children()[1:6] are used (cvae.py:431-435)."""
    class BasicBlock(nn.Module):
        def __init__(s, i, o, stride=1):
            super().__init__()
            s.conv1 = nn.Conv2d(i, o, 3, stride, 1, bias=False)
            s.bn1 = nn.BatchNorm2d(o)
            s.relu = nn.ReLU(inplace=True)
            s.conv2 = nn.Conv2d(o, o, 3, 1, 1, bias=False)
            s.bn2 = nn.BatchNorm2d(o)
            s.downsample = None
            if stride != 1 or i != o:
                s.downsample = nn.Sequential(nn.Conv2d(i, o, 1, stride, bias=False), nn.BatchNorm2d(o))

        def forward(s, x):
            idt = x if s.downsample is None else s.downsample(x)
            out = s.relu(s.bn1(s.conv1(x)))
            out = s.bn2(s.conv2(out))
            return s.relu(out + idt)

    class ResNet(nn.Module):
        def __init__(s):
            super().__init__()
            s.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            s.bn1 = nn.BatchNorm2d(64)
            s.relu = nn.ReLU(inplace=True)
            s.maxpool = nn.MaxPool2d(3, 2, 1)
            s.layer1 = nn.Sequential(BasicBlock(64, 64), BasicBlock(64, 64))
            s.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128))
            s.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256))
            s.layer4 = nn.Sequential(BasicBlock(256, 512, 2), BasicBlock(512, 512))
            s.avgpool = nn.AdaptiveAvgPool2d((1, 1))
            s.fc = nn.Linear(512, 1000)

    return ResNet()


def T(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32)


def save(name, **arrs):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + '.npz')
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024))


# ------------------------------------------------------------------------------------------
def gen_rot_glue():
    from cvae import GeometryTransformer as GT, ContinousRotReprDecoder as CR
    rs = np.random.RandomState(101)
    x72 = rs.standard_normal((16, 72)).astype(np.float32)
    x72[:, 3:6] *= 1.2
    x72[0, 3:6] = [1e-5, -2e-5, 1e-5]           # Taylor branch of angle_axis_to_rotation_matrix
    x72[1, 3:6] = [3.1, 0.02, -0.01]            # near pi: quaternion branches with small w
    x72[2, 3:6] = [0.0, 2.5, 1.8]
    x75 = GT.convert_to_6D_rot(T(x72))
    x75_free = rs.standard_normal((16, 75)).astype(np.float32)   # non-orthonormal 6D: Gram-Schmidt does real work
    x72_back = GT.convert_to_3D_rot(x75)
    x72_free = GT.convert_to_3D_rot(T(x75_free))
    R_free = CR.decode(T(x75_free[:, 3:9]))
    cam_int = np.tile(np.array([[1060.5, 0, 951.3], [0, 1060.2, 536.7], [0, 0, 1]], np.float32)[None], (16, 1, 1))
    max_d = (np.abs(rs.standard_normal(16)) * 2 + 3).astype(np.float32)
    xt = x72.copy()
    xt[:, 2] = np.abs(xt[:, 2]) + 1.0
    xn = GT.normalize_global_T(T(xt), T(cam_int), T(max_d))
    xb = GT.recover_global_T(xn, T(cam_int), T(max_d))
    verts = rs.standard_normal((16, 50, 3)).astype(np.float32)
    cam_ext = synth.make_cam_ext(5, 16)
    vt = GT.verts_transform(T(verts), T(cam_ext))
    save('rot_glue', x72=x72, x75=x75, x75_free=x75_free, x72_back=x72_back, x72_free=x72_free, R_free=R_free,
         cam_int=cam_int, max_d=max_d, xt=xt, xn=xn, xb=xb, verts=verts, cam_ext=cam_ext, verts_t=vt)


def ref_vposer(sd_np):
    from human_body_prior.train.vposer_smpl import VPoser
    vp = VPoser(512, 32, [1, 21, 3])
    vp.load_state_dict({k: torch.tensor(v) for k, v in sd_np.items()})
    vp.eval()
    return vp


def gen_vposer(sd_np):
    vp = ref_vposer(sd_np)
    rs = np.random.RandomState(102)
    z = rs.standard_normal((8, 32)).astype(np.float32)
    with torch.no_grad():
        aa = vp.decode(T(z), output_type='aa').view(8, -1)
        mr = vp.decode(T(z), output_type='matrot').view(8, -1)
    save('vposer_decode', z=z, aa=aa, matrot=mr, seed=3)


def gen_lbs(L, data):
    m = O.SMPLXOracle(data)
    rs = np.random.RandomState(103)
    B = 4
    betas = rs.standard_normal((B, 20)).astype(np.float32)
    pose = (rs.standard_normal((B, 165)) * 0.4).astype(np.float32)
    pose[0, :] = 0.0                                   # rest pose: exercises the +1e-8 trick in batch_rodrigues
    with torch.no_grad():
        v, j = L.lbs(T(betas), T(pose), m.v_template.unsqueeze(0).repeat(B, 1, 1), m.shapedirs, m.posedirs,
                     m.J_regressor, m.parents, m.lbs_weights, num_joints=55)
        R = L.batch_rodrigues(T(pose).view(-1, 3))
    save('lbs', betas=betas, pose=pose, verts=v, joints=j, rodrigues=R, smplx_seed=7)


def gen_fitting(data, vposer_sd):
    B, m_pts, n_c, D = 4, 4096, 512, 32
    scene = synth.make_scene(seed=0, m=m_pts, D=D, n_contact=n_c)
    holder = {'verts': scene.verts.astype(np.float64)}
    SCENE_HOLDER.update(holder)
    import human_body_prior.tools.model_loader as ML
    ML.load_vposer = lambda *a, **k: (ref_vposer(vposer_sd), None)
    import fitting_proxe as FP
    FP.load_vposer = ML.load_vposer
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        paths = scene.write_prox_layout(tmp, 'S')
        bodies = synth.make_bodies(seed=11, B=B)
        bodies['cam_ext'] = synth.make_cam_ext(5, B)
        pkl = os.path.join(tmp, 'in.pkl')
        with open(pkl, 'wb') as f:
            pickle.dump(bodies, f)
        for ac in (True, False):
            # torch 1.2.0 (pinned) behaved as align_corners=True; the reference passes no argument
            orig = F.grid_sample
            FP.F.grid_sample = lambda *a, _o=orig, _ac=ac, **k: _o(*a, align_corners=_ac, **k)
            try:
                cfg = {'scene_verts_path': paths['scene_verts_path'], 'scene_sdf_path': paths['scene_sdf_path'],
                       'human_model_path': '', 'vposer_ckpt_path': '', 'init_lr_h': 0.1, 'num_iter': 5,
                       'batch_size': B, 'device': torch.device('cpu'), 'contact_part': synth.CONTACT_PARTS,
                       'contact_id_folder': paths['contact_id_folder'],
                       'body_segments_folder': paths['contact_id_folder'], 'verbose': False}
                lw = {'weight_loss_rec': 1, 'weight_loss_vposer': 0.01, 'weight_contact': 0.1, 'weight_collision': 0.5}
                torch.manual_seed(0)
                fop = FP.FittingOP(cfg, lw)
                # --- one cal_loss at the initial point (with the parameters perturbed off the input so l_rec != 0)
                xh, cam_ext, cam_int = FP.BodyParamParser.body_params_parse_fitting(bodies)
                xhr = FP.GeometryTransformer.convert_to_6D_rot(xh)
                rs = np.random.RandomState(104)
                fop.xhr_rec.data = xhr.clone() + T(rs.standard_normal((B, 75)) * 0.05)
                losses = fop.cal_loss(xhr, cam_ext)
                fop.optimizer.zero_grad()
                sum(losses).backward()
                tag = 'ac1' if ac else 'ac0'
                out['xhr_' + tag] = xhr.detach().numpy()
                out['xhr_rec0_' + tag] = fop.xhr_rec.detach().numpy().copy()
                out['loss0_' + tag] = np.array([float(l) for l in losses], np.float32)
                out['grad0_' + tag] = fop.xhr_rec.grad.detach().numpy().copy()
                # --- the 5-iteration loop exactly as shipped (fresh op => fresh Adam state)
                fop = FP.FittingOP(cfg, lw)
                rec = []
                _cal = fop.cal_loss

                def cal(xhr_, ce, _c=_cal, _r=rec):
                    ls = _c(xhr_, ce)
                    _r.append([float(l) for l in ls])
                    return ls
                fop.cal_loss = cal
                xh_rec = fop.fitting(pkl)
                out['traj_loss_' + tag] = np.array(rec, np.float32)
                out['traj_final_' + tag] = xh_rec.detach().numpy()
                out['traj_final_xhr_' + tag] = fop.xhr_rec.detach().numpy()
            finally:
                FP.F.grid_sample = orig
        # body verts at the perturbed initial point (camera frame), from the reference's own modules
        fop.xhr_rec.data = T(out['xhr_rec0_ac1'])
        xh_rec = FP.GeometryTransformer.convert_to_3D_rot(fop.xhr_rec)
        p = FP.BodyParamParser.body_params_encapsulate_batch(xh_rec)
        aa = fop.vposer.decode(p['body_pose_vp'], output_type='aa').view(B, -1)
        bp = {k: v for k, v in p.items() if k != 'body_pose_vp'}
        with torch.no_grad():
            v = fop.body_mesh_model(return_verts=True, body_pose=aa, **bp).vertices
            v = FP.GeometryTransformer.verts_transform(v, cam_ext)
        out['verts0'] = v.numpy()
        vid, _ = FP.GeometryTransformer.get_contact_id(paths['contact_id_folder'], synth.CONTACT_PARTS)
        out['contact_ids'] = np.asarray(vid, np.int64)
    out['cam_ext'] = bodies['cam_ext']
    save('fitting_proxe', B=B, m=m_pts, n_c=n_c, D=D, **out)


def gen_chamfer_known_answer():
    """chamfer_pytorch/test_chamfer.py:35-54: rand(4,100,3) clouds, expanded-form brute force, tolerance 1e-8."""
    rs = np.random.RandomState(105)
    p1 = rs.uniform(0, 1, (4, 100, 3)).astype(np.float32)
    p2 = rs.uniform(0, 1, (4, 100, 3)).astype(np.float32)
    x, y = T(p1), T(p2)
    xx = torch.bmm(x, x.transpose(2, 1))
    yy = torch.bmm(y, y.transpose(2, 1))
    zz = torch.bmm(x, y.transpose(2, 1))
    di = torch.arange(0, 100)
    rx = xx[:, di, di].unsqueeze(1).expand_as(xx)
    ry = yy[:, di, di].unsqueeze(1).expand_as(yy)
    P = rx.transpose(2, 1) + ry - 2 * zz
    save('chamfer_known_answer', p1=p1, p2=p2, mydist1=torch.min(P, 2)[0], mydist2=torch.min(P, 1)[0])


def gen_training(data, vposer_sd):
    """TrainOP.cal_loss of train_s1.py / train_s2.py (values + a few parameter gradients) on a synthetic batch."""
    import human_body_prior.tools.model_loader as ML
    ML.load_vposer = lambda *a, **k: (ref_vposer(vposer_sd), None)
    h5 = types.ModuleType('h5py')
    sys.modules['h5py'] = h5
    torch.cuda.get_device_name = lambda *a, **k: 'cpu'
    import train_s1 as TS1
    import train_s2 as TS2
    B, m_pts, n_c, D = 4, 2048, 256, 16
    scene = synth.make_scene(seed=2, m=m_pts, D=D, n_contact=n_c)
    inp = synth.make_cvae_inputs(13, B)
    bodies = synth.make_bodies(17, B)
    xh = synth.body_vector_72(bodies)
    xh[:, 2] = np.abs(xh[:, 2]) + 2.0
    cam_ext = synth.make_cam_ext(9, B)
    cam_int = bodies['cam_int']
    max_d = np.full(B, 6.0, np.float32)
    out = {'xh': xh, 'cam_ext': cam_ext, 'cam_int': cam_int, 'max_d': max_d}
    with tempfile.TemporaryDirectory() as tmp:
        paths = scene.write_prox_layout(tmp, 'S')
        os.makedirs(os.path.join(tmp, 'smplx'))
        np.savez(os.path.join(tmp, 'smplx', 'SMPLX_NEUTRAL.npz'), f=data.f)
        cfg = {'human_model_path': tmp, 'vposer_ckpt_path': '', 'scene_model_ckpt': None, 'init_lr_h': 1e-4, 'batch_size': B,
               'epoch': 100, 'loss_weight_anealing': True, 'device': torch.device('cpu'), 'save_dir': os.path.join(tmp, 'ckpt'),
               'contact_id_folder': paths['contact_id_folder'], 'contact_part': synth.CONTACT_PARTS, 'verbose': False,
               'use_cont_rot': True, 'resume_training': False}
        lw = {'weight_loss_rec_s': 1.0, 'weight_loss_rec_h': 1.0, 'weight_loss_vposer': 1e-3, 'weight_loss_kl': 1e-1,
              'weight_contact': 1e-1, 'weight_collision': 1e-1}
        args = dict(xs=T(inp['xs']), xh=T(xh), cam_ext=T(cam_ext), cam_int=T(cam_int), max_d=T(max_d),
                    scene_verts=T(scene.verts)[None].repeat(B, 1, 1), scene_face=None,
                    s_grid_min_batch=T(scene.grid_min)[None].repeat(B, 1), s_grid_max_batch=T(scene.grid_max)[None].repeat(B, 1),
                    s_grid_sdf_batch=T(scene.sdf)[None].repeat(B, 1, 1, 1))
        for name, MOD, seed in (('s1', TS1, 0), ('s2', TS2, 1)):
            orig = F.grid_sample
            MOD.F.grid_sample = lambda *a, _o=orig, **k: _o(*a, align_corners=True, **k)
            MOD.load_vposer = ML.load_vposer
            try:
                op = MOD.TrainOP(cfg, lw)
                shapes = {k: tuple(v.shape) for k, v in op.model_h.state_dict().items()}
                op.model_h.load_state_dict({k: torch.tensor(v) for k, v in synth.make_state_like(shapes, seed).items()})
                op.model_h.train()
                if name == 's1':
                    op.model_h._sampler = lambda mu, lv: T(inp['eps32']) * torch.exp(0.5 * lv) + mu
                else:
                    op.model_h.trans_vae.sampler = lambda mu, lv: T(inp['eps32']) * torch.exp(0.5 * lv) + mu
                    op.model_h.pose_vae.sampler = lambda mu, lv: T(inp['eps32b']) * torch.exp(0.5 * lv) + mu
                for ep in (10, 90):                       # before / after the 0.75*epoch gate of the scene losses
                    op.model_h.load_state_dict({k: torch.tensor(v) for k, v in synth.make_state_like(shapes, seed).items()})
                    op.model_h.zero_grad()
                    if name == 's1':
                        losses = op.cal_loss(ep=ep, **args)
                    else:
                        losses = op.cal_loss(eps_g=None, eps_l=None, ep=ep, **args)
                    sum(losses).backward()
                    out['%s_ep%d_losses' % (name, ep)] = np.array([float(l) for l in losses], np.float32)
                    sd = dict(op.model_h.named_parameters())
                    keys = ['linear_out.weight', 'resnet.0.weight', 'mu_enc.bias'] if name == 's1' else \
                        ['pose_vae.decode.3.weight', 'trans_vae.resnet.0.weight', 'trans_vae.decode.3.bias']
                    for k in keys:
                        g = sd[k].grad
                        out['%s_ep%d_grad_%s' % (name, ep, k)] = g.numpy().copy() if g.numel() < 20000 else g.numpy().reshape(-1)[:20000].copy()
            finally:
                MOD.F.grid_sample = orig
    save('training', B=B, m=m_pts, n_c=n_c, D=D, **out)


def gen_preproc():
    """TestOP.data_preprocessing of test_habitat_s2.py:75-149 (wide and tall images, both modalities)."""
    h5 = types.ModuleType('h5py')
    sys.modules.setdefault('h5py', h5)
    import test_habitat_s2 as TH
    me = types.SimpleNamespace(device=torch.device('cpu'))
    rs = np.random.RandomState(77)
    out = {}
    for tag, (H, W) in (('wide', (96, 160)), ('tall', (150, 90)), ('square', (64, 64))):
        depth = (rs.uniform(0.3, 9.0, (H, W))).astype(np.float32)
        seg = rs.randint(0, 60, (H, W)).astype(np.float32)
        out[tag + '_depth_in'], out[tag + '_seg_in'] = depth, seg
        for mod, arr in (('depth', depth), ('seg', seg)):
            c, f, mx = TH.TestOP.data_preprocessing(me, T(arr.copy()), mod, target_domain_size=[128, 128])
            out['%s_%s_canvas' % (tag, mod)] = c.numpy()
            out['%s_%s_factor' % (tag, mod)] = np.float32(f)
            out['%s_%s_max' % (tag, mod)] = np.float32(mx)
    save('preproc', **out)


SCENE_HOLDER = {}


def main(which):
    warnings.filterwarnings('ignore')
    torch.set_num_threads(8)
    data = synth.make_smplx(7)
    vsd = synth.make_vposer_state(3)
    L = install_stubs(SCENE_HOLDER, data)
    manifest = {
        'smplx_seed7': {k: synth.checksum(getattr(data, k)) for k in ('v_template', 'shapedirs', 'posedirs',
                                                                      'J_regressor', 'weights')},
        'vposer_seed3': {k: synth.checksum(v) for k, v in vsd.items() if 'dec' in k},
        'scene_seed0_m4096_D32': {'verts': synth.checksum(synth.make_scene(0, 4096, 32, 512).verts),
                                  'sdf': synth.checksum(synth.make_scene(0, 4096, 32, 512).sdf)},
        'torch': torch.__version__, 'numpy': np.__version__,
    }
    if 'rot' in which:
        gen_rot_glue()
    if 'vposer' in which:
        gen_vposer(vsd)
    if 'lbs' in which:
        gen_lbs(L, data)
    if 'chamfer' in which:
        gen_chamfer_known_answer()
    if 'fitting' in which:
        gen_fitting(data, vsd)
    if 'preproc' in which:
        gen_preproc()
    if 'training' in which:
        gen_training(data, vsd)
    if 'cvae' in which:
        import make_golden_cvae
        make_golden_cvae.gen(save)
    with open(os.path.join(GOLD, 'manifest.json'), 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main(sys.argv[1:] or ['rot', 'vposer', 'lbs', 'chamfer', 'fitting'])
