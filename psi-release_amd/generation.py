"""Generation drivers: sample bodies for a scene view from a trained CVAE and write the ``body_gen_*.pkl`` files that
the fitting stage consumes.

Reference: source/test_habitat_s1.py / test_habitat_s2.py (``TestOP.data_preprocessing`` :75-149, ``TestOP.test``
:155-229) and source/test_proxe_s1.py / test_proxe_s2.py (``TestOP.test`` :74-134).  Forward-only use of the CVAEs
(a19/a20): ``model_h.sample`` -> ``convert_to_3D_rot`` -> ``recover_global_T`` -> ``body_params_encapsulate`` -> pkl
(schema cvae.py:226-232 + ``cam_ext``, ``cam_int``).  Quirk kept: the Habitat drivers preprocess the semantic map with
modality 'depth' (clip at 6.0, not 41; test_habitat_s2.py:192-193) — that is what the shipped checkpoints saw.
"""
from __future__ import annotations

import glob
import os
import pickle

import numpy as np
import torch
import torch.nn.functional as F

from .geometry import BodyParamParser, GeometryTransformer
from .models import HumanCVAES1, HumanCVAES2


def data_preprocessing(img, modality, target_domain_size=(128, 128), device=None):
    """test_habitat_s2.py:75-149: clip, scale to [-1,1] by the max, bilinear-resize into a centred canvas.

    img: [H,W] tensor (modified in place like the reference); returns (canvas [1,1,h,w], factor, max_val)."""
    device = img.device if device is None else device
    th, tw = int(target_domain_size[0]), int(target_domain_size[1])
    H, W = img.shape
    canvas = torch.zeros([1, 1, th, tw], dtype=torch.float32, device=device)
    if modality == 'depth':
        img[img > 6.0] = 6.0
    if modality == 'seg':
        img[img > 41] = 41
    max_val = torch.max(img)
    _img = (2 * img / max_val - 1.0).view(1, 1, H, W)
    if H >= W:
        factor = float(th) / H
        target_width = int(W * factor) // 2 * 2
        res = F.interpolate(_img, size=[th, target_width], mode='bilinear', align_corners=False)
        lower, upper = (tw // 2) - (target_width // 2), (tw // 2) + (target_width // 2)
        canvas[:, :, :, lower:upper] = res
    else:
        factor = float(tw) / W
        target_height = int(factor * H) // 2 * 2
        res = F.interpolate(_img, size=[target_height, tw], mode='bilinear', align_corners=False)
        lower, upper = (th // 2) - (target_height // 2), (th // 2) + (target_height // 2)
        canvas[:, :, lower:upper, :] = res
    return canvas, factor, max_val


def newest_checkpoint(ckpt_dir):
    ckp_list = sorted(glob.glob(os.path.join(ckpt_dir, 'epoch-*.ckp')), key=os.path.getmtime)   # test_proxe_s1.py:83-88
    if not ckp_list:
        raise FileNotFoundError('no epoch-*.ckp under %s' % ckpt_dir)
    return ckp_list[-1]


class TestOP:
    """Sampler for one trained model.  testconfig keys as the reference: ckpt_dir, device, n_samples, use_cont_rot,
    outdir / output_dir, test_data_path (+ ``stage`` = 's1' | 's2', default 's2')."""

    def __init__(self, testconfig):
        self.stage = 's2'
        self.use_cont_rot = True
        self.autocast_bf16 = False
        for key, val in testconfig.items():
            setattr(self, key, val)
        self.device = torch.device(self.device)
        n_dim_body = 72 + 3 if self.use_cont_rot else 72
        self.model_h_latentD = 256
        if self.stage == 's1':
            self.model_h = HumanCVAES1(latentD=self.model_h_latentD, n_dim_body=n_dim_body, test=True,
                                       autocast_bf16=self.autocast_bf16)
        else:
            self.model_h = HumanCVAES2(latentD_g=self.model_h_latentD, latentD_l=self.model_h_latentD, n_dim_body=n_dim_body,
                                       test=True, autocast_bf16=self.autocast_bf16)
        self.model_h.eval().to(self.device)
        self._loaded = False
        # optional: callable(view_index, n_samples) -> (eps_g [n,32], eps_l [n,32]) (S2) or eps [n,32] (S1).  The reference draws the
        # latent with the global CPU generator inside the model (net_layers.py:97-98,197-198); injecting it makes a run reproducible
        # and lets the tests replay the exact latents a reference run consumed.
        self.latent_source = getattr(self, 'latent_source', None)

    def load(self, state_dict=None):
        if state_dict is None:
            ckp_path = newest_checkpoint(self.ckpt_dir)
            print('[INFO] load checkpoints: ' + ckp_path)
            state_dict = torch.load(ckp_path, map_location=self.device)['model_h_state_dict']
        self.model_h.load_state_dict(state_dict)
        self._loaded = True

    @torch.no_grad()
    def sample_view(self, depth, seg, cam_int, cam_ext, max_d, n_samples=None, latents=None, repeat_cams=True):
        """depth/seg [1,1,128,128] (preprocessed), cam_int [1,3,3], cam_ext [1,4,4], max_d [1] -> list of pkl dicts.
        ``repeat_cams``: the Habitat drivers store the cameras repeated to [n,...] in every pkl (test_habitat_s2.py:216-217), the
        PROX-E drivers store the single [1,4,4] / [1,3,3] (test_proxe_s1.py:120-128)."""
        n = n_samples or self.n_samples
        # the reference feeds the trunk n copies of the view (xs.repeat(n_samples, 1, 1, 1), test_habitat_s2.py:192); the model is in eval
        # mode, so the one view is encoded ONCE and its feature row shared by the n samples (models._SceneCond._scene_feature)
        xs_1 = torch.cat([depth, seg], dim=1)
        cam_int_b, cam_ext_b, max_d_b = cam_int.repeat(n, 1, 1), cam_ext.repeat(n, 1, 1), max_d.view(1).repeat(n)
        if latents is None:
            xhnr_gen = self.model_h.sample(xs_1, rows=n)
        elif self.stage == 's1':
            xhnr_gen = self.model_h.sample(xs_1, eps=latents, rows=n)
        else:
            xhnr_gen = self.model_h.sample(xs_1, eps_g=latents[0], eps_l=latents[1], use_eps=True, rows=n)
        xhn_gen = GeometryTransformer.convert_to_3D_rot(xhnr_gen)
        xh_gen = GeometryTransformer.recover_global_T(xhn_gen, cam_int_b, max_d_b)
        body_param_list = BodyParamParser.body_params_encapsulate(xh_gen)
        for body_param in body_param_list:
            body_param['cam_ext'] = (cam_ext_b if repeat_cams else cam_ext).detach().cpu().numpy()
            body_param['cam_int'] = (cam_int_b if repeat_cams else cam_int).detach().cpu().numpy()
        return body_param_list

    @staticmethod
    def write(body_param_list, outdir, first_index=0):
        os.makedirs(outdir, exist_ok=True)
        print('[INFO] save results to: ' + outdir)
        for jj, body_param in enumerate(body_param_list):
            with open(os.path.join(outdir, 'body_gen_{:06d}.pkl'.format(first_index + jj)), 'wb') as f:
                pickle.dump(body_param, f)

    def test_habitat(self):
        """test_habitat_s2.py:155-229: every ``cam_*`` file of ``test_data_path`` (+ matching ``depth_*`` / ``seg_*`` .npy)."""
        if not self._loaded:
            self.load()
        for ii, cam_file in enumerate(sorted(glob.glob(self.test_data_path + '/cam_*'))):
            cam_params = np.load(cam_file, allow_pickle=True, encoding='latin1').item()
            t = lambda a: torch.tensor(a, dtype=torch.float32, device=self.device)
            depth0, seg0 = t(np.load(cam_file.replace('cam', 'depth'))), t(np.load(cam_file.replace('cam', 'seg')))
            depth, _, max_d = data_preprocessing(depth0, 'depth', [128, 128])
            seg, _, _ = data_preprocessing(seg0, 'depth', [128, 128])          # sic: 'depth' (see module docstring)
            lat = self.latent_source(ii, self.n_samples) if self.latent_source is not None else None
            bodies = self.sample_view(depth, seg, t(cam_params['cam_int']).unsqueeze(0), t(cam_params['cam_ext']).unsqueeze(0), max_d, latents=lat)
            self.write(bodies, self.outdir, self.n_samples * ii)

    def test_proxe(self, test_data, scene_name):
        """test_proxe_s1.py:74-134: ``test_data`` = (depth, seg, max_d, cam_int, cam_ext) of one snapshot; files start at 900."""
        if not self._loaded:
            self.load()
        depth, seg, max_d, cam_int, cam_ext = test_data[:5]
        lat = self.latent_source(0, self.n_samples) if self.latent_source is not None else None
        bodies = self.sample_view(depth, seg, cam_int, cam_ext, max_d, latents=lat, repeat_cams=False)
        self.write(bodies, os.path.join(self.output_dir, scene_name), 900)
