"""VPoser v1.0 pose prior: state_dict-compatible module and loader.

Reference: human_body_prior/train/vposer_smpl.py:66-171 (class VPoser) and
human_body_prior/tools/model_loader.py:26-72 (expid2model / load_vposer).  PSI only ever calls
``vposer.decode(z, output_type='aa')`` on a pre-trained, ``.eval()`` model (fitting_proxe.py:115-116);
the encoder PARAMETERS are kept so that ``load_state_dict`` of a real ``vposer_v1_0`` snapshot is strict; the encoder's methods
(``encode`` / ``forward`` / ``sample_poses``) and the trainer (vposer_smpl.py:174-479) are not on the path and are not provided.
"""
from __future__ import annotations

import configparser
import glob
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .geometry import ContinousRotReprDecoder, rotation_matrix_to_angle_axis


class VPoser(nn.Module):
    def __init__(self, num_neurons=512, latentD=32, data_shape=(1, 21, 3), use_cont_repr=True):
        super().__init__()
        self.latentD = latentD
        self.use_cont_repr = use_cont_repr
        n_features = int(np.prod(data_shape))
        self.num_joints = data_shape[1]
        self.bodyprior_enc_bn1 = nn.BatchNorm1d(n_features)
        self.bodyprior_enc_fc1 = nn.Linear(n_features, num_neurons)
        self.bodyprior_enc_bn2 = nn.BatchNorm1d(num_neurons)
        self.bodyprior_enc_fc2 = nn.Linear(num_neurons, num_neurons)
        self.bodyprior_enc_mu = nn.Linear(num_neurons, latentD)
        self.bodyprior_enc_logvar = nn.Linear(num_neurons, latentD)
        self.dropout = nn.Dropout(p=.1, inplace=False)
        self.bodyprior_dec_fc1 = nn.Linear(latentD, num_neurons)
        self.bodyprior_dec_fc2 = nn.Linear(num_neurons, num_neurons)
        if use_cont_repr:
            self.rot_decoder = ContinousRotReprDecoder()
        self.bodyprior_dec_out = nn.Linear(num_neurons, self.num_joints * 6)

    def decode(self, Zin, output_type='matrot'):
        assert output_type in ['matrot', 'aa']
        x = F.leaky_relu(self.bodyprior_dec_fc1(Zin), negative_slope=.2)
        x = self.dropout(x)
        x = F.leaky_relu(self.bodyprior_dec_fc2(x), negative_slope=.2)
        x = self.bodyprior_dec_out(x)
        x = self.rot_decoder(x) if self.use_cont_repr else torch.tanh(x)
        x = x.view([-1, 1, self.num_joints, 9])
        return VPoser.matrot2aa(x) if output_type == 'aa' else x

    @staticmethod
    def matrot2aa(pose_matrot):
        bs = pose_matrot.size(0)
        homogen = F.pad(pose_matrot.view(-1, 3, 3), [0, 1])
        return rotation_matrix_to_angle_axis(homogen).view(bs, 1, -1, 3).contiguous()


def _read_settings(expr_dir):
    """Hyper-parameters from ``{expr_dir}/*.ini`` (model_loader.py:34-39; vposer_smpl_defaults.ini:35-37)."""
    ps = {'num_neurons': 512, 'latentD': 32, 'data_shape': [1, 21, 3]}
    inis = glob.glob(os.path.join(expr_dir, '*.ini'))
    if inis:
        cp = configparser.ConfigParser()
        cp.read(inis[0])
        for sec in cp.sections():
            for k in ('num_neurons', 'latentD', 'latentd'):
                if cp.has_option(sec, k):
                    ps['latentD' if k.lower() == 'latentd' else k] = int(cp.get(sec, k))
            if cp.has_option(sec, 'data_shape'):
                ps['data_shape'] = [int(x) for x in cp.get(sec, 'data_shape').strip('[]() ').split(',')]
    return ps


def load_vposer(expr_dir, vp_model='snapshot'):
    """``load_vposer(expr_dir, vp_model='snapshot') -> (vposer.eval(), ps)`` (model_loader.py:43-72).

    ``expr_dir`` is a VPoser experiment folder (``snapshots/*.pt`` newest by mtime + ``*.ini``).  For tests and the
    synthetic bench it may also be a state_dict (dict of arrays/tensors) used directly."""
    if isinstance(expr_dir, dict):
        ps = {'num_neurons': 512, 'latentD': 32, 'data_shape': [1, 21, 3]}
        sd = {k: torch.as_tensor(np.asarray(v)) for k, v in expr_dir.items()}
    else:
        if not os.path.exists(expr_dir):
            raise ValueError('Could not find the experiment directory: %s' % expr_dir)
        snaps = sorted(glob.glob(os.path.join(expr_dir, 'snapshots', '*.pt')), key=os.path.getmtime)
        if not snaps:
            raise ValueError('no snapshots/*.pt under %s' % expr_dir)
        ps = _read_settings(expr_dir)
        sd = torch.load(snaps[-1], map_location='cpu')
        print('Found Trained Model: %s' % snaps[-1])
    vp = VPoser(num_neurons=ps['num_neurons'], latentD=ps['latentD'], data_shape=ps['data_shape'])
    vp.load_state_dict(sd)
    vp.eval()
    return vp, ps
