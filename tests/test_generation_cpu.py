"""Generation driver pieces that are plain PyTorch (device-agnostic): data_preprocessing vs the reference's own
TestOP.data_preprocessing (tests/golden/preproc.npz), sampling + pkl schema, newest-checkpoint selection."""
import os
import pickle
import time

import numpy as np
import torch

from conftest import golden
from psi_release_amd import generation, synth

T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)


def test_data_preprocessing_golden():
    g = golden('preproc')
    for tag in ('wide', 'tall', 'square'):
        for mod in ('depth', 'seg'):
            c, f, mx = generation.data_preprocessing(T(g['%s_%s_in' % (tag, mod)].copy()), mod, [128, 128])
            assert np.abs(c.numpy() - g['%s_%s_canvas' % (tag, mod)]).max() < 1e-6
            assert abs(f - float(g['%s_%s_factor' % (tag, mod)])) < 1e-7 and abs(float(mx) - float(g['%s_%s_max' % (tag, mod)])) < 1e-6


def test_sampling_pkl_schema_and_newest_checkpoint(tmp_path):
    cfg = {'ckpt_dir': str(tmp_path / 'ck'), 'device': 'cpu', 'n_samples': 3, 'use_cont_rot': True, 'stage': 's1',
           'outdir': str(tmp_path / 'out')}
    op = generation.TestOP(cfg)
    os.makedirs(cfg['ckpt_dir'])
    shapes = {k: tuple(v.shape) for k, v in op.model_h.state_dict().items()}
    for ep, seed in ((10, 0), (20, 5)):
        sd = {k: torch.tensor(v) for k, v in synth.make_state_like(shapes, seed).items()}
        torch.save({'epoch': ep, 'model_h_state_dict': sd, 'optimizer_h_state_dict': {}}, os.path.join(cfg['ckpt_dir'], 'epoch-%06d.ckp' % ep))
        time.sleep(0.05)
    assert generation.newest_checkpoint(cfg['ckpt_dir']).endswith('epoch-000020.ckp')
    op.load()
    rs = np.random.RandomState(0)
    depth, _, max_d = generation.data_preprocessing(T(rs.uniform(0.5, 5, (96, 160))), 'depth')
    seg, _, _ = generation.data_preprocessing(T(rs.randint(0, 40, (96, 160)).astype(np.float32)), 'depth')
    cam_int = T([[1000., 0, 960.], [0, 1000., 540.], [0, 0, 1.]])[None]
    bodies = op.sample_view(depth, seg, cam_int, torch.eye(4)[None], max_d)
    assert len(bodies) == 3
    op.write(bodies, cfg['outdir'], 6)
    with open(os.path.join(cfg['outdir'], 'body_gen_000007.pkl'), 'rb') as f:
        b = pickle.load(f)
    assert set(b.keys()) == {'transl', 'global_orient', 'betas', 'body_pose', 'left_hand_pose', 'right_hand_pose', 'cam_ext', 'cam_int'}
    assert b['transl'].shape == (1, 3) and b['body_pose'].shape == (1, 32) and b['cam_ext'].shape == (3, 4, 4)
