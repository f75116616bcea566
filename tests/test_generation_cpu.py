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


def test_batch_generator_test_mat_files(tmp_path):
    """BatchGeneratorTest over scipy .mat snapshots (batch_gen_hdf5.py:726-797)."""
    import scipy.io as sio
    from psi_release_amd import batch_gen
    rs = np.random.RandomState(1)
    bodies = synth.make_bodies(3, 1)
    ext = synth.make_cam_ext(2, 1)[0]
    sio.savemat(str(tmp_path / 'rec_000000.mat'), {
        'depth': rs.uniform(0.5, 8, (90, 160)).astype(np.float32), 'seg': rs.randint(0, 50, (90, 160)).astype(np.float32),
        'cam': {'intrinsic': bodies['cam_int'][0], 'extrinsic': ext},
        'body': {k: bodies[k] for k in ('transl', 'global_orient', 'betas', 'body_pose', 'left_hand_pose', 'right_hand_pose')}})
    bg = batch_gen.BatchGeneratorTest(str(tmp_path), 'cpu')
    d, s, md, ci, ce, body = bg.next_batch(2)
    assert d.shape == (2, 1, 128, 128) and s.shape == (2, 1, 128, 128) and md.shape == (2,) and ci.shape == (2, 3, 3)
    assert ce.shape == (2, 4, 4) and body.shape == (2, 72)
    assert np.allclose(ce[0].numpy() @ ext, np.eye(4), atol=1e-5)          # cam_ext is the INVERSE of the stored extrinsic
    assert float(md[0]) == 6.0                                             # depth clipped at 6 m


def test_diversity_scores():
    from psi_release_amd import evaluation
    rs = np.random.RandomState(0)
    x = np.concatenate([rs.standard_normal((50, 72)) * 0.05 + c for c in (0.0, 3.0, -3.0)])
    ent, dist = evaluation.diversity_scores(x, n_clusters=3)
    assert abs(ent - np.log(3)) < 1e-6 and dist < 1.0
