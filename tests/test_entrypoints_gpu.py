"""End-to-end runs of the entry-point scripts (reference CLIs) on synthetic stand-in assets, on the GPU."""
import glob
import os
import pickle
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
SRC = os.path.join(ROOT, 'psi-release_amd', 'source')


def run(args, timeout=600):
    r = subprocess.run([sys.executable] + args, capture_output=True, text=True, timeout=timeout, cwd=SRC)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_fitting_proxe_script(tmp_path):
    fit = str(tmp_path / 'fit')
    out = run([os.path.join(SRC, 'fitting_proxe.py'), fit, '--synthetic', str(tmp_path / 'syn'), '--scenes', 'MPH16', '--num_iter', '5',
               '--batch_size', '2', '--verbose'])
    files = sorted(glob.glob(os.path.join(fit, 'MPH16', 'body_gen_*.pkl')))
    assert len(files) == 2
    assert out.count('[INFO][fitting] iter=') == 10 and 'l_collision=' in out
    with open(files[0], 'rb') as f:
        b = pickle.load(f)
    assert set(b.keys()) == {'transl', 'global_orient', 'betas', 'body_pose', 'left_hand_pose', 'right_hand_pose', 'cam_ext', 'cam_int'}
    # idempotent: a second run finds the outputs and does nothing (fitting_proxe.py:259-260)
    out2 = run([os.path.join(SRC, 'fitting_proxe.py'), fit, '--synthetic', str(tmp_path / 'syn'), '--scenes', 'MPH16', '--num_iter', '5',
                '--batch_size', '2'])
    assert 'save results' not in out2


def test_fitting_habitat_script(tmp_path):
    fit = str(tmp_path / 'fit')
    run([os.path.join(SRC, 'fitting_habitat.py'), fit, '--synthetic', str(tmp_path / 'syn'), '--scenes', 'roomA', '--num_iter', '3'])
    assert len(glob.glob(os.path.join(fit, 'roomA', 'body_gen_*.pkl'))) == 2


@pytest.mark.parametrize('stage', ['s1', 's2'])
def test_train_script_and_generation(tmp_path, stage):
    save = str(tmp_path / 'ckpt')
    out = run([os.path.join(SRC, 'train_%s.py' % stage), '--save_dir', save, '--batch_size', '4', '--num_epoch', '10', '--synthetic', '8',
               '--lr_h', '0.0003'])
    assert '---in [epoch 10]:' in out and 'collision=' in out
    ck = glob.glob(os.path.join(save, 'epoch-*.ckp'))
    assert len(ck) == 1
    sd = torch.load(ck[0], map_location='cpu')
    assert sd['epoch'] == 10 and 'model_h_state_dict' in sd and 'optimizer_h_state_dict' in sd
