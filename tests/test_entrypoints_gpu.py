"""End-to-end runs of the entry-point scripts (reference CLIs) on synthetic stand-in assets, on the GPU."""
import glob
import os
import pickle
import shutil
import subprocess
import sys
import tempfile

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


def _torchrun(args, nproc, env_extra, timeout=900):
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, **env_extra)
    logdir = tempfile.mkdtemp(prefix='psi_torchrun_')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
                        '--master-port', str(port), '--tee', '3', '--log-dir', logdir] + args,
                       capture_output=True, text=True, timeout=timeout, cwd=SRC, env=env)
    if r.returncode != 0:          # every rank's own stderr (the launcher's output only carries the tail of the first failure)
        per_rank = []
        for fn in sorted(glob.glob(os.path.join(logdir, '**', 'stderr.log'), recursive=True)):
            with open(fn, errors='replace') as f:
                per_rank.append('---- %s ----\n%s' % (os.path.relpath(fn, logdir), f.read()[-3000:]))
        raise AssertionError('torchrun exit %d\n%s\n==== launcher ====\n%s' % (r.returncode, '\n'.join(per_rank), r.stderr[-3000:]))
    shutil.rmtree(logdir, ignore_errors=True)
    return r.stdout


def _load(fn):
    with open(fn, 'rb') as f:
        return pickle.load(f)


@pytest.mark.parametrize('script', ['fitting_proxe.py', 'fitting_habitat.py'])
def test_fitting_scripts_rank_sharded(tmp_path, script):
    """The file loop of the fitting entry points on 2 ranks (gloo, both on the one GPU of the test box):
    --shard files: rank r fits files r, r+2, ... as independent problems -> identical to the single-process outputs;
    --shard rows : every file's batch is split over the ranks with the global loss normalisers (one all-reduce per iteration)
                   -> the gathered rows equal the single-process batch fit."""
    import numpy as np
    common = ['--synthetic', str(tmp_path / 'syn'), '--scenes', 'S0', '--num_iter', '4', '--batch_size', '2', '--save_all_rows', '--reset_optimizer']
    one, files_d, rows_d = str(tmp_path / 'one'), str(tmp_path / 'files'), str(tmp_path / 'rows')
    run([os.path.join(SRC, script), one] + common)
    _torchrun([os.path.join(SRC, script), files_d] + common + ['--shard', 'files'], 2, {'PSI_DIST_BACKEND': 'gloo'})
    _torchrun([os.path.join(SRC, script), rows_d] + common + ['--shard', 'rows'], 2, {'PSI_DIST_BACKEND': 'gloo'})
    ref = sorted(glob.glob(os.path.join(one, 'S0', 'body_gen_*.pkl')))
    assert len(ref) == 2
    for fn in ref:
        a = _load(fn)
        assert a['transl'].shape == (2, 3)                     # --save_all_rows: both rows of the batch
        for other, tol in ((files_d, 0.0), (rows_d, 5e-5)):
            b = _load(os.path.join(other, 'S0', os.path.basename(fn)))
            assert list(a.keys()) == list(b.keys())
            for k in a:
                assert a[k].shape == b[k].shape, (k, a[k].shape, b[k].shape)
                assert np.abs(a[k] - b[k]).max() <= tol, (other, k, np.abs(a[k] - b[k]).max())
