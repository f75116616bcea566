"""BASELINE.json configs[3] and configs[4] at their per-rank shapes against the ORACLE (not against another engine of this package).

configs[3]  "fitting_proxe.py batch=256 sharded 8xMI355X": every rank holds 32 bodies at the full scene size (n_c=2048, m=32768,
            256^3 SDF) and the loss normalisers are global: 2 ranks x 32 bodies against ``FittingOracle`` on the GLOBAL batch of 64 —
            that test spawns processes and lives in tests/test_configs_dp_gpu.py (collected after every single-process parity module);
            the comparison helpers are here.
configs[4]  "fitting_habitat.py MP3D-R sweep, batch=512 over 8 GPUs" = 64 bodies per GPU, contact constant 1.0
            (fitting_habitat.py:141), camera pre-multiplied by diag(1,-1,-1,1) and shared by the batch (fitting_habitat.py:179-184),
            full scene size: ``FittingOPHabitat`` at B=64 against the oracle with the same constants.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import psi_oracle as O
from conftest import ROOT, rel_err
from psi_release_amd import fitting, geometry, synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'
LOSS = {'weight_loss_rec': 1, 'weight_loss_vposer': 0.01, 'weight_contact': 0.1, 'weight_collision': 0.5}
GT = geometry.GeometryTransformer
M, NC, D, ITERS = 32768, 2048, 256, 3


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _cfg(smplx_data, vposer_sd, scene, B, engine='fused'):
    return {'scene_verts_path': None, 'scene_sdf_path': None, 'human_model_path': None, 'vposer_ckpt_path': None, 'init_lr_h': 0.1,
            'num_iter': ITERS, 'batch_size': B, 'device': torch.device(DEV, 0), 'contact_part': synth.CONTACT_PARTS, 'contact_id_folder': None,
            'verbose': False, 'smplx_data': smplx_data, 'vposer_state': vposer_sd, 'scene': scene, 'engine': engine}


def _oracle(smplx_data, vposer_sd, scene, B, **kw):
    O.set_threads(min(16, os.cpu_count() or 1))
    return O.FittingOracle(O.SMPLXOracle(smplx_data), vposer_sd, scene.verts, scene.sdf, scene.grid_min, scene.grid_max,
                           synth.contact_ids_from_parts(scene.contact_parts), B, **kw)


def _first_moment(op):
    """Adam's first moment after ONE step = (1 - beta1) * gradient of the first iteration, [B,75]."""
    if op.engine == 'fused':
        return op._fused.buffer('adam_m', (op.batch_size, 75)).cpu().numpy()
    return op.optimizer.state[op.xhr_rec]['exp_avg'].detach().cpu().numpy()


def _run(op, bodies):
    runner = op.make_step_runner(bodies)
    losses, m1 = [], None
    for it in range(ITERS):
        runner.step()
        losses.append(runner.last_losses())
        if it == 0:
            m1 = _first_moment(op)
    runner.finish()
    return GT.convert_to_3D_rot(op.xhr_rec).detach().cpu().numpy(), np.asarray(losses), m1


def _oracle_run(fo, bodies, cam):
    xhr = O.convert_to_6d_rot(torch.as_tensor(synth.body_vector_72(bodies), dtype=torch.float32))
    cam = torch.as_tensor(cam, dtype=torch.float32)
    fo.xhr_rec.data = xhr.clone()
    losses, m1 = [], None
    for it in range(ITERS):                                  # FittingOracle.fitting (fitting_proxe.py:177-189), one step at a time
        fo.optimizer.zero_grad()
        ls = fo.cal_loss(xhr, cam)
        losses.append([float(l.detach()) for l in ls])
        sum(ls).backward()
        fo.optimizer.step()
        if it == 0:
            m1 = fo.optimizer.state[fo.xhr_rec]['exp_avg'].detach().numpy().copy()
    return O.convert_to_3d_rot(fo.xhr_rec).detach().numpy(), np.asarray(losses), m1


def _check(gpu, ref):
    """(x after ITERS steps, losses per iteration, first Adam moment) of the product against the oracle.

    What is compared exactly and what is not.  With 64 bodies in the global batch every mean-type loss is divided by 64 x (75 | 32 |
    n_c) and the gradient entries are of order 1e-5, one in a hundred below 1e-7.  Adam's step is lr * m / (sqrt(v) + 1e-8): where
    |g| is within a few 1e-8 of zero the step reacts to absolute differences of 1e-9 — fp32 summation order — with changes of 1e-3 to
    2 * lr.  That is a property of the reference's optimiser at this batch size (its own CPU and CUDA runs differ the same way), not of
    an implementation, so the comparison is made where it is well-posed:
      * the GRADIENT of the first iteration (read back from Adam's first moment), every entry, to 1e-4 of the largest entry — except
        that up to three bodies may carry ONE vertex whose SDF value is within fp32 rounding of zero and is masked differently
        (sdf < 0, fitting_proxe.py:155; among 670 000 vertices about one per iteration is): their gradient then differs by that
        vertex's share, bounded here by 2 % of the body's largest entry;
      * the loss values of every iteration (1e-5; the last one 2e-4: it is evaluated after two such steps);
      * the parameters after ITERS steps: median error below 1e-4 and at least 90 % of the entries within 1e-3."""
    (x_gpu, l_gpu, m_gpu), (x_ref, l_ref, m_ref) = gpu, ref
    g_gpu, g_ref = m_gpu / 0.1, m_ref / 0.1
    gerr = np.abs(g_gpu - g_ref).max(axis=1)
    loose = gerr > 1e-4 * np.abs(g_ref).max()
    assert loose.sum() <= 3, (int(loose.sum()), np.sort(gerr)[-5:], np.abs(g_ref).max())
    assert np.all(gerr[loose] <= 0.02 * np.abs(g_ref[loose]).max(axis=1)), (gerr[loose], np.abs(g_ref[loose]).max(axis=1))
    assert np.abs(l_gpu[:2] - l_ref[:2]).max() < 1e-5, (l_gpu, l_ref)
    assert np.abs(l_gpu[2:] - l_ref[2:]).max() < 2e-4, (l_gpu, l_ref)
    # parameters after ITERS Adam steps: the typical entry agrees to 1e-4; entries whose gradient passed near zero in one of the steps
    # carry up to a fraction of lr (see above) — they bound the tail, the gradient check above is the parity statement
    err = np.abs(x_gpu - x_ref)
    assert np.median(err) < 1e-4, float(np.median(err))
    assert np.mean(err < 1e-3) > 0.9, float(np.mean(err < 1e-3))


@pytest.mark.parametrize('engine', ['fused', 'modular'])
def test_configs4_habitat_64_bodies_full_size_vs_oracle(smplx_data, vposer_sd, engine):
    B = 64
    scene = synth.make_scene(4, M, D, NC)
    bodies = synth.make_bodies(17, B)
    bodies['cam_ext'] = synth.make_cam_ext(2, 1)                  # one camera per view (test_habitat_s2.py writes one cam_ext per body file)
    op = fitting.FittingOPHabitat(_cfg(smplx_data, vposer_sd, scene, B, engine), dict(LOSS))
    gpu = _run(op, dict(bodies))
    fo = _oracle(smplx_data, vposer_sd, scene, B, contact_const=1.0)                          # fitting_habitat.py:141
    cam = bodies['cam_ext'][:1] @ np.diag([1.0, -1.0, -1.0, 1.0]).astype(np.float32)          # fitting_habitat.py:179-184
    _check(gpu, _oracle_run(fo, bodies, np.repeat(cam, B, axis=0)))
