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

import arbiter
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


def _run(op, bodies):
    """ITERS single iterations of the product with the state around each of them (tests/arbiter.py)."""
    return arbiter.gpu_trace(op, bodies, ITERS)


def _check(trace, smplx_data, vposer_sd, scene, bodies, cam, name='configs', **oracle_kw):
    """Every iteration of the product against the oracle in fp32 and in fp64 (the arbiter) AT THE PRODUCT'S OWN STATE: loss values, the
    gradient and Adam's update, each within K_NOISE x the fp32 oracle's own distance from the arbiter — tests/arbiter.py states the bounds
    and the explicit rule for vertices whose SDF value is within 1e-6 of zero.  No literal tolerance looser than that is used."""
    B = trace[0]['x0'].shape[0]
    O.set_threads(min(16, os.cpu_count() or 1))
    make = lambda dt: O.FittingOracle(O.SMPLXOracle(smplx_data, dtype=dt), vposer_sd, scene.verts, scene.sdf, scene.grid_min, scene.grid_max,
                                      synth.contact_ids_from_parts(scene.contact_parts), B, **oracle_kw)
    report = arbiter.check_trace(trace, make, np.asarray(cam, np.float64))
    arbiter.record(name, report)                                  # counts per rule -> gpurun_out/arbiter/<name>.json, floor on rule (a)


@pytest.mark.parametrize('engine', ['fused', 'modular'])
def test_configs4_habitat_64_bodies_full_size_vs_oracle(smplx_data, vposer_sd, engine):
    B = 64
    scene = synth.make_scene(4, M, D, NC)
    bodies = synth.make_bodies(17, B)
    bodies['cam_ext'] = synth.make_cam_ext(2, 1)                  # one camera per view (test_habitat_s2.py writes one cam_ext per body file)
    op = fitting.FittingOPHabitat(_cfg(smplx_data, vposer_sd, scene, B, engine), dict(LOSS))
    trace = _run(op, dict(bodies))
    cam = bodies['cam_ext'][:1] @ np.diag([1.0, -1.0, -1.0, 1.0]).astype(np.float32)          # fitting_habitat.py:179-184
    _check(trace, smplx_data, vposer_sd, scene, bodies, np.repeat(cam, B, axis=0), name='configs4_habitat_64_%s' % engine, contact_const=1.0)     # fitting_habitat.py:141
