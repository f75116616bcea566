"""BASELINE.json configs[3] at its per-rank shape against the ORACLE: "fitting_proxe.py batch=256 sharded 8xMI355X" = every rank holds 32
bodies at the full scene size (n_c=2048, m=32768, 256^3 SDF) and the loss normalisers are global.  Here: 2 ranks x 32 bodies (gloo, both on
the one GPU of the test box), 3 iterations; the gathered first-iteration gradient, the per-iteration loss values and the parameters must
equal ``FittingOracle`` run on the GLOBAL batch of 64 (fitting_proxe.py:101-162,177-189 on 64 bodies) — see test_configs_gpu._check for
what is compared exactly.  A separate module because it spawns processes: tests/conftest.py collects it after the single-process parity
modules."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT
from psi_release_amd import fitting, synth
from test_configs_gpu import ITERS, LOSS, M, NC, D, _cfg, _check, _free_port, _oracle, _oracle_run, _run

pytestmark = pytest.mark.gpu


def _rank_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    per = 32
    scene = synth.make_scene(0, M, D, NC)
    bodies = synth.make_bodies(13, per * world)
    bodies['cam_ext'] = synth.make_cam_ext(9, per * world)
    op = fitting.FittingOP(_cfg(synth.make_smplx(7), synth.make_vposer_state(3), scene, per), dict(LOSS))
    x, losses, m1 = _run(op, {k: v[rank * per:(rank + 1) * per] for k, v in bodies.items()})
    np.save(os.path.join(tmp, 'x%d.npy' % rank), x)
    np.save(os.path.join(tmp, 'l%d.npy' % rank), losses)
    np.save(os.path.join(tmp, 'm%d.npy' % rank), m1)
    dist.barrier()
    dist.destroy_process_group()


def test_configs3_two_ranks_of_32_bodies_equal_the_oracle_on_64(tmp_path, smplx_data, vposer_sd):
    world, per = 2, 32
    port = _free_port()
    mp.spawn(_rank_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    x_gpu = np.concatenate([np.load(tmp_path / ('x%d.npy' % r)) for r in range(world)])
    l0, l1 = np.load(tmp_path / 'l0.npy'), np.load(tmp_path / 'l1.npy')
    assert np.array_equal(l0, l1)                   # every rank reports the GLOBAL loss values (one all-reduce per iteration)
    scene = synth.make_scene(0, M, D, NC)
    bodies = synth.make_bodies(13, per * world)
    bodies['cam_ext'] = synth.make_cam_ext(9, per * world)
    m_gpu = np.concatenate([np.load(tmp_path / ('m%d.npy' % r)) for r in range(world)])
    fo = _oracle(smplx_data, vposer_sd, scene, per * world)
    _check((x_gpu, l0, m_gpu), _oracle_run(fo, bodies, bodies['cam_ext']))
