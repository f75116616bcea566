"""BASELINE.json configs[3] at its per-rank shape against the ORACLE: "fitting_proxe.py batch=256 sharded 8xMI355X" = every rank holds 32
bodies at the full scene size (n_c=2048, m=32768, 256^3 SDF) and the loss normalisers are global.  Here: 2 ranks x 32 bodies (gloo, both on
the one GPU of the test box), 3 iterations; the gathered first-iteration gradient, the per-iteration loss values and the parameters must
equal ``FittingOracle`` run on the GLOBAL batch of 64 (fitting_proxe.py:101-162,177-189 on 64 bodies), iteration by iteration from the
ranks' own state, within bounds derived from the fp64 arbiter — see tests/arbiter.py for what is compared exactly.  A separate module because it spawns processes: tests/conftest.py collects it after the single-process parity
modules."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT
from psi_release_amd import fitting, synth
import arbiter
from test_configs_gpu import ITERS, LOSS, M, NC, D, _cfg, _check, _free_port, _run

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
    trace = _run(op, {k: v[rank * per:(rank + 1) * per] for k, v in bodies.items()})
    arbiter.save_trace(os.path.join(tmp, 'trace%d.npz' % rank), trace)
    dist.barrier()
    dist.destroy_process_group()


def test_configs3_two_ranks_of_32_bodies_equal_the_oracle_on_64(tmp_path, smplx_data, vposer_sd):
    world, per = 2, 32
    port = _free_port()
    mp.spawn(_rank_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    trace = arbiter.load_traces([tmp_path / ('trace%d.npz' % r) for r in range(world)])     # rows concatenated; asserts the ranks' loss values agree
    scene = synth.make_scene(0, M, D, NC)
    bodies = synth.make_bodies(13, per * world)
    bodies['cam_ext'] = synth.make_cam_ext(9, per * world)
    _check(trace, smplx_data, vposer_sd, scene, bodies, bodies['cam_ext'], name='configs3_two_ranks_of_32')
