"""Data-parallel fused engine on the GPU: two processes (gloo all-reduce, both on cuda:0 — the test box has one GPU) each
fit half of a 4-body batch; their rows must equal the single-process full-batch result (psi_fit_forward -> all-reduce of
stats[6] -> psi_fit_backward_step with global-batch normalisers and the GLOBAL penetration count)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]
LOSS = {'weight_loss_rec': 1, 'weight_loss_vposer': 0.01, 'weight_contact': 0.1, 'weight_collision': 0.5}


def _cfg(B, engine='fused'):
    from psi_release_amd import synth
    return {'scene_verts_path': None, 'scene_sdf_path': None, 'human_model_path': None, 'vposer_ckpt_path': None, 'init_lr_h': 0.1,
            'num_iter': 3, 'batch_size': B, 'device': torch.device('cuda', 0), 'contact_part': synth.CONTACT_PARTS,
            'contact_id_folder': None, 'verbose': False, 'smplx_data': synth.make_smplx(7), 'vposer_state': synth.make_vposer_state(3),
            'scene': synth.make_scene(3, 3000, 16, 300), 'engine': engine}


def _rows(bodies, lo, hi):
    return {k: v[lo:hi] for k, v in bodies.items()}


def _worker(rank, world, port, tmp, engine):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from psi_release_amd import fitting, synth
    torch.manual_seed(0)
    bodies = synth.make_bodies(51, 4)
    bodies['cam_ext'] = synth.make_cam_ext(4, 4)
    op = fitting.FittingOP(_cfg(2, engine), dict(LOSS))
    op.dp_use_graph = (rank == 0)         # one rank replays the two half-graphs, the other launches eagerly: same numbers
    op.fitting(_rows(bodies, 2 * rank, 2 * rank + 2))
    np.save(os.path.join(tmp, 'x%d.npy' % rank), op.xhr_rec.detach().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('engine', ['fused', 'modular'])
def test_two_ranks_equal_full_batch(tmp_path, engine):
    from psi_release_amd import fitting, synth
    torch.manual_seed(0)
    bodies = synth.make_bodies(51, 4)
    bodies['cam_ext'] = synth.make_cam_ext(4, 4)
    op = fitting.FittingOP(_cfg(4, engine), dict(LOSS))
    op.fitting(dict(bodies))
    full = op.xhr_rec.detach().cpu().numpy()
    del op
    torch.cuda.synchronize()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), engine), nprocs=2, join=True)
    got = np.concatenate([np.load(tmp_path / 'x0.npy'), np.load(tmp_path / 'x1.npy')])
    assert np.abs(got - full).max() < 5e-5


def _train_worker(rank, world, port, tmp, ep):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from test_training_gpu import _train_setup
    op, batch = _train_setup(tmp, 2, rows=slice(2 * rank, 2 * rank + 2))
    losses = op.train_step(batch, ep=ep)
    if rank == 0:
        torch.save({k: v.grad.detach().cpu() for k, v in op.model_h.named_parameters()}, os.path.join(tmp, 'dp.pt'))
        torch.save(torch.stack([l.detach().float().cpu() for l in losses]), os.path.join(tmp, 'dp_losses.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('ep', [0, 90])
def test_training_gradient_allreduce_equals_full_batch(tmp_path, ep):
    """TrainOP data parallel: averaged per-rank gradients == full-batch gradient (equal shards, BN in eval mode so that batch
    statistics do not differ between the shardings) — in the first loss phase (ep 0: scene terms gated off) and in the second
    (ep 90 of 100: contact + penetration live, the penetration mean over the GLOBAL count of penetrating vertices)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_training_gpu import _train_setup
    op, batch = _train_setup(str(tmp_path), 4, rows=slice(0, 4))
    losses = op.train_step(batch, ep=ep)
    full_losses = torch.stack([l.detach().float().cpu() for l in losses])
    if ep == 90:
        assert float(full_losses[3]) > 0 and float(full_losses[5]) > 0        # contact and penetration terms are live
    full = {k: v.grad.detach().cpu() for k, v in op.model_h.named_parameters()}
    del op
    torch.cuda.synchronize()
    port = _free_port()
    mp.spawn(_train_worker, args=(2, port, str(tmp_path), ep), nprocs=2, join=True)
    dp = torch.load(tmp_path / 'dp.pt')
    dp_losses = torch.load(tmp_path / 'dp_losses.pt')
    assert abs(float(dp_losses[5]) - float(full_losses[5])) < 1e-6 * max(1.0, abs(float(full_losses[5])))   # global penetration mean on every rank
    # gradients (after the all-reduce) rather than parameters: Adam's first step is +-lr for any non-tiny gradient
    for k in full:
        scale = float(full[k].abs().max()) + 1e-12
        assert float((dp[k] - full[k]).abs().max()) / scale < 1e-3, k


@pytest.mark.parametrize('iters', [5, 27])
def test_rccl_leg_at_world_size_one(iters):
    """The data-parallel loop issued from C (psi_fit_iterate_dp: forward half -> ncclAllReduce of stats[6] on the library's own RCCL
    communicator -> backward half; the first iteration eager, the rest as hipGraphs that contain the RCCL kernel — 27 iterations = 1
    eager + two 10-iteration graphs + 6 single-iteration graphs) over a 1-rank nccl group (PSI_FORCE_DP_PATH=1) reproduces the
    single-process result bit for bit: the collective leg the multi-GPU bench relies on, exercised on the one GPU this box has."""
    import subprocess
    out = {}
    for force in ('0', '1'):
        env = dict(os.environ, PSI_FORCE_DP_PATH=force, GRAFT_REPO_ROOT=ROOT, PSI_TEST_ITERS=str(iters))
        for attempt in range(2):                               # one retry: a failed rendezvous is not what this test is about
            port = _free_port()
            r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
                                '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'tests', 'dp_check_worker.py')], env=env,
                               capture_output=True, text=True, timeout=600)
            if r.returncode == 0:
                break
        assert r.returncode == 0, r.stderr[-2000:]
        lines = {l.split()[0]: l for l in r.stdout.splitlines() if l.startswith(('checksum', 'stats', 'backend'))}
        out[force] = lines
    assert 'backend' not in out['0'] or 'nccl' in out['1'].get('backend', 'nccl')
    assert 'nccl' in out['1']['backend']
    assert out['0']['checksum'] == out['1']['checksum'], (out['0'], out['1'])
    assert out['1']['stats'] != out['0']['stats']            # the all-reduced statistics buffer was actually used
