"""The data-parallel fitting loop over RCCL on MORE THAN ONE GPU, checked against the oracle — runs wherever at least two GPUs are visible,
skips otherwise (the single-GPU test boxes; there the world-2 semantics are covered with gloo by test_configs_dp_gpu.py and the capture /
replay mechanics over a 1-rank RCCL group by test_dist_gpu.py).

BASELINE.json configs[3] ("fitting_proxe.py batch=256 sharded 8xMI355X, RCCL all-reduce over xGMI") and configs[4] (fitting_habitat, 64
bodies per GPU): N = min(8, device_count) ranks, one per GPU, 32 (64) bodies each at the full scene size, through ``psi_fit_iterate_dp`` —
forward half, ONE ncclAllReduce of the 6 loss normalisers issued from C on the library's own communicator, backward half, replayed as
hipGraphs.  Checked:
  * every iteration of the gathered run against ``FittingOracle`` on the GLOBAL batch (fitting_proxe.py:101-162,177-189 on N x 32 bodies),
    from the ranks' own state, by the rules of tests/arbiter.py (loss values, gradient, Adam update; fp64 arbiter);
  * RCCL itself reports N ranks on every rank (``psi_dp_comm_info`` = ncclCommCount) and every engine ran its loop as hipGraphs with the
    collective inside (``psi_fit_dp_mode`` == 1);
  * a 23-iteration run issued as one call (eager first iteration, two 10-iteration graphs, single-iteration graphs) is bit-identical on
    every rank to the same run issued iteration by iteration, and every rank reports the same global loss values.
So the first run on an 8-GPU node is a parity result, not just a number."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT
from psi_release_amd import fitting, synth
import arbiter
from test_configs_gpu import LOSS, M, NC, D, _cfg, _check, _free_port

N_GPUS = torch.cuda.device_count() if torch.cuda.is_available() else 0
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(N_GPUS < 2, reason='needs at least two GPUs (one RCCL rank per GPU)')]


def _rank_worker(rank, world, port, tmp, habitat, per, iters):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    from psi_release_amd import dist as pd
    torch.cuda.set_device(rank)                                     # one rank per GPU
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    scene = synth.make_scene(4 if habitat else 0, M, D, NC)
    bodies = synth.make_bodies(17 if habitat else 13, per * world)
    bodies['cam_ext'] = np.repeat(synth.make_cam_ext(2, 1), per * world, axis=0) if habitat else synth.make_cam_ext(9, per * world)
    cfg = dict(_cfg(synth.make_smplx(7), synth.make_vposer_state(3), scene, per), device=torch.device('cuda', rank))
    op = (fitting.FittingOPHabitat if habitat else fitting.FittingOP)(cfg, dict(LOSS))
    mine = {k: v[rank * per:(rank + 1) * per] for k, v in bodies.items()}
    trace = arbiter.gpu_trace(op, dict(mine), iters)
    arbiter.save_trace(os.path.join(tmp, 'trace%d.npz' % rank), trace)
    eng = op._fused
    _, seen, ver = pd.rccl_comm_info()
    # one call of 23 iterations against 23 calls of one: the same launches grouped differently
    runner = op.make_step_runner(dict(mine))
    runner.restart()
    runner.steps(23)
    x_once, _, step_once = eng.read(0)
    loss_once = runner.last_losses()
    runner.restart()
    for _ in range(23):
        runner.step()
    x_each, _, step_each = eng.read(0)
    np.savez(os.path.join(tmp, 'info%d.npz' % rank), seen=seen, version=ver, mode=eng.dp_mode(), world_engine=eng.world,
             same=bool(torch.equal(x_once, x_each)), steps=np.array([step_once, step_each]), losses=np.array(loss_once),
             losses_each=np.array(runner.last_losses()))
    dist.barrier()
    pd.rccl_comm_release()
    dist.destroy_process_group()


@pytest.mark.parametrize('habitat', [False, True], ids=['configs3_proxe_32_per_gpu', 'configs4_habitat_64_per_gpu'])
def test_rccl_ranks_equal_the_oracle_on_the_global_batch(tmp_path, smplx_data, vposer_sd, habitat):
    world = min(8, N_GPUS)
    per, iters = (64, 2) if habitat else (32, 3)
    mp.spawn(_rank_worker, args=(world, _free_port(), str(tmp_path), habitat, per, iters), nprocs=world, join=True)
    infos = [np.load(tmp_path / ('info%d.npz' % r)) for r in range(world)]
    for r, i in enumerate(infos):
        assert int(i['seen']) == world, ('RCCL saw %d ranks on rank %d, expected %d' % (int(i['seen']), r, world))
        assert int(i['world_engine']) == world
        assert int(i['mode']) == 1, ('rank %d did not run its loop as hipGraphs with the collective inside (psi_fit_dp_mode = %d)' % (r, int(i['mode'])))
        assert bool(i['same']), 'rank %d: one call of 23 iterations differs from 23 calls of one' % r
        assert list(i['steps']) == [23, 23]
        assert np.array_equal(i['losses'], infos[0]['losses']) and np.array_equal(i['losses'], i['losses_each'])      # GLOBAL loss values
    trace = arbiter.load_traces([tmp_path / ('trace%d.npz' % r) for r in range(world)])       # rows concatenated; asserts the ranks agree
    scene = synth.make_scene(4 if habitat else 0, M, D, NC)
    bodies = synth.make_bodies(17 if habitat else 13, per * world)
    if habitat:
        cam = synth.make_cam_ext(2, 1) @ np.diag([1.0, -1.0, -1.0, 1.0]).astype(np.float32)   # fitting_habitat.py:179-184
        _check(trace, smplx_data, vposer_sd, scene, bodies, np.repeat(cam, per * world, axis=0), name='rccl_%d_ranks_habitat_64' % world, contact_const=1.0)
    else:
        cam = synth.make_cam_ext(9, per * world)
        _check(trace, smplx_data, vposer_sd, scene, bodies, cam, name='rccl_%d_ranks_proxe_32' % world)


def _bucket_worker(rank, world, port, tmp):
    """The NCCL branch of dist.GradBuckets (side-stream all-reduce issued from autograd hooks): one data-parallel step, eager and captured in a
    torch.cuda.graph (the collectives become branches of the graph), against the gradients of the full batch."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    from psi_release_amd import dist as pd
    torch.cuda.set_device(rank)
    dev = torch.device('cuda', rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    def make():
        torch.manual_seed(3)
        m = torch.nn.Sequential(torch.nn.Conv2d(2, 8, 3, 1, 1), torch.nn.ReLU(), torch.nn.Flatten(), torch.nn.Linear(8 * 6 * 6, 40), torch.nn.LeakyReLU(),
                                torch.nn.Linear(40, 5)).to(dev)
        m[0].to(memory_format=torch.channels_last)
        return m
    rs = np.random.RandomState(2)
    n = 4 * world
    X = torch.tensor(rs.standard_normal((n, 2, 6, 6)), dtype=torch.float32, device=dev)
    Y = torch.tensor(rs.standard_normal((n, 5)), dtype=torch.float32, device=dev)
    lo, hi = pd.shard_rows(n, rank, world)
    full, mine = make(), make()
    ((full(X) - Y) ** 2).mean().backward()
    want = [p.grad.clone() for p in full.parameters()]
    b = pd.GradBuckets(mine, bucket_mb=0.004)
    ok = b.n_buckets() >= 3

    def step(xs, ys):
        b.begin()
        ((mine(xs) - ys) ** 2).mean().backward()
        b.finish()
    step(X[lo:hi], Y[lo:hi])                                           # eager: also RCCL's lazy channel setup, which cannot be captured
    torch.cuda.synchronize()
    ok = ok and all(float((p.grad - w).abs().max()) < 1e-5 for p, w in zip(mine.parameters(), want))
    xs, ys = X[lo:hi].clone(), Y[lo:hi].clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step(xs, ys)
    for p in mine.parameters():
        p.grad.zero_()                                                 # (the capture did not execute)
    graph.replay()
    torch.cuda.synchronize()
    ok = ok and all(float((p.grad - w).abs().max()) < 1e-5 for p, w in zip(mine.parameters(), want))
    open(os.path.join(tmp, 'bk%d' % rank), 'w').write('1' if ok else '0')
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_buckets_over_rccl_eager_and_captured(tmp_path):
    world = min(N_GPUS, 8)
    mp.spawn(_bucket_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(open(tmp_path / ('bk%d' % r)).read() == '1' for r in range(world))
