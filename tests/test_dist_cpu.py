"""world_size-2 gloo test (CPU) of the data-parallel loss reduction: each rank's losses and gradients equal the
single-process full-batch ones for its rows (psi-release_amd/dist.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _torch_pen_stats(vals):
    neg = vals < 0
    return torch.stack([(-vals[neg]).sum(), neg.sum().to(vals.dtype)])


def _full_batch(x, sdf_in):
    """Single-process statement: mean losses over the whole batch + global-count penetration loss."""
    l_rec = x.abs().mean()
    l_vp = (x[:, :5] ** 2).mean()
    l_c = torch.sigmoid(x).mean()
    sdf = sdf_in * x[:, :1]                       # make sdf depend on the parameters
    neg = sdf < 0
    pen = sdf[neg].abs().mean() if neg.any() else sdf.sum() * 0
    return l_rec, l_vp, l_c, pen


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from psi_release_amd import dist as pd
    pd.penetration_stats = _torch_pen_stats        # CPU stand-in for the HIP reduction (logic test only)
    rs = np.random.RandomState(0)
    X = torch.tensor(rs.standard_normal((8, 6)), dtype=torch.float32)
    S = torch.tensor(rs.standard_normal((8, 50)), dtype=torch.float32)
    lo, hi = pd.shard_rows(8, rank, world)
    assert (lo, hi) == (rank * 4, rank * 4 + 4)
    x = X[lo:hi].clone().requires_grad_()
    sdf = S[lo:hi] * x[:, :1]
    l_rec, l_vp, l_c = x.abs().mean(), (x[:, :5] ** 2).mean(), torch.sigmoid(x).mean()
    g = pd.fitting_loss_reduce(l_rec, l_vp, l_c, sdf)
    total = g[0] + 0.01 * g[1] + 0.1 * g[2] + 0.5 * g[3]
    total.backward()
    xf = X.clone().requires_grad_()
    f = _full_batch(xf, S)
    (f[0] + 0.01 * f[1] + 0.1 * f[2] + 0.5 * f[3]).backward()
    ok = all(abs(float(a) - float(b)) < 1e-6 for a, b in zip(g, f)) and \
        float((x.grad - xf.grad[lo:hi]).abs().max()) < 1e-6
    open(os.path.join(tmp, 'ok%d' % rank), 'w').write('1' if ok else '0')
    dist.destroy_process_group()


def test_loss_reduce_world2(tmp_path):
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / 'ok0').read() == '1' and open(tmp_path / 'ok1').read() == '1'


def test_shard_rows_covers_everything():
    sys.path.insert(0, ROOT)
    from psi_release_amd import dist as pd
    for n in (1, 7, 32, 255, 256):
        for w in (1, 2, 4, 8):
            spans = [pd.shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _worker_training_pen(rank, world, port, tmp):
    """dist.penetration_loss_global + averaged gradients == the full-batch penetration mean and its gradient; gather_rows and
    assert_equal_across_ranks; the collective skip decision of the training loop."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from psi_release_amd import dist as pd
    pd.penetration_stats = _torch_pen_stats
    rs = np.random.RandomState(1)
    Wt = torch.tensor(rs.standard_normal((3,)), dtype=torch.float32)           # "model parameters" shared by the ranks
    S = torch.tensor(rs.standard_normal((8, 40)), dtype=torch.float32)
    S[:4] += 1.5                                                                   # very different penetration counts per rank
    lo, hi = pd.shard_rows(8, rank, world)
    w = Wt.clone().requires_grad_()
    sdf = S[lo:hi] * w[0] + w[1] * 0.1 + w[2] * S[lo:hi] ** 2 * 0.01
    loss = pd.penetration_loss_global(sdf)
    loss.backward()
    g = w.grad.clone()
    dist.all_reduce(g)
    g /= world                                                                     # the mean over the ranks (dist.GradBuckets)
    wf = Wt.clone().requires_grad_()
    sf = S * wf[0] + wf[1] * 0.1 + wf[2] * S ** 2 * 0.01
    neg = sf < 0
    lf = sf[neg].abs().mean()
    lf.backward()
    ok = abs(float(loss) - float(lf)) < 1e-6 and float((g - wf.grad).abs().max()) < 1e-5
    # rows come back in rank order
    rows = pd.gather_rows(torch.arange(lo, hi, dtype=torch.float32).view(-1, 1))
    ok = ok and rows.view(-1).tolist() == list(range(8))
    pd.assert_equal_across_ranks(4, 'rows')
    try:
        pd.assert_equal_across_ranks(4 + rank, 'rows')
        ok = False
    except ValueError:
        pass
    # collective skip: a batch that only ONE rank lacks is skipped by both
    import types
    from psi_release_amd import training
    me = types.SimpleNamespace(device=torch.device('cpu'))
    ok = ok and training._TrainBase._all_ranks_have(me, None if rank == 1 else [1]) is False
    ok = ok and training._TrainBase._all_ranks_have(me, [1]) is True
    open(os.path.join(tmp, 'pen%d' % rank), 'w').write('1' if ok else '0')
    dist.barrier()
    dist.destroy_process_group()


def test_training_penetration_global_and_helpers_world2(tmp_path):
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker_training_pen, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / 'pen0').read() == '1' and open(tmp_path / 'pen1').read() == '1'


def _worker_buckets(rank, world, port, tmp):
    """dist.GradBuckets: gradients that alias a few flat buckets, all-reduced from autograd hooks, equal the full-batch gradients; several
    buckets, a channels_last convolution weight, a parameter that gets no gradient; two steps through an optimiser; the per-epoch skip
    decision of the training loop."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from psi_release_amd import dist as pd

    def make():
        torch.manual_seed(3)
        m = torch.nn.Sequential(torch.nn.Conv2d(2, 8, 3, 1, 1), torch.nn.ReLU(), torch.nn.Flatten(), torch.nn.Linear(8 * 6 * 6, 40), torch.nn.LeakyReLU(),
                                torch.nn.Linear(40, 5))
        m[0].to(memory_format=torch.channels_last)
        m.unused = torch.nn.Parameter(torch.ones(7))                     # never reaches the loss
        return m
    rs = np.random.RandomState(2)
    X = torch.tensor(rs.standard_normal((8, 2, 6, 6)), dtype=torch.float32)
    Y = torch.tensor(rs.standard_normal((8, 5)), dtype=torch.float32)
    lo, hi = pd.shard_rows(8, rank, world)
    full, mine = make(), make()
    opt_f, opt_m = torch.optim.Adam(full.parameters(), lr=1e-2), torch.optim.Adam(mine.parameters(), lr=1e-2)
    b = pd.GradBuckets(mine, bucket_mb=0.004)                            # 1000 floats per bucket: several buckets
    ok = b.n_buckets() >= 3
    for step in range(2):
        opt_f.zero_grad()
        ((full(X) - Y) ** 2).mean().backward()
        b.begin()
        ((mine(X[lo:hi]) - Y[lo:hi]) ** 2).mean().backward()
        b.finish()
        for (k, pf), (_, pm) in zip(full.named_parameters(), mine.named_parameters()):
            gf = pf.grad if pf.grad is not None else torch.zeros_like(pf)
            ok = ok and pm.grad is not None and float((pm.grad - gf).abs().max()) < 1e-6 and pm.grad.stride() == pm.stride()
        opt_f.step()
        opt_m.step()
        ok = ok and all(float((pf - pm).abs().max()) < 1e-6 for pf, pm in zip(full.parameters(), mine.parameters()))
    # gradients that lost their alias (set_to_none, or a Module.to(memory_format=...) that re-created parameter and gradient) are moved back
    opt_m.zero_grad(set_to_none=True)
    mine[0].to(memory_format=torch.contiguous_format)
    opt_f.zero_grad()
    ((full(X) - Y) ** 2).mean().backward()
    b.begin()
    ((mine(X[lo:hi]) - Y[lo:hi]) ** 2).mean().backward()
    b.finish()
    for pf, pm in zip(full.parameters(), mine.parameters()):
        gf = pf.grad if pf.grad is not None else torch.zeros_like(pf)
        if pm is mine.unused:
            continue                                                       # (no gradient arrives: nothing to re-alias, the bucket holds zeros)
        inside = any(bb['flat'].data_ptr() <= pm.grad.data_ptr() < bb['flat'].data_ptr() + bb['flat'].numel() * 4 for bb in b.buckets)
        ok = ok and inside and float((pm.grad - gf).abs().max()) < 1e-6
    dist.barrier()
    # per-epoch skip decision: rank 1 lacks batch 1 -> both skip it; one collective per epoch
    import types
    from psi_release_amd import training
    gen = types.SimpleNamespace(epoch_batch_validity=lambda bs: [True, rank == 0, True])
    me = types.SimpleNamespace(device=torch.device('cpu'), batch_size=4)
    ok = ok and training._TrainBase._epoch_validity(me, gen) == [True, False, True]
    ok = ok and training._TrainBase._epoch_validity(me, types.SimpleNamespace()) is None
    open(os.path.join(tmp, 'bk%d' % rank), 'w').write('1' if ok else '0')
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_buckets_and_epoch_skip_decision_world2(tmp_path):
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker_buckets, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / 'bk0').read() == '1' and open(tmp_path / 'bk1').read() == '1'
