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
