"""CPU link of the loss-glue parity chain.  The GPU tests (tests/test_cvae_glue_gpu.py) hold the fused HIP ops of the training step to the
operator sequence of psi_release_amd/geometry.py and to the loss expressions of cal_loss; here that operator sequence and those expressions
are held to the ORACLE's restatement (oracle/psi_oracle.py, itself pinned to the reference's recorded vectors in test_oracle_cpu.py) on
random batches — including rotations in the first-order branch of the axis-angle conversion and batches without a penetrating vertex."""
import numpy as np
import torch

import psi_oracle as O
from conftest import rel_err
from psi_release_amd.geometry import GeometryTransformer as GT


def _batch(B, seed):
    rs = np.random.RandomState(seed)
    xh = rs.standard_normal((B, 72)).astype(np.float32) * 0.4
    xh[:, 2] = np.abs(xh[:, 2]) + 1.5
    xh[0, 3:6] = [3e-4, -2e-4, 1e-4]                      # theta^2 < 1e-6
    xh[1, 3:6] = 0.0
    cam = np.tile(np.array([[500.0, 0, 320], [0, 480.0, 250], [0, 0, 1]], dtype=np.float32), (B, 1, 1))
    cam[:, 0, 2] += rs.uniform(-30, 30, B).astype(np.float32)
    return torch.tensor(xh), torch.tensor(cam), torch.tensor(rs.uniform(4, 8, B).astype(np.float32))


def test_target_representation_and_recover_equal_the_oracle():
    for B, seed in ((2, 0), (7, 1), (128, 2)):
        xh, cam, md = _batch(B, seed)
        tgt = GT.convert_to_6D_rot(GT.normalize_global_T(xh, cam, md))
        ref = O.convert_to_6d_rot(O.normalize_global_T(xh, cam, md))
        assert tgt.shape == (B, 75) and rel_err(tgt, ref) < 1e-6
        rec = tgt + 0.05 * torch.randn(B, 75, generator=torch.Generator().manual_seed(seed))
        assert rel_err(GT.recover_global_T(rec, cam, md), O.recover_global_T(rec, cam, md)) < 1e-6
        # the two maps invert each other on the translation (cvae.py:153-199), to the 1e-6 regulariser of the normalisation
        back = O.recover_global_T(O.normalize_global_T(xh, cam, md), cam, md)
        assert float((back[:, :3] - xh[:, :3]).abs().max()) < 1e-4


def test_scene_loss_expressions_equal_the_oracle():
    g = torch.Generator().manual_seed(5)
    dist = torch.rand(6, 300, generator=g) * 0.3
    s = torch.sqrt(dist + 1e-4)
    assert abs(float(torch.mean(s / (s + 1.0))) - float(O.contact_loss(dist, 1.0))) < 1e-7          # train_s1.py:175-177
    vals = torch.randn(6, 2000, generator=g)
    neg = vals < 0
    expr = (-vals[neg]).sum() / neg.sum().clamp(min=1)                                               # what scene_loss.hip evaluates
    assert abs(float(expr) - float(O.penetration_loss(vals))) < 1e-6
    assert float(O.penetration_loss(vals.abs() + 0.1)) == 0.0                                        # no penetrating vertex -> 0
