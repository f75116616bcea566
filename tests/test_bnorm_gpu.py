"""Fused BatchNorm (+ ReLU, + skip connection) of the scene trunk (csrc/bnorm.hip, ops.bn_act) against a plain PyTorch fp32 reference
of the same op on the same bf16-rounded inputs: output, running statistics, and every gradient; then the whole trunk of HumanCVAES2
with the fused kernels against the library path (torchvision BasicBlock semantics, cvae.py:427-435)."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import library_paths
from psi_release_amd import models, ops

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _ref(x, res, bn, relu):
    """fp32 reference: F.batch_norm in training mode on the bf16-rounded operands (fp32 arithmetic), then + residual, ReLU."""
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    y = F.batch_norm(x.float(), rm, rv, bn.weight, bn.bias, True, bn.momentum, bn.eps)
    if res is not None:
        y = y + res.float()
    if relu:
        y = torch.relu(y)
    return y, rm, rv


@pytest.mark.parametrize('shape', [(8, 64, 32, 32), (4, 128, 16, 16), (2, 64, 64, 64), (3, 128, 5, 7)])
@pytest.mark.parametrize('relu,with_res', [(True, False), (True, True), (False, False), (False, True)])
def test_bn_act_matches_fp32_reference(shape, relu, with_res):
    torch.manual_seed(sum(shape) + 2 * relu + with_res)
    N, C, H, W = shape
    x = (torch.randn(shape, device=DEV) * 1.7 + 0.3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    res = (torch.randn(shape, device=DEV)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_() if with_res else None
    bn = torch.nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
    bn.train()
    bn2 = torch.nn.BatchNorm2d(C).to(DEV)
    bn2.load_state_dict(bn.state_dict())
    # reference first (does not touch bn's buffers)
    xr = x.detach().clone().requires_grad_()
    rr = res.detach().clone().requires_grad_() if with_res else None
    yr, rm, rv = _ref(xr, rr, bn2, relu)
    g = torch.randn(shape, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    yr.backward(g.float())
    y = ops.bn_act(x, bn, relu=relu, residual=res)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(g)
    # output: bf16 rounding of an fp32 result
    assert float((y.float() - yr).abs().max()) <= 2 ** -7 * float(yr.abs().max()) + 1e-6
    # running statistics (fp32) and the batch counter
    assert torch.allclose(bn.running_mean, rm, rtol=1e-5, atol=1e-6) and torch.allclose(bn.running_var, rv, rtol=1e-4, atol=1e-6)
    assert int(bn.num_batches_tracked) == 1
    # gradients: dgamma / dbeta are fp32 sums, dx / dresidual are stored as bf16
    assert torch.allclose(bn.weight.grad, bn2.weight.grad, rtol=2e-3, atol=2e-3 * float(bn2.weight.grad.abs().max()))
    assert torch.allclose(bn.bias.grad, bn2.bias.grad, rtol=2e-3, atol=2e-3 * float(bn2.bias.grad.abs().max()))
    assert float((x.grad.float() - xr.grad).abs().max()) <= 2 ** -6 * float(xr.grad.abs().max())
    if with_res:
        assert float((res.grad.float() - rr.grad).abs().max()) <= 2 ** -7 * float(rr.grad.abs().max())


@pytest.mark.parametrize('shape', [(4, 64, 64, 64), (2, 64, 9, 7), (3, 128, 16, 16), (1, 8, 1, 1)])
def test_maxpool3x3s2_equals_torch(shape):
    """ops.maxpool3x3s2 == F.max_pool2d(x, 3, 2, 1) on bf16 channels_last maps, values and gradient, bit for bit — on ReLU outputs,
    i.e. with many exact ties inside the windows (the first maximum in scan order takes the gradient in both)."""
    torch.manual_seed(sum(shape))
    x = torch.relu(torch.randn(shape, device=DEV)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    xr = x.detach().clone().requires_grad_()
    y = ops.maxpool3x3s2(x)
    yr = F.max_pool2d(xr, 3, 2, 1)
    assert y.shape == yr.shape and torch.equal(y, yr)
    g = torch.randn(yr.shape, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y.backward(g)
    yr.backward(g)
    assert torch.equal(x.grad, xr.grad)


def test_trunk_with_fused_bn_is_as_close_to_fp32_as_the_library_path(monkeypatch):
    """BodyGlobalPoseVAE scene feature (trunk + conv + fc) in training mode, three ways on the same weights and input: fp32 (no
    autocast: the reference's arithmetic, cvae.py:427-455), bf16 autocast with the library BN (MIOpen), bf16 autocast with the fused
    HIP BN.  bf16 activation gradients through nine convolutions make the per-channel sums of the BN parameters noisy in EITHER bf16
    path (tens of per cent of a tensor's largest entry at this batch size), so the fused path is held to the library path's distance
    from fp32, not to the library path itself: outputs, every parameter gradient, and the updated running statistics."""
    torch.manual_seed(0)
    xs = torch.randn(8, 2, 128, 128, device=DEV)
    sd, outs = None, {}
    def run(mode):
        m = models.BodyGlobalPoseVAE(zdim=32, in_dim=2, num_hidden=256).to(DEV)
        m.autocast_bf16 = mode != 'fp32'
        m.load_state_dict(sd)
        m.train()
        z = m._scene_feature(xs)
        z.float().square().mean().backward()
        return (z.detach().float(), {k: p.grad.detach().float() for k, p in m.named_parameters() if p.grad is not None},
                {k: v.detach().float().clone() for k, v in m.state_dict().items() if 'running' in k or 'num_batches' in k})
    sd = {k: v.clone() for k, v in models.BodyGlobalPoseVAE(zdim=32, in_dim=2, num_hidden=256).to(DEV).state_dict().items()}
    outs['fp32'], outs['hip'] = run('fp32'), run('hip')
    with monkeypatch.context() as mp_:
        library_paths.bf16_batchnorm_on_the_library(mp_)             # nn.BatchNorm2d under autocast (MIOpen)
        outs['lib'] = run('lib')
    z32, g32, r32 = outs['fp32']
    dist = lambda a, b: float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)
    for mode in ('lib', 'hip'):
        z, g, r = outs[mode]
        assert set(g) == set(g32)
        assert dist(z, z32) < 3e-2, (mode, dist(z, z32))
        for k in r32:                                            # running statistics: fp32 sums of bf16 maps in both paths
            assert torch.allclose(r[k], r32[k], rtol=2e-2, atol=2e-3), (mode, k)
    e_lib = np.array([dist(outs['lib'][1][k], g32[k]) for k in sorted(g32)])
    e_hip = np.array([dist(outs['hip'][1][k], g32[k]) for k in sorted(g32)])
    assert e_hip.mean() <= 1.25 * e_lib.mean() + 0.01, (e_hip.mean(), e_lib.mean())
    assert e_hip.max() <= 1.5 * e_lib.max() + 0.02, (e_hip.max(), e_lib.max())


@pytest.mark.parametrize('shape,f32', [((64, 64, 32, 32), False), ((8, 128, 16, 16), True), ((3, 64, 9, 5), False), ((2, 128, 7, 3), True)])
def test_relu_mask_recomputed_from_x_equals_the_stored_mask(shape, f32, monkeypatch):
    """BatchNorm + ReLU without a skip connection: the backward recomputes "y > 0" from x (y = relu(fma(x, scale, shift)), rounded as the forward
    stored it) and reads one map less per pass; against the same backward with the mask taken from the stored y (ops._bn_mask_from_x patched): every
    gradient bit for bit — also with inputs placed ON the threshold (outputs that are exactly zero, and the smallest positive ones)."""
    N, C, H, W = shape
    torch.manual_seed(C + H + W)
    dt = torch.float32 if f32 else torch.bfloat16
    x0 = (torch.randn(shape, device=DEV) * 1.1 - 0.1).to(dt).contiguous(memory_format=torch.channels_last)
    g = torch.randn(shape, device=DEV).to(dt).contiguous(memory_format=torch.channels_last)
    got = {}
    for xmask in ('1', '0'):
        if xmask == '0':
            library_paths.bn_relu_mask_from_the_stored_output(monkeypatch)
        torch.manual_seed(2)
        bn = torch.nn.BatchNorm2d(C).to(DEV).train()
        with torch.no_grad():
            bn.weight.uniform_(-1.5, 1.5)                           # negative scales too
            bn.bias.uniform_(-0.5, 0.5)
            bn.bias[::3] = 0.0                                      # shift = -mean * scale: outputs around zero
        x = x0.clone().requires_grad_()
        y = (ops.bn_act_t if f32 else ops.bn_act)(x, bn, relu=True)
        y.backward(g)
        got[xmask] = (y.detach().clone(), x.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone())
        frac_open = float((y > 0).float().mean())
        assert 0.2 < frac_open < 0.8
    for a, b in zip(got['1'], got['0']):
        assert torch.equal(a, b)
