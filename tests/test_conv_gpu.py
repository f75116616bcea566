"""Hand-written 3x3 / stride 1 / padding 1 convolution of the scene trunk (csrc/conv.hip, ops.conv3x3) against a plain PyTorch fp32
reference of the same op on the same bf16-rounded operands: output, input gradient (the kernel again, on the rotated weight), weight and
bias gradients — at the trunk's shapes (cvae.py:427-435 layer1 / layer2, net_layers.py:160-164 head conv) and small odd batches."""
import pytest
import torch
import torch.nn.functional as F

from psi_release_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('N,Cin,Cout,H,W,bias', [(3, 64, 64, 32, 32, False), (2, 128, 128, 16, 16, False), (2, 128, 128, 16, 16, True),
                                                 (1, 64, 128, 8, 32, True), (5, 128, 256, 8, 16, False), (128, 64, 64, 32, 32, False),
                                                 (128, 128, 128, 16, 16, True)])
def test_conv3x3_matches_fp32_reference(N, Cin, Cout, H, W, bias):
    torch.manual_seed(N + Cin + Cout + H)
    conv = torch.nn.Conv2d(Cin, Cout, 3, 1, 1, bias=bias).to(DEV).to(memory_format=torch.channels_last)
    x = torch.randn(N, Cin, H, W, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    assert ops.conv3x3_supported(conv, x)
    g = torch.randn(N, Cout, H, W, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = ops.conv3x3(x, conv)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(g)
    gw, gb = conv.weight.grad.clone(), (conv.bias.grad.clone() if bias else None)
    conv.zero_grad()
    # reference: fp32 arithmetic on the bf16-rounded operands
    xr = x.detach().float().requires_grad_()
    w16 = conv.weight.detach().to(torch.bfloat16).float().requires_grad_()
    br = conv.bias.detach().clone().requires_grad_() if bias else None
    yr = F.conv2d(xr, w16, br, 1, 1)
    yr.backward(g.float())
    scale = float(yr.abs().max())
    assert float((y.float() - yr).abs().max()) <= 2 ** -7 * scale                       # bf16 rounding of the fp32 result
    assert float((x.grad.float() - xr.grad).abs().max()) <= 2 ** -7 * float(xr.grad.abs().max())
    # weight gradient: library bf16 wrw (bf16 output) — a few bf16 ulps of the largest entry
    assert float((gw - w16.grad).abs().max()) <= 2 ** -6 * float(w16.grad.abs().max())
    if bias:
        assert torch.allclose(gb, br.grad, rtol=1e-3, atol=1e-3 * float(br.grad.abs().max()))


def test_uncovered_shapes_are_reported():
    x = torch.zeros(1, 64, 16, 16, device=DEV, dtype=torch.bfloat16)
    assert not ops.conv3x3_supported(torch.nn.Conv2d(64, 64, 3, 1, 1).to(DEV), x)           # W % 32 != 0 for Cin = 64
    assert not ops.conv3x3_supported(torch.nn.Conv2d(64, 64, 3, 2, 1).to(DEV), torch.zeros(1, 64, 32, 32, device=DEV, dtype=torch.bfloat16))
    assert not ops.conv3x3_supported(torch.nn.Conv2d(128, 32, 3, 1, 1).to(DEV), torch.zeros(1, 128, 16, 16, device=DEV, dtype=torch.bfloat16))
