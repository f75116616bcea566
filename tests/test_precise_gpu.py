"""The CVAEs at the REFERENCE'S PRECISION on hand-written kernels (fp32 models: cvae.py:427-455,474-492; net_layers.py:28-43,56-93): the general
implicit-GEMM convolution with three-term split products (csrc/conv_gemm.hip, ops.conv2d_split), BatchNorm / max-pool on fp32 NHWC maps in
training and eval form (csrc/bnorm.hip, ops.bn_act_t / maxpool3x3s2_t) and the dense layers of any width (csrc/linear.hip, ops.linear_act3).

Operator level: each against a plain PyTorch evaluation of the same op in DOUBLE precision on the same fp32 operands — the three-term
product's error budget is 2^-16 of the sum of |products| (the bound asserted), a one-term bf16 product sits at 2^-8.
Model level: HumanCVAES1 / S2 with the hand-written path ON against the reference's recorded forward passes (tests/golden/cvae.npz:
s1_{eval,train}_*, s2_*) at 2e-4, against the library path on the same weights, and the parameter gradients of a training-mode loss."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import library_paths
from conftest import golden, rel_err
from psi_release_amd import models, ops, synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'
SPLIT3 = 2.0 ** -16               # |hi*hi + hi*lo + lo*hi - a*b| per product, relative to |a b|: the dropped lo*lo term and the two split residues are
                                  # 3 x 2^-18 with round-to-nearest splits; the worst ratio measured over these tests is 1.5e-5 = 2^-16 (3-term inner products)

# (N, Cin, Cout, K, stride, pad, H, bias): the convolutions of the scene trunk and the heads, plus odd batch sizes (M not a multiple of 128)
CONVS = [(2, 2, 64, 7, 2, 3, 128, False), (3, 64, 64, 3, 1, 1, 32, False), (2, 64, 128, 3, 2, 1, 32, False), (2, 64, 128, 1, 2, 0, 32, False),
         (5, 128, 128, 3, 1, 1, 16, False), (3, 128, 32, 3, 1, 1, 16, True), (1, 128, 128, 3, 1, 1, 16, True), (128, 64, 64, 3, 1, 1, 32, False),
         (3, 2, 64, 7, 2, 3, 50, True), (100, 2, 64, 7, 2, 3, 36, False)]      # the stem's own kernels (conv_stem.hip): partial 8 x 16 tiles, bias, more tiles than workgroups


@pytest.mark.parametrize('N,Cin,Cout,K,stride,pad,H,bias', CONVS)
def test_conv2d_split_matches_double_precision(N, Cin, Cout, K, stride, pad, H, bias):
    torch.manual_seed(N + Cin + Cout + K)
    conv = torch.nn.Conv2d(Cin, Cout, K, stride, pad, bias=bias).to(DEV).to(memory_format=torch.channels_last)
    assert ops.conv2d_supported(conv)
    x = torch.randn(N, Cin, H, H, device=DEV).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = ops.conv2d_split(x, conv, nterm=3)
        ref = F.conv2d(x.double().cpu(), conv.weight.double().cpu(), conv.bias.double().cpu() if bias else None, stride, pad)
        mag = F.conv2d(x.double().cpu().abs(), conv.weight.double().cpu().abs(), None, stride, pad)      # sum of |products| per output
    assert y.dtype == torch.float32 and y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    err = (y.cpu().double() - ref).abs()
    assert float((err / (mag + 1e-30)).max()) <= SPLIT3 + 2e-7, float((err / mag).max())      # + fp32 accumulation of K <= 1152 terms
    assert rel_err(y.cpu(), ref) < 2e-5
    # one-term products (the bf16 mode): against fp32 arithmetic on the bf16-rounded operands
    with torch.no_grad():
        y1 = ops.conv2d_split(x.to(torch.bfloat16), conv, nterm=1, out_bf16=True)
        r1 = F.conv2d(x.to(torch.bfloat16).float(), conv.weight.to(torch.bfloat16).float(), conv.bias if bias else None, stride, pad)
    assert y1.dtype == torch.bfloat16 and float((y1.float() - r1).abs().max()) <= 2 ** -7 * float(r1.abs().max())


@pytest.mark.parametrize('N,Cin,Cout,K,stride,pad,H,bias', CONVS)
def test_conv2d_split_gradients_match_double_precision(N, Cin, Cout, K, stride, pad, H, bias):
    """Input gradient (the forward kernel in its transposed-gather form) and weight gradient (conv_wgrad_kernel: pixel-contraction with
    transposed LDS tiles, ordered split sums) of ops.conv2d_split against double precision, three-term and one-term products; the
    library path (tests/library_paths.py) gives the same numbers to fp32 rounding."""
    torch.manual_seed(N + Cin + Cout + K + 1)
    conv = torch.nn.Conv2d(Cin, Cout, K, stride, pad, bias=bias).to(DEV).to(memory_format=torch.channels_last)
    x = torch.randn(N, Cin, H, H, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(Cin > 2)     # (the stem's input needs none)
    OH = (H + 2 * pad - K) // stride + 1
    g = torch.randn(N, Cout, OH, OH, device=DEV).contiguous(memory_format=torch.channels_last)
    ops.conv2d_split(x, conv, nterm=3).backward(g)
    gx = x.grad.clone() if Cin > 2 else None
    gw, gb = conv.weight.grad.clone(), (conv.bias.grad.clone() if bias else None)
    xd = x.detach().double().cpu().requires_grad_(Cin > 2)
    wd = conv.weight.detach().double().cpu().requires_grad_()
    bd = conv.bias.detach().double().cpu().requires_grad_() if bias else None
    F.conv2d(xd, wd, bd, stride, pad).backward(g.double().cpu())
    if Cin > 2:
        mag = torch.nn.grad.conv2d_input(xd.shape, wd.detach().abs(), g.double().cpu().abs(), stride, pad)
        assert float(((gx.cpu().double() - xd.grad).abs() / (mag + 1e-30)).max()) <= SPLIT3 + 4e-7
        assert rel_err(gx.cpu(), xd.grad) < 2e-5
    magw = torch.nn.grad.conv2d_weight(xd.detach().abs(), wd.shape, g.double().cpu().abs(), stride, pad)
    assert float(((gw.cpu().double() - wd.grad).abs() / (magw + 1e-30)).max()) <= SPLIT3 + 2e-6      # + fp32 sums over up to 131072 pixels
    assert rel_err(gw.cpu(), wd.grad) < 2e-5
    if bias:
        assert rel_err(gb.cpu(), bd.grad) < 1e-5
    # one-term products on bf16 maps: against fp32 arithmetic on the bf16-rounded operands
    conv.zero_grad()
    xb = x.detach().to(torch.bfloat16).requires_grad_(Cin > 2)
    ops.conv2d_split(xb, conv, nterm=1, out_bf16=True).backward(g.to(torch.bfloat16))
    xr = xb.detach().float().requires_grad_(Cin > 2)
    wr = conv.weight.detach().to(torch.bfloat16).float().requires_grad_()
    F.conv2d(xr, wr, None, stride, pad).backward(g.to(torch.bfloat16).float())
    if Cin > 2:
        assert float((xb.grad.float() - xr.grad).abs().max()) <= 2 ** -7 * float(xr.grad.abs().max())
    assert float((conv.weight.grad - wr.grad).abs().max()) <= 2e-5 * float(wr.grad.abs().max()) + 1e-6


@pytest.mark.parametrize('N,Cin,Cout,K,stride,pad,H,nterm', [(4, 64, 64, 3, 1, 1, 32, 3), (3, 64, 128, 3, 2, 1, 32, 3), (2, 64, 128, 1, 2, 0, 32, 1),
                                                            (5, 128, 32, 3, 1, 1, 16, 3), (2, 128, 128, 3, 1, 1, 16, 1)])
def test_prepared_weights_give_the_same_bits(N, Cin, Cout, K, stride, pad, H, nterm, monkeypatch):
    """psi_conv2d_prepare_weight + psi_conv2d_forward_p / _input_grad_p (the weight's bf16 parts written once per layer and step, in both
    layouts) against the kernels that round / split the fp32 weight tile in every workgroup (ops._conv2d_prepared_ok patched): the same parts, the same
    products in the same order — outputs and input gradients bit for bit."""
    torch.manual_seed(Cin + Cout + K)
    dt = torch.float32 if nterm == 3 else torch.bfloat16
    conv = torch.nn.Conv2d(Cin, Cout, K, stride, pad, bias=False).to(DEV).to(memory_format=torch.channels_last)
    x0 = torch.randn(N, Cin, H, H, device=DEV).to(dt).contiguous(memory_format=torch.channels_last)
    got = {}
    for prep in ('1', '0'):
        if prep == '0':
            library_paths.conv_weights_split_in_every_workgroup(monkeypatch)
        x = x0.clone().requires_grad_()
        y = ops.conv2d_split(x, conv, nterm=nterm, out_bf16=nterm == 1)
        conv.zero_grad()
        y.backward(torch.ones_like(y) * 0.37)
        got[prep] = (y.detach().clone(), x.grad.clone(), conv.weight.grad.clone())
    # (the prepared route of a stride-1 3x3 layer with 64 input channels at the fp32 model's precision is a kernel of its own — conv.hip:
    # conv3x3s_kernel, halo tile split once per workgroup — which adds the same three-term products in another order: fp32 rounding apart)
    own_kernel = nterm == 3 and K == 3 and stride == 1 and Cin == 64 and H % 32 == 0
    for a, b in zip(got['1'], got['0']):
        if own_kernel:
            assert float((a.float() - b.float()).abs().max()) <= 4e-6 * float(b.float().abs().max())
        else:
            assert torch.equal(a, b)


@pytest.mark.parametrize('C,H,N,relu,res,train', [(64, 64, 3, True, False, True), (64, 32, 4, True, True, True), (128, 16, 5, False, False, True),
                                                 (64, 32, 2, True, True, False), (128, 16, 1, False, False, False), (32, 16, 2, True, False, True)])
def test_bn_act_on_fp32_maps(C, H, N, relu, res, train):
    torch.manual_seed(C + H + N)
    bn = torch.nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    ref_bn = torch.nn.BatchNorm2d(C).to(DEV)
    ref_bn.load_state_dict(bn.state_dict())
    bn.train(train); ref_bn.train(train)
    x = (torch.randn(N, C, H, H, device=DEV) * 1.7 + 0.3).contiguous(memory_format=torch.channels_last).requires_grad_()
    r = torch.randn(N, C, H, H, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_() if res else None
    g = torch.randn(N, C, H, H, device=DEV)
    y = ops.bn_act_t(x, bn, relu=relu, residual=r)
    y.backward(g)
    got = (y.detach(), x.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(), r.grad.clone() if res else None)
    x.grad = None
    if res:
        r.grad = None
    yr = ref_bn(x)
    if res:
        yr = yr + r
    if relu:
        yr = F.relu(yr)
    yr.backward(g)
    want = (yr.detach(), x.grad, ref_bn.weight.grad, ref_bn.bias.grad, r.grad if res else None)
    for a, b, tol in zip(got, want, (2e-6, 2e-5, 2e-5, 2e-5, 1e-6)):
        if a is not None:
            assert rel_err(a.cpu(), b.cpu()) < tol
    assert rel_err(bn.running_mean.cpu(), ref_bn.running_mean.cpu()) < 1e-6 and rel_err(bn.running_var.cpu(), ref_bn.running_var.cpu()) < 1e-5
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked)


def test_maxpool_on_fp32_maps_is_exact():
    torch.manual_seed(1)
    x = torch.randn(3, 64, 64, 64, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    g = torch.randn(3, 64, 32, 32, device=DEV)
    y = ops.maxpool3x3s2_t(x)
    y.backward(g)
    gx = x.grad.clone()
    x.grad = None
    yr = F.max_pool2d(x, 3, 2, 1)
    yr.backward(g)
    assert torch.equal(y, yr) and torch.equal(gx, x.grad)
    with torch.no_grad():
        assert torch.equal(ops.maxpool3x3s2_t(x.detach()), yr)        # inference form: no window positions kept


@pytest.mark.parametrize('M,K,N,act,res', [(4, 75, 256, None, False), (128, 3, 256, None, False), (128, 72, 256, None, False), (200, 512, 512, 'leaky_relu', True),
                                          (128, 512, 75, None, False), (128, 32, 3, None, False), (128, 32768, 256, None, False), (7, 768, 768, 'leaky_relu', True),
                                          (128, 288, 32, None, False), (1, 8192, 256, None, False)])
def test_linear_act3_matches_double_precision(M, K, N, act, res):
    torch.manual_seed(M + K + N)
    x = torch.randn(M, K, device=DEV).requires_grad_()
    lin = torch.nn.Linear(K, N).to(DEV)
    r = torch.randn(M, N, device=DEV).requires_grad_() if res else None
    g = torch.randn(M, N, device=DEV)
    y = ops.linear_act3(x, lin.weight, lin.bias, act, 0.01, r)
    y.backward(g)
    got = (x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    xd, wd, bd = x.detach().double().cpu(), lin.weight.detach().double().cpu(), lin.bias.detach().double().cpu()
    pre = xd @ wd.t() + bd
    ref = F.leaky_relu(pre, 0.01) if act else pre
    if res:
        ref = ref + r.detach().double().cpu()
    mag = xd.abs() @ wd.abs().t() + bd.abs()
    err = (y.detach().cpu().double() - ref).abs()
    assert float((err / (mag + 1e-30)).max()) <= SPLIT3 + 4e-7, float((err / mag).max())
    assert rel_err(y.detach().cpu(), ref) < 2e-5
    # gradients (psi_linear_backward3: dX = G W, dW = G^T X, dbias with the same split products) against double precision
    gd = g.double().cpu() * (torch.where(pre > 0, 1.0, 0.01) if act else 1.0)
    for a, b in zip(got, (gd @ wd, gd.t() @ xd, gd.sum(0))):
        assert rel_err(a.cpu(), b) < 2e-5
    if res:
        assert torch.equal(r.grad, g)


# ------------------------------------------------------------------------------------------------------------------
# model level
# ------------------------------------------------------------------------------------------------------------------
T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)


def _load(m, seed):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_state_like(shapes, seed).items()}, strict=True)


def _count_calls(monkeypatch):
    """Counters on the hand-written ops: the model-level checks below must have gone THROUGH them (not through a library fallback)."""
    calls = {'conv': 0, 'bn': 0, 'lin': 0, 'pool': 0}
    for name, key in (('conv2d_split', 'conv'), ('bn_act_t', 'bn'), ('linear_act3', 'lin'), ('maxpool3x3s2_t', 'pool')):
        orig = getattr(ops, name)

        def wrapped(*a, _o=orig, _k=key, **kw):
            calls[_k] += 1
            return _o(*a, **kw)
        monkeypatch.setattr(ops, name, wrapped)
    return calls


def test_s1_fp32_model_on_the_hand_written_path_matches_the_reference(monkeypatch):
    g = golden('cvae')
    inp = synth.make_cvae_inputs(13, 4)
    calls = _count_calls(monkeypatch)
    for mode in ('eval', 'train'):
        m = models.HumanCVAES1(latentD=256, n_dim_body=75).to(DEV)
        _load(m, 0)
        getattr(m, mode)()
        with torch.no_grad():
            xr, mu, lv = m(T(inp['x75']), T(inp['xs']), eps=T(inp['eps32']))
        for a, k in ((xr, 'xrec'), (mu, 'mu'), (lv, 'logvar')):
            assert rel_err(a.cpu(), g['s1_%s_%s' % (mode, k)]) < 1e-4, (mode, k)          # BASELINE north_star: within 1e-4 rel fp32 (measured 0.6-3.2e-5)
    # every convolution (stem, 8 trunk 3x3, downsample, head = 11), every BatchNorm (10), the max-pool and all 14 dense layers, twice
    assert (calls['conv'], calls['bn'], calls['pool']) == (22, 20, 2) and calls['lin'] == 28, calls
    _load(m, 0)                                                 # (the training-mode pass above moved the running statistics)
    m.eval()
    with torch.no_grad():
        assert rel_err(m.sample(T(inp['xs']), eps=T(inp['eps32'])).cpu(), g['s1_sample']) < 1e-4


def test_s2_fp32_model_on_the_hand_written_path_matches_the_reference_and_the_library(monkeypatch):
    g = golden('cvae')
    inp = synth.make_cvae_inputs(13, 4)
    m = models.HumanCVAES2(latentD_g=256, latentD_l=256, n_dim_body=75).to(DEV)
    _load(m, 1)
    m.eval()
    calls = _count_calls(monkeypatch)
    args = (T(inp['x75']), T(inp['eps32']), T(inp['eps32b']), T(inp['xs']))
    with torch.no_grad():
        out = m(*args, use_eps=True)
    assert calls['conv'] == 22 and calls['bn'] == 20 and calls['pool'] == 2 and calls['lin'] == 29, calls      # two trunks; 14 + 15 dense layers
    for a, k in zip(out, ('s2_xrec', 's2_mu_g', 's2_lv_g', 's2_mu_l', 's2_lv_l')):
        assert rel_err(a.cpu(), g[k]) < 1e-4, k
    library_paths.fp32_models_on_the_library(monkeypatch)           # the library path (MIOpen / hipBLASLt fp32) on the same weights
    n0 = dict(calls)
    with torch.no_grad():
        lib = m(*args, use_eps=True)
    assert calls == n0
    for a, b in zip(out, lib):
        assert rel_err(a.cpu(), b.cpu()) < 1e-4


def test_training_mode_backward_on_the_hand_written_kernels(monkeypatch):
    """One training-mode forward + backward of HumanCVAES1 (batch statistics, running statistics updated).
    (1) The loss against the same model in DOUBLE precision on the CPU: no further from it than 4 x the all-library fp32 model is (+ 2e-5).
    (2) The BACKWARD kernels (psi_conv2d_input_grad / psi_conv2d_weight_grad / psi_linear_backward3) against the library's fp32 gradient
        kernels behind the SAME hand-written forward (tests/library_paths.py: aten.convolution_backward / matmul on the
        saved activations): every parameter gradient to 2e-4 of its largest entry.  Behind the same forward the ReLU masks and max-pool
        winners are the same, so this isolates the gradient arithmetic; comparing against a DIFFERENT forward (the library's, or double
        precision) measures something else — which near-zero activations fall on which side of zero: the library's own fp32 trunk gradients
        sit up to 4e-3 from the double-precision ones at this batch of 4, the three-term forward (1e-5 instead of 1e-6 per layer) up to
        7e-2 — the bounds of test_training_gpu.py::test_cal_loss_golden (3e-2 / 2e-3 against the reference's recorded gradients at its own
        batch) are what holds the end-to-end gradients."""
    inp = synth.make_cvae_inputs(13, 4)
    res = {}
    for mode in ('hand', 'lib_bwd', 'lib', 'f64'):
        dev, dt = ('cpu', torch.float64) if mode == 'f64' else (DEV, torch.float32)
        with monkeypatch.context() as mp_:
            if mode in ('f64', 'lib'):
                library_paths.fp32_models_on_the_library(mp_)
            if mode == 'lib_bwd':
                library_paths.fp32_backward_on_the_library(mp_)
            m = models.HumanCVAES1(latentD=256, n_dim_body=75)
            _load(m, 0)
            m = m.to(dev).to(dt)
            m.train()
            t = lambda a: torch.tensor(np.asarray(a), dtype=dt, device=dev)
            xr, mu, lv = m(t(inp['x75']), t(inp['xs']), eps=t(inp['eps32']))
            loss = (xr - t(inp['x75'])).abs().mean() + 0.1 * (mu ** 2 + lv.exp() - lv).mean()
            loss.backward()
        res[mode] = (float(loss), {k: p.grad.detach().double().cpu().contiguous() for k, p in m.named_parameters()},
                     {k: b.detach().double().cpu() for k, b in m.named_buffers()})
    assert abs(res['hand'][0] - res['f64'][0]) <= 4 * abs(res['lib'][0] - res['f64'][0]) + 2e-5 * abs(res['f64'][0])
    assert res['hand'][0] == res['lib_bwd'][0]                       # the same forward
    worst = max((rel_err(res['hand'][1][k], g), k) for k, g in res['lib_bwd'][1].items())
    assert worst[0] < 2e-4, worst
    for k, b in res['f64'][2].items():
        assert rel_err(res['hand'][2][k], b) < 2e-5, k
