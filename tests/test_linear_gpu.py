"""The hand-written bf16-MFMA dense layers (csrc/linear.hip, ops.linear_act) against a plain PyTorch fp32 reference of the same
op evaluated on bf16-rounded operands (the kernel rounds x, W and the incoming gradient to bf16 as it loads them and accumulates in
fp32), and the CVAEs built on them against the reference's golden outputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import library_paths
from conftest import golden, rel_err
from psi_release_amd import models, ops, synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'
r16 = lambda t: t.to(torch.bfloat16).float()


def _case(M, N, K, seed, xb=False):
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV) * 0.1
    res = torch.randn(M, N, generator=g).to(DEV)
    gy = torch.randn(M, N, generator=g).to(DEV)
    if xb:
        x = x.to(torch.bfloat16)
    return x, W, b, res, gy


@pytest.mark.parametrize('M,N,K', [(128, 512, 512), (128, 768, 768), (4, 32, 32), (100, 48, 64), (128, 256, 8192), (128, 256, 32768), (130, 128, 544),
                                   (128, 1536, 1536), (128, 32, 1024), (200, 1024, 1024)])
@pytest.mark.parametrize('mode', ['plain', 'leaky', 'leaky_res'])
@pytest.mark.parametrize('bwd', ['hip', 'library'])
def test_linear_act_forward_backward(M, N, K, mode, bwd, monkeypatch):
    # bwd='hip': the hand-written dX / dW kernels (fp32 outputs: exact to the rounded-operand reference); 'library' (default): the
    # backward GEMMs go to hipBLASLt in bf16 (bf16 outputs, like the autocast path this replaces)
    if bwd == 'library':
        library_paths.bf16_dense_backward_on_the_library(monkeypatch)
    btol = 2e-5 if bwd == 'hip' else 1e-2
    x, W, b, res, gy = _case(M, N, K, M + N + K)
    act = None if mode == 'plain' else 'leaky_relu'
    use_res = mode == 'leaky_res'
    xt, Wt, bt, rt = x.clone().requires_grad_(), W.clone().requires_grad_(), b.clone().requires_grad_(), res.clone().requires_grad_()
    y = ops.linear_act(xt, Wt, bt, act, 0.01, residual=rt if use_res else None)
    (y * gy).sum().backward()
    # reference: same rounding points, fp32 everywhere else
    xr, Wr, br, rr = x.clone().requires_grad_(), W.clone().requires_grad_(), b.clone().requires_grad_(), res.clone().requires_grad_()
    pre = r16(xr.detach()) @ r16(Wr.detach()).T + br.detach()
    a = F.leaky_relu(pre, 0.01) if act else pre
    y_ref = a + rr.detach() if use_res else a
    assert rel_err(y.detach().cpu(), y_ref.cpu()) < 2e-5
    G = gy * (torch.where(pre > 0, 1.0, 0.01) if act else 1.0)
    gx_ref = r16(G) @ r16(W)
    gW_ref = r16(G).T @ r16(x)
    assert rel_err(xt.grad.cpu(), gx_ref.cpu()) < btol
    assert rel_err(Wt.grad.cpu(), gW_ref.cpu()) < btol
    assert rel_err(bt.grad.cpu(), G.sum(0).cpu()) < 2e-5
    if use_res:
        assert torch.equal(rt.grad, gy)
    # and against the unrounded fp32 op, at bf16 accuracy
    full = F.linear(x, W, b)
    full = (F.leaky_relu(full, 0.01) if act else full) + (res if use_res else 0)
    assert rel_err(y.detach().cpu(), full.cpu()) < 2e-2


def test_linear_act_bf16_input_and_determinism(monkeypatch):
    x, W, b, res, gy = _case(128, 256, 32768, 5, xb=True)
    xt, Wt = x.clone().requires_grad_(), W.clone().requires_grad_()
    y = ops.linear_act(xt, Wt, b)
    (y * gy).sum().backward()
    y_ref = x.float() @ r16(W).T + b
    assert rel_err(y.detach().cpu(), y_ref.cpu()) < 2e-5
    assert xt.grad.dtype == torch.bfloat16
    assert rel_err(xt.grad.float().cpu(), (r16(gy) @ r16(W)).cpu()) < 1e-2          # the gradient itself is stored as bf16
    assert rel_err(Wt.grad.cpu(), (r16(gy).T @ x.float()).cpu()) < 2e-5
    for _ in range(3):                                                                 # split-K partials are summed in a fixed order
        assert torch.equal(ops.linear_act(x, W, b), y.detach())
    with pytest.raises(ValueError):
        ops.linear_act(x[:, :100], W[:, :100], b)


def _load(m, seed):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_state_like(shapes, seed).items()})


@pytest.mark.parametrize('stage', ['s1', 's2'])
def test_cvae_on_hip_linear_matches_reference_golden(stage, monkeypatch):
    """HumanCVAES1 / S2 with the bf16 trunk and the dense layers on the hand-written MFMA kernels vs the reference model's fp32
    outputs (tests/golden/cvae.npz): same tolerance as the PyTorch bf16 path, and closer to fp32 than that path in the dense part."""
    g = golden('cvae')
    inp = synth.make_cvae_inputs(13, 4)
    T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)
    out = {}
    def run():
        if stage == 's1':
            m = models.HumanCVAES1(latentD=256, n_dim_body=75, autocast_bf16=True).to(DEV)
            _load(m, 0)
            m.eval()
            with torch.no_grad():
                o = m(T(inp['x75']), T(inp['xs']), eps=T(inp['eps32']))
        else:
            m = models.HumanCVAES2(latentD_g=256, latentD_l=256, n_dim_body=75, autocast_bf16=True).to(DEV)
            _load(m, 1)
            m.eval()
            with torch.no_grad():
                o = m(T(inp['x75']), T(inp['eps32']), T(inp['eps32b']), T(inp['xs']), use_eps=True)
        return o[0].float().cpu().numpy()
    out['hip'] = run()
    with monkeypatch.context() as mp_:
        library_paths.bf16_dense_layers_on_the_library(mp_)          # nn.Linear as PyTorch runs it
        out['torch'] = run()
    ref = g['s1_eval_xrec' if stage == 's1' else 's2_xrec']
    assert rel_err(out['hip'], ref) < 5e-2
    assert rel_err(out['hip'], out['torch']) < 5e-2
    assert not np.array_equal(out['hip'], out['torch'])                    # the two paths are really different code


def test_hip_linear_policy(monkeypatch):
    """A model built for bf16 takes the hand-written dense kernels in training and in no_grad mode alike (one arithmetic for both)."""
    calls = []
    real = ops.linear_act
    monkeypatch.setattr(ops, 'linear_act', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    rb = models.ResBlock(512).to(DEV)
    x = torch.randn(8, 512, device=DEV)
    rb(x)
    assert len(calls) == 0                                                  # not a bf16 model: library path
    models.set_hip_linear(rb, True)
    with torch.no_grad():
        y2 = rb(x)
    assert len(calls) == 2
    y = rb(x.requires_grad_())
    y.sum().backward()
    assert len(calls) == 4 and rb.fc1.weight.grad is not None and x.grad is not None
    assert torch.equal(y2, y.detach())                                      # one arithmetic for both modes
    models.set_hip_linear(rb, False)
    with torch.no_grad():
        rb(x)
    assert len(calls) == 4
