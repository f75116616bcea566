"""The trainers' optimiser (train_s1.py:229 / train_s2.py:295-296: optim.Adam(model_h.parameters(), lr)) on the hand-written multi-tensor kernel
(csrc/adam.hip: psi_adam_step, psi_release_amd/optim.py) against torch.optim.Adam on the same GPU: the fused implementation it replaces
(same operation order: equal to the last bit or one ulp of the update) and the plain per-tensor one (fp32 rounding apart); tensors of every
kind the models hold — 16-byte-unaligned views, channels_last convolution weights, single elements, more than one workgroup's chunk, more
tensors than one launch takes (80) — weight decay, hipGraph capture of the step, and the state_dict round trip in both directions."""
import copy

import pytest
import torch

from psi_release_amd import optim as psi_optim

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _params(seed, many=False):
    g = torch.Generator().manual_seed(seed)
    shapes = [(1,), (3,), (64,), (75, 3), (512, 512), (1024, 700), (64, 2, 7, 7), (128, 64, 3, 3), (8193,), (20000,)]
    if many:
        shapes = shapes + [(17 + i,) for i in range(90)]
    ps = []
    for s in shapes:
        t = torch.randn(*s, generator=g)
        if len(s) == 4:
            t = t.contiguous(memory_format=torch.channels_last)
        ps.append(t)
    flat = torch.randn(1000, generator=g)
    return ps, flat


def _make(seed, many=False):
    ps, flat = _params(seed, many)
    out = [torch.nn.Parameter(p.to(DEV).contiguous(memory_format=torch.channels_last) if p.dim() == 4 else p.to(DEV)) for p in ps]
    base = flat.to(DEV)
    out.append(torch.nn.Parameter(base[1:998]))                     # a view at a 4-byte offset: no 16-byte accesses for this one
    return out


def _grads(params, seed):
    g = torch.Generator().manual_seed(seed)
    for p in params:
        gr = torch.randn(p.shape, generator=g).to(DEV) * (1.0 + p.detach().abs())
        p.grad = gr.contiguous(memory_format=torch.channels_last) if p.dim() == 4 else gr


@pytest.mark.parametrize('wd,many', [(0.0, False), (0.01, False), (0.0, True)])
def test_step_equals_torch_adam(wd, many):
    mine, fused, plain = _make(1, many), _make(1, many), _make(1, many)
    om = psi_optim.Adam(mine, lr=3e-3, weight_decay=wd)
    of = torch.optim.Adam(fused, lr=3e-3, weight_decay=wd, fused=True)
    op = torch.optim.Adam(plain, lr=3e-3, weight_decay=wd, foreach=False)
    for it in range(7):
        for ps, o in ((mine, om), (fused, of), (plain, op)):
            _grads(ps, 100 + it)
            o.step()
    assert om.hip_steps == 7
    for pa, pb, pc in zip(mine, fused, plain):
        a, b, c = pa.detach(), pb.detach(), pc.detach()
        scale = float(b.abs().max()) + 1e-3
        assert float((a - b).abs().max()) <= 2.0 ** -22 * scale, (tuple(a.shape), float((a - b).abs().max()))      # <= 2 ulp of the largest element
        assert float((a - c).abs().max()) <= 1e-6 * scale + 3e-3 * 1e-5, tuple(a.shape)                             # the plain fp32 operator sequence
        sa, sb = om.state[pa], of.state[pb]
        assert float(sa['step']) == 7.0 == float(sb['step'])
        for k in ('exp_avg', 'exp_avg_sq'):
            assert sa[k].stride() == a.stride()
            assert float((sa[k] - sb[k]).abs().max()) <= 2.0 ** -22 * (float(sb[k].abs().max()) + 1e-30), (k, tuple(a.shape))
    assert len({om.state[p]['step'].data_ptr() for p in mine}) == 1                                                  # one shared device counter


def test_step_in_a_hip_graph_and_state_dict_round_trip():
    mine, ref = _make(2), _make(2)
    om = psi_optim.Adam(mine, lr=1e-3)
    orf = torch.optim.Adam(ref, lr=1e-3, fused=True)
    _grads(mine, 7)
    _grads(ref, 7)
    om.step()                                                       # eager first step: state and the shared counter exist before the capture
    orf.step()
    static = [p.grad for p in mine]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        om.step()
    # (the capture itself does not execute: two replays = steps 2 and 3)
    for it in range(2):
        _grads(ref, 20 + it)
        for s, r in zip(static, ref):
            s.copy_(r.grad)
        graph.replay()
        orf.step()
    torch.cuda.synchronize()
    assert float(om.state[mine[0]]['step']) == 3.0
    for a, b in zip(mine, ref):
        assert float((a.detach() - b.detach()).abs().max()) <= 2.0 ** -22 * (float(b.detach().abs().max()) + 1e-3), tuple(a.shape)
    # ---- checkpoints travel both ways
    sd = copy.deepcopy(om.state_dict())
    other = torch.optim.Adam(_make(2), lr=1e-3, fused=True)
    other.load_state_dict(sd)                                       # torch.optim.Adam reads a psi checkpoint
    assert float(other.state[other.param_groups[0]['params'][3]]['step']) == 3.0
    back = psi_optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in ref], lr=1e-3)
    back.load_state_dict(copy.deepcopy(orf.state_dict()))          # and the other way: separate step tensors are unified at the next step
    cont = back.param_groups[0]['params']
    _grads(cont, 31)
    _grads(ref, 31)
    back.step()
    orf.step()
    assert back.hip_steps == 1 and float(back.state[cont[0]]['step']) == 4.0
    for a, b in zip(cont, ref):
        assert float((a.detach() - b.detach()).abs().max()) <= 2.0 ** -22 * (float(b.detach().abs().max()) + 1e-3), tuple(a.shape)
