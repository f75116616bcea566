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
    # ... and the consumer really steps: torch.optim.Adam increments every step entry it is handed, so the psi state_dict must not alias them
    ps_o = other.param_groups[0]['params']
    for p, r in zip(ps_o, ref):
        p.data.copy_(r.detach())
    _grads(ps_o, 32)
    other.step()
    assert all(float(other.state[p]['step']) == 4.0 for p in ps_o)              # 3 + 1, not 3 + (number of aliased entries)


def test_state_from_another_layout_and_separate_counters_are_adopted():
    """A checkpoint written while a convolution weight was contiguous (before the CVAE switched its trunk to channels_last, or by torch.optim.Adam)
    loaded into a model whose weight is channels_last NOW: the moments are re-laid out at the next step instead of the step failing; separate
    per-parameter step tensors are unified by load_state_dict itself, so that a graph capture right after a resume works."""
    src = [torch.nn.Parameter(torch.randn(16, 8, 3, 3, device=DEV)), torch.nn.Parameter(torch.randn(40, device=DEV))]        # contiguous 4-D weight
    osrc = torch.optim.Adam(src, lr=1e-3, fused=True)
    _grads(src, 5)
    src[0].grad = src[0].grad.contiguous()
    osrc.step()
    sd = copy.deepcopy(osrc.state_dict())
    dst = [torch.nn.Parameter(src[0].detach().clone().contiguous(memory_format=torch.channels_last)), torch.nn.Parameter(src[1].detach().clone())]
    od = psi_optim.Adam(dst, lr=1e-3)
    od.load_state_dict(sd)
    assert len({od.state[p]['step'].data_ptr() for p in dst}) == 1                    # unified eagerly
    assert od.state[dst[0]]['exp_avg'].stride() != dst[0].stride()                   # still in the checkpoint's layout
    for p, q in zip(dst, src):
        p.grad = torch.randn_like(p)
        q.grad = p.grad.detach().clone().contiguous()
    static = [p.grad for p in dst]
    od.step()                                                                         # re-lays the moments out, steps on the HIP kernel
    osrc.step()
    assert od.hip_steps == 1 and od.state[dst[0]]['exp_avg'].stride() == dst[0].stride()
    for a, b in zip(dst, src):
        assert float((a.detach() - b.detach()).abs().max()) <= 2.0 ** -22 * (float(b.detach().abs().max()) + 1e-3)
    # resume + capture: a second optimiser takes a state with SEPARATE counters and is captured without an eager step in between
    od2 = psi_optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in dst], lr=1e-3)
    od2.load_state_dict(copy.deepcopy(osrc.state_dict()))
    ps2 = od2.param_groups[0]['params']
    for p, s in zip(ps2, static):
        p.grad = s.clone()
    od2.step()                                                                        # (first step re-lays out; then the capture)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        od2.step()
    graph.replay()
    torch.cuda.synchronize()
    assert float(od2.state[ps2[0]]['step']) == 4.0


def test_group_outside_the_kernels_coverage_keeps_its_own_counters():
    """A group with a parameter that is neither contiguous nor channels_last takes torch's update: its counters must stay per-parameter tensors
    (never the shared alias, which torch would increment once per parameter)."""
    base = torch.randn(12, 20, device=DEV)
    ps = [torch.nn.Parameter(base.t()), torch.nn.Parameter(torch.randn(30, device=DEV))]       # a transposed view: not dense in either layout
    o = psi_optim.Adam(ps, lr=1e-3)
    for it in range(3):
        for p in ps:
            p.grad = torch.randn_like(p)
        o.step()
    assert o.hip_steps == 0
    assert [float(o.state[p]['step']) for p in ps] == [3.0, 3.0]
    # a state the HIP kernel stepped (aliased counters) continued by torch's update: de-aliased first
    qs = [torch.nn.Parameter(torch.randn(8, 4, device=DEV)), torch.nn.Parameter(torch.randn(30, device=DEV))]
    o2 = psi_optim.Adam(qs, lr=1e-3)
    for p in qs:
        p.grad = torch.randn_like(p)
    o2.step()
    assert o2.hip_steps == 1 and len({o2.state[p]['step'].data_ptr() for p in qs}) == 1
    o2.param_groups[0]['amsgrad'] = False
    o2.param_groups[0]['maximize'] = True                                             # outside the kernel's coverage from now on
    for p in qs:
        p.grad = torch.randn_like(p)
    o2.step()
    assert [float(o2.state[p]['step']) for p in qs] == [2.0, 2.0]
