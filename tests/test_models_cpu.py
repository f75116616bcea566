"""CVAE S1/S2 (psi-release_amd/models.py) against the forward passes recorded from the reference's own modules
(tests/golden/cvae.npz, oracle/make_golden_cvae.py): identical state_dict keys/shapes (checkpoint layout, SURVEY
Appendix B) and outputs within fp32 round-off.  Pure PyTorch modules -> checked on CPU here and on the GPU in -m gpu."""
import numpy as np
import pytest
import torch

from conftest import golden, rel_err
from psi_release_amd import models, synth

T = lambda a, dev='cpu': torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)


def _load(m, seed):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_state_like(shapes, seed).items()}, strict=True)
    return shapes


def _check_s1(dev, tol):
    g = golden('cvae')
    inp = synth.make_cvae_inputs(13, 4)
    m = models.HumanCVAES1(latentD=256, n_dim_body=75).to(dev)
    shapes = _load(m, 0)
    assert list(shapes.keys()) == list(g['s1_keys']) and [str(s) for s in shapes.values()] == list(g['s1_shapes'])
    assert sum(p.numel() for p in m.parameters()) == int(g['s1_nparams']) == 5014699
    for mode in ('eval', 'train'):
        getattr(m, mode)()
        with torch.no_grad():
            xr, mu, lv = m(T(inp['x75'], dev), T(inp['xs'], dev), eps=T(inp['eps32'], dev))
        assert rel_err(xr.cpu(), g['s1_%s_xrec' % mode]) < tol
        assert rel_err(mu.cpu(), g['s1_%s_mu' % mode]) < tol and rel_err(lv.cpu(), g['s1_%s_logvar' % mode]) < tol
        _load(m, 0)
    m.eval()
    with torch.no_grad():
        assert rel_err(m.sample(T(inp['xs'], dev), eps=T(inp['eps32'], dev)).cpu(), g['s1_sample']) < tol
        xl, e = m.sample_line(T(inp['xs'], dev))
    assert xl.shape == (4, 75) and torch.allclose(e[:, 0].cpu(), torch.tensor([-3.0, -1.5, 0.0, 1.5]))


def _check_s2(dev, tol):
    g = golden('cvae')
    inp = synth.make_cvae_inputs(13, 4)
    m = models.HumanCVAES2(latentD_g=256, latentD_l=256, n_dim_body=75).to(dev)
    shapes = _load(m, 1)
    assert list(shapes.keys()) == list(g['s2_keys']) and [str(s) for s in shapes.values()] == list(g['s2_shapes'])
    assert sum(p.numel() for p in m.parameters()) == int(g['s2_nparams']) == 15705067
    m.eval()
    with torch.no_grad():
        out = m(T(inp['x75'], dev), T(inp['eps32'], dev), T(inp['eps32b'], dev), T(inp['xs'], dev), use_eps=True)
    for a, k in zip(out, ('s2_xrec', 's2_mu_g', 's2_lv_g', 's2_mu_l', 's2_lv_l')):
        assert rel_err(a.cpu(), g[k]) < tol, k
    m2 = models.HumanCVAES2(latentD_g=256, latentD_l=256, n_dim_body=75, test=True).to(dev)
    m2.load_state_dict(m.state_dict())
    m2.eval()
    with torch.no_grad():
        assert m2.sample(T(inp['xs'], dev)).shape == (4, 75)


def test_s1_cpu():
    _check_s1('cpu', 1e-5)


def test_s2_cpu():
    _check_s2('cpu', 1e-5)


@pytest.mark.gpu
def test_s1_gpu():
    _check_s1('cuda', 2e-4)


@pytest.mark.gpu
def test_s2_gpu():
    _check_s2('cuda', 2e-4)


@pytest.mark.gpu
def test_s2_bf16_autocast_runs_and_is_close():
    inp = synth.make_cvae_inputs(13, 4)
    m = models.HumanCVAES2(latentD_g=256, latentD_l=256, n_dim_body=75, autocast_bf16=True).to('cuda')
    _load(m, 1)
    m.eval()
    g = golden('cvae')
    with torch.no_grad():
        out = m(T(inp['x75'], 'cuda'), T(inp['eps32'], 'cuda'), T(inp['eps32b'], 'cuda'), T(inp['xs'], 'cuda'), use_eps=True)
    assert out[0].dtype == torch.float32
    assert rel_err(out[0].cpu(), g['s2_xrec']) < 5e-2           # bf16 trunk: loose by construction
