"""GPU parity of the fused body-vector glue of the training step (csrc/cvae_loss.hip: psi_cvae_target, psi_cvae_losses_forward /
_backward) against the operator sequence it replaces — GeometryTransformer (cvae.py:118-199, itself pinned to the reference's fixtures in
test_geometry_cpu.py) and the loss expressions of cal_loss (train_s2.py:119-139) under autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from psi_release_amd import ops, synth, training
from psi_release_amd.geometry import GeometryTransformer

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _batch(B, seed):
    rs = np.random.RandomState(seed)
    body = synth.body_vector_72(synth.make_bodies(seed, B))
    body[:, 2] = np.abs(body[:, 2]) + 1.5
    body[0, 3:6] = [3e-4, -2e-4, 1e-4]                     # theta^2 < 1e-6: the first-order branch of the axis-angle conversion
    if B > 1:
        body[1, 3:6] = 0.0
    cam_int = np.asarray(synth.make_bodies(seed, B)['cam_int'], dtype=np.float32)
    T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)
    return T(body), T(cam_int), T(rs.uniform(4.0, 8.0, B))


@pytest.mark.parametrize('B', [1, 7, 128])
def test_target_matches_the_operator_sequence(B):
    xh, cam_int, max_d = _batch(B, 3)
    ref = GeometryTransformer.convert_to_6D_rot(GeometryTransformer.normalize_global_T(xh, cam_int, max_d))
    got = ops.cvae_target(xh, cam_int, max_d)
    assert got.shape == (B, 75)
    assert torch.equal(got[:, 9:], xh[:, 6:])
    assert (got - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item())


def _reference_losses(rec, tgt, xh, cam_int, max_d, lat, fca, w):
    xh_rec = GeometryTransformer.recover_global_T(rec, cam_int, max_d)
    l_t = w[0] * (0.5 * F.l1_loss(rec[:, :3], tgt[:, :3]) + 0.5 * F.l1_loss(xh_rec[:, :3], xh[:, :3]))
    l_p = w[0] * F.l1_loss(rec[:, 3:], tgt[:, 3:])
    kls = [fca ** 2 * w[1] * 0.5 * torch.mean(torch.exp(lv) + mu ** 2 - 1.0 - lv) for mu, lv in lat]
    l_v = w[2] * torch.mean(xh_rec[:, 19:51] ** 2)
    return xh_rec, [l_t, l_p] + kls + [l_v]


@pytest.mark.parametrize('B,nlat,fca_kind', [(4, 2, 'float'), (128, 2, 'tensor'), (33, 1, 'float')])
def test_losses_and_gradients_match_autograd(B, nlat, fca_kind):
    torch.manual_seed(B)
    xh, cam_int, max_d = _batch(B, 5)
    tgt = ops.cvae_target(xh, cam_int, max_d)
    w = (1.0, 0.1, 1e-3)
    fca = 0.62
    coef = torch.randn(5, device=DEV).abs() + 0.5                      # the trainer sums the losses; any weighting must come out right
    Wx = torch.randn(B, 75, device=DEV) * 0.01                         # stands for what the scene losses send back through xh_rec

    def run(fused):
        rec = (tgt + 0.1 * torch.randn(B, 75, device=DEV, generator=torch.Generator(DEV).manual_seed(1))).requires_grad_(True)
        lat = []
        for k in range(nlat):
            g = torch.Generator(DEV).manual_seed(10 + k)
            lat.append(((0.3 * torch.randn(B, 32 * (k + 1), device=DEV, generator=g)).requires_grad_(True),
                        (0.2 * torch.randn(B, 32 * (k + 1), device=DEV, generator=g)).requires_grad_(True)))
        if fused:
            f = torch.tensor(fca, device=DEV) if fca_kind == 'tensor' else fca
            flat = [t for pair in lat for t in pair] + [None] * (4 - 2 * nlat)
            xh_rec, L = ops.cvae_losses(rec, tgt, xh, cam_int, max_d, *flat, fca=f, w_rec=w[0], w_kl=w[1], w_vposer=w[2])
            Ls = list(L.unbind(0))
            if nlat == 1:
                assert float(Ls[3].detach()) == 0.0
                Ls = Ls[:3] + Ls[4:]
        else:
            xh_rec, Ls = _reference_losses(rec, tgt, xh, cam_int, max_d, lat, fca, w)
        total = sum(c * l for c, l in zip(coef, Ls)) + (xh_rec * Wx).sum()
        total.backward()
        return xh_rec.detach(), [float(l.detach()) for l in Ls], [rec.grad] + [t.grad for pair in lat for t in pair]

    xr_a, L_a, g_a = run(True)
    xr_b, L_b, g_b = run(False)
    assert (xr_a - xr_b).abs().max().item() < 1e-5 * max(1.0, xr_b.abs().max().item())
    for a, b in zip(L_a, L_b):
        assert abs(a - b) < 1e-5 * max(abs(b), 1e-3), (L_a, L_b)
    for a, b in zip(g_a, g_b):
        assert (a - b).abs().max().item() < 1e-5 * max(b.abs().max().item(), 1e-6)


def test_losses_without_downstream_use_of_xh_rec():
    """Scene losses gated off (the first 75 % of the epochs): nothing flows back through xh_rec."""
    B = 8
    xh, cam_int, max_d = _batch(B, 9)
    tgt = ops.cvae_target(xh, cam_int, max_d)
    rec = (tgt + 0.05).requires_grad_(True)
    mu = torch.zeros(B, 16, device=DEV, requires_grad=True)
    lv = torch.zeros(B, 16, device=DEV, requires_grad=True)
    _, L = ops.cvae_losses(rec, tgt, xh, cam_int, max_d, mu, lv, fca=1.0)
    L[:3].sum().backward()
    assert float(L[2].detach()) == 0.0 and mu.grad.abs().max().item() == 0.0 and lv.grad.abs().max().item() == 0.0
    assert abs(float(L[1].detach()) - 0.05) < 1e-6
    assert torch.isfinite(rec.grad).all()


@pytest.mark.parametrize('stage', ['s1', 's2'])
def test_cal_loss_is_the_same_with_and_without_the_fused_glue(tmp_path, smplx_data, vposer_sd, stage, monkeypatch):
    from test_training_gpu import LW, make_cfg, _load
    B = 4
    scene = synth.make_scene(2, 4096, 32, 256)
    inp = synth.make_cvae_inputs(13, B)
    T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)
    xh, cam_int, max_d = _batch(B, 21)
    cls, seed = (training.TrainOP, 0) if stage == 's1' else (training.TrainOPS2, 1)
    op = cls(make_cfg(tmp_path, smplx_data, vposer_sd, scene, B), dict(LW))
    args = dict(xs=T(inp['xs']), xh=xh, cam_ext=T(synth.make_cam_ext(0, B)), cam_int=cam_int, max_d=max_d,
                scene_verts=T(scene.verts)[None].repeat(B, 1, 1), scene_face=None,
                s_grid_min_batch=T(scene.grid_min)[None].repeat(B, 1), s_grid_max_batch=T(scene.grid_max)[None].repeat(B, 1),
                s_grid_sdf_batch=(T(scene.sdf)[None].contiguous(), torch.zeros(B, dtype=torch.int32, device=DEV), T(scene.grid_min)[None],
                                  T(scene.grid_max)[None], ops.SceneSet(T(scene.verts)[None])))
    out = {}
    for glue in ('1', '0'):
        op.fused_glue = glue == '1'
        _load(op.model_h, seed)
        op.model_h.train()
        op.model_h.zero_grad()
        if stage == 's1':
            losses = op.cal_loss(ep=90, eps=T(inp['eps32']), **args)
        else:
            losses = op.cal_loss(eps_g=T(inp['eps32']), eps_l=T(inp['eps32b']), ep=90, use_eps=True, **args)
        sum(losses).backward()
        out[glue] = ([float(l) for l in losses], {k: p.grad.clone() for k, p in op.model_h.named_parameters() if p.grad is not None})
    for a, b in zip(out['1'][0], out['0'][0]):
        assert abs(a - b) < 1e-5 * max(abs(b), 1e-3), out
    dev = {k: (out['1'][1][k] - g).abs().max().item() / max(g.abs().max().item(), 1e-8) for k, g in out['0'][1].items()}
    # fp32 model; the trunk sits below train-mode BatchNorm layers at batch 4 and the library's weight-gradient kernels sum with atomics:
    # two runs of the SAME path differ there at the percent level (test_training_gpu.py allows 3e-2 against the reference for resnet.0)
    assert max(v for k, v in dev.items() if 'resnet' not in k) < 2e-3, sorted(dev.items(), key=lambda kv: -kv[1])[:5]
    assert max(v for k, v in dev.items() if 'resnet' in k) < 3e-2, sorted(dev.items(), key=lambda kv: -kv[1])[:5]


@pytest.mark.parametrize('B,penetrating', [(3, True), (16, True), (5, False)])
def test_scene_losses_match_the_operator_sequence(smplx_data, B, penetrating):
    """ops.scene_losses (csrc/scene_loss.hip) against chamfer_to_scenes + sdf_sample + penetration_loss and the elementwise expressions of
    train_s1.py:171-177, 193-204 under autograd: values and the gradient with respect to the body vertices."""
    n_scenes, m, D, n_c = 2, 4096, 32, 300
    sc = [synth.make_scene(i, m, D, n_c) for i in range(n_scenes)]
    T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)
    table = torch.stack([T(s.verts) for s in sc])
    sdf = torch.stack([T(s.sdf) for s in sc]).contiguous()
    gmin, gmax = torch.stack([T(s.grid_min) for s in sc]), torch.stack([T(s.grid_max) for s in sc])
    scenes = ops.SceneSet(table)
    slot = torch.tensor([b % n_scenes for b in range(B)], dtype=torch.int32, device=DEV)
    rs = np.random.RandomState(B)
    V = 2000
    lo, hi = sc[0].grid_min, sc[0].grid_max
    verts0 = T(rs.uniform(lo, hi, (B, V, 3)))
    if not penetrating:
        sdf = sdf.abs() + 0.01
    vid = torch.tensor(rs.choice(V, n_c, replace=False), dtype=torch.int64, device=DEV)
    vid[1] = vid[0]                                        # a vertex listed twice (two contact parts sharing it) accumulates both
    w_c, w_p = 0.1, 0.3
    coef = (1.7, 0.6)

    def run(fused):
        verts = verts0.clone().requires_grad_(True)
        if fused:
            l_c, l_p = ops.scene_losses(verts, vid, scenes, slot, sdf, gmin, gmax, True, w_c, w_p, 1.0)
        else:
            d = ops.chamfer_to_scenes(verts[:, vid, :].contiguous(), scenes, slot)
            s = torch.sqrt(d + 1e-4)
            l_c = 1.0 * w_c * torch.mean(s / (s + 1.0))
            vals = ops.sdf_sample(verts, sdf, gmin, gmax, scene_id=slot, align_corners=True)
            l_p = 1.0 * w_p * ops.penetration_loss(vals)
        (coef[0] * l_c + coef[1] * l_p).backward()
        return float(l_c.detach()), float(l_p.detach()), verts.grad

    a, b = run(True), run(False)
    assert abs(a[0] - b[0]) < 1e-6 * max(abs(b[0]), 1e-3) and abs(a[1] - b[1]) < 1e-6 * max(abs(b[1]), 1e-3), (a[:2], b[:2])
    assert (b[1] > 0) == penetrating
    assert (a[2] - b[2]).abs().max().item() < 1e-5 * b[2].abs().max().item()
    # fixed-order reductions: the same call twice gives the same bits
    a2 = run(True)
    assert a2[0] == a[0] and a2[1] == a[1]


def test_scene_loss_backward_and_penetration_statistics_repeat_bit_for_bit_with_repeated_contact_ids():
    """SURVEY section 5 / VERDICT r03 #8: no floating-point atomics on the training path.  Contact ids in which several vertices are listed two,
    three and five times (cvae.py:99-115 keeps duplicates; an atomic scatter adds their terms in arrival order): twenty backward passes of
    ``ops.scene_losses`` give ONE set of bits, equal to the ordered sum formed on the host; ``psi_sdf_penetration_stats`` likewise."""
    n_scenes, m, D, n_c, B, V = 2, 4096, 32, 512, 24, 3000
    sc = [synth.make_scene(i, m, D, n_c) for i in range(n_scenes)]
    T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)
    table = torch.stack([T(s.verts) for s in sc])
    sdf = torch.stack([T(s.sdf) for s in sc]).contiguous()
    gmin, gmax = torch.stack([T(s.grid_min) for s in sc]), torch.stack([T(s.grid_max) for s in sc])
    scenes = ops.SceneSet(table)
    slot = torch.tensor([b % n_scenes for b in range(B)], dtype=torch.int32, device=DEV)
    rs = np.random.RandomState(5)
    verts0 = T(rs.uniform(sc[0].grid_min, sc[0].grid_max, (B, V, 3)))
    ids = rs.choice(V, n_c, replace=False)
    ids[100:140] = ids[:40]                                 # forty vertices twice ...
    ids[200:210] = ids[:10]                                 # ... ten of them three times ...
    ids[300:302] = ids[0]
    ids[400:402] = ids[0]                                   # ... and one seven times, far apart in slot order
    vid = torch.tensor(ids, dtype=torch.int64, device=DEV)
    grads = []
    for _ in range(20):
        verts = verts0.clone().requires_grad_(True)
        l_c, l_p = ops.scene_losses(verts, vid, scenes, slot, sdf, gmin, gmax, True, 0.1, 0.3, 1.0)
        (1.3 * l_c + 0.7 * l_p).backward()
        grads.append(verts.grad.clone())
    assert all(torch.equal(g, grads[0]) for g in grads[1:])
    # the repeated vertex's row = penetration part + its seven contact terms; against the operator sequence under autograd (fp32 reassociation)
    verts = verts0.clone().requires_grad_(True)
    d = ops.chamfer_to_scenes(verts[:, vid, :].contiguous(), scenes, slot)
    s = torch.sqrt(d + 1e-4)
    vals = ops.sdf_sample(verts, sdf, gmin, gmax, scene_id=slot, align_corners=True)
    (1.3 * 0.1 * torch.mean(s / (s + 1.0)) + 0.7 * 0.3 * ops.penetration_loss(vals)).backward()
    assert (grads[0] - verts.grad).abs().max().item() < 1e-5 * verts.grad.abs().max().item()
    assert (grads[0][:, ids[0]] - verts.grad[:, ids[0]]).abs().max().item() < 1e-5 * verts.grad[:, ids[0]].abs().max().item()
    st = [ops.penetration_loss(vals.detach()) for _ in range(20)]
    assert all(torch.equal(x, st[0]) for x in st[1:])
