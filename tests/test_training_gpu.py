"""GPU parity of the training path: TrainOP.cal_loss (stage 1 and 2) vs the losses / parameter gradients recorded from
the reference's own TrainOP (tests/golden/training.npz, oracle/make_golden.py::gen_training), plus the batch contract,
checkpoint schema and resume behaviour (train_s1.py:220-233,303-321)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import golden, rel_err
from psi_release_amd import batch_gen, models, synth, training

pytestmark = pytest.mark.gpu
DEV = 'cuda'
T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)
LW = {'weight_loss_rec_s': 1.0, 'weight_loss_rec_h': 1.0, 'weight_loss_vposer': 1e-3, 'weight_loss_kl': 1e-1,
      'weight_contact': 1e-1, 'weight_collision': 1e-1}


def make_cfg(tmp, smplx_data, vposer_sd, scene, B, epoch=100):
    return {'human_model_path': None, 'vposer_ckpt_path': None, 'scene_model_ckpt': None, 'init_lr_h': 1e-4, 'batch_size': B,
            'epoch': epoch, 'loss_weight_anealing': True, 'device': torch.device(DEV), 'save_dir': str(tmp),
            'contact_id_folder': None, 'contact_part': synth.CONTACT_PARTS, 'verbose': False, 'use_cont_rot': True,
            'resume_training': True, 'smplx_data': smplx_data, 'vposer_state': vposer_sd, 'contact_parts_data': scene.contact_parts}


def _load(m, seed):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_state_like(shapes, seed).items()})


@pytest.mark.parametrize('stage', ['s1', 's2'])
def test_cal_loss_golden(tmp_path, smplx_data, vposer_sd, stage):
    g = golden('training')
    B, m, n_c, D = int(g['B']), int(g['m']), int(g['n_c']), int(g['D'])
    scene = synth.make_scene(2, m, D, n_c)
    inp = synth.make_cvae_inputs(13, B)
    cls, seed = (training.TrainOP, 0) if stage == 's1' else (training.TrainOPS2, 1)
    op = cls(make_cfg(tmp_path, smplx_data, vposer_sd, scene, B), dict(LW))
    args = dict(xs=T(inp['xs']), xh=T(g['xh']), cam_ext=T(g['cam_ext']), cam_int=T(g['cam_int']), max_d=T(g['max_d']),
                scene_verts=T(scene.verts)[None].repeat(B, 1, 1), scene_face=None,
                s_grid_min_batch=T(scene.grid_min)[None].repeat(B, 1), s_grid_max_batch=T(scene.grid_max)[None].repeat(B, 1),
                s_grid_sdf_batch=T(scene.sdf)[None].repeat(B, 1, 1, 1))
    for ep in (10, 90):
        _load(op.model_h, seed)
        op.model_h.train()
        op.model_h.zero_grad()
        if stage == 's1':
            losses = op.cal_loss(ep=ep, eps=T(inp['eps32']), **args)
            keys = ['linear_out.weight', 'resnet.0.weight', 'mu_enc.bias']
        else:
            losses = op.cal_loss(eps_g=T(inp['eps32']), eps_l=T(inp['eps32b']), ep=ep, use_eps=True, **args)
            keys = ['pose_vae.decode.3.weight', 'trans_vae.resnet.0.weight', 'trans_vae.decode.3.bias']
        sum(losses).backward()
        got = np.array([float(l) for l in losses])
        ref = g['%s_ep%d_losses' % (stage, ep)]
        assert np.abs(got - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), (got, ref)       # BASELINE north_star: losses within 1e-4 of the reference
        params = dict(op.model_h.named_parameters())
        for k in keys:
            gr = params[k].grad.detach().cpu().numpy().reshape(-1)
            rg = g['%s_ep%d_grad_%s' % (stage, ep, k)].reshape(-1)
            # the first conv sits below 17 train-mode BatchNorm layers at batch 4: ReLU / max-pool masks of near-zero activations differ between any
            # two fp32 evaluations (the reference's CPU run included); test_configs2_gpu.py holds the same gradients to an fp64 arbiter instead of a literal bound
            assert rel_err(gr[:rg.size], rg) < (3e-2 if 'resnet.0' in k else 2e-3), k
    # indirect SDF (scene table + ids) gives the same losses as the dense per-sample volumes
    _load(op.model_h, seed)
    tup = (T(scene.sdf)[None].contiguous(), torch.zeros(B, dtype=torch.int32, device=DEV), T(scene.grid_min)[None], T(scene.grid_max)[None])
    a2 = dict(args)
    a2['s_grid_sdf_batch'] = tup
    with torch.no_grad():
        l_ind = op.cal_loss(ep=90, eps=T(inp['eps32']), **a2) if stage == 's1' else \
            op.cal_loss(eps_g=T(inp['eps32']), eps_l=T(inp['eps32b']), ep=90, use_eps=True, **a2)
    assert abs(float(l_ind[-1]) - float(g['%s_ep90_losses' % stage][-1])) < 1e-4


def _table(n, n_scenes, seed=0):
    rs = np.random.RandomState(seed)
    body = synth.body_vector_72(synth.make_bodies(seed, n))
    body[:, 2] = np.abs(body[:, 2]) + 2.0
    t = {'depth': rs.uniform(-1, 1, (n, 1, 128, 128)), 'seg': rs.uniform(-1, 1, (n, 1, 128, 128)), 'body': body,
         'cam_ext': synth.make_cam_ext(seed, n), 'cam_int': synth.make_bodies(seed, n)['cam_int'],
         'max_d': np.full(n, 6.0), 'sceneid': rs.randint(0, n_scenes, n).astype(np.float32)}
    return {k: np.concatenate([np.zeros_like(np.asarray(v)[:1]), np.asarray(v)]).astype(np.float32) for k, v in t.items()}


def test_batch_contract_and_training_loop_checkpoint_resume(tmp_path, smplx_data, vposer_sd):
    scenes_d = {n: synth.make_scene(i, 1500, 16, 200) for i, n in enumerate(['A', 'B', 'C'])}
    scenes = {n: {'verts': s.verts, 'sdf': s.sdf, 'grid_min': s.grid_min, 'grid_max': s.grid_max, 'grid_dim': s.grid_dim}
              for n, s in scenes_d.items()}
    B = 4
    table = _table(18, 3)
    for indirect in (False, True):
        bg = batch_gen.BatchGeneratorWithSceneMesh.from_arrays(table, scenes, DEV, indirect_sdf=indirect)
        assert bg.n_samples == 18
        d = bg.next_batch(B)
        assert len(d) == 12
        assert d[0].shape == (B, 1, 128, 128) and d[2].shape == (B, 72) and d[3].shape == (B, 4, 4) and d[4].shape == (B, 3, 3)
        assert d[5].shape == (B,) and d[6].shape == (B, 1500, 3) and d[8].shape == (B, 3) and d[10].shape == (B,)
        assert (isinstance(d[11], tuple) and d[11][0].shape == (3, 16, 16, 16)) if indirect else d[11].shape == (B, 16, 16, 16)
        n = 1
        while bg.has_next_batch():
            n += bg.next_batch(B) is not None
        assert n == 4                                           # 18 // 4 full batches, the short one dropped
    cfg = make_cfg(tmp_path, smplx_data, vposer_sd, scenes_d['A'], B, epoch=10)
    op = training.TrainOPS2(cfg, dict(LW))
    bg = batch_gen.BatchGeneratorWithSceneMesh.from_arrays(table, scenes, DEV, indirect_sdf=True)
    w0 = op.model_h.pose_vae.decode[3].weight.detach().clone()
    op.train(bg)                                                # 10 epochs x 4 steps; scene losses switch on after epoch 7
    ck = sorted(glob.glob(os.path.join(str(tmp_path), 'epoch-*.ckp')))
    assert [os.path.basename(c) for c in ck] == ['epoch-000010.ckp']
    sd = torch.load(ck[0], map_location='cpu')
    assert set(sd.keys()) == {'epoch', 'model_h_state_dict', 'optimizer_h_state_dict'} and sd['epoch'] == 10
    ref_keys = list(golden('cvae')['s2_keys'])
    assert list(sd['model_h_state_dict'].keys()) == ref_keys    # checkpoint layout == the reference's (Appendix B)
    assert not torch.equal(w0, op.model_h.pose_vae.decode[3].weight.detach())
    assert all(torch.isfinite(p).all() for p in op.model_h.parameters())
    # resume: a fresh TrainOP picks up the newest checkpoint and has nothing left to do
    op2 = training.TrainOPS2(cfg, dict(LW))
    assert op2._resume() == 10
    assert torch.equal(op2.model_h.pose_vae.decode[3].weight.detach().cpu(), sd['model_h_state_dict']['pose_vae.decode.3.weight'])


def _train_setup(tmp, B, rows):
    """Deterministic stage-1 TrainOP (eval-mode BN, fixed weights) and rows `rows` of a fixed 4-sample batch."""
    import types
    smplx_data, vposer_sd = synth.make_smplx(7), synth.make_vposer_state(3)
    scene = synth.make_scene(2, 1500, 16, 200)
    cfg = make_cfg(tmp, smplx_data, vposer_sd, scene, B)
    cfg['resume_training'] = False
    op = training.TrainOP(cfg, dict(LW))
    _load(op.model_h, 0)
    op.model_h.eval()
    torch.manual_seed(0)
    eps = torch.randn(4, 32, device=DEV)[rows]
    inp = synth.make_cvae_inputs(13, 4)
    bodies = synth.make_bodies(17, 4)
    xh = synth.body_vector_72(bodies)
    xh[:, 2] = np.abs(xh[:, 2]) + 2.0
    n = len(range(4)[rows])
    rep = lambda a: T(a)[None].repeat(n, *([1] * np.asarray(a).ndim))
    batch = [T(inp['xs'][rows, :1]), T(inp['xs'][rows, 1:]), T(xh[rows]), T(synth.make_cam_ext(9, 4)[rows]), T(bodies['cam_int'][rows]),
             T(np.full(n, 6.0, np.float32)), rep(scene.verts), None, rep(scene.grid_min), rep(scene.grid_max), None, rep(scene.sdf)]
    cal = op.cal_loss
    op.cal_loss = lambda **kw: cal(eps=eps, **kw)           # fixed reparameterisation noise for both shardings
    return op, batch


def test_scene_index_path_equals_dense_chamfer(tmp_path, smplx_data, vposer_sd):
    """cal_loss through the per-scene NN indices (indirect batches) == through the dense [B,m,3] Chamfer op, loss and grads."""
    scenes_d = {n: synth.make_scene(i, 1500, 16, 200) for i, n in enumerate(['A', 'B', 'C'])}
    scenes = {n: {'verts': s.verts, 'sdf': s.sdf, 'grid_min': s.grid_min, 'grid_max': s.grid_max, 'grid_dim': s.grid_dim}
              for n, s in scenes_d.items()}
    B = 6
    bg = batch_gen.BatchGeneratorWithSceneMesh.from_arrays(_table(12, 3), scenes, DEV, indirect_sdf=True)
    d = bg.next_batch(B)
    assert len(set(d[11][1].tolist())) > 1                                  # the batch really mixes scenes
    cfg = make_cfg(tmp_path, smplx_data, vposer_sd, scenes_d['A'], B, epoch=10)
    op = training.TrainOPS2(cfg, dict(LW))
    op.model_h.eval()
    res = {}
    op._losses_from_batch(d, 9)                                            # MIOpen solver search happens here, not between the runs
    for use in (True, False):
        op.use_scene_index = use
        torch.manual_seed(0)
        op.model_h.zero_grad()
        losses = op._losses_from_batch(d, 9)                              # ep 9 of 10: contact + penetration terms active
        sum(losses).backward()
        res[use] = (torch.stack([l.detach() for l in losses]), torch.cat([p.grad.reshape(-1) for p in op.model_h.parameters() if p.grad is not None]))
    assert float(res[True][0][4]) > 0                                     # the contact term is live ...
    # the op itself is bit-identical (test_scene_set_query_equals_chamfer_on_gathered_clouds); the CVAE forward in front of
    # it is not run-to-run deterministic on this stack (MIOpen / hipBLASLt), hence tolerances here
    assert torch.allclose(res[True][0], res[False][0], rtol=1e-4, atol=1e-7), (res[True][0], res[False][0])
    assert rel_err(res[True][1].cpu(), res[False][1].cpu()) < 1e-3


def test_graph_step_equals_eager_step(tmp_path, smplx_data, vposer_sd, monkeypatch):
    """The HIP-graph replay of a whole optimiser step (use_graph) trains like the eager step: same losses over 3 steps
    on changing batches (sampling noise pinned to 0 so that both runs see the same latent)."""
    monkeypatch.setattr(torch, 'randn_like', lambda t, **kw: torch.zeros_like(t))
    scenes_d = {n: synth.make_scene(i, 1500, 16, 200) for i, n in enumerate(['A', 'B', 'C'])}
    scenes = {n: {'verts': s.verts, 'sdf': s.sdf, 'grid_min': s.grid_min, 'grid_max': s.grid_max, 'grid_dim': s.grid_dim}
              for n, s in scenes_d.items()}
    B = 4
    bg = batch_gen.BatchGeneratorWithSceneMesh.from_arrays(_table(12, 3), scenes, DEV, indirect_sdf=True)
    batches = [bg.next_batch(B) for _ in range(3)]
    out = {}
    for use_graph in (False, True):
        torch.manual_seed(0)
        cfg = make_cfg(tmp_path, smplx_data, vposer_sd, scenes_d['A'], B, epoch=10)
        cfg['use_graph'] = use_graph
        op = training.TrainOPS2(cfg, dict(LW))
        op.model_h.eval()                                   # BN on running stats: no batch-4 statistics noise
        hist = []
        for i, d in enumerate(batches):
            ep = 9 if i else 2                              # first step in the gated-off phase, then the full loss
            hist.append(torch.stack([l.detach().clone() for l in op.train_step(d, ep)]).cpu())
        out[use_graph] = torch.stack(hist)
    assert float(out[True][1][4]) > 0 and float(out[True][0][4]) == 0      # contact term: off, then live
    assert torch.allclose(out[True], out[False], rtol=2e-3, atol=1e-6), (out[True], out[False])


def test_body_decoder_equals_modular_chain(tmp_path, smplx_data, vposer_sd):
    """fitting.BodyDecoder (one HIP op, hand-derived backward) == convert_to_3D_rot -> vposer.decode -> SMPL-X layer -> cam_ext
    with autograd: vertices and the gradient wrt the 75-D body vector."""
    from psi_release_amd.geometry import BodyParamParser, GeometryTransformer
    scene = synth.make_scene(2, 500, 8, 64)
    B = 5
    op = training.TrainOP(make_cfg(tmp_path, smplx_data, vposer_sd, scene, B), dict(LW))
    rs = np.random.RandomState(3)
    body = synth.body_vector_72(synth.make_bodies(4, B))
    x72 = T(body)
    x75 = GeometryTransformer.convert_to_6D_rot(x72).detach().clone().requires_grad_(True)
    cam = T(synth.make_cam_ext(4, B))
    gv = T(rs.randn(B, 10475, 3))
    # modular chain (what the reference's cal_loss does)
    xa = GeometryTransformer.convert_to_3D_rot(x75)
    bp = BodyParamParser.body_params_encapsulate_batch(xa)
    jr = op.vposer.decode(bp['body_pose_vp'], output_type='aa').view(B, -1)
    v_ref = op.body_mesh_model(return_verts=True, body_pose=jr, cam_ext=cam, **{k: v for k, v in bp.items() if k != 'body_pose_vp'}).vertices
    (v_ref * gv).sum().backward()
    g_ref = x75.grad.clone()
    x75.grad = None
    from psi_release_amd.fitting import BodyDecoder
    dec = BodyDecoder(op.vposer, op.body_mesh_model, B, DEV)
    v = dec(x75, cam)
    (v * gv).sum().backward()
    assert rel_err(v.detach().cpu(), v_ref.detach().cpu()) < 1e-5
    assert rel_err(x75.grad.cpu(), g_ref.cpu()) < 1e-4
    with pytest.raises(RuntimeError):                      # a second forward invalidates the first one's saved activations
        v1 = dec(x75, cam)
        dec(x75, cam)
        v1.sum().backward()
