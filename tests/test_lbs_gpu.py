"""GPU parity: SMPL-X LBS forward/backward through the C ABI vs the oracle and the reference's golden vectors.
Tolerance: vertices / gradients within 1e-4 relative (fp32), the bar BASELINE.json's north_star states."""
import numpy as np
import pytest
import torch

import psi_oracle as O
from conftest import golden, rel_err
from psi_release_amd import body_model, synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'
T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)
C = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)


@pytest.fixture(scope='module')
def layer(smplx_data):
    return body_model.create(smplx_data, num_pca_comps=12, batch_size=4, device=DEV)


@pytest.fixture(scope='module')
def oracle_model(smplx_data):
    return O.SMPLXOracle(smplx_data)


def test_lbs_forward_golden(layer):
    """verts / joints of the reference's vendored lbs() (lbs.py:34-118) on the golden inputs."""
    g = golden('lbs')
    v, j = body_model.lbs(layer.lbs_model, T(g['betas']), T(g['pose']), return_joints=True)
    assert rel_err(v.cpu(), g['verts']) < 1e-4
    assert rel_err(j.cpu(), g['joints']) < 1e-4
    assert np.abs(v.cpu().numpy() - g['verts']).max() < 2e-5


@pytest.mark.parametrize('B', [1, 3, 16, 17, 32, 40, 70, 128, 131])          # >= 128: the multi-body skinning kernels
def test_lbs_forward_batch_sizes(layer, oracle_model, B):
    rs = np.random.RandomState(B)
    betas = rs.standard_normal((B, 20)).astype(np.float32)
    pose = (rs.standard_normal((B, 165)) * 0.5).astype(np.float32)
    transl = rs.standard_normal((B, 3)).astype(np.float32)
    cam = synth.make_cam_ext(B, B)
    v = body_model.lbs(layer.lbs_model, T(betas), T(pose), T(transl), T(cam))
    m = oracle_model
    with torch.no_grad():
        vo, _ = O.lbs(C(betas), C(pose), m.v_template, m.shapedirs, m.posedirs, m.J_regressor, m.parents, m.lbs_weights)
        vo = O.verts_transform(vo + C(transl).unsqueeze(1), C(cam))
    assert rel_err(v.cpu(), vo) < 1e-4


@pytest.mark.parametrize('B', [32, 70])
def test_blend_products_are_in_the_fp32_accuracy_class(layer, smplx_data, oracle_model, B):
    """blend_fwd multiplies on the fp16 matrix pipe with both operands stored as two fp16 parts per value (22 mantissa bits; lbs.hip).  The
    result must be as close to an fp64 evaluation of lbs() as the fp32 ORACLE is (K = 4, the arbiter's constant), not merely within 1e-4:
    large shape parameters and rotations, so the blend offsets are decimetres."""
    rs = np.random.RandomState(7 + B)
    betas = (rs.standard_normal((B, 20)) * 2.0).astype(np.float32)
    pose = (rs.standard_normal((B, 165)) * 0.8).astype(np.float32)
    v = body_model.lbs(layer.lbs_model, T(betas), T(pose)).cpu().double()
    m, m64 = oracle_model, O.SMPLXOracle(smplx_data, dtype=torch.float64)
    with torch.no_grad():
        v32, _ = O.lbs(C(betas), C(pose), m.v_template, m.shapedirs, m.posedirs, m.J_regressor, m.parents, m.lbs_weights)
        v64, _ = O.lbs(C(betas).double(), C(pose).double(), m64.v_template, m64.shapedirs, m64.posedirs, m64.J_regressor, m64.parents, m64.lbs_weights)
    d_prod, d_orc = float((v - v64).abs().max()), float((v32.double() - v64).abs().max())
    assert d_prod <= 4.0 * d_orc + 1e-7, (d_prod, d_orc)


@pytest.mark.parametrize('B,use_cam', [(2, True), (5, False), (32, True), (40, True), (70, False), (131, True)])   # > 32: the 4-tile MFMA variants; >= 128: multi-body skinning
def test_lbs_backward_vs_autograd(layer, oracle_model, B, use_cam):
    rs = np.random.RandomState(100 + B)
    betas = rs.standard_normal((B, 20)).astype(np.float32)
    pose = (rs.standard_normal((B, 165)) * 0.5).astype(np.float32)
    pose[0, 3:9] = 0.0                                         # zero rotations: the +1e-8 branch of Rodrigues
    transl = rs.standard_normal((B, 3)).astype(np.float32)
    cam = synth.make_cam_ext(B, B) if use_cam else None
    w = rs.standard_normal((B, 10475, 3)).astype(np.float32)
    bt, pt, tt = T(betas).requires_grad_(), T(pose).requires_grad_(), T(transl).requires_grad_()
    v = body_model.lbs(layer.lbs_model, bt, pt, tt, T(cam) if use_cam else None)
    (v * T(w)).sum().backward()
    m = oracle_model
    bo, po, to = C(betas).requires_grad_(), C(pose).requires_grad_(), C(transl).requires_grad_()
    vo, _ = O.lbs(bo, po, m.v_template, m.shapedirs, m.posedirs, m.J_regressor, m.parents, m.lbs_weights)
    vo = vo + to.unsqueeze(1)
    if use_cam:
        vo = O.verts_transform(vo, C(cam))
    (vo * C(w)).sum().backward()
    assert rel_err(v.detach().cpu(), vo.detach()) < 1e-4
    assert rel_err(tt.grad.cpu(), to.grad) < 1e-4
    assert rel_err(bt.grad.cpu(), bo.grad) < 1e-4
    assert rel_err(pt.grad.cpu(), po.grad) < 1e-4


def test_smplx_layer_matches_oracle_forward_and_grad(layer, oracle_model):
    """The smplx-style call PSI makes (hand PCA, pose_mean, zero jaw/eyes/expression) incl. gradients."""
    B = 4
    bodies = synth.make_bodies(11, B)
    rs = np.random.RandomState(9)
    body_pose = (rs.standard_normal((B, 63)) * 0.4).astype(np.float32)
    keys = ('transl', 'global_orient', 'betas', 'left_hand_pose', 'right_hand_pose')
    gi = {k: T(bodies[k]).requires_grad_() for k in keys}
    gbp = T(body_pose).requires_grad_()
    out = layer(return_verts=True, body_pose=gbp, **gi)
    w = T(rs.standard_normal((B, 10475, 3)))
    (out.vertices * w).sum().backward()
    ci = {k: C(bodies[k]).requires_grad_() for k in keys}
    cbp = C(body_pose).requires_grad_()
    ref = oracle_model(body_pose=cbp, **ci)
    (ref.vertices * w.cpu()).sum().backward()
    assert rel_err(out.vertices.detach().cpu(), ref.vertices.detach()) < 1e-4
    assert rel_err(out.joints.cpu(), ref.joints.detach()) < 1e-4
    for k in keys:
        assert rel_err(gi[k].grad.cpu(), ci[k].grad) < 1e-4, k
    assert rel_err(gbp.grad.cpu(), cbp.grad) < 1e-4


def test_lbs_generic_tree_and_small_model():
    """A non-SMPL-X topology (binary tree, J=23, V=777) exercises padding and the level schedule."""
    data = synth.make_smplx(seed=5, V=777, J=23)
    parents = data.kintree_table[0].copy()
    parents[0] = -1
    posedirs = data.posedirs.reshape(-1, data.posedirs.shape[-1]).T.copy()
    mdl = body_model.LbsModel(data.v_template, data.shapedirs, posedirs, data.J_regressor, data.weights, parents, DEV)
    rs = np.random.RandomState(3)
    B = 6
    betas = rs.standard_normal((B, 20)).astype(np.float32)
    pose = (rs.standard_normal((B, 69)) * 0.6).astype(np.float32)
    bt, pt = T(betas).requires_grad_(), T(pose).requires_grad_()
    v = body_model.lbs(mdl, bt, pt)
    w = rs.standard_normal((B, 777, 3)).astype(np.float32)
    (v * T(w)).sum().backward()
    bo, po = C(betas).requires_grad_(), C(pose).requires_grad_()
    vo, _ = O.lbs(bo, po, C(data.v_template), C(data.shapedirs), C(posedirs), C(data.J_regressor), torch.tensor(parents),
                  C(data.weights))
    (vo * C(w)).sum().backward()
    assert rel_err(v.detach().cpu(), vo.detach()) < 1e-4
    assert rel_err(bt.grad.cpu(), bo.grad) < 1e-4 and rel_err(pt.grad.cpu(), po.grad) < 1e-4


def test_compressed_skinning_rows_bit_identical(monkeypatch):
    """Weight rows with <= 8 non-zeros (what real SMPL-X rows look like) take the compressed-row skinning path; it skips
    exact zeros in ascending joint order, so vertices and gradients equal the dense loop's bit for bit."""
    data = synth.make_smplx(seed=7)
    W = np.array(data.weights, dtype=np.float32)
    keep = np.argsort(-W, axis=1)[:, :5]                       # 5 strongest joints per vertex, renormalised
    Ws = np.zeros_like(W)
    np.put_along_axis(Ws, keep, np.take_along_axis(W, keep, axis=1), axis=1)
    Ws /= Ws.sum(axis=1, keepdims=True)
    parents = data.kintree_table[0].copy()
    parents[0] = -1
    posedirs = data.posedirs.reshape(-1, data.posedirs.shape[-1]).T.copy()
    rs = np.random.RandomState(11)
    B = 4
    betas, pose = rs.standard_normal((B, 20)).astype(np.float32), (rs.standard_normal((B, 165)) * 0.4).astype(np.float32)
    transl, cam, w = rs.standard_normal((B, 3)).astype(np.float32), synth.make_cam_ext(3, B), rs.standard_normal((B, 10475, 3)).astype(np.float32)
    out = {}
    for mode in ('compressed', 'dense'):
        monkeypatch.setenv('PSI_LBS_DENSE', '1' if mode == 'dense' else '0')
        mdl = body_model.LbsModel(data.v_template, data.shapedirs, posedirs, data.J_regressor, Ws, parents, DEV)
        bt, pt, tt = T(betas).requires_grad_(), T(pose).requires_grad_(), T(transl).requires_grad_()
        v = body_model.lbs(mdl, bt, pt, transl=tt, cam_ext=T(cam))
        (v * T(w)).sum().backward()
        out[mode] = [v.detach().clone(), bt.grad.clone(), pt.grad.clone(), tt.grad.clone()]
    for a, b in zip(out['compressed'], out['dense']):
        assert torch.equal(a, b)
    # and the path is actually different from the dense model's (the sparsified weights changed the mesh)
    dense_model = body_model.LbsModel(data.v_template, data.shapedirs, posedirs, data.J_regressor, W, parents, DEV)
    assert not torch.equal(body_model.lbs(dense_model, T(betas), T(pose), transl=T(transl), cam_ext=T(cam)), out['dense'][0])


@pytest.mark.parametrize('compressed', [False, True])
def test_multi_body_skinning_kernels_bit_identical_to_single_body(compressed):
    """B >= 128 takes the multi-body skinning kernels (a lane keeps its vertex's weights in registers and walks 8 bodies); they sum
    the same terms in the same order as the one-body-per-workgroup kernels, so a batch of 136 equals its 32-row chunks bit for bit
    (forward vertices and all three gradients), dense and compressed weight rows alike."""
    data = synth.make_smplx(seed=7)
    W = np.array(data.weights, dtype=np.float32)
    if compressed:
        keep = np.argsort(-W, axis=1)[:, :6]
        Ws = np.zeros_like(W)
        np.put_along_axis(Ws, keep, np.take_along_axis(W, keep, axis=1), axis=1)
        W = Ws / Ws.sum(axis=1, keepdims=True)
    parents = data.kintree_table[0].copy()
    parents[0] = -1
    posedirs = data.posedirs.reshape(-1, data.posedirs.shape[-1]).T.copy()
    mdl = body_model.LbsModel(data.v_template, data.shapedirs, posedirs, data.J_regressor, W, parents, DEV)
    rs = np.random.RandomState(12)
    B = 136
    betas, pose = rs.standard_normal((B, 20)).astype(np.float32), (rs.standard_normal((B, 165)) * 0.4).astype(np.float32)
    transl, cam, w = rs.standard_normal((B, 3)).astype(np.float32), synth.make_cam_ext(3, B), rs.standard_normal((B, 10475, 3)).astype(np.float32)

    def run(sl):
        bt, pt, tt = T(betas[sl]).requires_grad_(), T(pose[sl]).requires_grad_(), T(transl[sl]).requires_grad_()
        v = body_model.lbs(mdl, bt, pt, transl=tt, cam_ext=T(cam[sl]))
        (v * T(w[sl])).sum().backward()
        return [v.detach(), bt.grad, pt.grad, tt.grad]
    whole = run(slice(0, B))
    parts = [run(slice(i, min(i + 32, B))) for i in range(0, B, 32)]
    assert torch.equal(whole[0], torch.cat([p[0] for p in parts]))                 # vertices: bit-identical
    for k in (1, 2, 3):                                                            # gradients: the per-body contractions are the same code,
        ref = torch.cat([p[k] for p in parts])                                     # only the MFMA row-tile variant differs (B > 32 vs 32)
        assert rel_err(whole[k].cpu(), ref.cpu()) < 1e-5
