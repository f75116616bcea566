"""CPU suite: the oracle (oracle/psi_oracle.py, oracle/chamfer_oracle.c) against the golden vectors that
oracle/make_golden.py recorded from the imported reference, and against the reference's own
known-answer check for Chamfer (chamfer_pytorch/test_chamfer.py:35-54)."""
import json
import os

import numpy as np
import pytest
import torch

import psi_oracle as O
from conftest import GOLD, golden, rel_err
from psi_release_amd import synth

T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)


def test_synth_manifest(smplx_data, vposer_sd):
    """The generator reproduces the tensors the fixtures were made from (checksums committed)."""
    man = json.load(open(os.path.join(GOLD, 'manifest.json')))
    for k, c in man['smplx_seed7'].items():
        assert synth.checksum(getattr(smplx_data, k)) == c, k
    for k, c in man['vposer_seed3'].items():
        assert synth.checksum(vposer_sd[k]) == c, k
    sc = synth.make_scene(0, 4096, 32, 512)
    assert synth.checksum(sc.verts) == man['scene_seed0_m4096_D32']['verts']
    assert synth.checksum(sc.sdf) == man['scene_seed0_m4096_D32']['sdf']


def test_rot_glue_golden():
    g = golden('rot_glue')
    assert rel_err(O.convert_to_6d_rot(T(g['x72'])), g['x75']) < 1e-6
    assert rel_err(O.convert_to_3d_rot(T(g['x75'])), g['x72_back']) < 1e-6
    assert rel_err(O.convert_to_3d_rot(T(g['x75_free'])), g['x72_free']) < 1e-6
    assert rel_err(O.rot6d_decode(T(g['x75_free'][:, 3:9])), g['R_free']) < 1e-6
    xn = O.normalize_global_T(T(g['xt']), T(g['cam_int']), T(g['max_d']))
    assert rel_err(xn, g['xn']) < 1e-6
    assert rel_err(O.recover_global_T(xn, T(g['cam_int']), T(g['max_d'])), g['xb']) < 1e-6
    assert rel_err(O.verts_transform(T(g['verts']), T(g['cam_ext'])), g['verts_t']) < 1e-6


def test_rot_roundtrip_property():
    """aa -> R -> 6D -> R -> aa returns the input rotation (as a rotation) for generic angles."""
    rs = np.random.RandomState(0)
    aa = T(rs.standard_normal((64, 3)))
    R = O.aa2matrot(aa)
    R2 = O.rot6d_decode(R[:, :, :2].reshape(-1, 6))
    assert float((R - R2).abs().max()) < 1e-5
    R3 = O.aa2matrot(O.matrot2aa(R2))
    assert float((R - R3).abs().max()) < 1e-5


def test_vposer_golden(vposer_sd):
    g = golden('vposer_decode')
    sd = {k: T(v) for k, v in vposer_sd.items() if 'dec' in k}
    assert rel_err(O.vposer_decode_aa(sd, T(g['z'])), g['aa']) < 1e-5


def test_lbs_golden(smplx_data):
    g = golden('lbs')
    m = O.SMPLXOracle(smplx_data)
    with torch.no_grad():
        v, j = O.lbs(T(g['betas']), T(g['pose']), m.v_template, m.shapedirs, m.posedirs, m.J_regressor, m.parents,
                     m.lbs_weights)
        R = O.batch_rodrigues(T(g['pose']).view(-1, 3))
    assert rel_err(v, g['verts']) < 1e-6
    assert rel_err(j, g['joints']) < 1e-6
    assert rel_err(R, g['rodrigues']) < 1e-6


def test_chamfer_known_answer():
    """Reference's own check (test_chamfer.py:50-54): sum of squared differences to expanded-form brute force < 1e-8."""
    g = golden('chamfer_known_answer')
    d1, i1, d2, i2 = O.chamfer_nn_np(g['p1'], g['p2'])
    s = ((d1 - g['mydist1']) ** 2).sum() + ((d2 - g['mydist2']) ** 2).sum()
    assert s < 1e-8
    # indices: the argmin of an fp64 brute force agrees wherever the fp64 gap is not within fp32 rounding
    dd = ((g['p1'][:, :, None, :].astype(np.float64) - g['p2'][:, None, :, :]) ** 2).sum(-1)
    assert (dd.argmin(2) == i1).mean() > 0.99 and (dd.argmin(1) == i2).mean() > 0.99


def test_chamfer_chunked_equals_scan_and_ties():
    """chamfer.cu's 512-chunk structure == ascending strict-< scan, including exact ties (lowest index wins)."""
    rs = np.random.RandomState(1)
    x = rs.uniform(-1, 1, (2, 77, 3)).astype(np.float32)
    y = rs.uniform(-1, 1, (2, 1300, 3)).astype(np.float32)      # 3 chunks, ragged last chunk
    y[:, 700] = y[:, 5]                                           # duplicates across chunks
    y[:, 1299] = y[:, 600]
    y[:, 6] = y[:, 5]                                             # and inside one
    a = O.chamfer_nn_np(x, y, both=True)
    b = O.chamfer_nn_np(x, y, both=True, chunked=True)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    assert not np.isin(a[1], [6, 700, 1299]).any()
    # pure-python statement on a tiny case
    q, t = x[0, :5], y[0, :40]
    for j in range(5):
        best, bi = None, 0
        for k in range(40):
            dx, dy, dz = np.float32(t[k, 0] - q[j, 0]), np.float32(t[k, 1] - q[j, 1]), np.float32(t[k, 2] - q[j, 2])
            d = np.float32(np.float32(np.float32(dx * dx) + np.float32(dy * dy)) + np.float32(dz * dz))
            if best is None or d < best:
                best, bi = d, k
        d1, i1, _, _ = O.chamfer_nn_np(q[None], t[None], both=False)
        assert i1[0, j] == bi and d1[0, j] == best


def test_chamfer_edge_shapes():
    rs = np.random.RandomState(2)
    for n, m in ((1, 1), (3, 1), (1, 513), (17, 512), (100, 100)):
        x = rs.standard_normal((3, n, 3)).astype(np.float32)
        y = rs.standard_normal((3, m, 3)).astype(np.float32)
        d1, i1, d2, i2 = O.chamfer_nn_np(x, y)
        dd = ((x[:, :, None] - y[:, None]) ** 2).sum(-1)
        assert np.allclose(d1, dd.min(2), rtol=1e-5, atol=1e-7) and np.allclose(d2, dd.min(1), rtol=1e-5, atol=1e-7)
        assert i1.min() >= 0 and i1.max() < m and i2.max() < n


def test_chamfer_grad_matches_autograd():
    rs = np.random.RandomState(3)
    x = T(rs.standard_normal((2, 50, 3))).requires_grad_()
    y = T(rs.standard_normal((2, 80, 3))).requires_grad_()
    d1, d2 = O.chamfer_dist(x, y)
    (d1.sum() + 0.5 * d2.sum()).backward()
    x2 = x.detach().clone().requires_grad_()
    y2 = y.detach().clone().requires_grad_()
    dd = ((x2[:, :, None] - y2[:, None]) ** 2).sum(-1)
    (dd.min(2)[0].sum() + 0.5 * dd.min(1)[0].sum()).backward()
    assert rel_err(x.grad, x2.grad) < 1e-5 and rel_err(y.grad, y2.grad) < 1e-5


@pytest.mark.parametrize('ac', [True, False])
def test_sdf_c_matches_grid_sample(ac):
    """Scalar trilinear restatement (Appendix C) == F.grid_sample incl. border clamping and its gradient."""
    sc = synth.make_scene(0, 64, 16, 14)
    rs = np.random.RandomState(4)
    verts = rs.uniform(-2.6, 2.6, (3, 500, 3)).astype(np.float32)      # ~25% outside the grid on each axis
    verts[0, 0] = [-2.0, 2.0, 0.0]
    verts[0, 1] = [2.0, 2.0, 2.0]
    vt = T(verts).requires_grad_()
    B = 3
    out = O.sdf_sample(T(sc.sdf).unsqueeze(0).expand(B, -1, -1, -1), T(sc.grid_min)[None].expand(B, -1),
                       T(sc.grid_max)[None].expand(B, -1), vt, align_corners=ac)
    out.sum().backward()
    val, grad = O.sdf_sample_c(sc.sdf[None], np.zeros(B, np.int32), sc.grid_min, sc.grid_max, verts, ac)
    assert np.abs(val - out.detach().numpy().reshape(B, -1)).max() < 2e-6
    assert np.abs(grad - vt.grad.numpy()).max() < 2e-5


@pytest.mark.parametrize('tag', ['ac1', 'ac0'])
def test_fitting_golden(smplx_data, vposer_sd, tag):
    """cal_loss value/gradient and the 5-iteration Adam trajectory of the reference's FittingOP."""
    g = golden('fitting_proxe')
    B, m, n_c, D = int(g['B']), int(g['m']), int(g['n_c']), int(g['D'])
    sc = synth.make_scene(0, m, D, n_c)
    vid = synth.contact_ids_from_parts(sc.contact_parts)
    assert np.array_equal(vid, g['contact_ids'])
    mk = lambda: O.FittingOracle(O.SMPLXOracle(smplx_data), vposer_sd, sc.verts, sc.sdf, sc.grid_min, sc.grid_max, vid,
                                 B, align_corners=(tag == 'ac1'))
    fo = mk()
    fo.xhr_rec.data = T(g['xhr_rec0_' + tag])
    losses = fo.cal_loss(T(g['xhr_' + tag]), T(g['cam_ext']))
    assert rel_err([float(l) for l in losses], g['loss0_' + tag]) < 1e-5
    sum(losses).backward()
    assert rel_err(fo.xhr_rec.grad, g['grad0_' + tag]) < 1e-4
    if tag == 'ac1':
        assert rel_err(fo.last.verts.detach(), g['verts0']) < 1e-5
    # trajectory
    fo = mk()
    bodies = synth.make_bodies(11, B)
    rec = []
    xh = fo.fitting(synth.body_vector_72(bodies), g['cam_ext'], 5, record=rec)
    assert np.abs(np.array(rec) - g['traj_loss_' + tag]).max() < 2e-5
    assert rel_err(fo.xhr_rec.detach(), g['traj_final_xhr_' + tag]) < 1e-4
    assert np.abs(xh.detach().numpy() - g['traj_final_' + tag]).max() < 1e-3


def test_arbiter_mode_bounds_the_fp32_oracle_and_the_reference(smplx_data, vposer_sd):
    """``SMPLXOracle(dtype=float64)`` / ``FittingOracle`` on it = the ARBITER: the same restatement in double precision on the same
    fp32-valued constants.  It is the yardstick of the GPU parity tests (|gpu - fp64| against |oracle_fp32 - fp64|); here it is held to
    the reference's own recorded numbers: the reference's fp32 loss values and gradient (tests/golden/fitting_proxe.npz, recorded by
    importing /root/reference) lie as close to the arbiter as this oracle's fp32 evaluation does."""
    g = golden('fitting_proxe')
    B, m, n_c, D = int(g['B']), int(g['m']), int(g['n_c']), int(g['D'])
    sc = synth.make_scene(0, m, D, n_c)
    vid = synth.contact_ids_from_parts(sc.contact_parts)
    out = {}
    for name, dt in (('f32', torch.float32), ('f64', torch.float64)):
        fo = O.FittingOracle(O.SMPLXOracle(smplx_data, dtype=dt), vposer_sd, sc.verts, sc.sdf, sc.grid_min, sc.grid_max, vid, B)
        fo.xhr_rec.data = torch.tensor(g['xhr_rec0_ac1'], dtype=dt)
        losses = fo.cal_loss(torch.tensor(g['xhr_ac1'], dtype=dt), torch.tensor(g['cam_ext'], dtype=dt))
        assert all(l.dtype == dt for l in losses) and fo.last.verts.dtype == dt
        sum(losses).backward()
        out[name] = (np.array([float(l) for l in losses]), fo.xhr_rec.grad.numpy().astype(np.float64),
                     fo.last.verts.detach().numpy().astype(np.float64))
    l32, g32, v32 = out['f32']
    l64, g64, v64 = out['f64']
    # the fp32 oracle's own distance from the exact value: a few ulp of the operands
    assert np.abs(l32 - l64).max() < 1e-6 and np.abs(v32 - v64).max() < 1e-5 and np.abs(g32 - g64).max() < 1e-5 * np.abs(g64).max()
    # the reference's recorded fp32 numbers are no further from the arbiter than 4x that (different summation orders, same precision)
    assert np.abs(g['loss0_ac1'] - l64).max() <= 4 * np.abs(l32 - l64).max() + 1e-7
    assert np.abs(g['grad0_ac1'] - g64).max() <= 4 * np.abs(g32 - g64).max() + 1e-7 * np.abs(g64).max()
    assert np.abs(g['verts0'] - v64).max() <= 4 * np.abs(v32 - v64).max() + 1e-7


def test_arbiter_trace_check_accepts_the_fp32_oracle_and_rejects_a_perturbed_run(smplx_data, vposer_sd):
    """tests/arbiter.py (the step-by-step check the GPU parity tests apply to the product) on a trace produced by the fp32 oracle's own
    loop — it must pass — and on the same trace with one gradient entry of one iteration off by 1e-3 of the gradient scale, the Adam
    update of another off by 1e-3, and one loss value off by 1e-4 — each must be rejected."""
    import arbiter
    B, m, n_c, D, iters = 12, 4096, 256, 64, 3
    sc = synth.make_scene(0, m, D, n_c)
    vid = synth.contact_ids_from_parts(sc.contact_parts)
    bodies = synth.make_bodies(11, B)
    cam = synth.make_cam_ext(5, B)
    make = lambda dt: O.FittingOracle(O.SMPLXOracle(smplx_data, dtype=dt), vposer_sd, sc.verts, sc.sdf, sc.grid_min, sc.grid_max, vid, B)
    fo = make(torch.float32)
    xhr = O.convert_to_6d_rot(torch.as_tensor(synth.body_vector_72(bodies), dtype=torch.float32))
    camt = torch.as_tensor(cam, dtype=torch.float32)
    fo.xhr_rec.data = xhr.clone()
    trace = []
    f = lambda t: t.detach().numpy().astype(np.float64)
    for it in range(iters):
        st = fo.optimizer.state.get(fo.xhr_rec, {})
        x0, m0, v0 = f(fo.xhr_rec), f(st.get('exp_avg', torch.zeros(B, 75))), f(st.get('exp_avg_sq', torch.zeros(B, 75)))
        fo.optimizer.zero_grad()
        ls = fo.cal_loss(xhr, camt)
        sum(ls).backward()
        fo.optimizer.step()
        st = fo.optimizer.state[fo.xhr_rec]
        trace.append(dict(x0=x0, m0=m0, v0=v0, x1=f(fo.xhr_rec), m1=f(st['exp_avg']), v1=f(st['exp_avg_sq']),
                          losses=np.array([float(l.detach()) for l in ls])))
    report = arbiter.check_trace(trace, make, np.asarray(cam, np.float64))
    assert len(report) == iters and all(r['bodies_by_rule']['a'] == B for r in report), report     # the fp32 oracle against itself: rule (a)
    # ... and the bookkeeping the GPU tests apply to their reports: the floor on the share of rule-(a) bodies
    arbiter.record('cpu_selfcheck', report)
    loose = [dict(r, bodies_by_rule=dict(a=1, b_only=B - 1, c_only=0)) for r in report]
    with pytest.raises(AssertionError):
        arbiter.record('cpu_selfcheck_loose', loose)
    import copy
    gscale = np.abs(trace[1]['m1']).max() / 0.1
    bad = copy.deepcopy(trace)
    bad[1]['m1'][2, 40] += 0.1 * 1e-3 * gscale                       # the gradient of iteration 2, one entry, off by 1e-3 of the scale
    with pytest.raises(AssertionError):
        arbiter.check_trace(bad, make, np.asarray(cam, np.float64))
    bad = copy.deepcopy(trace)
    j = int(np.argmax(np.abs(trace[2]['m1'][1])))                     # a well-conditioned entry of body 1
    bad[2]['x1'][1, j] += 1e-3
    with pytest.raises(AssertionError):
        arbiter.check_trace(bad, make, np.asarray(cam, np.float64))
    bad = copy.deepcopy(trace)
    bad[0]['losses'][3] += 1e-4
    with pytest.raises(AssertionError):
        arbiter.check_trace(bad, make, np.asarray(cam, np.float64))
