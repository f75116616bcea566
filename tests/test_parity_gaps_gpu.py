"""Product-path parity cases that round 1 only ran against the oracle (VERDICT r01 "what's weak" #1):
  * the reference's rotation-glue and VPoser.decode fixtures through the PRODUCT: geometry.py on the GPU and the fused
    engine's head kernel (all four quaternion branches of rotmat_to_aa, the Taylor branch, the identity);
  * a 3-iteration oracle trajectory at the full BASELINE configs[1] size;
  * train_s2 at batch 128: HIP-graph step vs eager step, bf16 trunk vs fp32 with a stated tolerance;
  * the drop-in modules (chamfer_pytorch.dist_chamfer, chamfer_pytorch.dist_chamfer_idx, the `chamfer` extension stand-in with
    caller-allocated tensors, smplx) imported the way the reference imports them;
  * the second arithmetic mode of the Chamfer distance (nvcc --fmad=true form), bit-exact against the oracle's matching build;
  * two streams calling the Chamfer op without a workspace; run-to-run bit equality of the fused iteration."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

import arbiter
import psi_oracle as O
from conftest import ROOT, golden, rel_err
from psi_release_amd import body_model, fitting, geometry, hip, ops, synth, training
from psi_release_amd.vposer import load_vposer

pytestmark = pytest.mark.gpu
DEV = 'cuda'
T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)
LOSS = {'weight_loss_rec': 1, 'weight_loss_vposer': 0.01, 'weight_contact': 0.1, 'weight_collision': 0.5}
GT = geometry.GeometryTransformer


# ------------------------------------------------------------------------------------------------------------------
# rotation glue and VPoser.decode through the product
# ------------------------------------------------------------------------------------------------------------------
def test_geometry_module_on_gpu_matches_reference_golden():
    """psi_release_amd.geometry (the host-side mirror of cvae.py:58-200) on CUDA tensors vs the reference's own outputs."""
    g = golden('rot_glue')
    assert rel_err(GT.convert_to_6D_rot(T(g['x72'])).cpu(), g['x75']) < 1e-5            # includes the Taylor-branch row 0
    assert np.abs(GT.convert_to_3D_rot(T(g['x75'])).cpu().numpy() - g['x72_back']).max() < 1e-4
    assert np.abs(GT.convert_to_3D_rot(T(g['x75_free'])).cpu().numpy() - g['x72_free']).max() < 1e-4
    assert rel_err(geometry.ContinousRotReprDecoder.decode(T(g['x75_free'][:, 3:9])).cpu(), g['R_free']) < 1e-5
    xn = GT.normalize_global_T(T(g['xt']), T(g['cam_int']), T(g['max_d']))
    assert rel_err(xn.cpu(), g['xn']) < 1e-5
    assert rel_err(GT.recover_global_T(xn, T(g['cam_int']), T(g['max_d'])).cpu(), g['xb']) < 1e-5
    assert rel_err(GT.verts_transform(T(g['verts']), T(g['cam_ext'])).cpu(), g['verts_t']) < 1e-5


def _quat_branch(R):
    """Which of the four candidate quaternions torchgeometry 0.1.2 picks (the branch structure of fit.hip rotmat_to_aa)."""
    m00, m11, m22 = R[0, 0], R[1, 1], R[2, 2]
    if m22 < 1e-6:
        return 0 if m00 > m11 else 1
    return 2 if m00 < -m11 else 3


def _decoder(smplx_data, vposer_sd, B):
    vp, _ = load_vposer(vposer_sd, vp_model='snapshot')
    vp.to(DEV)
    bm = body_model.create(smplx_data, model_type='smplx', gender='neutral', ext='npz', num_pca_comps=12, batch_size=B, device=DEV)
    return fitting.BodyDecoder(vp, bm, B, DEV), bm


def test_head_kernel_rotation_branches(smplx_data, vposer_sd):
    """The fused engine's head kernel (6D -> Gram-Schmidt -> R -> quaternion -> angle-axis for the global orientation) on the
    reference's fixture rows: every quaternion branch is hit and every row agrees with the reference output to 1e-4."""
    g = golden('rot_glue')
    rows75 = np.concatenate([g['x75_free'], g['x75']])                      # 16 non-orthonormal 6D rows + 16 converted rows
    want = np.concatenate([g['x72_free'][:, 3:6], g['x72_back'][:, 3:6]])
    # plus the exact identity (sin^2 == 0 branch of quaternion_to_angle_axis: k = 2) checked against the pinned oracle
    ident = g['x75'][:1].copy()
    ident[0, 3:9] = [1, 0, 0, 1, 0, 0]
    rows75 = np.concatenate([rows75, ident])
    want = np.concatenate([want, O.convert_to_3d_rot(torch.tensor(ident))[:, 3:6].numpy()])
    B = rows75.shape[0]
    dec, bm = _decoder(smplx_data, vposer_sd, B)
    cam = T(np.tile(np.eye(4, dtype=np.float32)[None], (B, 1, 1)))
    dec(T(rows75), cam)                                                       # psi_fit_decode_forward: head kernel + LBS
    pose = dec.engine.buffer('pose', (B, 165)).cpu().numpy()
    got = pose[:, :3] - bm.pose_mean.cpu().numpy()[:3]
    R = geometry.ContinousRotReprDecoder.decode(T(rows75[:, 3:9])).cpu().numpy()
    branches = {_quat_branch(r) for r in R}
    assert branches == {0, 1, 2, 3}, branches
    # angle-axis is compared as a rotation where the angle is near pi (the sign of the axis is ill-conditioned there), else directly
    Rg = O.aa2matrot(torch.tensor(got)).numpy()
    Rw = O.aa2matrot(torch.tensor(want)).numpy()
    assert np.abs(Rg - Rw).max() < 1e-4
    ang = np.linalg.norm(want, axis=1)
    ok = ang < 3.0
    assert np.abs(got[ok] - want[ok]).max() < 1e-4
    assert np.abs(got[-1]).max() == 0.0                                       # identity -> exactly zero rotation vector


def test_head_kernel_vposer_decode_golden(smplx_data, vposer_sd):
    """VPoser.decode(z, 'aa') of the reference (vposer_smpl.py:107-121,152-161) vs the engine's head kernel (the 32->512->512->126
    MLP + 21 x (6D -> R -> aa)) and vs the product's torch VPoser module on the GPU."""
    g = golden('vposer_decode')
    B = g['z'].shape[0]
    dec, bm = _decoder(smplx_data, vposer_sd, B)
    x75 = np.zeros((B, 75), np.float32)
    x75[:, 3:9] = [1, 0, 0, 1, 0, 0]
    x75[:, 19:51] = g['z']
    dec(T(x75), T(np.tile(np.eye(4, dtype=np.float32)[None], (B, 1, 1))))
    pose = dec.engine.buffer('pose', (B, 165)).cpu().numpy()
    got = pose[:, 3:66] - bm.pose_mean.cpu().numpy()[3:66]
    assert np.abs(got - g['aa']).max() < 1e-4 and rel_err(got, g['aa']) < 1e-4
    vp, _ = load_vposer(vposer_sd, vp_model='snapshot')
    vp.to(DEV)
    with torch.no_grad():
        aa = vp.decode(T(g['z']), output_type='aa').view(B, -1).cpu().numpy()
        mr = vp.decode(T(g['z']), output_type='matrot').view(B, -1).cpu().numpy()
    assert rel_err(aa, g['aa']) < 1e-4 and rel_err(mr, g['matrot']) < 1e-5


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[1] at full size against the oracle
# ------------------------------------------------------------------------------------------------------------------
def test_full_baseline_size_three_iteration_oracle_trajectory(smplx_data, vposer_sd):
    """fitting_proxe loop at B=32, n_c=2048, m=32768, 256^3 SDF: loss values of 3 iterations and the fitted parameters against
    the oracle (the CPU port pinned by the reference's golden vectors), not against another engine of this package."""
    B, m, n_c, D, iters = 32, 32768, 2048, 256, 3
    scene = synth.make_scene(0, m, D, n_c)
    bodies = synth.make_bodies(11, B)
    bodies['cam_ext'] = synth.make_cam_ext(5, B)
    cfg = {'scene_verts_path': None, 'scene_sdf_path': None, 'human_model_path': None, 'vposer_ckpt_path': None, 'init_lr_h': 0.1,
           'num_iter': iters, 'batch_size': B, 'device': torch.device(DEV), 'contact_part': synth.CONTACT_PARTS, 'contact_id_folder': None,
           'verbose': False, 'smplx_data': smplx_data, 'vposer_state': vposer_sd, 'scene': scene, 'engine': 'fused'}
    op = fitting.FittingOP(cfg, dict(LOSS))
    trace = arbiter.gpu_trace(op, dict(bodies), iters)
    O.set_threads(min(16, os.cpu_count() or 1))
    make = lambda dt: O.FittingOracle(O.SMPLXOracle(smplx_data, dtype=dt), vposer_sd, scene.verts, scene.sdf, scene.grid_min, scene.grid_max,
                                      synth.contact_ids_from_parts(scene.contact_parts), B)
    # every iteration from the product's own state: loss values, gradient and Adam update within K_NOISE x the fp32 oracle's own distance
    # from the fp64 arbiter (tests/arbiter.py) — no "99 % of the entries" allowance
    report = arbiter.check_trace(trace, make, np.asarray(bodies['cam_ext'], np.float64))
    arbiter.record('configs1_full_baseline_size', report)      # counts per rule -> gpurun_out/arbiter/, floor on rule (a)
    # and the free-running loss trajectory against the free-running fp32 oracle, as before
    fo = make(torch.float32)
    rec = []
    fo.fitting(synth.body_vector_72(bodies), bodies['cam_ext'], iters, record=rec)
    got_losses, rec = np.array([t['losses'] for t in trace]), np.array(rec)
    assert np.abs(got_losses[:2] - rec[:2]).max() < 1e-5, (got_losses, rec)


def test_fused_iteration_is_run_to_run_bit_identical(smplx_data, vposer_sd):
    """Race detection: two fresh engines on the same problem give bit-identical parameters, Adam moments and
    loss history after 25 iterations (no atomics on the data path, fixed reduction orders)."""
    B, m, n_c, D = 8, 8192, 512, 32
    scene = synth.make_scene(1, m, D, n_c)
    bodies = synth.make_bodies(21, B)
    bodies['cam_ext'] = synth.make_cam_ext(6, B)
    outs = []
    for _ in range(2):
        cfg = {'scene_verts_path': None, 'scene_sdf_path': None, 'human_model_path': None, 'vposer_ckpt_path': None, 'init_lr_h': 0.1,
               'num_iter': 25, 'batch_size': B, 'device': torch.device(DEV), 'contact_part': synth.CONTACT_PARTS, 'contact_id_folder': None,
               'verbose': False, 'smplx_data': smplx_data, 'vposer_state': vposer_sd, 'scene': scene, 'engine': 'fused'}
        op = fitting.FittingOP(cfg, dict(LOSS))
        op.fitting(dict(bodies))
        eng = op._fused
        x, hist, step = eng.read(25)
        outs.append((x.clone(), hist.clone(), eng.buffer('adam_m', (B, 75)), eng.buffer('adam_v', (B, 75)), step))
        del op
    assert outs[0][4] == outs[1][4] == 25
    for a, b in zip(outs[0][:4], outs[1][:4]):
        assert torch.equal(a, b)


def test_full_size_rerun_is_bit_identical_across_processes():
    """The same property at the BASELINE size and across fresh PROCESSES (fresh device memory, different launch timing): three processes fit the
    same 32 bodies for 25 iterations (m = 32768, n_c = 2048) and must agree to the last bit — what caught an unreproducible kernel variant in
    round 6 that the small in-process test above did not (profiles/r06_ab_skin_blend_mfma.txt)."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'check_rerun.py')
    out = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'bit-identical: True' in out.stdout, out.stdout[-500:]


# ------------------------------------------------------------------------------------------------------------------
# train_s2 at batch 128
# ------------------------------------------------------------------------------------------------------------------
def _s2_setup(tmp, smplx_data, vposer_sd, B, use_graph, bf16):
    from test_training_gpu import LW, _table, make_cfg
    from psi_release_amd import batch_gen
    scenes_d = {n: synth.make_scene(i, 2048, 32, 256) for i, n in enumerate(['A', 'B'])}
    scenes = {n: {'verts': s.verts, 'sdf': s.sdf, 'grid_min': s.grid_min, 'grid_max': s.grid_max, 'grid_dim': s.grid_dim}
              for n, s in scenes_d.items()}
    bg = batch_gen.BatchGeneratorWithSceneMesh.from_arrays(_table(2 * B, 2), scenes, DEV, indirect_sdf=True)
    batches = [bg.next_batch(B) for _ in range(2)]
    cfg = make_cfg(tmp, smplx_data, vposer_sd, scenes_d['A'], B, epoch=10)
    cfg.update(use_graph=use_graph, autocast_bf16=bf16, resume_training=False)
    torch.manual_seed(0)
    op = training.TrainOPS2(cfg, dict(LW))
    op.model_h.eval()                                       # BN on running statistics: the comparison is about the step, not batch noise
    return op, batches


def test_train_s2_batch128_graph_equals_eager_and_bf16_tolerance(tmp_path, smplx_data, vposer_sd, monkeypatch):
    """BASELINE configs[2] batch size: (a) the whole-step HIP graph reproduces the eager optimiser step at B=128 (losses of 3 steps,
    scene terms active); (b) the bf16-trunk losses stay within 2e-2 relative of the fp32 losses (bf16 has 8 mantissa bits and the
    trunk is ~20 layers deep; the scene / SMPL-X losses themselves are always evaluated in fp32)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    monkeypatch.setattr(torch, 'randn_like', lambda t, **kw: torch.zeros_like(t))
    B = 128
    hist = {}
    for key, (use_graph, bf16) in {'eager': (False, False), 'graph': (True, False), 'bf16': (False, True)}.items():
        op, batches = _s2_setup(tmp_path, smplx_data, vposer_sd, B, use_graph, bf16)
        h = []
        for i in range(3):
            h.append(torch.stack([l.detach().float().clone() for l in op.train_step(batches[i % 2], ep=9)]).cpu())
        hist[key] = torch.stack(h)
        del op
        torch.cuda.empty_cache()
    assert float(hist['eager'][0][4]) > 0 and float(hist['eager'][0][6]) >= 0        # contact term live at ep 9 of 10
    assert torch.allclose(hist['graph'], hist['eager'], rtol=2e-3, atol=1e-6), (hist['graph'], hist['eager'])
    e, b = hist['eager'][0], hist['bf16'][0]                                          # first step: same weights in both runs
    assert float(((e - b).abs() / (e.abs() + 1e-3)).max()) < 2e-2, (e, b)


# ------------------------------------------------------------------------------------------------------------------
# drop-in modules, imported the way the reference imports them
# ------------------------------------------------------------------------------------------------------------------
def test_dropin_modules_reference_import_pattern():
    """`import chamfer_pytorch.dist_chamfer as ext` (fitting_proxe.py:34), `import chamfer` with caller-allocated zero tensors
    (dist_chamfer.py:19-30,40-45), `import smplx; smplx.create(...)` (fitting_proxe.py:32,55)."""
    dropin = os.path.join(ROOT, 'psi-release_amd', 'dropin')
    sys.path.insert(0, dropin)
    try:
        for mod in ('chamfer', 'chamfer_pytorch', 'chamfer_pytorch.dist_chamfer', 'chamfer_pytorch.dist_chamfer_idx', 'smplx'):
            sys.modules.pop(mod, None)
        import chamfer
        import chamfer_pytorch.dist_chamfer as ext
        import chamfer_pytorch.dist_chamfer_idx as ext_idx
        import smplx
        rs = np.random.RandomState(5)
        x, y = rs.standard_normal((3, 200, 3)).astype(np.float32), rs.standard_normal((3, 700, 3)).astype(np.float32)
        rd1, ri1, rd2, ri2 = O.chamfer_nn_np(x, y)
        # the compiled-extension stand-in: caller allocates, 1 = ok
        xt, yt = T(x), T(y)
        d1, d2 = torch.zeros(3, 200).cuda(), torch.zeros(3, 700).cuda()
        i1, i2 = torch.zeros(3, 200).type(torch.IntTensor).cuda(), torch.zeros(3, 700).type(torch.IntTensor).cuda()
        assert chamfer.forward(xt, yt, d1, d2, i1, i2) == 1
        assert np.array_equal(d1.cpu().numpy(), rd1) and np.array_equal(i1.cpu().numpy(), ri1)
        assert np.array_equal(d2.cpu().numpy(), rd2) and np.array_equal(i2.cpu().numpy(), ri2)
        g1, g2 = T(rs.standard_normal((3, 200))), T(rs.standard_normal((3, 700)))
        gx1, gx2 = torch.zeros(xt.size()).cuda(), torch.zeros(yt.size()).cuda()
        assert chamfer.backward(xt, yt, gx1, gx2, g1, g2, i1, i2) == 1
        ox1, ox2 = O.chamfer_grad_np(x, y, g1.cpu().numpy(), g2.cpu().numpy(), ri1, ri2)
        assert rel_err(gx1.cpu(), ox1) < 1e-6 and rel_err(gx2.cpu(), ox2) < 1e-5
        # the module surface the fitting scripts use
        a1, a2 = ext.chamferDist()(xt, yt)
        assert torch.equal(a1, d1) and torch.equal(a2, d2)
        b1, b2, bi1, bi2 = ext_idx.chamferDist()(xt, yt)
        assert torch.equal(b1, d1) and torch.equal(bi1, i1) and torch.equal(bi2, i2)
        bm = smplx.create(synth.make_smplx(7), model_type='smplx', gender='neutral', ext='npz', num_pca_comps=12, create_global_orient=True,
                          create_body_pose=True, create_betas=True, create_left_hand_pose=True, create_right_hand_pose=True,
                          create_expression=True, create_jaw_pose=True, create_leye_pose=True, create_reye_pose=True, create_transl=True,
                          batch_size=2, device=DEV)
        out = bm(return_verts=True, body_pose=T(rs.standard_normal((2, 63)) * 0.3), transl=T(rs.standard_normal((2, 3))),
                 global_orient=T(rs.standard_normal((2, 3))), betas=T(rs.standard_normal((2, 10))),
                 left_hand_pose=T(rs.standard_normal((2, 12)) * 0.2), right_hand_pose=T(rs.standard_normal((2, 12)) * 0.2))
        assert out.vertices.shape == (2, 10475, 3) and torch.isfinite(out.vertices).all()
    finally:
        sys.path.remove(dropin)


# ------------------------------------------------------------------------------------------------------------------
# Chamfer: second arithmetic mode, concurrent streams
# ------------------------------------------------------------------------------------------------------------------
def test_chamfer_fma_build_bit_exact_against_fma_oracle():
    """libpsi_hip_fma.so (PSI_CHAMFER_FMA=1: distance as mul, fma, fma — nvcc's default --fmad=true contraction of chamfer.cu:32-35)
    against the oracle built in the same mode: distances and indices bit-exact, for the brute-force op and the kd-tree index;
    and the two modes really differ in the last bit of some distances."""
    path = os.path.join(os.path.dirname(hip.LIB_PATH), 'libpsi_hip_fma.so')
    L = ctypes.CDLL(path)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.psi_chamfer_forward.argtypes = [vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp]
    L.psi_nn_index_create.argtypes = [vp, vp, ci]
    L.psi_nn_index_query.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp]
    L.psi_nn_index_destroy.argtypes = [vp]
    assert L.psi_chamfer_arith_mode() == 1 and hip.lib().psi_chamfer_arith_mode() == 0
    rs = np.random.RandomState(9)
    B, n, m = 3, 700, 9000
    x = rs.standard_normal((B, n, 3)).astype(np.float32)
    y = rs.standard_normal((B, m, 3)).astype(np.float32)
    y[:, 100:140] = y[:, 50:90]                                                # exact ties: lowest index must win in both modes
    x[:, :20] = y[:, 100:120]                                                  # coincident query / target
    xt, yt = T(x), T(y)
    d1, d2 = torch.zeros(B, n, device=DEV), torch.zeros(B, m, device=DEV)
    i1, i2 = torch.zeros(B, n, dtype=torch.int32, device=DEV), torch.zeros(B, m, dtype=torch.int32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert L.psi_chamfer_forward(xt.data_ptr(), yt.data_ptr(), B, n, m, d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr(), None, st) == 0
    rd1, ri1, rd2, ri2 = O.chamfer_nn_np(x, y, fma=True)
    assert np.array_equal(d1.cpu().numpy(), rd1) and np.array_equal(i1.cpu().numpy(), ri1)
    assert np.array_equal(d2.cpu().numpy(), rd2) and np.array_equal(i2.cpu().numpy(), ri2)
    nd1 = O.chamfer_nn_np(x, y, both=False, fma=False)[0]
    assert (nd1 != rd1).any()                                                  # the modes are distinguishable ...
    assert np.abs(nd1 - rd1).max() <= 4 * np.finfo(np.float32).eps * np.abs(rd1).max()    # ... by rounding only
    # exact index over one static cloud, same mode
    h = ctypes.c_void_p()
    y0 = np.ascontiguousarray(y[0])
    assert L.psi_nn_index_create(ctypes.byref(h), y0.ctypes.data_as(vp), m) == 0
    kd, ki = torch.zeros(B, n, device=DEV), torch.zeros(B, n, dtype=torch.int32, device=DEV)
    assert L.psi_nn_index_query(h, xt.data_ptr(), B, n, kd.data_ptr(), ki.data_ptr(), None, st) == 0
    torch.cuda.synchronize()
    od, oi, _, _ = O.chamfer_nn_np(x, np.repeat(y0[None], B, 0), both=False, fma=True)
    assert np.array_equal(kd.cpu().numpy(), od) and np.array_equal(ki.cpu().numpy(), oi)
    L.psi_nn_index_destroy(h)


def test_chamfer_two_streams_without_workspace():
    """Two streams calling psi_chamfer_forward(workspace=NULL) concurrently: the internal scratch is per (device, stream), so the
    results equal the serial ones (they shared one growable per-device buffer before)."""
    rs = np.random.RandomState(3)
    cases = [(rs.standard_normal((4, 1500, 3)).astype(np.float32), rs.standard_normal((4, 20000, 3)).astype(np.float32)) for _ in range(2)]
    ref = [O.chamfer_nn_np(x, y, both=False)[:2] for x, y in cases]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ten = [(T(x), T(y)) for x, y in cases]
    torch.cuda.synchronize()
    L = hip.lib()
    outs = [(torch.zeros(4, 1500, device=DEV), torch.zeros(4, 1500, dtype=torch.int32, device=DEV)) for _ in range(2)]
    for rep in range(5):
        for o in outs:
            o[0].zero_(), o[1].zero_()
        torch.cuda.synchronize()
        for s, (xt, yt), (d1, i1) in zip(streams, ten, outs):
            hip.check(L.psi_chamfer_forward(xt.data_ptr(), yt.data_ptr(), 4, 1500, 20000, d1.data_ptr(), i1.data_ptr(), None, None, None,
                                            s.cuda_stream), 'psi_chamfer_forward')          # workspace = NULL: internal scratch
        torch.cuda.synchronize()
        for (d1, i1), (rd, ri) in zip(outs, ref):
            assert np.array_equal(d1.cpu().numpy(), rd) and np.array_equal(i1.cpu().numpy(), ri)
