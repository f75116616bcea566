"""dev (GPU box): two fresh PROCESSES fit the same full-size problem (B = 32, m = 32768, n_c = 2048, 64^3 SDF) for 25 iterations; parameters, Adam
moments and loss history must be bit-identical (the in-suite test of the same property, test_fused_iteration_is_run_to_run_bit_identical, is small)."""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from psi_release_amd import fitting, synth
    B = 32
    scene = synth.make_scene(5, 32768, 64, 2048)
    cfg = {'scene_verts_path': None, 'scene_sdf_path': None, 'human_model_path': None, 'vposer_ckpt_path': None, 'init_lr_h': 0.1, 'num_iter': 25,
           'batch_size': B, 'device': torch.device('cuda'), 'contact_part': synth.CONTACT_PARTS, 'contact_id_folder': None, 'verbose': False,
           'smplx_data': synth.make_smplx(7), 'vposer_state': synth.make_vposer_state(3), 'scene': scene, 'engine': 'fused', 'align_corners': True}
    op = fitting.FittingOP(cfg, {'weight_loss_rec': 1, 'weight_loss_vposer': 0.01, 'weight_contact': 0.1, 'weight_collision': 0.5})
    bodies = synth.make_bodies(23, B); bodies['cam_ext'] = synth.make_cam_ext(9, B)
    op.fitting(dict(bodies))
    eng = op._fused
    x, hist, step = eng.read(25)
    np.save(sys.argv[1], np.concatenate([x.cpu().numpy().ravel(), hist.cpu().numpy().ravel(), eng.buffer('adam_m', (B, 75)).cpu().numpy().ravel(),
                                         eng.buffer('adam_v', (B, 75)).cpu().numpy().ravel()]))
    sys.exit(0)
outs = []
for i in range(3):
    subprocess.check_call([sys.executable, __file__, '/tmp/rr%d.npy' % i], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    outs.append(np.load('/tmp/rr%d.npy' % i))
print('three fresh processes bit-identical:', all(np.array_equal(outs[0], o) for o in outs[1:]), 'max diff', max(float(np.abs(outs[0] - o).max()) for o in outs[1:]))
