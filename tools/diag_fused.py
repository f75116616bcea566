import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'oracle'))
from psi_release_amd import fitting, synth
from test_fitting_gpu import make_op
smplx, vp = synth.make_smplx(7), synth.make_vposer_state(3)
B=3
scene = synth.make_scene(3, 3000, 24, 300)
bodies = synth.make_bodies(21, B); bodies['cam_ext'] = synth.make_cam_ext(7, B)
ops = {e: make_op(smplx, vp, scene, B, e, num_iter=1, lr=0.05) for e in ('modular','fused')}
for e,o in ops.items(): o.use_graph = False
runners = {e: o.make_step_runner(dict(bodies)) for e,o in ops.items()}
for it in range(8):
    for e in runners: runners[e].step()
    runners['fused'].finish()
    xm = ops['modular'].xhr_rec.detach(); xf = ops['fused'].xhr_rec.detach()
    d = (xm-xf).abs()
    print(it, 'max diff', float(d.max()), 'argmax', int(d.argmax()), 'losses m', ['%.6f'%v for v in runners['modular'].last_losses()], 'f', ['%.6f'%v for v in runners['fused'].last_losses()])
    if it == 0:
        gm = ops['modular'].xhr_rec.grad.cpu().numpy()
        mf = ops['fused']._fused.buffer('adam_m', (B,75)).cpu().numpy()/0.1
        print('grad rel err', np.abs(gm-mf).max()/np.abs(gm).max(), 'abs', np.abs(gm-mf).max())
        i = np.unravel_index(np.abs(gm-mf).argmax(), gm.shape); print(i, gm[i], mf[i])
