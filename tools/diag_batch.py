"""dev: first-iteration gradient of the fused engine vs autograd over the HIP operators (modular engine) at several batch sizes:
Adam's first moment after one step is 0.1 * gradient in both."""
import sys, os
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
for p in ('', 'tests', 'oracle'): sys.path.insert(0, os.path.join(R, p))
import numpy as np, torch
from psi_release_amd import fitting, synth
import test_fitting_gpu as T
smplx_data, vposer_sd = synth.make_smplx(7), synth.make_vposer_state(3)
scene = synth.make_scene(3, 3000, 24, 300)
for B in (6, 33, 40, 64):
    bodies = synth.make_bodies(21, B); bodies['cam_ext'] = synth.make_cam_ext(7, B)
    g = {}
    for engine in ('modular', 'fused'):
        op = T.make_op(smplx_data, vposer_sd, scene, B, engine, num_iter=1, cls=fitting.FittingOP, lr=0.05)
        op.fitting(dict(bodies))
        if engine == 'fused':
            g[engine] = op._fused.buffer('adam_m', (B, 75)).cpu().numpy() * 10
        else:
            g[engine] = op.optimizer.state[op.xhr_rec]['exp_avg'].detach().cpu().numpy() * 10
    d = np.abs(g['fused'] - g['modular'])
    scale = np.abs(g['modular']).max()
    print('B %3d  max|g| %.3e  max abs diff %.3e  (%.2e of max|g|)  smallest |g| %.2e  elements with |g| < 1e-7: %d' %
          (B, scale, d.max(), d.max() / scale, np.abs(g['modular']).min(), int((np.abs(g['modular']) < 1e-7).sum())))
