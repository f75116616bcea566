"""dev: where do a 2-rank x 32-body fit, a single-process 64-body fit and the oracle on 64 bodies differ? (tests/test_configs_gpu.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
import torch.multiprocessing as mp
import test_configs_gpu as tc
import test_configs_dp_gpu as tcd
from psi_release_amd import fitting, synth
import psi_oracle as O

def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 13
    B = 64
    smplx, vp = synth.make_smplx(7), synth.make_vposer_state(3)
    scene = synth.make_scene(0, tc.M, tc.D, tc.NC)
    bodies = synth.make_bodies(seed, B); bodies['cam_ext'] = synth.make_cam_ext(9, B)
    op = fitting.FittingOP(tc._cfg(smplx, vp, scene, B), dict(tc.LOSS))
    r = op.make_step_runner(dict(bodies)); L = []
    xs = []
    for _ in range(tc.ITERS):
        r.step(); L.append(r.last_losses()); xs.append(op._fused.read(0)[0].cpu().numpy().copy())
    r.finish()
    x1 = tc.GT.convert_to_3D_rot(op.xhr_rec).detach().cpu().numpy()
    fo = tc._oracle(smplx, vp, scene, B)
    rec = []
    # oracle step by step
    xh72 = synth.body_vector_72(bodies)
    xhr = O.convert_to_6d_rot(torch.as_tensor(xh72)); cam = torch.as_tensor(bodies['cam_ext'])
    fo.xhr_rec.data = xhr.clone(); xo = []
    for _ in range(tc.ITERS):
        fo.optimizer.zero_grad(); ls = fo.cal_loss(xhr, cam); rec.append([float(l) for l in ls]); sum(ls).backward(); fo.optimizer.step(); xo.append(fo.xhr_rec.detach().numpy().copy())
    for it in range(tc.ITERS):
        e = np.abs(xs[it] - xo[it])
        bad = np.argwhere(e > 2e-3)
        print('iter', it + 1, 'single64 vs oracle64: max', e.max(), 'n>2e-3', len(bad), 'cols', sorted(set(bad[:, 1].tolist()))[:30], 'bodies', sorted(set(bad[:, 0].tolist()))[:20])
        print('   losses gpu', L[it], 'oracle', rec[it])
    if len(sys.argv) > 2:
        port = tc._free_port(); tmp = '/tmp/diagc3'; os.makedirs(tmp, exist_ok=True)
        mp.spawn(tcd._rank_worker, args=(2, port, tmp), nprocs=2, join=True)
        xg = np.concatenate([np.load(tmp + '/x%d.npy' % r) for r in range(2)])
        e = np.abs(xg - x1); bad = np.argwhere(e > 2e-3)
        print('dp2x32 vs single64: max', e.max(), 'n>2e-3', len(bad), 'cols', sorted(set(bad[:, 1].tolist()))[:30])

if __name__ == '__main__':
    main()
