"""Development tool: where the CPU oracle spends its time at the BASELINE shape, per thread count."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
import psi_oracle as O
from psi_release_amd import synth
B, nc, m, D = 32, 2048, 32768, 256
smplx, vp, scene = synth.make_smplx(7), synth.make_vposer_state(3), synth.make_scene(0, m, D, nc)
bodies = synth.make_bodies(11, B); xh = synth.body_vector_72(bodies)
for nt in [int(x) for x in sys.argv[1:]] or [8, 32, 64]:
    torch.set_num_threads(nt); os.environ['OMP_NUM_THREADS'] = str(nt)
    fo = O.FittingOracle(O.SMPLXOracle(smplx), vp, scene.verts, scene.sdf, scene.grid_min, scene.grid_max, synth.contact_ids_from_parts(scene.contact_parts), B)
    x = np.random.RandomState(0).uniform(-1.5, 1.5, (B, nc, 3)).astype(np.float32)
    y = np.broadcast_to(scene.verts, (B, m, 3)).copy()
    t = time.time(); O.chamfer_nn_np(x, y, both=False); t_ch1 = time.time() - t
    t = time.time(); O.chamfer_nn_np(x, y, both=True); t_ch2 = time.time() - t
    fo.fitting(xh, bodies['cam_ext'], 1)
    t = time.time(); fo.fitting(xh, bodies['cam_ext'], 2); t_it = (time.time() - t) / 2
    xr = O.convert_to_6d_rot(torch.tensor(xh)); fo.xhr_rec.data = xr.clone()
    t = time.time(); v = fo.body_verts(O.convert_to_3d_rot(fo.xhr_rec), torch.tensor(bodies['cam_ext'])); t_lbs = time.time() - t
    t = time.time(); s = O.sdf_sample(fo.sdf.expand(B, -1, -1, -1), fo.gmin, fo.gmax, v, True); t_sdf = time.time() - t
    t = time.time(); (v.sum() + s.sum()).backward(); t_bwd = time.time() - t
    print('threads %3d: iter %.2fs | chamfer dir1 %.2fs both %.2fs | lbs fwd %.2fs | sdf fwd %.2fs | lbs+sdf bwd %.2fs' % (nt, t_it, t_ch1, t_ch2, t_lbs, t_sdf, t_bwd), flush=True)
