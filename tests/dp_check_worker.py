"""Fit 4 bodies for 5 iterations under torchrun and print the result checksum; with PSI_FORCE_DP_PATH=1 through the
data-parallel sequence over a 1-rank nccl group (tests/test_dist_gpu.py::test_rccl_leg_at_world_size_one)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from psi_release_amd import dist as psi_dist, fitting, synth
psi_dist.init_from_env()
import torch.distributed as tdist
print('backend', tdist.get_backend() if tdist.is_initialized() else 'none', 'force', os.environ.get('PSI_FORCE_DP_PATH'))
cfg = {'scene_verts_path': None, 'scene_sdf_path': None, 'human_model_path': None, 'vposer_ckpt_path': None, 'init_lr_h': 0.1,
       'num_iter': int(os.environ.get('PSI_TEST_ITERS', '5')), 'batch_size': 4, 'device': torch.device('cuda', 0), 'contact_part': synth.CONTACT_PARTS,
       'contact_id_folder': None, 'verbose': False, 'smplx_data': synth.make_smplx(7), 'vposer_state': synth.make_vposer_state(3),
       'scene': synth.make_scene(3, 3000, 16, 300), 'engine': 'fused'}
bodies = synth.make_bodies(51, 4); bodies['cam_ext'] = synth.make_cam_ext(4, 4)
op = fitting.FittingOP(cfg, {'weight_loss_rec': 1, 'weight_loss_vposer': 0.01, 'weight_contact': 0.1, 'weight_collision': 0.5})
op.fitting(dict(bodies))
x = op.xhr_rec.detach().cpu().numpy()
print('stats buffer', [round(float(v), 3) for v in op._fused.stats.cpu()[:6]]); print('checksum %.6f' % float(np.abs(x).sum()))
if tdist.is_initialized():
    tdist.barrier(); psi_dist.rccl_comm_release(); tdist.destroy_process_group()
