"""Shared main() of train_s1.py / train_s2.py: the reference's argparse flags (train_s1.py:345-366) + path flags."""
import argparse
import os
import sys

import _common  # noqa: F401
import numpy as np
import torch


def main(stage, argv=None):
    from psi_release_amd import batch_gen, dist as psi_dist, synth, training
    p = argparse.ArgumentParser()
    p.add_argument('--save_dir', type=str, default=os.getcwd(), help='dir for checkpoints')
    p.add_argument('--batch_size', type=int, default=128 if stage == 's2' else 32)
    p.add_argument('--lr_s', type=float, default=0.001)
    p.add_argument('--lr_h', type=float, default=0.0001)
    p.add_argument('--num_epoch', type=int, default=50)
    p.add_argument('--weight_loss_vposer', type=float, default=1e-3)
    p.add_argument('--weight_loss_kl', type=float, default=1e-1)
    p.add_argument('--weight_loss_contact', type=float, default=1e-1)
    p.add_argument('--weight_loss_collision', type=float, default=1e-1)
    p.add_argument('--only_vircam', type=int, default=0)
    p.add_argument('--use_all', type=int, default=0)
    p.add_argument('--dataset_path', default='/is/cluster/yzhang/PROXE')
    p.add_argument('--human_model_path', default='/is/ps2/yzhang/body_models/VPoser')
    p.add_argument('--vposer_ckpt_path', default='/is/ps2/yzhang/body_models/VPoser/vposer_v1_0')
    p.add_argument('--scene_model_ckpt', default=None, help='data/resnet18.pth (a missing blob in the reference tree)')
    p.add_argument('--bf16', type=int, default=0, help='bf16 autocast for the CVAE trunk (losses stay fp32)')
    p.add_argument('--use_graph', type=int, default=0, help='replay each optimiser step as one HIP graph (single process)')
    p.add_argument('--synthetic', type=int, default=0, help='N>0: train on N synthetic samples (licensed data absent)')
    a = p.parse_args(argv)
    if a.save_dir == 'None':
        print('[error] the checkpoint save directory should be specified.')
        sys.exit(0)
    psi_dist.init_from_env()
    device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
    trainconfig = {
        'scene_model_ckpt': a.scene_model_ckpt, 'human_model_path': a.human_model_path, 'vposer_ckpt_path': a.vposer_ckpt_path,
        'init_lr_s': a.lr_s, 'init_lr_h': a.lr_h, 'batch_size': a.batch_size, 'epoch': a.num_epoch, 'loss_weight_anealing': True,
        'device': device, 'fine_tuning': None, 'save_dir': a.save_dir,
        'contact_id_folder': os.path.join(a.dataset_path, 'body_segments'),
        'contact_part': ['back', 'butt', 'L_Hand', 'R_Hand', 'L_Leg', 'R_Leg', 'thighs'], 'saving_per_X_ep': 2, 'verbose': True,
        'use_cont_rot': True, 'resume_training': True, 'autocast_bf16': bool(a.bf16), 'use_graph': bool(a.use_graph)}
    lossconfig = {'weight_loss_rec_s': 1.0, 'weight_loss_rec_h': 1.0, 'weight_loss_vposer': a.weight_loss_vposer,
                  'weight_loss_kl': a.weight_loss_kl, 'weight_contact': a.weight_loss_contact, 'weight_collision': a.weight_loss_collision}
    if a.synthetic > 0:
        names = ['SynA', 'SynB']
        sd = {n: synth.make_scene(i, 8192, 64, 512) for i, n in enumerate(names)}
        scenes = {n: {'verts': s.verts, 'sdf': s.sdf, 'grid_min': s.grid_min, 'grid_max': s.grid_max, 'grid_dim': s.grid_dim} for n, s in sd.items()}
        rs = np.random.RandomState(psi_dist.rank())
        n = a.synthetic
        body = synth.body_vector_72(synth.make_bodies(psi_dist.rank(), n))
        body[:, 2] = np.abs(body[:, 2]) + 2.0
        t = {'depth': rs.uniform(-1, 1, (n, 1, 128, 128)), 'seg': rs.uniform(-1, 1, (n, 1, 128, 128)), 'body': body,
             'cam_ext': synth.make_cam_ext(0, n), 'cam_int': synth.make_bodies(0, n)['cam_int'], 'max_d': np.full(n, 6.0),
             'sceneid': rs.randint(0, 2, n).astype(np.float32)}
        table = {k: np.concatenate([np.zeros_like(np.asarray(v)[:1]), np.asarray(v)]).astype(np.float32) for k, v in t.items()}
        bg = batch_gen.BatchGeneratorWithSceneMesh.from_arrays(table, scenes, device, indirect_sdf=True)
        trainconfig.update(smplx_data=synth.make_smplx(7), vposer_state=synth.make_vposer_state(3), contact_parts_data=sd['SynA'].contact_parts)
    else:
        files = [os.path.join(a.dataset_path, 'virtualcams_v2.hdf5')] if a.only_vircam == 1 else \
            [os.path.join(a.dataset_path, 'virtualcams_v2.hdf5'), os.path.join(a.dataset_path, 'realcams_v2.hdf5')]
        bg = batch_gen.BatchGeneratorWithSceneMesh(dataset_path=files, scene_verts_path=os.path.join(a.dataset_path, 'scenes_downsampled'),
                                                   scene_sdf_path=os.path.join(a.dataset_path, 'scenes_sdf'),
                                                   mode='all' if a.use_all == 1 else 'train', device=device, read_all_to_ram=True,
                                                   indirect_sdf=True, rank=psi_dist.rank(), world=psi_dist.world_size(), seed=0)
    cls = training.TrainOP if stage == 's1' else training.TrainOPS2
    cls(trainconfig, lossconfig).train(bg)
