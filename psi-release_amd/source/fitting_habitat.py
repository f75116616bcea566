#!/usr/bin/env python
"""python fitting_habitat.py GEN_PATH FIT_PATH     (source/fitting_habitat.py:230-289: MP3D-R rooms, batch 1, 50 iterations)"""
import argparse
import os

import _common  # noqa: F401
import torch

from psi_release_amd.fitting import FittingOPHabitat

ROOMS = ['17DRP5sb8fy-bedroom', '17DRP5sb8fy-familyroomlounge', '17DRP5sb8fy-livingroom', 'sKLMLpTHeUy-familyname_0_1',
         'X7HyMhZNoso-livingroom_0_16', 'zsNo4HB9uLZ-bedroom0_0', 'zsNo4HB9uLZ-livingroom0_13']


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('gen_path', nargs='?')
    ap.add_argument('fit_path')
    ap.add_argument('--mp3dr_path', default='/is/cluster/yzhang/mp3d-rooms')
    ap.add_argument('--human_model_path', default='/is/ps2/yzhang/body_models/VPoser')
    ap.add_argument('--vposer_ckpt_path', default='/is/ps2/yzhang/body_models/VPoser/vposer_v1_0')
    ap.add_argument('--contact_id_folder', default='/is/cluster/yzhang/PROXE/body_segments')
    ap.add_argument('--scenes', nargs='*', default=ROOMS)
    ap.add_argument('--num_iter', type=int, default=50)
    ap.add_argument('--max_files', type=int, default=10000)
    ap.add_argument('--engine', default='fused', choices=['fused', 'modular'])
    ap.add_argument('--verbose', action='store_true')
    ap.add_argument('--synthetic', default=None)
    ap.add_argument('--batch_size', type=int, default=1, help='bodies per pkl (reference: 1; BASELINE configs[4]: 512 over 8 GPUs)')
    ap.add_argument('--shard', default='files', choices=['files', 'rows'], help='under torchrun: shard the pkl files or the rows of every batch')
    ap.add_argument('--concurrency', type=int, default=1,
                    help='independent pkl files in flight per GPU (each on its own fused engine and HIP stream; implies a fresh Adam state per file)')
    ap.add_argument('--pack', type=int, default=1,
                    help='fit this many pkl files as ONE engine run with per-body loss normalisers (identical results to one-by-one fits run with --reset_optimizer: every file starts from a fresh Adam state; '
                         'the reference fits one batch-1 file at a time, which leaves the GPU idle)')
    ap.add_argument('--reset_optimizer', action='store_true',
                    help='fresh Adam state for every file (the reference carries one optimizer across the files of a scene, fitting_proxe.py:73-74; '
                         'with --shard files the carried state depends on which files a rank sees)')
    ap.add_argument('--save_all_rows', action='store_true', help='batch_size > 1: write every fitted row (the reference keeps the last one)')
    a = ap.parse_args(argv)
    rank, world = _common.dist_setup()
    extra = {}
    if a.synthetic:
        if rank == 0:
            _common.synthetic_prox_tree(a.synthetic, a.scenes, batch=a.batch_size)
        if world > 1:
            torch.distributed.barrier()
        root, a.gen_path, extra['smplx_data'], extra['vposer_state'] = _common.synthetic_prox_tree(a.synthetic, a.scenes, batch=a.batch_size, write=False)
        sdf_dir, ply_dir, a.contact_id_folder = os.path.join(root, 'scenes_sdf'), os.path.join(root, 'scenes_downsampled'), os.path.join(root, 'body_segments')
    else:
        sdf_dir, ply_dir = os.path.join(a.mp3dr_path, 'sdf'), os.path.join(a.mp3dr_path, 'mesh')
    for scenename in a.scenes:
        cfg = {'scene_verts_path': os.path.join(ply_dir, scenename + '.ply'), 'scene_sdf_path': os.path.join(sdf_dir, scenename),
               'human_model_path': a.human_model_path, 'vposer_ckpt_path': a.vposer_ckpt_path, 'init_lr_h': 0.1,
               'num_iter': a.num_iter, 'batch_size': a.batch_size,
               'device': torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu'),
               'contact_part': ['back', 'butt', 'L_Hand', 'R_Hand', 'L_Leg', 'R_Leg', 'thighs'],
               'contact_id_folder': a.contact_id_folder, 'verbose': a.verbose, 'engine': a.engine, 'save_all_rows': a.save_all_rows, 'reset_optimizer': a.reset_optimizer}
        cfg.update(extra)
        lossconfig = {'weight_loss_rec': 1, 'weight_loss_vposer': 0.01, 'weight_contact': 0.1, 'weight_collision': 0.5}
        _common.fit_files(FittingOPHabitat, cfg, lossconfig, os.path.join(a.gen_path, scenename), os.path.join(a.fit_path, scenename),
                          a.max_files, a.shard, rank, world, a.concurrency, a.pack)
    if world > 1:
        torch.distributed.barrier()
        from psi_release_amd import dist as psi_dist
        psi_dist.rccl_comm_release()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
