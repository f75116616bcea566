#!/usr/bin/env python
"""Entry point with the reference's CLI:  python fitting_proxe.py GEN_PATH FIT_PATH     (source/fitting_proxe.py:217-263)

For every test scene and every ``GEN_PATH/<scene>/body_gen_%06d.pkl`` that has no output yet, fit the body to the scene
and write ``FIT_PATH/<scene>/body_gen_%06d.pkl``.  Paths that the reference hard-codes are flags here; ``--synthetic DIR``
creates stand-in assets (the licensed PROX-E / SMPL-X / VPoser files do not ship) and runs on those.
"""
import argparse
import os

import _common  # noqa: F401
import torch

from psi_release_amd.fitting import FittingOP


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('gen_path', nargs='?')
    ap.add_argument('fit_path')
    ap.add_argument('--proxe_path', default='/home/yzhang/Videos/PROXE')
    ap.add_argument('--human_model_path', default='/home/yzhang/body_models/VPoser')
    ap.add_argument('--vposer_ckpt_path', default='/home/yzhang/body_models/VPoser/vposer_v1_0')
    ap.add_argument('--scenes', nargs='*', default=['MPH16', 'MPH1Library', 'N0SittingBooth', 'N3OpenArea'])
    ap.add_argument('--num_iter', type=int, default=20)
    ap.add_argument('--batch_size', type=int, default=1)
    ap.add_argument('--init_lr_h', type=float, default=0.1)
    ap.add_argument('--max_files', type=int, default=1200)
    ap.add_argument('--engine', default='fused', choices=['fused', 'modular'])
    ap.add_argument('--align_corners', type=int, default=1, help='1 = torch 1.2.0 (pinned) semantics of the SDF lookup')
    ap.add_argument('--verbose', action='store_true')
    ap.add_argument('--synthetic', default=None, help='directory to create synthetic stand-in assets in')
    ap.add_argument('--shard', default='files', choices=['files', 'rows'],
                    help="under torchrun: 'files' = every rank fits its own pkl files (independent problems); 'rows' = every file's batch is "
                         "split over the ranks with one all-reduce of the loss normalisers per iteration (BASELINE configs[3])")
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
        a.proxe_path, a.gen_path, extra['smplx_data'], extra['vposer_state'] = _common.synthetic_prox_tree(
            a.synthetic, a.scenes, batch=a.batch_size, write=False)
    for scenename in a.scenes:
        fittingconfig = {
            'scene_verts_path': os.path.join(a.proxe_path, 'scenes_downsampled/' + scenename + '.ply'),
            'scene_sdf_path': os.path.join(a.proxe_path, 'scenes_sdf/' + scenename),
            'human_model_path': a.human_model_path, 'vposer_ckpt_path': a.vposer_ckpt_path,
            'init_lr_h': a.init_lr_h, 'num_iter': a.num_iter, 'batch_size': a.batch_size,
            'device': torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu'),
            'contact_part': ['back', 'butt', 'L_Hand', 'R_Hand', 'L_Leg', 'R_Leg', 'thighs'],
            'contact_id_folder': os.path.join(a.proxe_path, 'body_segments'), 'verbose': a.verbose,
            'engine': a.engine, 'align_corners': bool(a.align_corners), 'save_all_rows': a.save_all_rows, 'reset_optimizer': a.reset_optimizer}
        fittingconfig.update(extra)
        lossconfig = {'weight_loss_rec': 1, 'weight_loss_vposer': 0.01, 'weight_contact': 0.1, 'weight_collision': 0.5}
        _common.fit_files(FittingOP, fittingconfig, lossconfig, os.path.join(a.gen_path, scenename), os.path.join(a.fit_path, scenename),
                          a.max_files, a.shard, rank, world, a.concurrency, a.pack)
    if world > 1:
        torch.distributed.barrier()
        from psi_release_amd import dist as psi_dist
        psi_dist.rccl_comm_release()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
