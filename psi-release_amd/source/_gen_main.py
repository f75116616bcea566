"""The `__main__` blocks of the reference's four generation drivers behind one pair of functions.

source/test_proxe_s1.py:205-246 and test_proxe_s2.py:206-247 differ only in the model class and checkpoint they load, as do
test_habitat_s1.py:224-260 and test_habitat_s2.py:232-268; the four scripts of this directory keep the reference's file names and call
`main_proxe` / `main_habitat` with their stage (`--stage` still overrides it)."""
import argparse
import os

import _common  # noqa: F401
import torch

from psi_release_amd.generation import TestOP

ROOMS = ['17DRP5sb8fy-bedroom', '17DRP5sb8fy-familyroomlounge', '17DRP5sb8fy-livingroom', 'sKLMLpTHeUy-familyname_0_1',
         'X7HyMhZNoso-livingroom_0_16', 'zsNo4HB9uLZ-bedroom0_0', 'zsNo4HB9uLZ-livingroom0_13']
SNAPSHOTS = ['MPH16_00157_01', 'N0SittingBooth_00162_01', 'MPH1Library_00034_01', 'N3OpenArea_00157_01']


def main_habitat(stage, argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--ckpt_dir', required=True)
    ap.add_argument('--mp3dr_path', default='/is/cluster/yzhang/mp3d-rooms')
    ap.add_argument('--outdir', default='results_habitat_stage%s_sceneloss/virtualcams' % stage[1])
    ap.add_argument('--n_samples', type=int, default=200)
    ap.add_argument('--stage', default=stage, choices=['s1', 's2'])
    ap.add_argument('--bf16', type=int, default=0)
    a = ap.parse_args(argv)
    for scene in ROOMS:
        print('[INFO] processing: ' + scene)
        TestOP({'outdir': os.path.join(a.outdir, scene), 'ckpt_dir': a.ckpt_dir, 'stage': a.stage, 'n_samples': a.n_samples,
                'device': torch.device('cuda' if torch.cuda.is_available() else 'cpu'), 'use_cont_rot': True,
                'autocast_bf16': bool(a.bf16), 'test_data_path': os.path.join(a.mp3dr_path, scene + '-sensor')}).test_habitat()


def main_proxe(stage, argv=None):
    from psi_release_amd.batch_gen import BatchGeneratorTest
    ap = argparse.ArgumentParser()
    ap.add_argument('--ckpt_dir', required=True)
    ap.add_argument('--proxe_path', default='/home/yzhang/Videos/PROXE')
    ap.add_argument('--output_dir', default='results_prox_stage%s_nosceneloss/virtualrealcams' % stage[1])
    ap.add_argument('--n_samples', type=int, default=300)
    ap.add_argument('--stage', default=stage, choices=['s1', 's2'])
    a = ap.parse_args(argv)
    dev = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
    for snap in SNAPSHOTS:
        path = os.path.join(a.proxe_path, 'snapshot_for_testing/' + snap)
        bg = BatchGeneratorTest(dataset_path=path, device=dev)
        bg.reset()
        op = TestOP({'ckpt_dir': a.ckpt_dir, 'stage': a.stage, 'n_samples': a.n_samples, 'device': dev, 'use_cont_rot': True,
                     'output_dir': a.output_dir, 'test_data_path': path})
        op.test_proxe(bg.next_batch(batch_size=1), scene_name=snap.split('_')[0])
