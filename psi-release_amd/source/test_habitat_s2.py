#!/usr/bin/env python
"""Generation for the MP3D-R rooms (source/test_habitat_s2.py __main__; --stage s1 gives test_habitat_s1.py)."""
import argparse
import os

import _common  # noqa: F401
import torch

from psi_release_amd.generation import TestOP

ROOMS = ['17DRP5sb8fy-bedroom', '17DRP5sb8fy-familyroomlounge', '17DRP5sb8fy-livingroom', 'sKLMLpTHeUy-familyname_0_1',
         'X7HyMhZNoso-livingroom_0_16', 'zsNo4HB9uLZ-bedroom0_0', 'zsNo4HB9uLZ-livingroom0_13']

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--ckpt_dir', required=True)
    ap.add_argument('--mp3dr_path', default='/is/cluster/yzhang/mp3d-rooms')
    ap.add_argument('--outdir', default='results_habitat_stage2_sceneloss/virtualcams')
    ap.add_argument('--n_samples', type=int, default=200)
    ap.add_argument('--stage', default='s2', choices=['s1', 's2'])
    ap.add_argument('--bf16', type=int, default=0)
    a = ap.parse_args()
    for scene in ROOMS:
        print('[INFO] processing: ' + scene)
        TestOP({'outdir': os.path.join(a.outdir, scene), 'ckpt_dir': a.ckpt_dir, 'stage': a.stage, 'n_samples': a.n_samples,
                'device': torch.device('cuda' if torch.cuda.is_available() else 'cpu'), 'use_cont_rot': True,
                'autocast_bf16': bool(a.bf16), 'test_data_path': os.path.join(a.mp3dr_path, scene + '-sensor')}).test_habitat()
