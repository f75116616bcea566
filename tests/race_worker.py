"""Worker of tests/test_output_race_cpu.py: waits on a file barrier, then writes results into directories that do not exist yet."""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psi_release_amd import fitting, generation  # noqa: E402

root, me, n_rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
op = types.SimpleNamespace(cam_ext=torch.eye(4)[None], cam_int=torch.eye(3)[None], save_all_rows=False)
x = torch.zeros(1, 72)
for r in range(n_rounds):
    go = os.path.join(root, 'go_%d' % r)
    open(os.path.join(root, 'ready_%d_%d' % (r, me)), 'w').close()
    while not os.path.exists(go):                   # released by the test once both workers are parked here
        time.sleep(0.0002)
    d = os.path.join(root, 'fit_%d' % r, 'scene')
    fitting.FittingOP.save_result(op, x, os.path.join(d, 'body_gen_%06d.pkl' % me))
    generation.TestOP.write([{'transl': x.numpy()}], os.path.join(root, 'gen_%d' % r, 'scene'), first_index=me)
print('done', me)
