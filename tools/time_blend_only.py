"""dev (GPU box, under rocprofv3 --kernel-trace --stats): the LBS forward alone in a loop — blend_fwd re-reads the same 64.5 MB matrix
back to back with only small kernels between (is the stream faster when the matrix can stay in the 256 MB Infinity Cache?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from psi_release_amd import body_model, synth
lay = body_model.create(synth.make_smplx(7), batch_size=1, device='cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rs = np.random.RandomState(B)
betas = torch.tensor(rs.standard_normal((B, 20)), dtype=torch.float32, device='cuda')
pose = torch.tensor(rs.standard_normal((B, 165)) * 0.3, dtype=torch.float32, device='cuda')
with torch.no_grad():
    for _ in range(2000):
        v = body_model.lbs(lay.lbs_model, betas, pose)
torch.cuda.synchronize()
