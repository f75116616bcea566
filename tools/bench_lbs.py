"""Development tool: LBS forward/backward kernels in isolation at several batch sizes (run under rocprofv3 for per-kernel times)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from psi_release_amd import body_model, synth
lay = body_model.create(synth.make_smplx(7), batch_size=1, device='cuda')
for B in [int(x) for x in sys.argv[1:]] or [16, 32, 64]:
    rs = np.random.RandomState(B)
    betas = torch.tensor(rs.standard_normal((B, 20)), dtype=torch.float32, device='cuda', requires_grad=True)
    pose = torch.tensor(rs.standard_normal((B, 165)) * 0.3, dtype=torch.float32, device='cuda', requires_grad=True)
    w = torch.randn(B, 10475, 3, device='cuda')
    for _ in range(20):
        v = body_model.lbs(lay.lbs_model, betas, pose)
        (v * w).sum().backward()
    torch.cuda.synchronize()
