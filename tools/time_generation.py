"""dev: the generation (sampling) path of HumanCVAES2 at the reference's n_samples = 200 (test_habitat_s2.py:243) with the dense layers on
the hand-written MFMA kernels vs the PyTorch library path (PSI_HIP_LINEAR=0), bf16 trunk.  -> gpurun_out/generation_times.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from psi_release_amd import generation, synth

dev = torch.device('cuda')
out = {}
for mode in ('1', '0'):
    os.environ['PSI_HIP_LINEAR'] = mode
    op = generation.TestOP({'ckpt_dir': '', 'device': dev, 'n_samples': 200, 'use_cont_rot': True, 'stage': 's2', 'autocast_bf16': True, 'outdir': '/tmp/x'})
    shapes = {k: tuple(v.shape) for k, v in op.model_h.state_dict().items()}
    op.load({k: torch.tensor(v) for k, v in synth.make_state_like(shapes, 1).items()})
    depth = torch.rand(1, 1, 128, 128, device=dev) * 2 - 1
    seg = torch.rand(1, 1, 128, 128, device=dev) * 2 - 1
    cam_int = torch.tensor([[[500.0, 0, 320], [0, 500, 240], [0, 0, 1]]], device=dev)
    cam_ext = torch.eye(4, device=dev)[None]
    max_d = torch.tensor([6.0], device=dev)
    xs = torch.cat([depth, seg], 1).repeat(200, 1, 1, 1)
    with torch.no_grad():
        for _ in range(5):
            op.model_h.sample(xs)
            op.model_h.sample(xs[:1], rows=200)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            op.model_h.sample(xs[:1], rows=200)         # the view encoded once (round 3)
        torch.cuda.synchronize()
        dt_once = (time.perf_counter() - t0) / 30
        t0 = time.perf_counter()
        for _ in range(30):
            op.model_h.sample(xs)                        # n copies of the view through the trunk, like the reference
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
        dtg = {}
        for tag, fn in (('copies', lambda: op.model_h.sample(xs)), ('once', lambda: op.model_h.sample(xs[:1], rows=200))):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                y = fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                g.replay()
            torch.cuda.synchronize()
            dtg[tag] = (time.perf_counter() - t0) / 30
    out['hip_linear=' + mode] = {'ms_per_view_eager_200_copies': round(dt * 1e3, 3), 'ms_per_view_eager_encoded_once': round(dt_once * 1e3, 3),
                                 'ms_per_view_graph_200_copies': round(dtg['copies'] * 1e3, 3), 'ms_per_view_graph_encoded_once': round(dtg['once'] * 1e3, 3),
                                 'bodies_per_s_graph_encoded_once': round(200 / dtg['once'], 1)}
    print(mode, out['hip_linear=' + mode], flush=True)
json.dump(out, open('gpurun_out/generation_times.json', 'w'), indent=1)
